// The four GEMMs of an encoder layer for a clip or two (the demo's single-clip transfer, configs[0]; mdm_forstyledataset.py:539-546):
// 64 tokens x 128 features per workgroup over a 2-D grid, so that tens of workgroups share a weight matrix.
//
// Round 1-3 ran these on the LDS-DMA slab ring (k_gemm_dma<64,128,...>): 8 .. 16 slabs of DMA -> wait -> barrier -> MFMA in series per
// workgroup, 4 .. 10 us per launch for 2 .. 8 MFLOP-sized tiles -- the slab loop's latency, not its bandwidth (profiles/r04_single_clip_families.txt).
// Here, as in the round-4 embed kernels (mst_embed.h): the tile's 64 token rows land in LDS in ONE LDS-DMA burst (K = 512: 64 KB,
// K = 1024: 128 KB) and stay for the whole K range; the weights are wave-private (wave w owns the tile's 16-row block w) and stream
// L2 -> VGPR as pre-packed 1-KB fragments behind hand-counted waits; a lane of the 16x16x32 accumulator holds four consecutive features
// of one token, which go straight to global memory (rows of a few hundred tokens: no transposition through LDS).
#pragma once
#include "mst_common.h"
#include "mst_embed.h"

namespace mst {

// W [N][K] f16 (row stride ldw) -> fragments [N / 16 blocks][K / 32][1 KB]: block b, k-step k32, lane l = row 16 b + (l & 15), k 32 k32 + 8 (l >> 4) .. + 7
__global__ __launch_bounds__(256) void k_pack_blocks(const f16* __restrict__ W, int ldw, int N, int K, f16* __restrict__ dst) {
    const int KS = K / 32, total = (N / 16) * KS * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, fi = i >> 6, k32 = fi % KS, b = fi / KS;
        reinterpret_cast<uint4*>(dst)[i] = *reinterpret_cast<const uint4*>(W + (size_t)(16 * b + (lane & 15)) * ldw + 32 * k32 + 8 * (lane >> 4));
    }
}

// MODE 0: + bias -> f16 [M][ldo];  1: + bias, erf GELU -> f16;  2: fp32 [M][ldo] as it is (the LayerNorm behind it adds the bias).
// KS = K / 32 (16 or 32).  LDS: token row r = KS / 16 pieces of 1 KB, 16-B chunk c of a piece at c ^ (r & 15).
template <int KS, int MODE>
__global__ __launch_bounds__(512) void k_rows_gemm(const f16* __restrict__ X, const f16* __restrict__ wpk, const float* __restrict__ bias,
                                                   void* __restrict__ out, int ldo, int M) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(KS == 16 || KS == 32, "K = 512 or 1024");
    constexpr int NP = KS / 16, ROWB = NP * 1024, D = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t16 = lane & 15, q4 = lane >> 4;
    const int tok0 = blockIdx.x * 64;
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // rows [8 w, 8 w + 8) of the tile: NP pieces each
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int r = 8 * wave + j;
        int tok = tok0 + r;
        if (tok >= M) tok = M - 1;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const unsigned voff = (unsigned)tok * (unsigned)(KS * 64) + (unsigned)(p * 1024) + (unsigned)((lane ^ (r & 15)) << 4);
            emb_glds(voff, (unsigned long long)X, __builtin_amdgcn_readfirstlane(smem_base + r * ROWB + p * 1024));
        }
    }
    const int blk = 8 * blockIdx.y + wave;                             // this wave's 16 output features
    const char* wsrc = reinterpret_cast<const char*>(wpk) + (size_t)blk * KS * 1024;
    f32x4 acc[4];
#pragma unroll
    for (int tb = 0; tb < 4; tb++) acc[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 xf[2][4];
    const unsigned xlane = (unsigned)t16 * ROWB, xswz = (unsigned)((q4 ^ t16) << 4);
    auto xread = [&](int k32, int p) {
        const char* src = smem + xlane + (unsigned)(k32 >> 4) * 1024u + (((unsigned)(k32 & 15) << 6) ^ xswz);
#pragma unroll
        for (int tb = 0; tb < 4; tb++) xf[p][tb] = *reinterpret_cast<const f16x8*>(src + tb * 16 * ROWB);
    };
    emb_stream<KS, D>(wsrc, (unsigned)lane * 16u,
        [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j == 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");      // the row pieces are older than the D fragments
                __syncthreads();
                xread(0, 0);
            }
            if constexpr (j + 1 < KS) xread(j + 1, (j + 1) & 1);
        },
        [&](auto jc, f16x8 wf) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int tb = 0; tb < 4; tb++) acc[tb] = mfma16(wf, xf[j & 1][tb], acc[tb]);
        });
    const int f = 16 * blk + 4 * q4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (MODE != 2) bv = *reinterpret_cast<const f32x4*>(bias + f);
#pragma unroll
    for (int tb = 0; tb < 4; tb++) {
        const int tok = tok0 + 16 * tb + t16;
        if (tok >= M) continue;
        f32x4 v = acc[tb] + bv;
        if (MODE == 1) v = f32x4{gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
        if (MODE == 2) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (size_t)tok * ldo + f) = v;
        else *reinterpret_cast<uint2*>(reinterpret_cast<f16*>(out) + (size_t)tok * ldo + f) = pack4_f16(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace mst
