// The four GEMMs of an encoder layer for a clip or a few (the demo's single-clip transfer, BASELINE configs[0]; the fine-tune objective's
// chained single-clip calls; mdm_forstyledataset.py:539-546): 16 NTB tokens x 128 features per workgroup over a 2-D grid, so that tens of
// workgroups share a weight matrix.
//
// Rounds 1-3 ran these on the LDS-DMA slab ring (k_gemm_dma<64,128,...>): 8 .. 16 slabs of DMA -> wait -> barrier -> MFMA in series per
// workgroup, 4 .. 10 us per launch for tiles of a few MFLOP -- the slab loop's latency, not its bandwidth (profiles/r04_single_clip_ring_kernels.txt).
// Here, as in the round-4 embed kernels (mst_embed.h): the tile's token rows land in LDS in ONE LDS-DMA burst (K = 512: 1 KB per row,
// K = 1024: 2 KB) and stay for the whole K range; the weights are wave-private (wave w owns the tile's 16-row feature block w) and stream
// L2 -> VGPR as pre-packed 1-KB fragments behind hand-counted waits; a lane of the 16x16x32 accumulator holds four consecutive features
// of one token, which go straight to global memory (rows of a few hundred tokens: no transposition through LDS).
//
// Tile height (engine: launch_rows_gemm): 16 tokens up to 800 stream rows (the burst in front of the first MFMA is a quarter of the
// 64-token tile's, and the extra workgroups find idle CUs), 64 tokens above, 32 above 1 300 (measured: tools/experiments/r4_ntb1_sweep.sh, r4_ntb2_sweep.sh).
// LNF = 1 makes the rows instead of reading them: the LayerNorm between two GEMMs without a launch of its own (below).
// 2.8 .. 4.4 us per launch at one clip; single-clip step 378 -> 253 us (docs/LAB_NOTES.md section 3.6).
#pragma once
#include "mst_common.h"
#include "mst_embed.h"
#include "mst_train.h"

namespace mst {

// W [N][K] f16 (row stride ldw) -> fragments [N / 16 blocks][K / 32][1 KB]: block b, k-step k32, lane l = row 16 b + (l & 15), k 32 k32 + 8 (l >> 4) .. + 7
__global__ __launch_bounds__(256) void k_pack_blocks(const f16* __restrict__ W, int ldw, int N, int K, f16* __restrict__ dst) {
    const int KS = K / 32, total = (N / 16) * KS * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, fi = i >> 6, k32 = fi % KS, b = fi / KS;
        reinterpret_cast<uint4*>(dst)[i] = *reinterpret_cast<const uint4*>(W + (size_t)(16 * b + (lane & 15)) * ldw + 32 * k32 + 8 * (lane >> 4));
    }
}

// LNF = 1: the token rows are not read but MADE: row = LayerNorm(acc + bias + residual) of the GEMM launch in front (k_ln_rows' arithmetic,
// lane for lane: the same bits), so that the stream's LayerNorm needs no launch of its own between two GEMMs (16 launches of ~3 us plus
// their boundaries per step at one clip).  Every workgroup of a token tile (N / 128 of them) normalises the tile's 64 rows for itself
// -- 192 KB of L2 reads against a launch boundary -- and column 0 also writes the rows out as the stream's new (hi, lo) pair: into
// ANOTHER buffer than the residual it reads, which the other columns are still reading.
struct LnRows {
    const float *acc, *bias, *gamma, *beta;     // fp32 GEMM result [M][512]; bias of that GEMM; LayerNorm weight, bias
    const f16 *res_hi, *res_lo;                 // the residual stream in front of the GEMM
    f16 *out_hi, *out_lo;                       // the stream behind the LayerNorm
    // training (round 6: the chained single-clip calls of the fine-tune objective, 16 LayerNorm launches fewer per call): the GEMM result
    // passes the site's dropout before the residual joins -- z = res + dropout(acc + bias), k_ln_rows_train's arithmetic -- and z goes to
    // its tape slot (column 0 writes it, like the rows).  z_hi == nullptr: inference.
    f16 *z_hi = nullptr, *z_lo = nullptr;
    Drop d = Drop{0u, 0u, 1.0f};
};

// The same re-layout for up to 32 matrices in one launch (blockIdx.y = job): the training path re-uploads every layer each iteration.
struct PackJob { const f16* W; f16* dst; int ldw, N, K, pad; };
struct PackJobs { PackJob j[32]; };
__global__ __launch_bounds__(256) void k_pack_blocks_multi(PackJobs jobs) {
    const PackJob& jb = jobs.j[blockIdx.y];
    const int KS = jb.K / 32, total = (jb.N / 16) * KS * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, fi = i >> 6, k32 = fi % KS, b = fi / KS;
        reinterpret_cast<uint4*>(jb.dst)[i] = *reinterpret_cast<const uint4*>(jb.W + (size_t)(16 * b + (lane & 15)) * jb.ldw + 32 * k32 + 8 * (lane >> 4));
    }
}

// MODE 3 (training FFN1, OpFfn1Train's arithmetic): pre = f16(acc + bias) -> ft.pre, out = f16(gelu(pre) * dropout keep-multiplier)
struct FfnTrain { f16* pre; Drop d; };

// MODE 0: + bias -> f16 [M][ldo];  1: + bias, erf GELU -> f16;  2: fp32 [M][ldo] as it is (the LayerNorm behind it adds the bias).
// KS = K / 32 (16 or 32).  LDS: token row r = KS / 16 pieces of 1 KB, 16-B chunk c of a piece at c ^ (r & 15).
// NTB: the tile is 16 NTB tokens high.  64 where the rows are read; 16 where they are made -- a workgroup's LayerNorm of 64 rows is
// ~830 VALU instructions per lane on two waves per SIMD (2.9 us measured on top of the 2.9 us GEMM, more than the launch it replaces),
// of 16 rows a quarter of that, and at the row counts this path serves (a clip or two) the extra workgroups find idle CUs.
template <int KS, int MODE, int LNF = 0, int NTB = 4>
__global__ __launch_bounds__(512) void k_rows_gemm(const f16* __restrict__ X, const f16* __restrict__ wpk, const float* __restrict__ bias,
                                                   void* __restrict__ out, int ldo, int M, LnRows ln = {}, FfnTrain ft = {}) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(KS == 16 || KS == 32, "K = 512 or 1024");
    static_assert(!LNF || KS == 16, "a LayerNorm row is 512 wide");
    constexpr int NP = KS / 16, ROWB = NP * 1024, D = 8, RPW = 2 * NTB, LB = RPW < 4 ? RPW : 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t16 = lane & 15, q4 = lane >> 4;
    const int tok0 = blockIdx.x * (16 * NTB);
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // rows [RPW w, RPW w + RPW) of the tile: NP pieces each
#pragma unroll
    for (int j = 0; j < (LNF ? 0 : RPW); j++) {
        const int r = RPW * wave + j;
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
    f32x4 acc[NTB];
#pragma unroll
    for (int tb = 0; tb < NTB; tb++) acc[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 xf[2][NTB];
    const unsigned xlane = (unsigned)t16 * ROWB, xswz = (unsigned)((q4 ^ t16) << 4);
    auto xread = [&](int k32, int p) {
        const char* src = smem + xlane + (unsigned)(k32 >> 4) * 1024u + (((unsigned)(k32 & 15) << 6) ^ xswz);
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) xf[p][tb] = *reinterpret_cast<const f16x8*>(src + tb * 16 * ROWB);
    };
    emb_stream<KS, D>(wsrc, (unsigned)lane * 16u,
        [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j == 0) {
                if constexpr (LNF) {
                    // (behind the first D weight fragments, which are in flight meanwhile; the compiler's own vmcnt waits for the
                    // loads below can only be stricter for the older fragment loads being there)
                    const int fa = lane * 4, fb = 256 + lane * 4;
                    const f32x4 ba = *reinterpret_cast<const f32x4*>(ln.bias + fa), bb = *reinterpret_cast<const f32x4*>(ln.bias + fb);
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(ln.gamma + fa), gb = *reinterpret_cast<const f32x4*>(ln.gamma + fb);
                    const f32x4 ea = *reinterpret_cast<const f32x4*>(ln.beta + fa), eb = *reinterpret_cast<const f32x4*>(ln.beta + fb);
#pragma unroll
                    for (int h = 0; h < RPW / LB; h++) {
                        f32x4 xa[LB], xb[LB], ta[LB], tb[LB];
#pragma unroll
                        for (int i = 0; i < LB; i++) {
                            int tok = tok0 + RPW * wave + LB * h + i;
                            if (tok >= M) tok = M - 1;
                            const size_t off = (size_t)tok * MST_D;
                            xa[i] = join4_f16(*reinterpret_cast<const uint2*>(ln.res_hi + off + fa), *reinterpret_cast<const uint2*>(ln.res_lo + off + fa));
                            xb[i] = join4_f16(*reinterpret_cast<const uint2*>(ln.res_hi + off + fb), *reinterpret_cast<const uint2*>(ln.res_lo + off + fb));
                            ta[i] = *reinterpret_cast<const f32x4*>(ln.acc + off + fa);
                            tb[i] = *reinterpret_cast<const f32x4*>(ln.acc + off + fb);
                        }
#pragma unroll
                        for (int i = 0; i < LB; i++) {
                            const int r = RPW * wave + LB * h + i, tok = tok0 + r;
                            if (ln.z_hi) {                                     // (uniform) training: dropout, then the residual; z -> tape
                                const uint32_t ia = (uint32_t)tok * (uint32_t)MST_D + (uint32_t)fa, ib = (uint32_t)tok * (uint32_t)MST_D + (uint32_t)fb;
#pragma unroll
                                for (int c = 0; c < 4; c++) {
                                    xa[i][c] += (ta[i][c] + ba[c]) * drop_mul(ln.d, ia + c);
                                    xb[i][c] += (tb[i][c] + bb[c]) * drop_mul(ln.d, ib + c);
                                }
                                if (blockIdx.y == 0 && tok < M) {
                                    const size_t off = (size_t)tok * MST_D;
                                    uint2 zh, zl;
                                    split4_f16(xa[i], zh, zl);
                                    *reinterpret_cast<uint2*>(ln.z_hi + off + fa) = zh;
                                    *reinterpret_cast<uint2*>(ln.z_lo + off + fa) = zl;
                                    split4_f16(xb[i], zh, zl);
                                    *reinterpret_cast<uint2*>(ln.z_hi + off + fb) = zh;
                                    *reinterpret_cast<uint2*>(ln.z_lo + off + fb) = zl;
                                }
                            } else {
#pragma unroll
                            for (int c = 0; c < 4; c++) {
                                xa[i][c] = ta[i][c] + ba[c] + xa[i][c];
                                xb[i][c] = tb[i][c] + bb[c] + xb[i][c];
                            }
                            }
                            ln_row_wave(xa[i], xb[i], ga, gb, ea, eb);
                            const f32x4 ya = xa[i], yb = xb[i];
                            uint2 ha, la, hb, lb;
                            split4_f16(ya, ha, la);
                            split4_f16(yb, hb, lb);
                            // features fa .. fa + 3 = bytes 8 lane .. of the row: chunk lane / 2 (fb: 32 + lane / 2), half (lane & 1)
                            char* dst = smem + r * ROWB + (lane & 1) * 8;
                            *reinterpret_cast<uint2*>(dst + (((lane >> 1) ^ (r & 15)) << 4)) = ha;
                            *reinterpret_cast<uint2*>(dst + (((32 + (lane >> 1)) ^ (r & 15)) << 4)) = hb;
                            if (blockIdx.y == 0 && tok < M) {
                                const size_t off = (size_t)tok * MST_D;
                                *reinterpret_cast<uint2*>(ln.out_hi + off + fa) = ha;
                                *reinterpret_cast<uint2*>(ln.out_lo + off + fa) = la;
                                *reinterpret_cast<uint2*>(ln.out_hi + off + fb) = hb;
                                *reinterpret_cast<uint2*>(ln.out_lo + off + fb) = lb;
                            }
                        }
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");      // the row pieces are older than the D fragments
                }
                __syncthreads();
                xread(0, 0);
            }
            if constexpr (j + 1 < KS) xread(j + 1, (j + 1) & 1);
        },
        [&](auto jc, f16x8 wf) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int tb = 0; tb < NTB; tb++) acc[tb] = mfma16(wf, xf[j & 1][tb], acc[tb]);
        });
    const int f = 16 * blk + 4 * q4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (MODE != 2) bv = *reinterpret_cast<const f32x4*>(bias + f);
#pragma unroll
    for (int tb = 0; tb < NTB; tb++) {
        const int tok = tok0 + 16 * tb + t16;
        if (tok >= M) continue;
        f32x4 v = acc[tb] + bv;
        if constexpr (MODE == 3) {
            const size_t o = (size_t)tok * ldo + f;
            const uint2 pv = pack4_f16(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<uint2*>(ft.pre + o) = pv;
            const f16x4 ph = __builtin_bit_cast(f16x4, pv);
            float h[4];
#pragma unroll
            for (int j = 0; j < 4; j++) h[j] = gelu_erf((float)ph[j]) * drop_mul(ft.d, (uint32_t)o + j);
            *reinterpret_cast<uint2*>(reinterpret_cast<f16*>(out) + o) = pack4_f16(h[0], h[1], h[2], h[3]);
            continue;
        }
        if (MODE == 1) v = f32x4{gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
        if (MODE == 2) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (size_t)tok * ldo + f) = v;
        else *reinterpret_cast<uint2*>(reinterpret_cast<f16*>(out) + (size_t)tok * ldo + f) = pack4_f16(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace mst
