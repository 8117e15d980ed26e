// The GEMMs of the TRAINING path at batch size (the fine-tune objective's 64-clip model call and its frozen motion encoder, forward and
// dgrad: mdm_forstyledataset.py:539-546 under autograd; train/finetune_style_diffusion.py's few_shot_style_finetune_losses), built like the
// sampling path's fused kernels instead of the LDS slab ring.
//
// Why: k_gemm_dma stages BOTH operands through LDS and every wave reads its 64 x 64 sub-tile's fragments from there -- 4 KB of LDS reads
// per 128 KFLOP, i.e. the CU's 128 B / clk are saturated exactly at the MFMA rate, and the 8 .. 48 slabs of a tile are DMA -> wait ->
// barrier -> MFMA rounds in series.  Measured at 12 608 token rows (profiles/r05_finetune_kernel_stats_streams1.csv): 30 .. 44 us per launch,
// 160 .. 660 TFLOP/s, a quarter of a CU's MFMA rate per workgroup -- while the layer tail of the sampling path does the same three
// products PLUS two LayerNorms in 43 us.  Here, as there (mst_tail.h):
//   * a workgroup owns BT tokens x 512 features; the tile's token rows land in LDS in ONE LDS-DMA burst (K f16 per row) and stay;
//   * the weights never touch LDS: wave w owns features [64 w, 64 w + 64) of the 512 for all BT tokens and streams its A fragments
//     (32 rows x 16 k, 1 KB, pre-packed in consumption order by k_pack_tok) L2 -> VGPR with D in flight behind hand-counted waits;
//   * LDS reads are the token fragments only: 2 KB per 128 KFLOP, half the ring's.
// The MFMA is the ring's (32x32x16, weights = A) in the ring's k order, so the accumulators are the ring's bit for bit and the
// epilogues are the ring's own classes, called with the ring's <BT, 512, BT / 32, 2> lane map.
#pragma once
#include "mst_common.h"
#include "mst_gemm_dma.h"
#include "mst_embed.h"

namespace mst {

// W [N][K] f16 (row stride ldw; N % 512 == 0, K % 16 == 0) -> per (512-column block cb, wave w) stream of (K / 16) x 2 fragments:
// fragment (k16, n) = rows cb 512 + 64 w + 32 n + (lane & 31), k 16 k16 + 8 (lane >> 5) .. + 7.
struct TokPackJob { const f16* W; f16* dst; int ldw, N, K, pad; };
struct TokPackJobs { TokPackJob j[64]; };
__global__ __launch_bounds__(256) void k_pack_tok(TokPackJobs jobs) {
    const TokPackJob& jb = jobs.j[blockIdx.y];
    const int KS = jb.K / 16, total = (jb.N / 32) * KS * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, fi = i >> 6, n = fi & 1, k16 = (fi >> 1) % KS, cw = (fi >> 1) / KS;      // cw = 8 cb + w
        const int row = 64 * cw + 32 * n + (lane & 31);
        reinterpret_cast<uint4*>(jb.dst)[i] = *reinterpret_cast<const uint4*>(jb.W + (size_t)row * jb.ldw + 16 * k16 + 8 * (lane >> 5));
    }
}

// X [M][ldx] f16 row-major (the first K columns are the operand), wpk = k_pack_tok's image of W [N][K]; grid (ceil(M / BT), N / 512).
// KP = K / 512 (1-KB pieces per token row).  LDS: token row r = KP pieces of 1 KB, 16-B chunk c of a piece at c ^ (r & 15)
// (conflict-free ds_read_b128 of the B operand: sixteen consecutive rows hit sixteen different 4-bank groups); the epilogue's
// transposition tile overlays the image behind a barrier.
template <int BT, int KP, int D, class EPI>
__global__ __launch_bounds__(512) void k_tok_gemm(const f16* __restrict__ X, int ldx, const f16* __restrict__ wpk, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(BT == 32 || BT == 64, "one or two 32-token blocks");
    constexpr int MT = BT / 32, NT = 2, ROWB = KP * 1024, KS = KP * 32, NFR = KS * NT, RPW = BT / 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = epi.rows();
    const int tok0 = blockIdx.x * BT, f0 = blockIdx.y * 512;
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // rows [RPW w, RPW w + RPW) of the tile: KP pieces each (rows beyond M: the last row again; never stored)
#pragma unroll
    for (int j = 0; j < RPW; j++) {
        const int r = RPW * wave + j;
        int tok = tok0 + r;
        if (tok >= M) tok = M - 1;
#pragma unroll
        for (int p = 0; p < KP; p++) {
            const unsigned voff = (unsigned)tok * (unsigned)(ldx * 2) + (unsigned)(p * 1024) + (unsigned)((lane ^ (r & 15)) << 4);
            emb_glds(voff, (unsigned long long)X, __builtin_amdgcn_readfirstlane(smem_base + r * ROWB + p * 1024));
        }
    }
    const char* wsrc = reinterpret_cast<const char*>(wpk) + (size_t)(8 * blockIdx.y + wave) * NFR * 1024;
    f32x16 acc[1][MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[0][m][n][r] = 0.f;
    f16x8 xf[2][MT];
    const int l31 = lane & 31, hh = lane >> 5;
    // k-step k16, lane half hh: 16-B chunk 2 k16 + hh of the row = piece k16 >> 5, chunk (2 k16 & 63) | hh, stored at chunk ^ (row & 15):
    // (a | hh) ^ b = a ^ (hh ^ b) for even a -- one per-lane constant, the rest is an immediate
    const unsigned xlane = (unsigned)l31 * ROWB, xswz = (unsigned)((hh ^ (l31 & 15)) << 4);
    auto xread = [&](int k16, int p) {
        const char* src = smem + xlane + (unsigned)(k16 >> 5) * 1024u + ((((unsigned)(2 * k16) & 63u) << 4) ^ xswz);
#pragma unroll
        for (int m = 0; m < MT; m++) xf[p][m] = *reinterpret_cast<const f16x8*>(src + m * 32 * ROWB);
    };
    emb_stream<NFR, D>(wsrc, (unsigned)lane * 16u,
        [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j == 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");      // the row pieces are older than the D fragments
                __syncthreads();
                xread(0, 0);
            }
            if constexpr (j % NT == 0 && j / NT + 1 < KS) xread(j / NT + 1, (j / NT + 1) & 1);
        },
        [&](auto jc, f16x8 wf) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int m = 0; m < MT; m++) acc[0][m][j % NT] = mfma_f16(wf, xf[(j / NT) & 1][m], acc[0][m][j % NT]);
        });
    __syncthreads();                                                  // every wave done reading the image: the epilogue overlays it
    epi.template run<BT, 512, MT, NT>(acc, tok0, f0, smem);
}

}  // namespace mst
