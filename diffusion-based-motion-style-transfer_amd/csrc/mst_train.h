// The trainable encoder stack in TRAINING mode (few_shot_style_finetune_losses, gaussian_diffusion.py:1317-1399,
// back-propagates through the 8 `seqTransEncoder` layers of StyleDiffusion, mdm_forstyledataset.py:539-546):
//   * forward epilogues that also write the activation tape and apply dropout,
//   * the backward pass: LayerNorm backward, GELU backward, attention backward, dgrad / wgrad GEMMs on the
//     same LDS-DMA main loop as the inference kernels (wgrad = split-K over the token dimension).
// Numerics: f16 MFMA operands, fp32 accumulation; incoming gradients are rescaled on the device to a fixed
// f16-friendly magnitude (k_grad_scale) and every fp32 result is unscaled where it is written.
#pragma once
#include "mst_common.h"
#include "mst_gemm_dma.h"
#include "mst_attn.h"

namespace mst {

// (the counter-based dropout mask -- struct Drop, mix32, drop_mul -- lives in mst_common.h: the pose-embedding epilogue applies it too)

// d/dx of the erf GELU:  Phi(x) + x phi(x)
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.0f + erf_as(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// ------------------------------------------------------------------------------------------------------------
// Epilogue: accumulators (+ bias) -> f16 tile in LDS -> OP on whole 16-byte row chunks (8 consecutive
// features of one token), fully coalesced.  Same pass structure as DEpiBiasF16.
// ------------------------------------------------------------------------------------------------------------
template <class OP>
struct DEpiRowOp {
    const float* bias; int M; OP op;              // bias may be null
    __device__ __forceinline__ int rows() const { return M; }
    template <int BT, int BF> static constexpr int pass_rows() { return (BT * (BF * 2 + 16) <= 69632) ? BT : (BF == 512 ? 64 : 128); }
    template <int BT, int BF> static constexpr int smem_bytes() { return pass_rows<BT, BF>() * (BF * 2 + 16); }
    template <int BT, int BF, int MT, int NT>
    __device__ __forceinline__ void run(f32x16 (&acc)[1][MT][NT], int tok0, int f0, char* smem) const {
        static_assert(BF == 128 || BF == 256 || BF == 512, "copy-out assumes 256-byte, 512-byte or 1-KiB tile rows");
        constexpr int LD = BF * 2 + 16;
        constexpr int PR = pass_rows<BT, BF>();
        constexpr int PASSES = BT / PR;
        constexpr int LPR = BF / 8, RPA = 64 / LPR;           // lanes per tile row, rows per 1-KiB wave access
        DLane<BT, BF, MT, NT> lc;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int sub = lane / LPR, col = (lane % LPR) * 16;
#pragma unroll
        for (int pass = 0; pass < PASSES; pass++) {
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const int tl = lc.tok(m);
                if (tl / PR != pass) continue;                // wave-uniform
                char* trow = smem + (tl - pass * PR) * LD;
#pragma unroll
                for (int n = 0; n < NT; n++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int f = lc.feat(n, g);
                        f32x4 b = {0.f, 0.f, 0.f, 0.f};
                        if (bias) b = *reinterpret_cast<const f32x4*>(bias + f0 + f);
                        *reinterpret_cast<uint2*>(trow + f * 2) =
                            pack4_f16(acc[0][m][n][4 * g] + b[0], acc[0][m][n][4 * g + 1] + b[1],
                                      acc[0][m][n][4 * g + 2] + b[2], acc[0][m][n][4 * g + 3] + b[3]);
                    }
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < PR / (8 * RPA); p++) {
                const int row = p * 8 * RPA + wave * RPA + sub;
                const int tok = tok0 + pass * PR + row;
                const uint4 v = *reinterpret_cast<const uint4*>(smem + row * LD + col);
                if (tok < M) op(tok, f0 + (col >> 1), v);
            }
            if (pass + 1 < PASSES) __syncthreads();
        }
    }
};

// FFN1 in training: tile = pre-activation (f16).  Writes `pre` and hid = dropout(gelu(pre)).
struct OpFfn1Train {
    f16* pre; f16* hid; int ld; Drop d;
    __device__ __forceinline__ void operator()(int tok, int c, uint4 v) const {
        const size_t o = (size_t)tok * ld + c;
        *reinterpret_cast<uint4*>(pre + o) = v;
        const f16x8 p = __builtin_bit_cast(f16x8, v);
        f16x8 h;
#pragma unroll
        for (int j = 0; j < 8; j++) h[j] = (f16)(gelu_erf((float)p[j]) * drop_mul(d, (uint32_t)o + j));
        *reinterpret_cast<uint4*>(hid + o) = __builtin_bit_cast(uint4, h);
    }
};
// FFN2 dgrad: tile = d hid.  out = tile * dropout-mask * gelu'(pre).
struct OpGeluBwd {
    const f16* pre; f16* out; int ld; Drop d;
    f16* hid = nullptr;          // non-null: also hid = dropout(GELU(pre)) (OpFfn1Train's arithmetic) -> the operand of dW2
    __device__ __forceinline__ void operator()(int tok, int c, uint4 v) const {
        const size_t o = (size_t)tok * ld + c;
        const f16x8 g = __builtin_bit_cast(f16x8, v);
        const f16x8 p = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(pre + o));
        f16x8 r, h;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float m = drop_mul(d, (uint32_t)o + j);
            r[j] = (f16)((float)g[j] * m * gelu_grad((float)p[j]));
            h[j] = (f16)(gelu_erf((float)p[j]) * m);
        }
        *reinterpret_cast<uint4*>(out + o) = __builtin_bit_cast(uint4, r);
        if (hid) *reinterpret_cast<uint4*>(hid + o) = __builtin_bit_cast(uint4, h);
    }
};

// ------------------------------------------------------------------------------------------------------------
// Epilogue: fp32 out (+ optional fp32 residual), row-major, coalesced through LDS in passes of 64 rows.
// dgrad GEMMs that end in the fp32 gradient stream, and the split-K partial products of the wgrad GEMMs.
// ------------------------------------------------------------------------------------------------------------
struct DEpiF32 {
    const float* resid; float* out; int ldo; int M;
    const float* gscale = nullptr;                    // non-null: accumulators are multiplied by gscale[1] (the unscale factor) first
    __device__ __forceinline__ int rows() const { return M; }
    template <int BT> static constexpr int pass_rows() { return BT >= 128 ? 64 : 32; }       // tile rows staged per pass
    template <int BT, int BF> static constexpr int smem_bytes() { return pass_rows<BT>() * (BF * 4 + 16); }
    template <int BT, int BF, int MT, int NT>
    __device__ __forceinline__ void run(f32x16 (&acc)[1][MT][NT], int tok0, int f0, char* smem) const {
        static_assert(BF == 256 || BF == 128, "1-KiB or 512-byte fp32 tile rows");
        constexpr int LD = BF * 4 + 16, PR = pass_rows<BT>(), PASSES = BT / PR;
        constexpr int LPR = BF / 4, RPA = 64 / LPR;           // lanes per row (16 B each), rows per wave access
        DLane<BT, BF, MT, NT> lc;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int sub = lane / LPR, lcol = lane % LPR;
#pragma unroll
        for (int pass = 0; pass < PASSES; pass++) {
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const int tl = lc.tok(m);
                if (tl / PR != pass) continue;                // wave-uniform
                char* trow = smem + (tl - pass * PR) * LD;
#pragma unroll
                for (int n = 0; n < NT; n++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        f32x4 v = {acc[0][m][n][4 * g], acc[0][m][n][4 * g + 1], acc[0][m][n][4 * g + 2], acc[0][m][n][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(trow + lc.feat(n, g) * 4) = v;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < PR / (8 * RPA); p++) {
                const int row = p * 8 * RPA + wave * RPA + sub, tok = tok0 + pass * PR + row;
                if (tok >= M) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * LD + lcol * 16);
                const size_t o = (size_t)tok * ldo + f0 + lcol * 4;
                if (gscale) {
                    const float us = gscale[1];
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] *= us;
                }
                if (resid) {
                    const f32x4 r = *reinterpret_cast<const f32x4*>(resid + o);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] += r[i];
                }
                *reinterpret_cast<f32x4*>(out + o) = v;
            }
            if (pass + 1 < PASSES) __syncthreads();
        }
    }
};

// split-K GEMM: grid (rows / BT, cols / BF, splits); split z contracts k in [z * kchunk, (z+1) * kchunk) and
// writes its own partial product at out + z * out_stride.
template <int BT, int BF, int MT, int NT, int NS>
__global__ __launch_bounds__(512) void k_gemm_splitk(const f16* __restrict__ A, int lda, const f16* __restrict__ B, int ldb,
                                                     int kchunk, size_t out_stride, DEpiF32 epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = DTile<BT, BF, MT, NT, NS, 1>;
    const int tok0 = blockIdx.x * BT, f0 = blockIdx.y * BF, z = blockIdx.z;
    f32x16 acc[1][MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[0][m][n][r] = 0.f;
    RowsDirect xs{A + (size_t)z * kchunk, lda};
    gemm_mainloop_dma<TL, BT, BF, MT, NT, 1, RowsDirect>(smem, xs, B + (size_t)z * kchunk, ldb, tok0, f0, kchunk, acc);
    epi.out += (size_t)z * out_stride;
    epi.template run<BT, BF, MT, NT>(acc, tok0, f0, smem);
}

// grad[i] += unscale * sum_z part[z][i]
__global__ void k_splitk_reduce(const float* __restrict__ part, int nsplit, size_t n, const float* __restrict__ gscale,
                                float* __restrict__ grad) {
    // (one float4 per thread and the splits' loads four at a time: 256 blocks walking the splits one load at a time took 37 us per
    // call for ~40 MB, a tenth of what the memory system moves; the order of the adds -- split 0, 1, 2, ... -- is unchanged)
    const float inv = gscale[1];
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 4 <= nsplit; z += 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(part + (size_t)(z + u) * n + i));
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int j = 0; j < 4; j++) s[j] += v[u][j];
        }
        for (; z < nsplit; z++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part + (size_t)z * n + i);
#pragma unroll
            for (int j = 0; j < 4; j++) s[j] += v[j];
        }
        f32x4 g = *reinterpret_cast<const f32x4*>(grad + i);
#pragma unroll
        for (int j = 0; j < 4; j++) g[j] += inv * s[j];
        *reinterpret_cast<f32x4*>(grad + i) = g;
    }
}

// The split-K reduces of a layer's (up to four) weight gradients in ONE launch (round 6: three launches less per layer on the weight-gradient
// stream, the tail of the backward pass; every element's adds in the order of k_splitk_reduce).  blockIdx.y = job.
struct RedJob { const float* part; float* grad; size_t n; int nsplit, pad; };
struct RedJobs { RedJob j[4]; };
__global__ void k_splitk_reduce_multi(RedJobs jobs, const float* __restrict__ gscale) {
    const RedJob jb = jobs.j[blockIdx.y];
    const float inv = gscale[1];
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < jb.n; i += (size_t)gridDim.x * blockDim.x * 4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 4 <= jb.nsplit; z += 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(jb.part + (size_t)(z + u) * jb.n + i));
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int j = 0; j < 4; j++) s[j] += v[u][j];
        }
        for (; z < jb.nsplit; z++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(jb.part + (size_t)z * jb.n + i);
#pragma unroll
            for (int j = 0; j < 4; j++) s[j] += v[j];
        }
        f32x4 g = *reinterpret_cast<const f32x4*>(jb.grad + i);
#pragma unroll
        for (int j = 0; j < 4; j++) g[j] += inv * s[j];
        *reinterpret_cast<f32x4*>(jb.grad + i) = g;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Training variant of DEpiResidLN: y = LN(z), z = resid + dropout(acc + bias); residual read from (rin),
// z and y written to their own tape slots (hi/lo pairs).
// ------------------------------------------------------------------------------------------------------------
struct DEpiResidLNTrain {
    const float* bias; const float* gamma; const float* beta;
    const f16* rin_hi; const f16* rin_lo; f16* z_hi; f16* z_lo; f16* y_hi; f16* y_lo; int M; Drop d;
    __device__ __forceinline__ int rows() const { return M; }
    template <int BT, int BF> static constexpr int smem_bytes() { return BT * (MST_D * 4 + 16); }
    template <int BT, int BF, int MT, int NT>
    __device__ __forceinline__ void run(f32x16 (&acc)[1][MT][NT], int tok0, int f0, char* smem) const {
        static_assert(BF == MST_D, "LayerNorm needs the whole row in one block");
        constexpr int LD = MST_D * 4 + 16;
        DLane<BT, BF, MT, NT> lc;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            char* trow = smem + lc.tok(m) * LD;
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 v = {acc[0][m][n][4 * g], acc[0][m][n][4 * g + 1], acc[0][m][n][4 * g + 2], acc[0][m][n][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(trow + lc.feat(n, g) * 4) = v;
                }
        }
        __syncthreads();
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int fa = lane * 4, fb = 256 + lane * 4;
        const f32x4 ba = *reinterpret_cast<const f32x4*>(bias + fa), bb = *reinterpret_cast<const f32x4*>(bias + fb);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + fa), gb = *reinterpret_cast<const f32x4*>(gamma + fb);
        const f32x4 ea = *reinterpret_cast<const f32x4*>(beta + fa), eb = *reinterpret_cast<const f32x4*>(beta + fb);
        constexpr int RPW = BT / 8;
        for (int r = 0; r < RPW; r++) {
            const int row = wave * RPW + r, tok = tok0 + row;
            if (tok >= M) continue;                            // wave-uniform
            const size_t off = (size_t)tok * MST_D;
            f32x4 xa = join4_f16(*reinterpret_cast<const uint2*>(rin_hi + off + fa), *reinterpret_cast<const uint2*>(rin_lo + off + fa));
            f32x4 xb = join4_f16(*reinterpret_cast<const uint2*>(rin_hi + off + fb), *reinterpret_cast<const uint2*>(rin_lo + off + fb));
            const f32x4 ta = *reinterpret_cast<const f32x4*>(smem + row * LD + fa * 4);
            const f32x4 tb = *reinterpret_cast<const f32x4*>(smem + row * LD + fb * 4);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xa[i] += (ta[i] + ba[i]) * drop_mul(d, (uint32_t)(off + fa + i));
                xb[i] += (tb[i] + bb[i]) * drop_mul(d, (uint32_t)(off + fb + i));
                s += xa[i] + xb[i];
            }
            uint2 h, l;
            split4_f16(xa, h, l);
            *reinterpret_cast<uint2*>(z_hi + off + fa) = h;
            *reinterpret_cast<uint2*>(z_lo + off + fa) = l;
            split4_f16(xb, h, l);
            *reinterpret_cast<uint2*>(z_hi + off + fb) = h;
            *reinterpret_cast<uint2*>(z_lo + off + fb) = l;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float mean = s * (1.0f / MST_D);
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xa[i] -= mean;
                xb[i] -= mean;
                s2 += xa[i] * xa[i] + xb[i] * xb[i];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
            const float rstd = ln_rstd(s2);
            f32x4 ya, yb;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ya[i] = xa[i] * rstd * ga[i] + ea[i];
                yb[i] = xb[i] * rstd * gb[i] + eb[i];
            }
            split4_f16(ya, h, l);
            *reinterpret_cast<uint2*>(y_hi + off + fa) = h;
            *reinterpret_cast<uint2*>(y_lo + off + fa) = l;
            split4_f16(yb, h, l);
            *reinterpret_cast<uint2*>(y_hi + off + fb) = h;
            *reinterpret_cast<uint2*>(y_lo + off + fb) = l;
        }
    }
};

// ------------------------------------------------------------------------------------------------------------
// Elementwise / row-wise helpers of the training path
// ------------------------------------------------------------------------------------------------------------
// Small launches in training: z = resid + dropout(acc + bias), y = LayerNorm(z); residual from one tape slot, z and y to
// two others (the row-wise counterpart of DEpiResidLNTrain; `acc` comes from a 64 x 128-tiled GEMM + DEpiPlainF32).
__global__ __launch_bounds__(256) void k_ln_rows_train(const float* __restrict__ acc, const float* __restrict__ bias,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const f16* __restrict__ rin_hi, const f16* __restrict__ rin_lo,
                                                       f16* __restrict__ z_hi, f16* __restrict__ z_lo, f16* __restrict__ y_hi,
                                                       f16* __restrict__ y_lo, int M, Drop d) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int fa = lane * 4, fb = 256 + lane * 4;
    const size_t off = (size_t)row * MST_D;
    f32x4 xa = join4_f16(*reinterpret_cast<const uint2*>(rin_hi + off + fa), *reinterpret_cast<const uint2*>(rin_lo + off + fa));
    f32x4 xb = join4_f16(*reinterpret_cast<const uint2*>(rin_hi + off + fb), *reinterpret_cast<const uint2*>(rin_lo + off + fb));
    const f32x4 ta = *reinterpret_cast<const f32x4*>(acc + off + fa), tb = *reinterpret_cast<const f32x4*>(acc + off + fb);
    const f32x4 ba = *reinterpret_cast<const f32x4*>(bias + fa), bb = *reinterpret_cast<const f32x4*>(bias + fb);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        xa[i] += (ta[i] + ba[i]) * drop_mul(d, (uint32_t)(off + fa + i));
        xb[i] += (tb[i] + bb[i]) * drop_mul(d, (uint32_t)(off + fb + i));
        s += xa[i] + xb[i];
    }
    uint2 h, l;
    split4_f16(xa, h, l);
    *reinterpret_cast<uint2*>(z_hi + off + fa) = h;
    *reinterpret_cast<uint2*>(z_lo + off + fa) = l;
    split4_f16(xb, h, l);
    *reinterpret_cast<uint2*>(z_hi + off + fb) = h;
    *reinterpret_cast<uint2*>(z_lo + off + fb) = l;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / MST_D);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        xa[i] -= mean;
        xb[i] -= mean;
        s2 += xa[i] * xa[i] + xb[i] * xb[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
    const float rstd = ln_rstd(s2);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + fa), gb = *reinterpret_cast<const f32x4*>(gamma + fb);
    const f32x4 ea = *reinterpret_cast<const f32x4*>(beta + fa), eb = *reinterpret_cast<const f32x4*>(beta + fb);
    f32x4 ya, yb;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ya[i] = xa[i] * rstd * ga[i] + ea[i];
        yb[i] = xb[i] * rstd * gb[i] + eb[i];
    }
    split4_f16(ya, h, l);
    *reinterpret_cast<uint2*>(y_hi + off + fa) = h;
    *reinterpret_cast<uint2*>(y_lo + off + fa) = l;
    split4_f16(yb, h, l);
    *reinterpret_cast<uint2*>(y_hi + off + fb) = h;
    *reinterpret_cast<uint2*>(y_lo + off + fb) = l;
}

// Model-level training node (input / output projections inside the native graph): positional-encoding dropout on the
// assembled token stream (PositionalEncoding.dropout, mdm_forstyledataset.py:404), its backward on the fp32 gradient, a
// plain f32 -> f16 copy, and the epilogue that scatters frame-row gradients back into token rows.
__global__ void k_dropout_stream(f16* __restrict__ hi, f16* __restrict__ lo, size_t n, Drop d) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        f32x4 v = join4_f16(*reinterpret_cast<const uint2*>(hi + i), *reinterpret_cast<const uint2*>(lo + i));
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] *= drop_mul(d, (uint32_t)(i + j));
        uint2 h, l;
        split4_f16(v, h, l);
        *reinterpret_cast<uint2*>(hi + i) = h;
        *reinterpret_cast<uint2*>(lo + i) = l;
    }
}
__global__ void k_mask_f32(float* __restrict__ g, size_t n, Drop d) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) g[i] *= drop_mul(d, (uint32_t)i);
}
__global__ void k_f32_to_f16(const float* __restrict__ in, size_t n, f16* __restrict__ out) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + i);
        *reinterpret_cast<uint2*>(out + i) = pack4_f16(v[0], v[1], v[2], v[3]);
    }
}
// backward of the output projection: frame rows x W_out -> gradient of the token stream, fp32 row clip*S + 1 + t
// (the conditioning token's row is zeroed by the caller); 64 x 512 tile, whole rows, like DEpiEmbedIn.
struct DEpiFramesGrad {
    float* out; int T, S, total;
    __device__ __forceinline__ int rows() const { return total; }
    template <int BT, int BF> static constexpr int smem_bytes() { return BT * (MST_D * 4 + 16); }
    template <int BT, int BF, int MT, int NT>
    __device__ __forceinline__ void run(f32x16 (&acc)[1][MT][NT], int tok0, int f0, char* smem) const {
        static_assert(BF == MST_D, "whole rows");
        constexpr int LD = MST_D * 4 + 16;
        DLane<BT, BF, MT, NT> lc;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            char* trow = smem + lc.tok(m) * LD;
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 v = {acc[0][m][n][4 * g], acc[0][m][n][4 * g + 1], acc[0][m][n][4 * g + 2], acc[0][m][n][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(trow + lc.feat(n, g) * 4) = v;
                }
        }
        __syncthreads();
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int fa = lane * 4, fb = 256 + lane * 4;
        constexpr int RPW = BT / 8;
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int row = wave * RPW + r, tok = tok0 + row;
            if (tok >= total) continue;
            const int clip = tok / T, t = tok - clip * T;
            const size_t off = ((size_t)clip * S + 1 + t) * MST_D;
            *reinterpret_cast<f32x4*>(out + off + fa) = *reinterpret_cast<const f32x4*>(smem + row * LD + fa * 4);
            *reinterpret_cast<f32x4*>(out + off + fb) = *reinterpret_cast<const f32x4*>(smem + row * LD + fb * 4);
        }
    }
};
// zero the conditioning token's row of every clip in an fp32 [rows][S][512] buffer
__global__ void k_zero_token0(float* __restrict__ g, int S, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * MST_D) g[(size_t)(i / MST_D) * S * MST_D + (i % MST_D)] = 0.f;
}

// fp32 rows -> hi/lo pair
__global__ void k_split_stream(const float* __restrict__ in, size_t n, f16* __restrict__ hi, f16* __restrict__ lo) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + i);
        uint2 h, l;
        split4_f16(v, h, l);
        *reinterpret_cast<uint2*>(hi + i) = h;
        *reinterpret_cast<uint2*>(lo + i) = l;
    }
}

// max |g| -> amax (float bits, non-negative -> unsigned order = float order)
__global__ void k_amax(const float* __restrict__ g, size_t n, unsigned* __restrict__ amax) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = fabsf(g[i]);
        m = a > m ? a : m;                               // NaN never wins: the scale stays finite
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}
// gscale[0] = power of two s with max|g| * s in [16, 32);  gscale[1] = 1 / s
__global__ void k_grad_scale(const unsigned* __restrict__ amax, float* __restrict__ gscale) {
    const float a = __uint_as_float(*amax);
    float s = 1.0f;
    if (a > 0.f && a < INFINITY) s = exp2f(floorf(log2f(32.0f / a)) - 0.0f);
    if (!(s > 0.f) || s == INFINITY) s = 1.0f;
    gscale[0] = s;
    gscale[1] = 1.0f / s;
}
__global__ void k_scale_f32(const float* __restrict__ in, size_t n, const float* __restrict__ gscale, int which, float* __restrict__ out) {
    const float s = gscale[which];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i] * s;
}

// LayerNorm backward for rows y = LN(z) * gamma + beta with z = resid + dropout(branch):
//   dz  = rstd * (gamma g - mean(gamma g) - xhat mean(gamma g xhat))                 (fp32, the residual gradient)
//   dbr = dz * dropout-mask                                                            (f16, the branch gradient = GEMM operand)
//   dgamma += sum_rows g xhat,  dbeta += sum_rows g,  dbias += sum_rows dbr            (unscaled, atomics)
// One wave per row; lane owns features [4 lane, +4) and [256 + 4 lane, +4).
// WPB waves per block: 4 for a clip or a few; 16 at batch size -- with 256 blocks of four waves every SIMD held ONE wave, and the row's
// three shuffle ladders, the mask hash and the memory round trips ran in series (35 us per launch at 12 608 rows, 2.6 TB/s).
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void k_ln_bwd(const float* __restrict__ g, const f16* __restrict__ z_hi, const f16* __restrict__ z_lo,
                                                const float* __restrict__ gamma, int M, Drop d, const float* __restrict__ gscale,
                                                float* __restrict__ dz, f16* __restrict__ dbr, float* __restrict__ part) {
    static_assert(WPB == 4 || WPB == 16, "partial sums: four LDS slots per block, each the ordered sum of WPB / 4 waves");
    __shared__ float red[3][4][MST_D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fa = lane * 4, fb = 256 + lane * 4;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + fa), gb = *reinterpret_cast<const f32x4*>(gamma + fb);
    f32x4 dga = {0.f, 0.f, 0.f, 0.f}, dgb = dga, dba = dga, dbb = dga, dca = dga, dcb = dga;
    // RB rows of a wave per pass, every load of the pass issued before the first row is worked on: the rows of a wave were load -> three
    // shuffle ladders -> store in series, twelve HBM round trips per launch at 12 608 rows with four waves on the CU (23 us).  The rows are
    // still PROCESSED one after the other in the old order, so the partial sums come out bit for bit as before.
    constexpr int RB = 4;
    const int rstride = gridDim.x * WPB;
    for (int row0 = blockIdx.x * WPB + wave; row0 < M; row0 += RB * rstride) {
        uint2 zha[RB], zla[RB], zhb[RB], zlb[RB];
        f32x4 gya[RB], gyb[RB];
#pragma unroll
        for (int k = 0; k < RB; k++) {
            const int rr = row0 + k * rstride;
            const size_t off = (size_t)(rr < M ? rr : row0) * MST_D;
            zha[k] = *reinterpret_cast<const uint2*>(z_hi + off + fa);
            zla[k] = *reinterpret_cast<const uint2*>(z_lo + off + fa);
            zhb[k] = *reinterpret_cast<const uint2*>(z_hi + off + fb);
            zlb[k] = *reinterpret_cast<const uint2*>(z_lo + off + fb);
            gya[k] = *reinterpret_cast<const f32x4*>(g + off + fa);
            gyb[k] = *reinterpret_cast<const f32x4*>(g + off + fb);
        }
#pragma unroll
        for (int k = 0; k < RB; k++) {
        const int row = row0 + k * rstride;
        if (row >= M) break;                                 // wave-uniform
        const size_t off = (size_t)row * MST_D;
        f32x4 xa = join4_f16(zha[k], zla[k]);
        f32x4 xb = join4_f16(zhb[k], zlb[k]);
        const f32x4 ya = gya[k], yb = gyb[k];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) s += xa[i] + xb[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / MST_D);
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            xa[i] -= mean;
            xb[i] -= mean;
            s2 += xa[i] * xa[i] + xb[i] * xb[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
        const float rstd = ln_rstd(s2);
        float c1 = 0.f, c2 = 0.f;
        f32x4 aa, ab;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            xa[i] *= rstd;                                   // xhat
            xb[i] *= rstd;
            aa[i] = ga[i] * ya[i];
            ab[i] = gb[i] * yb[i];
            c1 += aa[i] + ab[i];
            c2 += aa[i] * xa[i] + ab[i] * xb[i];
            dga[i] += ya[i] * xa[i];
            dgb[i] += yb[i] * xb[i];
            dba[i] += ya[i];
            dbb[i] += yb[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c1 += __shfl_xor(c1, o);
            c2 += __shfl_xor(c2, o);
        }
        c1 *= (1.0f / MST_D);
        c2 *= (1.0f / MST_D);
        f32x4 za, zb, ra, rb;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            za[i] = rstd * (aa[i] - c1 - xa[i] * c2);
            zb[i] = rstd * (ab[i] - c1 - xb[i] * c2);
            ra[i] = za[i] * drop_mul(d, (uint32_t)(off + fa + i));
            rb[i] = zb[i] * drop_mul(d, (uint32_t)(off + fb + i));
            dca[i] += ra[i];
            dcb[i] += rb[i];
        }
        *reinterpret_cast<f32x4*>(dz + off + fa) = za;
        *reinterpret_cast<f32x4*>(dz + off + fb) = zb;
        *reinterpret_cast<uint2*>(dbr + off + fa) = pack4_f16(ra[0], ra[1], ra[2], ra[3]);
        *reinterpret_cast<uint2*>(dbr + off + fb) = pack4_f16(rb[0], rb[1], rb[2], rb[3]);
        }
    }
    // slot (wave & 3) = the sums of waves wave & 3, + 4, + 8, + 12, added in that order (a fixed order: the results are reproducible)
#pragma unroll
    for (int grp = 0; grp < WPB / 4; grp++) {
        if ((wave >> 2) == grp) {
            const int sl = wave & 3;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (grp == 0) {
                    red[0][sl][fa + i] = dga[i]; red[0][sl][fb + i] = dgb[i];
                    red[1][sl][fa + i] = dba[i]; red[1][sl][fb + i] = dbb[i];
                    red[2][sl][fa + i] = dca[i]; red[2][sl][fb + i] = dcb[i];
                } else {
                    red[0][sl][fa + i] += dga[i]; red[0][sl][fb + i] += dgb[i];
                    red[1][sl][fa + i] += dba[i]; red[1][sl][fb + i] += dbb[i];
                    red[2][sl][fa + i] += dca[i]; red[2][sl][fb + i] += dcb[i];
                }
            }
        }
        __syncthreads();
    }
    // per-block partial sums; k_ln_bwd_finish adds them up in block order (float atomics here made dgamma / dbeta / dbias differ in the
    // last bits from run to run -- and between data-parallel replicas)
    for (int i = threadIdx.x; i < 3 * MST_D; i += 64 * WPB) {
        const int a = i / MST_D, f = i - a * MST_D;
        part[(size_t)blockIdx.x * (3 * MST_D) + i] = red[a][0][f] + red[a][1][f] + red[a][2][f] + red[a][3][f];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Epilogue: a dgrad GEMM whose result (+ the residual branch's gradient) is the gradient wrt a LayerNorm output, with that LayerNorm's
// backward behind it in the same launch (k_ln_bwd's arithmetic, row for row): the fp32 gradient stream's round trip between the two
// launches (51.6 MB at 12 608 rows) and one launch per LayerNorm less.  64 x 512 tiles (whole rows); the tile's [dgamma | dbeta | dbias]
// sums go to part[blockIdx.x] and k_ln_bwd_finish adds the tiles up in block order.
//   g = acc + resid;  dz = rstd (gamma g - mean(gamma g) - xhat mean(gamma g xhat)) -> dz (may alias resid: same lane, same addresses);
//   dbr = dz * keep-mask -> f16
// ------------------------------------------------------------------------------------------------------------
struct DEpiLnBwd {
    const float* resid; const f16* z_hi; const f16* z_lo; const float* gamma; int M; Drop d;
    float* dz; f16* dbr; float* part;
    __device__ __forceinline__ int rows() const { return M; }
    template <int BT, int BF> static constexpr int smem_bytes() { return BT * (MST_D * 4 + 16); }
    template <int BT, int BF, int MT, int NT>
    __device__ __forceinline__ void run(f32x16 (&acc)[1][MT][NT], int tok0, int f0, char* smem) const {
        static_assert(BF == MST_D && BT == 64, "whole rows, 64 per tile");
        constexpr int LD = MST_D * 4 + 16;
        DLane<BT, BF, MT, NT> lc;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            char* trow = smem + lc.tok(m) * LD;
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 v = {acc[0][m][n][4 * g], acc[0][m][n][4 * g + 1], acc[0][m][n][4 * g + 2], acc[0][m][n][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(trow + lc.feat(n, g) * 4) = v;
                }
        }
        __syncthreads();
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int fa = lane * 4, fb = 256 + lane * 4;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + fa), gb = *reinterpret_cast<const f32x4*>(gamma + fb);
        f32x4 dga = {0.f, 0.f, 0.f, 0.f}, dgb = dga, dba = dga, dbb = dga, dca = dga, dcb = dga;
        constexpr int RPW = BT / 8;
        // every row's tape and residual loads first (one memory latency per tile)
        uint2 zha[RPW], zla[RPW], zhb[RPW], zlb[RPW];
        f32x4 ra_[RPW], rb_[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int tok = tok0 + wave * RPW + r;
            const size_t off = (size_t)(tok < M ? tok : M - 1) * MST_D;
            zha[r] = *reinterpret_cast<const uint2*>(z_hi + off + fa);
            zla[r] = *reinterpret_cast<const uint2*>(z_lo + off + fa);
            zhb[r] = *reinterpret_cast<const uint2*>(z_hi + off + fb);
            zlb[r] = *reinterpret_cast<const uint2*>(z_lo + off + fb);
            ra_[r] = *reinterpret_cast<const f32x4*>(resid + off + fa);
            rb_[r] = *reinterpret_cast<const f32x4*>(resid + off + fb);
        }
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int row = wave * RPW + r, tok = tok0 + row;
            if (tok >= M) break;                               // wave-uniform; rows ascend
            const size_t off = (size_t)tok * MST_D;
            f32x4 xa = join4_f16(zha[r], zla[r]);
            f32x4 xb = join4_f16(zhb[r], zlb[r]);
            f32x4 ya = *reinterpret_cast<const f32x4*>(smem + row * LD + fa * 4);
            f32x4 yb = *reinterpret_cast<const f32x4*>(smem + row * LD + fb * 4);
#pragma unroll
            for (int i = 0; i < 4; i++) { ya[i] += ra_[r][i]; yb[i] += rb_[r][i]; }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) s += xa[i] + xb[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float mean = s * (1.0f / MST_D);
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xa[i] -= mean;
                xb[i] -= mean;
                s2 += xa[i] * xa[i] + xb[i] * xb[i];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
            const float rstd = ln_rstd(s2);
            float c1 = 0.f, c2 = 0.f;
            f32x4 aa, ab;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xa[i] *= rstd;                                   // xhat
                xb[i] *= rstd;
                aa[i] = ga[i] * ya[i];
                ab[i] = gb[i] * yb[i];
                c1 += aa[i] + ab[i];
                c2 += aa[i] * xa[i] + ab[i] * xb[i];
                dga[i] += ya[i] * xa[i];
                dgb[i] += yb[i] * xb[i];
                dba[i] += ya[i];
                dbb[i] += yb[i];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                c1 += __shfl_xor(c1, o);
                c2 += __shfl_xor(c2, o);
            }
            c1 *= (1.0f / MST_D);
            c2 *= (1.0f / MST_D);
            f32x4 za, zb, qa, qb;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                za[i] = rstd * (aa[i] - c1 - xa[i] * c2);
                zb[i] = rstd * (ab[i] - c1 - xb[i] * c2);
                qa[i] = za[i] * drop_mul(d, (uint32_t)(off + fa + i));
                qb[i] = zb[i] * drop_mul(d, (uint32_t)(off + fb + i));
                dca[i] += qa[i];
                dcb[i] += qb[i];
            }
            *reinterpret_cast<f32x4*>(dz + off + fa) = za;
            *reinterpret_cast<f32x4*>(dz + off + fb) = zb;
            *reinterpret_cast<uint2*>(dbr + off + fa) = pack4_f16(qa[0], qa[1], qa[2], qa[3]);
            *reinterpret_cast<uint2*>(dbr + off + fb) = pack4_f16(qb[0], qb[1], qb[2], qb[3]);
        }
        __syncthreads();                                       // the tile rows are consumed: the waves' sums overlay them
        float* red = reinterpret_cast<float*>(smem);           // [3][8][512]
#pragma unroll
        for (int i = 0; i < 4; i++) {
            red[(0 * 8 + wave) * MST_D + fa + i] = dga[i]; red[(0 * 8 + wave) * MST_D + fb + i] = dgb[i];
            red[(1 * 8 + wave) * MST_D + fa + i] = dba[i]; red[(1 * 8 + wave) * MST_D + fb + i] = dbb[i];
            red[(2 * 8 + wave) * MST_D + fa + i] = dca[i]; red[(2 * 8 + wave) * MST_D + fb + i] = dcb[i];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * MST_D; i += 512) {
            const int a = i / MST_D, f = i - a * MST_D;
            const float* rr = red + (size_t)a * 8 * MST_D + f;
            part[(size_t)blockIdx.x * (3 * MST_D) + i] = ((rr[0] + rr[MST_D]) + (rr[2 * MST_D] + rr[3 * MST_D])) + ((rr[4 * MST_D] + rr[5 * MST_D]) + (rr[6 * MST_D] + rr[7 * MST_D]));
        }
    }
};

// Ordered second stage of the partial-sum reductions: out[i] = sum_p part[p][i], p in a FIXED order -- kFinOut outputs per block, the
// partials dealt to kFinSlices thread slices (p = slice, slice + kFinSlices, ...) whose sums meet in a fixed binary tree.  Deterministic.
// Round 6: 16 slices x 16 outputs (96 blocks for LayerNorm's 1536 sums) instead of 4 x 64 (24 blocks): with 197 .. 256 partials a thread
// walked 50 .. 64 of them in series, 12 us per call behind every LayerNorm backward -- on the dgrad chain, 32 calls per iteration.
constexpr int kFinSlices = 16, kFinOut = 16;
__device__ __forceinline__ float ordered_partial_sum(const float* __restrict__ part, int nparts, int n, int i_local, int i, float (&red)[kFinSlices][kFinOut]) {
    const int slice = threadIdx.x / kFinOut;
    float s = 0.f;
    if (i < n) {
        int p = slice;
        for (; p + 7 * kFinSlices < nparts; p += 8 * kFinSlices) {                 // eight loads in flight, added in the same order as one at a time
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = part[(size_t)(p + kFinSlices * u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; p < nparts; p += kFinSlices) s += part[(size_t)p * n + i];
    }
    red[slice][i_local] = s;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x < kFinOut) {
        float a[kFinSlices];
#pragma unroll
        for (int k = 0; k < kFinSlices; k++) a[k] = red[k][i_local];
#pragma unroll
        for (int w = 1; w < kFinSlices; w *= 2)
#pragma unroll
            for (int k = 0; k < kFinSlices; k += 2 * w) a[k] += a[k + w];
        t = a[0];
    }
    return t;
}

// dgamma / dbeta / dbias += unscale * sum over the blocks of k_ln_bwd (grid = 3 * 512 / kFinOut blocks of 256 threads)
__global__ __launch_bounds__(256) void k_ln_bwd_finish(const float* __restrict__ part, int nblocks, const float* __restrict__ gscale,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias) {
    __shared__ float red[kFinSlices][kFinOut];
    const int il = threadIdx.x % kFinOut, i = blockIdx.x * kFinOut + il;
    const float s = ordered_partial_sum(part, nblocks, 3 * MST_D, il, i, red);
    if (threadIdx.x < kFinOut) {
        const int a = i / MST_D, f = i - a * MST_D;
        float* dst = a == 0 ? dgamma : (a == 1 ? dbeta : dbias);
        dst[f] += s * gscale[1];
    }
}

// Both LayerNorms of a layer in ONE launch (round 6: one launch less per layer on the dgrad chain; the sums and their order are those of two
// k_ln_bwd_finish launches).  blockIdx.y = 0: job a, 1: job b.
struct LnFinJob { const float* part; int nblocks; float *dgamma, *dbeta, *dbias; };
__global__ __launch_bounds__(256) void k_ln_bwd_finish2(LnFinJob ja, LnFinJob jb, const float* __restrict__ gscale) {
    __shared__ float red[kFinSlices][kFinOut];
    const LnFinJob j = blockIdx.y ? jb : ja;
    const int il = threadIdx.x % kFinOut, i = blockIdx.x * kFinOut + il;
    const float s = ordered_partial_sum(j.part, j.nblocks, 3 * MST_D, il, i, red);
    if (threadIdx.x < kFinOut) {
        const int a = i / MST_D, f = i - a * MST_D;
        float* dst = a == 0 ? j.dgamma : (a == 1 ? j.dbeta : j.dbias);
        dst[f] += s * gscale[1];
    }
}

// [N][K] float32 -> [K][N] f16 (transposed weight copies: the "weights" operand of the dgrad GEMMs)
__global__ void k_convert_transpose(const float* __restrict__ src, int N, int K, f16* __restrict__ dst, int ldd) {
    __shared__ float tile[32][33];
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) tile[j][tx] = (n0 + j < N && k0 + tx < K) ? src[(size_t)(n0 + j) * K + k0 + tx] : 0.f;
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (k0 + j < K && n0 + tx < N) dst[(size_t)(k0 + j) * ldd + n0 + tx] = (f16)tile[tx][j];
}

// debug / tests: the keep-multiplier of `n` consecutive elements of a site
__global__ void k_dropout_mask(Drop d, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = drop_mul(d, (uint32_t)i);
}

// ------------------------------------------------------------------------------------------------------------
// Attention in training: forward with dropout on the probabilities, and the backward pass.
// Dropout index of P[q][key] of (clip, head): ((clip * 4 + head) * S + q) * S + key.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t p_index(int ch, int S, int q, int key) { return ((uint32_t)ch * S + q) * S + key; }

// forward: k_attention (mst_attn.h) + dropout on P.  Both K and V use the transposed-read image layout.
// keep: optional [clips][S] bytes, 0 = key is padding (src_key_padding_mask of the motion encoder, mdm_forstyledataset.py:117-121)
__device__ __forceinline__ void stage_key_bias(float* kbias, const unsigned char* keep, int clip, int S, int KEYS, int tid) {
    for (int k = tid; k < KEYS; k += 512) kbias[k] = (k < S && (!keep || keep[(size_t)clip * S + k])) ? 0.f : -INFINITY;
}

template <int NKT>
__global__ __launch_bounds__(512) void k_attention_train(const f16* __restrict__ qkv, f16* __restrict__ out, int S, Drop d,
                                                         const unsigned char* __restrict__ keep, int qsplit,
                                                         float* __restrict__ lse_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KEYS = NKT * 32;
    char* ks = smem;
    char* vs = smem + KEYS * 256;
    float* kbias = reinterpret_cast<float*>(smem + 2 * KEYS * 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int clip = blockIdx.x / MST_H, head = blockIdx.x % MST_H;
    const f16* base = qkv + (size_t)clip * S * (3 * MST_D) + head * MST_HD;
    stage_key_bias(kbias, keep, clip, S, KEYS, tid);
    for (int q = tid; q < KEYS * 16; q += 512) {
        int row = q >> 4, ch = q & 15;
        uint4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
        if (row < S) {
            const f16* p = base + (size_t)row * (3 * MST_D) + ch * 8;
            kv = *reinterpret_cast<const uint4*>(p + MST_D);
            vv = *reinterpret_cast<const uint4*>(p + 2 * MST_D);
        }
        *reinterpret_cast<uint4*>(ks + k_off(row, ch)) = kv;
        *reinterpret_cast<uint4*>(vs + v_off(row, ch * 8)) = vv;
    }
    __syncthreads();
    if (wave >= NKT) return;
    int qtile = wave;
    if (qsplit) {                      // small batches: grid.y = NKT, one query tile per workgroup (see k_attention)
        if (wave != 0) return;
        qtile = blockIdx.y;
    }
    const int hh = lane >> 5;
    const int q_idx = qtile * 32 + (lane & 31);
    const int q_ld = q_idx < S ? q_idx : S - 1;
    f16x8 qf[8];
    {
        const f16* qp = base + (size_t)q_ld * (3 * MST_D) + 8 * hh;
#pragma unroll
        for (int s = 0; s < 8; s++) qf[s] = *reinterpret_cast<const f16x8*>(qp + s * 16);
    }
    f32x16 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; kt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) sc[kt][r] = 0.f;
        const int row = kt * 32 + (lane & 31);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            f16x8 kf = *reinterpret_cast<const f16x8*>(ks + k_off(row, 2 * s + hh));
            sc[kt] = mfma_f16(kf, qf[s], sc[kt]);
        }
    }
    const float scale = 0.08838834764831845f;
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float v = sc[kt][r] * scale + kbias[kt * 32 + mfma_row(r, lane)];
            sc[kt][r] = v;
            m = fmaxf(m, v);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float p = __expf(sc[kt][r] - m);
            sc[kt][r] = p;
            l += p;
        }
    l += __shfl_xor(l, 32);
    const float inv_l = 1.0f / l;
    const int ch = clip * MST_H + head;
    if (lse_out && hh == 0 && q_idx < S) lse_out[(size_t)ch * S + q_idx] = m + __logf(l);   // row statistics for the split backward
    f16x8 pf[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int key = kt * 32 + mfma_row(8 * s2 + j, lane);
                pf[kt][s2][j] = (f16)(sc[kt][8 * s2 + j] * inv_l * drop_mul(d, p_index(ch, S, q_idx, key)));
            }
    const int i16 = lane & 15, g = lane >> 4;
    const int key_lane = 4 * hh + (i16 >> 2);
    const int d_lane = 16 * (g & 1) + 4 * (i16 & 3);
    f16* orow = out + ((size_t)clip * S + q_ld) * MST_D + head * MST_HD;
#pragma unroll
    for (int dt = 0; dt < 4; dt++) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; kt++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                int key = kt * 32 + 16 * s2 + key_lane;
                int dd = dt * 32 + d_lane;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vs + v_off(key, dd)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vs + v_off(key + 8, dd)));
                const f16x4 lo_h = __builtin_bit_cast(f16x4, lo), hi_h = __builtin_bit_cast(f16x4, hi);
                f16x8 vf = __builtin_shufflevector(lo_h, hi_h, 0, 1, 2, 3, 4, 5, 6, 7);
                o = mfma_f16(vf, pf[kt][s2], o);
            }
        if (q_idx < S) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                int dd = dt * 32 + 8 * gq + 4 * hh;
                *reinterpret_cast<uint2*>(orow + dd) = pack4_f16(o[4 * gq], o[4 * gq + 1], o[4 * gq + 2], o[4 * gq + 3]);
            }
        }
    }
}

// Images of the backward kernel are read BOTH ways (row-major ds_read_b128 fragments and transposed ds_read_b64_tr_b16
// fragments).  Layout a_off: 256-B rows; the four 64-B pieces rotated by row & 3 (what the transposed read needs: its 4
// rows x 32 B then cover 32 distinct banks) AND the four 16-B chunks inside a piece rotated by (row >> 2) & 3, so the 16
// rows one ds_read_b128 phase touches land on 16 different chunk slots = all 64 banks.  (With v_off alone the row-major
// reads were 8-way conflicted: a row stride of 256 B maps every row to the same banks.)
__device__ __forceinline__ int a_off(int row, int d) {
    return row * 256 + (((d >> 5) ^ (row & 3)) << 6) + (((((d & 31) >> 3) ^ ((row >> 2) & 3))) << 4) + ((d & 7) << 1);
}
__device__ __forceinline__ f16x8 img_row_frag(const char* img, int row, int s, int hh) {       // d = 16 s + 8 hh .. + 8
    return *reinterpret_cast<const f16x8*>(img + a_off(row, 16 * s + 8 * hh));
}
// A-operand fragment of img^T for MFMA k-step (tile t, half s2) and d tile dt: lane (d = dt*32 + l31) receives
// rows t*32 + 16 s2 + 4 hh + {0..3} and + 8 + {0..3} -- the permuted k order the accumulator-as-operand trick needs.
__device__ __forceinline__ f16x8 img_tr_frag_a(const char* img, int t, int s2, int dt, int lane) {
    const int hh = lane >> 5, i16 = lane & 15, g = lane >> 4;
    const int row = t * 32 + 16 * s2 + 4 * hh + (i16 >> 2);
    const int dd = dt * 32 + 16 * (g & 1) + 4 * (i16 & 3);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + a_off(row, dd)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + a_off(row + 8, dd)));
    const f16x4 lo_h = __builtin_bit_cast(f16x4, lo), hi_h = __builtin_bit_cast(f16x4, hi);
    return __builtin_shufflevector(lo_h, hi_h, 0, 1, 2, 3, 4, 5, 6, 7);
}
// same for an image in the plain v_off layout (k_wgrad_tr: images written by the DMA ring, transposed reads only)
__device__ __forceinline__ f16x8 img_tr_frag(const char* img, int t, int s2, int dt, int lane) {
    const int hh = lane >> 5, i16 = lane & 15, g = lane >> 4;
    const int row = t * 32 + 16 * s2 + 4 * hh + (i16 >> 2);
    const int dd = dt * 32 + 16 * (g & 1) + 4 * (i16 & 3);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + v_off(row, dd)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + v_off(row + 8, dd)));
    const f16x4 lo_h = __builtin_bit_cast(f16x4, lo), hi_h = __builtin_bit_cast(f16x4, hi);
    return __builtin_shufflevector(lo_h, hi_h, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void stage_image(char* img, const f16* src, size_t row_stride, int S, int KEYS, int tid) {
    for (int q = tid; q < KEYS * 16; q += 512) {
        const int row = q >> 4, ch = q & 15;
        uint4 v = {0, 0, 0, 0};
        if (row < S) v = *reinterpret_cast<const uint4*>(src + (size_t)row * row_stride + ch * 8);
        *reinterpret_cast<uint4*>(img + a_off(row, ch * 8)) = v;
    }
}

// Backward of one (clip, head):  with P = softmax(Q K^T * scale), Pd = dropout(P), O = Pd V:
//   dV = Pd^T dO,  dPd = dO V^T,  dP = dPd * mask,  dS = P (dP - D) * scale with D_q = sum_d dO[q][d] O[q][d],
//   dQ = dS K,  dK = dS^T Q.
// Pass 1 (wave = 32-query tile, lane = query; K, V images in LDS): dQ.
// Pass 2 (wave = 32-key tile, lane = key; Q, dO images in LDS): dK, dV.  P is recomputed in both passes from the forward's row
// statistics (lse_in, one float per query row in the tape).
template <int NKT>
__global__ __launch_bounds__(512) void k_attention_bwd(const f16* __restrict__ qkv, const f16* __restrict__ att,
                                                       const f16* __restrict__ datt, f16* __restrict__ dqkv, int S, Drop d,
                                                       const unsigned char* __restrict__ keep, const float* __restrict__ lse_in,
                                                       int split) {
    // split = 1 (small batches): grid.y = 2 NKT; workgroup y < NKT runs pass 1 for query tile y, workgroup NKT + j runs pass 2
    // for key tile j, each on wave 0 with the row statistics of the forward (lse_in) instead of pass 1's -- 14 workgroups per
    // (clip, head) instead of one whose waves walk both passes back to back.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KEYS = NKT * 32;
    char* img0 = smem;
    char* img1 = smem + KEYS * 256;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * KEYS * 256);
    float* dq_s = lse_s + KEYS;
    float* kbias = dq_s + KEYS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, l31 = lane & 31;
    const int clip = blockIdx.x / MST_H, head = blockIdx.x % MST_H, ch = clip * MST_H + head;
    const f16* base = qkv + (size_t)clip * S * (3 * MST_D) + head * MST_HD;
    const f16* obase = att + (size_t)clip * S * MST_D + head * MST_HD;
    const f16* dobase = datt + (size_t)clip * S * MST_D + head * MST_HD;
    f16* dbase = dqkv + (size_t)clip * S * (3 * MST_D) + head * MST_HD;
    const float scale = 0.08838834764831845f;

    const bool role_q = !split || (int)blockIdx.y < NKT, role_k = !split || (int)blockIdx.y >= NKT;   // block-uniform
    const int tile = split ? (int)blockIdx.y % NKT : wave;
    const bool act = split ? wave == 0 : wave < NKT;
    stage_key_bias(kbias, keep, clip, S, KEYS, tid);
    // ---------------- pass 1: K -> img0, V -> img1
    if (role_q) {
    stage_image(img0, base + MST_D, 3 * MST_D, S, KEYS, tid);
    stage_image(img1, base + 2 * MST_D, 3 * MST_D, S, KEYS, tid);
    __syncthreads();
    if (act) {
        const int q_idx = tile * 32 + l31, q_ld = q_idx < S ? q_idx : S - 1;
        f16x8 qf[8], dof[8];
        float D = 0.f;
        {
            const f16* qp = base + (size_t)q_ld * (3 * MST_D) + 8 * hh;
            const f16* dp = dobase + (size_t)q_ld * MST_D + 8 * hh;
            const f16* op = obase + (size_t)q_ld * MST_D + 8 * hh;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                qf[s] = *reinterpret_cast<const f16x8*>(qp + s * 16);
                dof[s] = *reinterpret_cast<const f16x8*>(dp + s * 16);
                const f16x8 of = *reinterpret_cast<const f16x8*>(op + s * 16);
#pragma unroll
                for (int j = 0; j < 8; j++) D += (float)dof[s][j] * (float)of[j];
            }
        }
        D += __shfl_xor(D, 32);
        // Row statistics: the forward pass left lse = max + log(sum) of every query row in the tape (k_attention_train), so P = exp(s - lse)
        // directly -- no max / sum sweeps over the scores here and, above all, no need to hold all NKT score tiles at once (112 registers at
        // 197 tokens: the kernel spilled 30 registers, 48 scratch instructions; now 8; round 6).  Padded queries: lse = inf -> P = 0.
        // (Requesting the wave's own q / dO / O rows in front of the image staging was tried with it: 26 scratch instructions and +3 us.)
        const float ls = q_idx < S ? lse_in[(size_t)ch * S + q_idx] : INFINITY;
        if (hh == 0) {
            lse_s[q_idx] = ls;
            dq_s[q_idx] = D;
        }
        f16x8 dsf[NKT][2];
#pragma unroll
        for (int kt = 0; kt < NKT; kt++) {
            f32x16 sc, dp;
#pragma unroll
            for (int r = 0; r < 16; r++) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 8; s++) {
                sc = mfma_f16(img_row_frag(img0, kt * 32 + l31, s, hh), qf[s], sc);
                dp = mfma_f16(img_row_frag(img1, kt * 32 + l31, s, hh), dof[s], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int key = kt * 32 + mfma_row(r, lane);
                const float p = __expf(sc[r] * scale + kbias[key] - ls);
                const float mul = drop_mul(d, p_index(ch, S, q_idx, key));
                dsf[kt][r >> 3][r & 7] = (f16)(p * (dp[r] * mul - D) * scale);
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; r++) o[r] = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; kt++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) o = mfma_f16(img_tr_frag_a(img0, kt, s2, dt, lane), dsf[kt][s2], o);
            if (q_idx < S) {
#pragma unroll
                for (int gq = 0; gq < 4; gq++)
                    *reinterpret_cast<uint2*>(dbase + (size_t)q_idx * (3 * MST_D) + dt * 32 + 8 * gq + 4 * hh) =
                        pack4_f16(o[4 * gq], o[4 * gq + 1], o[4 * gq + 2], o[4 * gq + 3]);
            }
        }
    }
    }   // role_q
    if (!role_k) return;
    __syncthreads();

    // ---------------- pass 2: Q -> img0, dO -> img1
    if (split) {                                          // statistics of every query row: lse from the forward, D = sum_d dO O
        for (int q = tid; q < KEYS; q += 512) {
            float D = 0.f, ls = INFINITY;
            if (q < S) {
                ls = lse_in[(size_t)ch * S + q];
                const f16* dp = dobase + (size_t)q * MST_D;
                const f16* op = obase + (size_t)q * MST_D;
                for (int c8 = 0; c8 < 16; c8++) {
                    const f16x8 a8 = *reinterpret_cast<const f16x8*>(dp + c8 * 8), b8 = *reinterpret_cast<const f16x8*>(op + c8 * 8);
#pragma unroll
                    for (int j = 0; j < 8; j++) D += (float)a8[j] * (float)b8[j];
                }
            }
            lse_s[q] = ls;
            dq_s[q] = D;
        }
    }
    stage_image(img0, base, 3 * MST_D, S, KEYS, tid);
    stage_image(img1, dobase, MST_D, S, KEYS, tid);
    __syncthreads();
    if (!act) return;
    {
        const int key_idx = tile * 32 + l31, key_ld = key_idx < S ? key_idx : S - 1;
        const bool key_ok = kbias[key_idx] == 0.f;            // real, un-padded key
        f16x8 kf[8], vf[8];
        {
            const f16* kp = base + (size_t)key_ld * (3 * MST_D) + MST_D + 8 * hh;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                kf[s] = *reinterpret_cast<const f16x8*>(kp + s * 16);
                vf[s] = *reinterpret_cast<const f16x8*>(kp + MST_D + s * 16);
            }
        }
        f32x16 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
#pragma unroll 1
        for (int qt = 0; qt < NKT; qt++) {
            f32x16 s_, dp;
#pragma unroll
            for (int r = 0; r < 16; r++) { s_[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 8; s++) {
                s_ = mfma_f16(img_row_frag(img0, qt * 32 + l31, s, hh), kf[s], s_);
                dp = mfma_f16(img_row_frag(img1, qt * 32 + l31, s, hh), vf[s], dp);
            }
            f16x8 pdf[2], dsf[2];
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const int q0 = qt * 32 + 8 * gq + 4 * hh;
                const f32x4 ls = *reinterpret_cast<const f32x4*>(lse_s + q0), dd = *reinterpret_cast<const f32x4*>(dq_s + q0);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int r = 4 * gq + i;
                    float p = key_ok ? __expf(s_[r] * scale - ls[i]) : 0.f;
                    const float mul = drop_mul(d, p_index(ch, S, q0 + i, key_idx));
                    pdf[r >> 3][r & 7] = (f16)(p * mul);
                    dsf[r >> 3][r & 7] = (f16)(p * (dp[r] * mul - dd[i]) * scale);
                }
            }
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    dv[dt] = mfma_f16(img_tr_frag_a(img1, qt, s2, dt, lane), pdf[s2], dv[dt]);
                    dk[dt] = mfma_f16(img_tr_frag_a(img0, qt, s2, dt, lane), dsf[s2], dk[dt]);
                }
        }
        if (key_idx < S) {
            f16* krow = dbase + (size_t)key_idx * (3 * MST_D) + MST_D;
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int dd = dt * 32 + 8 * gq + 4 * hh;
                    *reinterpret_cast<uint2*>(krow + dd) = pack4_f16(dk[dt][4 * gq], dk[dt][4 * gq + 1], dk[dt][4 * gq + 2], dk[dt][4 * gq + 3]);
                    *reinterpret_cast<uint2*>(krow + MST_D + dd) = pack4_f16(dv[dt][4 * gq], dv[dt][4 * gq + 1], dv[dt][4 * gq + 2], dv[dt][4 * gq + 3]);
                }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// wgrad without transposed copies:  dW[r][c] = sum_tok dY[tok][r] * X[tok][c]  contracts over the ROW index of two
// row-major activations.  32-token slabs of dY[:, r0:r0+128] and X[:, c0:c0+256] stream through the LDS-DMA ring as
// three [32 tokens][128 features] images in the transposed-read layout (v_off); BOTH MFMA operands are then read with
// ds_read_b64_tr_b16, which hands every lane 4 consecutive tokens of its feature -- the k order
// slot(hh, j) <-> token 16 ks + 8 (j >> 2) + 4 hh + (j & 3) is the same on both sides, so any order is a valid
// contraction.  (First version: k_transpose_f16 copies + the ordinary K-contiguous GEMM: 92 us of transposes per layer.)
// Grid (n_out / 128, k_in / 256, splits): split z contracts tokens [z * kchunk, (z + 1) * kchunk) and writes its own
// fp32 partial; rows past M in the last slab are zeroed in LDS (the tape's pad rows are uninitialised memory).
// ------------------------------------------------------------------------------------------------------------
struct WgTile {
    static constexpr int BLK = 32 * 256;                 // one [32][128] f16 image
    static constexpr int STAGE = 3 * BLK;                // dY block + two X blocks
#ifndef MST_WG_NSTAGE
#define MST_WG_NSTAGE 3
#endif
    static constexpr int NSTAGE = MST_WG_NSTAGE;         // slabs in the ring (NSTAGE - 1 in flight per workgroup: 24 KB each)
    static constexpr int SMEM = NSTAGE * STAGE;
    static constexpr int PER = 3;                        // 1-KiB pieces per wave per slab (24 / 8)
};

__global__ __launch_bounds__(512) void k_wgrad_tr(const f16* __restrict__ dY, int n_out, const f16* __restrict__ X, int k_in,
                                                  int M, int kchunk, size_t out_stride, DEpiF32 epi, int xcd_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = WgTile;
    constexpr int BT = 128, BF = 256, MT = 2, NT = 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // xcd_tiles > 0 (a 1-D grid of tiles x splits workgroups, splits a multiple of 8): workgroup L runs on XCD L % 8 (round-robin dispatch),
    // and all tiles of a split are given to ONE XCD -- split z lives on XCD z % 8 -- so that the 32-token slabs of dY and X a split walks
    // are fetched into that XCD's L2 once and shared by its tiles (each slab is otherwise requested by n_out / 128 resp. k_in / 256
    // workgroups scattered over all eight L2s)
    int bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
    if (xcd_tiles > 0) {
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3, gx = n_out / BT;
        const int tile = slot % xcd_tiles;
        z = xcd + 8 * (slot / xcd_tiles);
        bx = tile % gx;
        by = tile / gx;
    }
    const int r0 = bx * BT, c0 = by * BF;
    const int k0 = z * kchunk;
    int ntok = M - k0;
    if (ntok > kchunk) ntok = kchunk;
    const int KT = ntok > 0 ? (ntok + 31) >> 5 : 0;
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // DMA plan: piece pid = 3 wave + j covers tokens 4 q .. 4 q + 3 of image blk (pid = 8 blk + q)
    unsigned voff[kMaxPer];
    bool isy[kMaxPer];
#pragma unroll
    for (int j = 0; j < kMaxPer; j++) {
        voff[j] = 0;
        isy[j] = false;
        if (j < TL::PER) {
            const int pid = wave * TL::PER + j, blk = pid >> 3, q = pid & 7;
            const int tok = 4 * q + (lane >> 4), c = lane & 15, piece = c >> 2, sub = c & 3;
            const int feat = ((piece ^ (tok & 3)) << 5) + (sub << 3);          // swizzle applied on the SOURCE side
            isy[j] = blk == 0;
            const unsigned ld = blk == 0 ? (unsigned)n_out : (unsigned)k_in;
            const unsigned fb = blk == 0 ? (unsigned)r0 : (unsigned)(c0 + 128 * (blk - 1));
            voff[j] = ((unsigned)tok * ld + fb + feat) * 2u + kDmaBias - (unsigned)(j & 3) * 1024u;
        }
    }
    const char* yb = reinterpret_cast<const char*>(dY) + (size_t)k0 * n_out * 2;
    const char* xb = reinterpret_cast<const char*>(X) + (size_t)k0 * k_in * 2;
    auto issue = [&](int kt) {
        const unsigned long long sy = (unsigned long long)(yb + (size_t)kt * 32 * n_out * 2 - kDmaBias);
        const unsigned long long sx = (unsigned long long)(xb + (size_t)kt * 32 * k_in * 2 - kDmaBias);
        unsigned long long sb[kMaxPer];
#pragma unroll
        for (int j = 0; j < kMaxPer; j++) sb[j] = isy[j] ? sy : sx;
        const unsigned base = __builtin_amdgcn_readfirstlane(smem_base + (kt % TL::NSTAGE) * TL::STAGE + wave * TL::PER * 1024);
        glds_group<3>(voff, sb, 0, base);
    };

    f32x16 acc[1][MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[0][m][n][r] = 0.f;
    const int wt = wave >> 2, wn = wave & 3;                  // 2 x 4 waves over the 128 x 256 tile
    constexpr int AHEAD = TL::NSTAGE - 1;
#pragma unroll
    for (int s_ = 0; s_ < AHEAD; s_++)
        if (s_ < KT) issue(s_);
    for (int kt = 0; kt < KT; kt++) {
        const int rem = KT - 1 - kt;
        if (rem >= AHEAD - 1) wait_vmcnt<(AHEAD - 1) * TL::PER>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + AHEAD < KT) issue(kt + AHEAD);
        char* st = smem + (kt % TL::NSTAGE) * TL::STAGE;
        const int lim = ntok - kt * 32;                       // valid tokens in this slab
        if (lim < 32) {                                       // block-uniform: last slab only
            for (int q = tid; q < 3 * 32 * 16; q += 512) {
                const int blk = q >> 9, rem_ = q & 511, row = rem_ >> 4;
                if (row >= lim) *reinterpret_cast<uint4*>(st + blk * TL::BLK + row * 256 + (rem_ & 15) * 16) = uint4{0, 0, 0, 0};
            }
            __syncthreads();
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            f16x8 yf[MT], xf[NT];
#pragma unroll
            for (int m = 0; m < MT; m++) yf[m] = img_tr_frag(st, 0, ks, wt * MT + m, lane);
#pragma unroll
            for (int n = 0; n < NT; n++) {
                const int ft = wn * NT + n;                   // 32-feature tile of the 256 X columns
                xf[n] = img_tr_frag(st + (1 + (ft >> 2)) * TL::BLK, 0, ks, ft & 3, lane);
            }
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int n = 0; n < NT; n++) acc[0][m][n] = mfma_f16(xf[n], yf[m], acc[0][m][n]);
        }
    }
    __builtin_amdgcn_s_barrier();
    epi.out += (size_t)z * out_stride;
    epi.template run<BT, BF, MT, NT>(acc, r0, c0, smem);
}

// colsum[n] += unscale * sum_rows in[row][n]   (bias gradients of linear1 / in_proj); in: [M][N] f16, N % 256 == 0.
// Block = 256 columns x rows_per_block rows: thread (cg, rl) sums 8 columns (16-byte loads) of every 8th row.
__global__ __launch_bounds__(256) void k_colsum_f16(const f16* __restrict__ in, int N, int M, int rows_per_block,
                                                    const float* __restrict__ gscale, float* __restrict__ colsum, float* __restrict__ part) {
    __shared__ float red[8][256];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 256 + cg * 8;
    const int r_lo = blockIdx.y * rows_per_block;
    int r_hi = r_lo + rows_per_block;
    if (r_hi > M) r_hi = M;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = r_lo + rl; r < r_hi; r += 8) {
        const f16x8 v = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(in + (size_t)r * N + c0));
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] += (float)v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) red[rl][cg * 8 + j] = s[j];
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) t += red[i][threadIdx.x];
    // one row block: the only writer of these columns adds straight in; several: partials, summed in order by k_sum_partials
    if (gridDim.y == 1) colsum[blockIdx.x * 256 + threadIdx.x] += t * gscale[1];
    else part[(size_t)blockIdx.y * N + blockIdx.x * 256 + threadIdx.x] = t;
}

// dst[i] += scale * sum_p part[p][i], p in a fixed order (the second stage of the bias-gradient reductions; grid = ceil(n / kFinOut))
__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ part, int nparts, int n, const float* __restrict__ gscale,
                                                      float* __restrict__ dst) {
    __shared__ float red[kFinSlices][kFinOut];
    const int il = threadIdx.x % kFinOut, i = blockIdx.x * kFinOut + il;
    const float s = ordered_partial_sum(part, nparts, n, il, i, red);
    if (threadIdx.x < kFinOut && i < n) dst[i] += gscale ? s * gscale[1] : s;
}


// ------------------------------------------------------------------------------------------------------------
// Fused multi-tensor AdamW + the trainer's norms (train/training_loop.py:196-200 `mp_trainer.optimize(opt)` of the
// reference = fp16_util.py:208-223: 192 per-tensor `.item()` syncs for grad/param norms, then torch AdamW's
// per-tensor kernels).  One launch walks every (tensor, 64 K-element chunk) pair of a device-side table: updates
// p, exp_avg, exp_avg_sq in place with torch.optim.AdamW's arithmetic (decoupled decay first, bias corrections as
// step_size = lr / (1 - b1^t) and denom = sqrt(v) / sqrt(1 - b2^t) + eps) and accumulates sum g^2 and sum p^2
// (parameters BEFORE the update, as the reference logs them) into norms[0..1].  HBM-bound: 28 bytes per parameter.
// ------------------------------------------------------------------------------------------------------------
struct AdamTensor { float* p; const float* g; float* m; float* v; long long numel; };
struct AdamChunk { int tensor; int pad; long long start; };
constexpr int kAdamChunk = 65536;

__global__ __launch_bounds__(256) void k_adamw_multi(const AdamTensor* __restrict__ tensors, const AdamChunk* __restrict__ chunks,
                                                     float lr, float beta1, float beta2, float eps, float wd,
                                                     float bias1, float bias2_sqrt, float* __restrict__ norm_part) {
    const AdamChunk ck = chunks[blockIdx.x];
    const AdamTensor t = tensors[ck.tensor];
    long long end = ck.start + kAdamChunk;
    if (end > t.numel) end = t.numel;
    const float step_size = lr / bias1, decay = 1.0f - lr * wd;
    float sg = 0.f, sp = 0.f;
    for (long long i = ck.start + threadIdx.x; i < end; i += 256) {
        const float g = t.g[i];
        float p = t.p[i];
        sg += g * g;
        sp += p * p;
        p *= decay;
        const float m0 = t.m[i];
        const float m = m0 + (1.0f - beta1) * (g - m0);       // torch: exp_avg.lerp_(grad, 1 - beta1)
        const float v = beta2 * t.v[i] + (1.0f - beta2) * g * g;
        const float denom = sqrtf(v) / bias2_sqrt + eps;
        t.m[i] = m;
        t.v[i] = v;
        t.p[i] = p - step_size * (m / denom);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sg += __shfl_xor(sg, o);
        sp += __shfl_xor(sp, o);
    }
    __shared__ float red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sg; red[1][threadIdx.x >> 6] = sp; }
    __syncthreads();
    if (threadIdx.x == 0 && norm_part) {                     // per-chunk partials; k_sum_partials adds them in chunk order
        norm_part[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        norm_part[gridDim.x + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// norms[0..1] += sum over the chunks of k_adamw_multi's partials, in chunk order
__global__ __launch_bounds__(256) void k_adamw_norms(const float* __restrict__ norm_part, int nchunks, float* __restrict__ norms) {
    __shared__ float red[2][256];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nchunks; i += 256) { a += norm_part[i]; b += norm_part[nchunks + i]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < 2) {
        float s = 0.f;
        for (int i = 0; i < 256; i++) s += red[threadIdx.x][i];
        norms[threadIdx.x] += s;
    }
}

}  // namespace mst
