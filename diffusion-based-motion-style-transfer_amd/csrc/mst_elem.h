// Bandwidth-bound and latency-bound helpers: weight conversion, the conditioning-token path (K1/K2),
// the stand-alone q_sample / step kernels (K10/K11 for callers that bring their own model) and the
// Philox normal fill.
#pragma once
#include "mst_common.h"

namespace mst {

// float32 [N][K] -> f16 [Npad][Kpad], zero padded; dst_lo != null: also the lo half f16(w - f16(w)), the second term of a hi + lo
// WEIGHT operand (precise mode), in the same pass (one launch per matrix: weight re-uploads sit on the fine-tune loop's critical path)
__global__ void k_convert_pad(const float* __restrict__ src, int N, int K, f16* __restrict__ dst, int Npad, int Kpad, f16* __restrict__ dst_lo) {
    size_t total = (size_t)Npad * Kpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int n = (int)(i / Kpad), k = (int)(i - (size_t)n * Kpad);
        const float v = (n < N && k < K) ? src[(size_t)n * K + k] : 0.f;
        const f16 h = (f16)v;
        dst[i] = h;
        if (dst_lo) dst_lo[i] = (f16)(v - (float)h);
    }
}

// The 12 tensors of up to 8 encoder layers in ONE launch (a fine-tune iteration re-uploads all 96 after every optimizer step: 16
// launches per layer one by one, 1.8 ms of launch chain per iteration).  grid = (2048 matrix tiles + 1 vector block, layers).
// A matrix block converts one 32 x 32 fp32 tile to the f16 [out][in] copy (the GEMMs' NT operand) and the [in][out] copy (the dgrad
// operand) -- the same roundings as k_convert_pad / k_convert_transpose; the vector block copies the 8 fp32 vectors.
struct UpLayer { const float* src[12]; f16* w[4]; f16* wT[4]; float* v[8]; };
struct UpArgs { UpLayer L[8]; };
__global__ __launch_bounds__(256) void k_upload_layers(UpArgs a) {
    __shared__ float tile[32][33];
    const UpLayer& L = a.L[blockIdx.y];
    int b = blockIdx.x;
    if (b == 2048) {                                    // vectors: in_proj_bias 1536, out_proj.bias 512, linear1.bias 1024, linear2.bias 512, 4 x 512 LayerNorm
        const int vs[8] = {1, 3, 5, 7, 8, 9, 10, 11}, vn[8] = {3 * MST_D, MST_D, MST_FF, MST_D, MST_D, MST_D, MST_D, MST_D};
        for (int j = 0; j < 8; j++)
            for (int i = threadIdx.x; i < vn[j]; i += 256) L.v[j][i] = L.src[vs[j]][i];
        return;
    }
    // matrix m: 0 in_proj [1536][512], 1 out_proj [512][512], 2 linear1 [1024][512], 3 linear2 [512][1024]; tiles 768 | 256 | 512 | 512
    int m, N, K;
    if (b < 768) { m = 0; N = 3 * MST_D; K = MST_D; }
    else if (b < 1024) { m = 1; b -= 768; N = MST_D; K = MST_D; }
    else if (b < 1536) { m = 2; b -= 1024; N = MST_FF; K = MST_D; }
    else { m = 3; b -= 1536; N = MST_D; K = MST_FF; }
    const int ktiles = K / 32, n0 = (b / ktiles) * 32, k0 = (b % ktiles) * 32;
    const float* src = L.src[2 * m];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        const float v = src[(size_t)(n0 + j) * K + k0 + tx];
        tile[j][tx] = v;
        L.w[m][(size_t)(n0 + j) * K + k0 + tx] = (f16)v;
    }
    __syncthreads();
#pragma unroll
    for (int j = ty; j < 32; j += 8) L.wT[m][(size_t)(k0 + j) * N + n0 + tx] = (f16)tile[tx][j];
}

__global__ void k_copy_pad_f32(const float* __restrict__ src, int n, float* __restrict__ dst, int npad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) dst[i] = i < n ? src[i] : 0.f;
}

// x [clips][F][T] float32 -> xt [clips*T][Kpad] f16 (frames as rows, features contiguous, zero padded):
// the activation operand of the pose-embedding GEMM in the layout the LDS-DMA ring loads.
// gscale != null: values are multiplied by gscale[0] first (backward of the output projection: the incoming gradient
// is brought into f16 range by the device-side scale of the training path).
// xt_lo != null: also f16(x - hi), the second half of a hi + lo operand (RowsDirect::Xlo).
__global__ __launch_bounds__(256) void k_frames_f16(const float* __restrict__ x, int F, int T, int Kpad, f16* __restrict__ xt,
                                                    const float* __restrict__ gscale, const LoopDev* __restrict__ ld, unsigned long long eo,
                                                    f16* __restrict__ xt_lo = nullptr) {
    __shared__ float tile[32][33];
    if (ld) x = ld->x + eo;                          // sampling loop: the clip tensor of THIS call (captured graphs are replayed across calls)
    const float sc = gscale ? gscale[0] : 1.0f;
    const int clip = blockIdx.z, f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        int f = f0 + j, t = t0 + tx;
        tile[j][tx] = (f < F && t < T) ? x[((size_t)clip * F + f) * T + t] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        int t = t0 + j, f = f0 + tx;
        if (t < T && f < Kpad) {
            const float v = tile[tx][j] * sc;
            const f16 h = (f16)v;
            xt[((size_t)clip * T + t) * Kpad + f] = h;
            if (xt_lo) xt_lo[((size_t)clip * T + t) * Kpad + f] = (f16)(v - (float)h);
        }
    }
}

// K1/K2: y[row][n] = act(sum_k in[row][k] * rowscale[row] * W[n][k] + b[n]) in float32.
// One wave per output element group: lanes stride K (coalesced W rows), shuffle reduce.
// gather != null: input row = table[gather[row]] (timestep embedding: pe[t]).
// act: 0 none, 1 SiLU.   rows_zero_from: rows >= that index use a zero input (uncond half).
__global__ __launch_bounds__(256) void k_rowwise_linear(const float* __restrict__ in, int ldin,
                                                        const long long* __restrict__ gather,
                                                        const float* __restrict__ rowscale, int rows_zero_from,
                                                        const float* __restrict__ W, const float* __restrict__ b,
                                                        int K, int N, int act, float* __restrict__ out, int in_row_mod, int rowscale_is_drop = 0) {
    const int row = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int src_row = in_row_mod > 0 ? row % in_row_mod : row;
    const float* x = in + (size_t)(gather ? gather[src_row] : src_row) * ldin;
    float rs = 1.0f;
    if (rowscale) rs = rowscale_is_drop ? 1.0f - rowscale[src_row] : rowscale[src_row];       // (a Bernoulli DROP mask: cond * (1 - mask), mdm :288-296)
    if (row >= rows_zero_from) rs = 0.0f;
    for (int n = blockIdx.y * 4 + wave; n < N; n += gridDim.y * 4) {
        const float* w = W + (size_t)n * K;
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += (x[k] * rs) * w[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            float v = s + b[n];
            if (act == 1) v = v / (1.0f + __expf(-v));
            out[(size_t)row * N + n] = v;
        }
    }
}

// token 0 of every clip: h[clip*S][:] = temb[row] + textproj[clip] + pe[0]  (mdm :609-621)
// (The conditioning token -- temb row + text projection + positional row 0 -> token 0 of every clip -- is written by the
// pose-embedding GEMM's epilogue: CondTok / DEpiEmbedIn in mst_gemm_dma.h.  uniform_row >= 0: every clip uses that temb row
// (sampling loop: one t per step); < 0: clip uses temb row (clip % temb_mod).  tp_half > 0 (CFG slice): rows [0, tp_half) are the
// slice's cond clips, rows [tp_half, 2 tp_half) their uncond twins, whose text projections sit tp_uncond rows further on.)

// Small launches: y = LayerNorm(acc + bias + residual) for rows of 512, one wave per row; the stream (hi/lo pair) is
// read as the residual and rewritten in place.  `acc` is the fp32 GEMM result of a launch tiled over N (k_gemm_dma with
// 64 x 128 tiles + DEpiPlainF32) -- the fused whole-row epilogue would leave all but 4 CUs idle at one clip.
__global__ __launch_bounds__(256) void k_ln_rows(const float* __restrict__ acc, const float* __restrict__ bias,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 f16* hi, f16* lo, int M, f16* ohi = nullptr, f16* olo = nullptr) {
    // (ohi, olo): the normalised rows go there instead of over the residual (the last LayerNorm of a stack whose other LayerNorms ran
    // inside the GEMM behind them, mst_small.h: its residual sits in the second stream buffer, its consumer reads the first)
    if (!ohi) { ohi = hi; olo = lo; }
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int fa = lane * 4, fb = 256 + lane * 4;
    const size_t off = (size_t)row * MST_D;
    f32x4 xa = join4_f16(*reinterpret_cast<const uint2*>(hi + off + fa), *reinterpret_cast<const uint2*>(lo + off + fa));
    f32x4 xb = join4_f16(*reinterpret_cast<const uint2*>(hi + off + fb), *reinterpret_cast<const uint2*>(lo + off + fb));
    const f32x4 ta = *reinterpret_cast<const f32x4*>(acc + off + fa), tb = *reinterpret_cast<const f32x4*>(acc + off + fb);
    const f32x4 ba = *reinterpret_cast<const f32x4*>(bias + fa), bb = *reinterpret_cast<const f32x4*>(bias + fb);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        xa[i] = ta[i] + ba[i] + xa[i];
        xb[i] = tb[i] + bb[i] + xb[i];
    }
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + fa), gb = *reinterpret_cast<const f32x4*>(gamma + fb);
    const f32x4 ea = *reinterpret_cast<const f32x4*>(beta + fa), eb = *reinterpret_cast<const f32x4*>(beta + fb);
    ln_row_wave(xa, xb, ga, gb, ea, eb);
    const f32x4 ya = xa, yb = xb;
    uint2 h, l;
    split4_f16(ya, h, l);
    *reinterpret_cast<uint2*>(ohi + off + fa) = h;
    *reinterpret_cast<uint2*>(olo + off + fa) = l;
    split4_f16(yb, h, l);
    *reinterpret_cast<uint2*>(ohi + off + fb) = h;
    *reinterpret_cast<uint2*>(olo + off + fb) = l;
}

// debug / tests: the stream as float32
__global__ void k_join_stream(const f16* __restrict__ hi, const f16* __restrict__ lo, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (float)hi[i] + (float)lo[i];
}

// K11 stand-alone: x_t = sqrt(abar_t) x0 + sqrt(1-abar_t) (noise * (1 - mask))
__global__ void k_q_sample(const float* __restrict__ tab, int nsteps, const float* __restrict__ x0,
                           const float* __restrict__ noise, const float* __restrict__ mask,
                           const long long* __restrict__ t, long long per_clip, float* __restrict__ out) {
    const int clip = blockIdx.y;
    const int tt = (int)t[clip];
    const float a = tab[TAB_SQRT_AC * nsteps + tt], b = tab[TAB_SQRT_1M_AC * nsteps + tt];
    const size_t base = (size_t)clip * per_clip;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_clip; i += (long long)gridDim.x * blockDim.x) {
        float n = noise[base + i];
        if (mask) n = n * (1.0f - mask[base + i]);
        out[base + i] = a * x0[base + i] + b * n;
    }
}

// K10 / K10' stand-alone (model output produced elsewhere)
template <int SAMPLER, int MEAN = 0>
__global__ void k_step_epilogue(const float* __restrict__ tab, int nsteps, float eta,
                                const float* __restrict__ model_out, const float* __restrict__ x,
                                const float* __restrict__ noise, const float* __restrict__ mask,
                                const float* __restrict__ motion, const long long* __restrict__ t,
                                long long per_clip, int mask_noise, int clip_denoised,
                                float* __restrict__ sample, float* __restrict__ xstart) {
    const int clip = blockIdx.y;
    const StepCoef sc = step_coef(tab, nsteps, (int)t[clip], eta);
    const bool blend = mask != nullptr && motion != nullptr;
    const size_t base = (size_t)clip * per_clip;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_clip; i += (long long)gridDim.x * blockDim.x) {
        size_t idx = base + i;
        float m = mask ? mask[idx] : 0.f;
        float mot = blend ? motion[idx] : 0.f;
        float pred;
        float nx = step_update<SAMPLER, MEAN>(sc, model_out[idx], x[idx], noise ? noise[idx] : 0.f, blend, m, mot,
                                              mask_noise && mask, clip_denoised, &pred);
        if (sample) sample[idx] = nx;
        if (xstart) xstart[idx] = pred;
    }
}

// Backward of the fused step for the `*_with_grad` samplers (inpainting_gaussian_diffusion.py:66-123, :179-239): both
// outputs are affine in the model output,
//     pred   = out (1 - mask) + motion mask
//     sample = c1 pred + c2 x + sigma noise                                   (ancestral, gaussian_diffusion.py:404-412)
//     sample = sqrt(abar_prev) pred + dir (srac x - pred) / srm1ac + ...      (DDIM, :157-177)
// so d out = (g_pred + g_sample d sample/d pred) (1 - mask).  One launch instead of autograd's chain of elementwise nodes.
template <int SAMPLER>
__global__ void k_step_backward(const float* __restrict__ tab, int nsteps, float eta, const float* __restrict__ g_sample,
                                const float* __restrict__ g_pred, const float* __restrict__ mask, int has_blend,
                                const long long* __restrict__ t, long long per_clip, const float* __restrict__ pred_clipped,
                                float* __restrict__ d_out) {
    const int clip = blockIdx.y;
    const StepCoef sc = step_coef(tab, nsteps, (int)t[clip], eta);
    const float dsdp = SAMPLER == 0 ? sc.c1 : sc.sq_abp - sc.dir / sc.srm1ac;
    const size_t base = (size_t)clip * per_clip;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_clip; i += (long long)gridDim.x * blockDim.x) {
        const size_t idx = base + i;
        float g = g_pred ? g_pred[idx] : 0.f;
        if (g_sample) g += g_sample[idx] * dsdp;
        if (has_blend) g *= 1.0f - mask[idx];
        // clip_denoised (gaussian_diffusion.py:389-395: x.clamp(-1, 1) on the blended prediction, the reference signature's default):
        // clamp passes the gradient only where it did not saturate.  `pred_clipped` is the forward's x0-hat; a value that came out
        // at exactly +-1 is taken as saturated (an unclamped prediction of exactly +-1.0f has measure zero).
        if (pred_clipped && !(fabsf(pred_clipped[idx]) < 1.0f)) g = 0.f;
        d_out[idx] = g;
    }
}

// K13a: masked_l2 (gaussian_diffusion.py:223-235): loss[n] = sum_{f,t} (a - b)^2 mask[n, t] / (sum_t mask[n, t] * F), one
// workgroup per sample; a / mask may be broadcast over n (stride 0: `x_style_start.expand(num_step, ...)`, :1380).
// Round 6: 1024 threads per sample, eight independent element streams per thread (256 threads walked ~200 dependent loads each: 80 us for
// six clips, behind the chained steps on the fine-tune iteration's critical path); the 16 waves' sums meet in a fixed order.
__global__ __launch_bounds__(1024) void k_masked_l2_fwd(const float* __restrict__ a, long long a_stride, const float* __restrict__ b,
                                                        const float* __restrict__ mask, long long m_stride, int F, int T,
                                                        float* __restrict__ loss) {
    __shared__ float red[2][16];
    const int n = blockIdx.x;
    const float* an = a + (size_t)n * a_stride;
    const float* bn = b + (size_t)n * F * T;
    const float* mn = mask + (size_t)n * m_stride;
    const int total = F * T;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = threadIdx.x;
    for (; i + 7 * 1024 < total; i += 8 * 1024) {
        float av[8], bv[8], mv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int j = i + u * 1024; av[u] = an[j]; bv[u] = bn[j]; mv[u] = mn[j % T]; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const float d = av[u] - bv[u]; acc[u] += d * d * mv[u]; }
    }
    for (int u = 0; i < total; i += 1024, u++) { const float d = an[i] - bn[i]; acc[u & 7] += d * d * mn[i % T]; }
    float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])), ms = 0.f;
    for (int tt = threadIdx.x; tt < T; tt += 1024) ms += mn[tt];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ms += __shfl_xor(ms, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ms; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float st = 0.f, mt = 0.f;
        for (int w = 0; w < 16; w++) { st += red[0][w]; mt += red[1][w]; }      // fixed order: run-to-run deterministic
        loss[n] = st / (mt * (float)F);
    }
}
// d b[n, f, t] = -2 (a - b) mask[t] / (sum mask * F) * g[n]   (d a is its negative)
__global__ __launch_bounds__(256) void k_masked_l2_bwd(const float* __restrict__ a, long long a_stride, const float* __restrict__ b,
                                                       const float* __restrict__ mask, long long m_stride, int F, int T,
                                                       const float* __restrict__ g, float* __restrict__ d_b) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    const float* mn = mask + (size_t)n * m_stride;
    float ms = 0.f;
    for (int tt = threadIdx.x; tt < T; tt += 256) ms += mn[tt];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ms += __shfl_xor(ms, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ms;
    __syncthreads();
    const float k = -2.0f * g[n] / (((red[0] + red[1]) + (red[2] + red[3])) * (float)F);
    const float* an = a + (size_t)n * a_stride;
    const float* bn = b + (size_t)n * F * T;
    float* dn = d_b + (size_t)n * F * T;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < F * T; i += gridDim.x * 256) dn[i] = k * (an[i] - bn[i]) * mn[i % T];
}

// K13b: text_cosine (gaussian_diffusion.py:1384-1388): mean_b (1 - cos(f_b / |f_b|, m_b / |m_b|)) with torch's
// cosine_similarity(eps = 1e-6) on the already normalised rows.  One wave per row, one workgroup in all (B <= a few hundred
// rows of 512: latency, not bandwidth); rows are summed in index order.  mode 0: loss[0]; mode 1: d m (g = dL/dloss).
__global__ __launch_bounds__(1024) void k_text_cosine(const float* __restrict__ f, const float* __restrict__ m, int B, int D,
                                                      int mode, const float* __restrict__ g, float* __restrict__ out) {
    // (round 6: 16 waves instead of 4 and the rows' terms summed by one wave -- 54 us for 64 rows on the iteration's critical path before)
    __shared__ float rowv[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = wave; b < B; b += 16) {
        const float* fb = f + (size_t)b * D;
        const float* mb = m + (size_t)b * D;
        float ff = 0.f, mm = 0.f, fm = 0.f;
        for (int i = lane; i < D; i += 64) { ff += fb[i] * fb[i]; mm += mb[i] * mb[i]; fm += fb[i] * mb[i]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ff += __shfl_xor(ff, o); mm += __shfl_xor(mm, o); fm += __shfl_xor(fm, o); }
        const float nf = sqrtf(ff), nm = sqrtf(mm);
        // cosine of the unit rows f / nf and m / nm; cosine_similarity's own eps clamp acts on norms that are 1 here
        const float c = fm / (nf * nm);
        if (mode == 0) {
            if (lane == 0 && b < 1024) rowv[b] = 1.0f - c;
        } else {
            // d(1 - c)/d m = -(fh / nm - c m / nm^2) with fh = f / nf; mean over B and the upstream gradient g[0]
            const float k = -g[0] / (float)B;
            for (int i = lane; i < D; i += 64) out[(size_t)b * D + i] = k * (fb[i] / (nf * nm) - c * mb[i] / (nm * nm));
        }
    }
    if (mode == 0) {
        __syncthreads();
        if (wave == 0) {                                   // rows lane, lane + 64, ... in index order per lane, then a fixed shuffle tree
            float s = 0.f;
            for (int b = lane; b < B && b < 1024; b += 64) s += rowv[b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) out[0] = s / (float)B;
        }
    }
}

// same counter mapping as the fused epilogue: element (clip, f, t) <- component t & 3 of counter (t >> 2, f, clip, step)
__global__ void k_philox_normal(float* __restrict__ out, int F, int T, unsigned long long seed, unsigned step) {
    const int clip = blockIdx.y;
    const int TQ = (T + 3) >> 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < F * TQ; i += gridDim.x * blockDim.x) {
        int f = i / TQ, tq = i - f * TQ;
        float n[4];
        philox_normal4((unsigned)tq, (unsigned)f, (unsigned)clip, step, seed, n);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int t = tq * 4 + j;
            if (t < T) out[((size_t)clip * F + f) * T + t] = n[j];
        }
    }
}


__global__ void k_loop_advance(LoopDev* ld, int n) { ld->jbase += n; }

// MotionEncoder.forward (mdm_forstyledataset.py:104-110): tokens 0 / 1 of every clip = muQuery + pe[0] / sigmaQuery + pe[1]
__global__ void k_query_tokens(const float* __restrict__ muq, const float* __restrict__ sigq, const float* __restrict__ pe, int S, int rows,
                               f16* __restrict__ hi, f16* __restrict__ lo) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 2 * MST_D) return;
    const int clip = i / (2 * MST_D), r = i - clip * 2 * MST_D, tok = r / MST_D, f = r - tok * MST_D;
    const float v = (tok == 0 ? muq[f] : sigq[f]) + pe[tok * MST_D + f];
    const size_t o = ((size_t)clip * S + tok) * MST_D + f;
    const f16 h = (f16)v;
    hi[o] = h;
    lo[o] = (f16)(v - (float)h);
}
// token 0 of every clip of an f16 hi/lo stream -> fp32 [rows][512]   (`final[0]`, :122)
__global__ void k_gather_token0(const f16* __restrict__ hi, const f16* __restrict__ lo, int S, int rows, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * MST_D) return;
    const size_t o = (size_t)(i / MST_D) * S * MST_D + (i % MST_D);
    out[i] = (float)hi[o] + (float)lo[o];
}
// backward seed: g[clip][0][:] = scale[0] * d_mu[clip][:] (g zeroed by the caller)
__global__ void k_scatter_token0(const float* __restrict__ d, const float* __restrict__ scale, int S, int rows, float* __restrict__ g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * MST_D) return;
    g[(size_t)(i / MST_D) * S * MST_D + (i % MST_D)] = d[i] * scale[0];
}

// the same seed with the zero fill inside (round 6: the 26 MB hipMemsetAsync in front of it was a launch of its own on the backward pass's
// dependent chain): g[clip][s][:] = s == 0 ? scale[0] * d_mu[clip][:] : 0, one float4 per thread
__global__ void k_seed_token0(const float* __restrict__ d, const float* __restrict__ scale, int S, int rows, float* __restrict__ g) {
    const size_t n4 = (size_t)rows * S * (MST_D / 4);
    const float sc = scale[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / (MST_D / 4);
        const int c4 = (int)(i % (MST_D / 4));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row % S == 0) {
            const f32x4 dv = *reinterpret_cast<const f32x4*>(d + (row / S) * MST_D + c4 * 4);
            v = f32x4{dv[0] * sc, dv[1] * sc, dv[2] * sc, dv[3] * sc};
        }
        reinterpret_cast<f32x4*>(g)[i] = v;
    }
}

// one wave per (clip, feature) row of a [rows][T] 0/1 mask: 0 = all zeros, 1 = all ones, 2 = anything else
__global__ __launch_bounds__(256) void k_mask_rowflags(const float* __restrict__ mask, int rows, int T, unsigned char* __restrict__ flags) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    bool z = true, o = true;
    for (int t = lane; t < T; t += 64) {
        const float v = mask[(size_t)row * T + t];
        z &= v == 0.f;
        o &= v == 1.f;
    }
    const bool az = __all(z), ao = __all(o);
    if (lane == 0) flags[row] = az ? 0 : (ao ? 1 : 2);
}

// Post-sampling: normalised hml_vec clip [B][F][T] -> joint positions [B][T][J][3] in one launch
// (sample.permute(0,2,3,1) * std + mean, then recover_from_ric: root yaw = running sum of the yaw velocities, root
// XZ = running sum of the yaw-rotated XZ velocities, local joints rotated by the same yaw and moved to the root;
// data_loaders/humanml/scripts/motion_process.py:389-410, :444-461, quaternion.py:88-99).  One workgroup per clip:
// the two running sums over <= 224 frames are done sequentially by one lane in torch.cumsum's order (microseconds),
// everything else is one thread per (frame, joint).  The quaternion is (cos a, 0, sin a, 0) with the FULL angle, as
// the reference has it.
__device__ __forceinline__ void yaw_rot(float c, float sn, float vx, float vy, float vz, float& ox, float& oy, float& oz) {
    // qrot with s = c, u = (0, sn, 0):  uv = u x v,  uuv = u x uv,  out = v + 2 (s uv + uuv)
    const float uvx = sn * vz, uvz = -sn * vx;                 // uv = (sn vz, 0, -sn vx)
    const float uuvx = sn * uvz, uuvz = -sn * uvx;             // uuv = (sn uvz, 0, -sn uvx)
    ox = vx + 2.0f * (c * uvx + uuvx);
    oy = vy;
    oz = vz + 2.0f * (c * uvz + uuvz);
}

__global__ __launch_bounds__(256) void k_recover_from_ric(const float* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, int F, int T, int J,
                                                          float* __restrict__ out) {
    extern __shared__ float sm[];            // ang[T], px[T], pz[T], cs[T], sn[T]
    float* ang = sm; float* px = sm + T; float* pz = sm + 2 * T; float* cs = sm + 3 * T; float* sn = sm + 4 * T;
    const int b = blockIdx.x;
    const float* xb = x + (size_t)b * F * T;
    auto den = [&](int f, int t) { return xb[(size_t)f * T + t] * stdv[f] + mean[f]; };
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int t = 0; t < T; t++) {        // exclusive running sum of the yaw velocity
            ang[t] = a;
            a += den(0, t);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
        float s_, c_;
        sincosf(ang[t], &s_, &c_);
        cs[t] = c_;
        sn[t] = s_;
        float ox = 0.f, oy, oz = 0.f;
        if (t > 0) yaw_rot(c_, s_, den(1, t - 1), 0.f, den(2, t - 1), ox, oy, oz);
        px[t] = ox;
        pz[t] = oz;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ax = 0.f, az = 0.f;
        for (int t = 0; t < T; t++) {        // inclusive running sums of the rotated root velocity
            ax += px[t];
            az += pz[t];
            px[t] = ax;
            pz[t] = az;
        }
    }
    __syncthreads();
    float* ob = out + (size_t)b * T * J * 3;
    for (int i = threadIdx.x; i < T * J; i += 256) {
        const int t = i / J, j = i - t * J;
        float ox, oy, oz;
        if (j == 0) {
            ox = px[t];
            oy = den(3, t);
            oz = pz[t];
        } else {
            const int f = 4 + 3 * (j - 1);
            yaw_rot(cs[t], sn[t], den(f, t), den(f + 1, t), den(f + 2, t), ox, oy, oz);
            ox += px[t];
            oz += pz[t];
        }
        ob[(size_t)i * 3 + 0] = ox;
        ob[(size_t)i * 3 + 1] = oy;
        ob[(size_t)i * 3 + 2] = oz;
    }
}

}  // namespace mst
