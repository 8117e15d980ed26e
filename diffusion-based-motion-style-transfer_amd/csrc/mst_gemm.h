// NT GEMM family for the denoiser: out[token][feature] = sum_k X[token][k] * W[feature][k].
//
// Orientation ("feature-major accumulators"): the MFMA A operand is the WEIGHT fragment and the B
// operand the ACTIVATION fragment, so in the 32x32 result each lane owns one token (column) and its
// 16 registers hold features in groups of 4 consecutive ones.  Consequences used throughout:
//   * bias/GELU/LayerNorm epilogues are per-lane loops; LayerNorm row statistics need one
//     cross-half shuffle plus a 4-wave LDS exchange instead of 32-lane butterflies;
//   * row-major [token][feature] stores are 8-byte (f16x4) / 16-byte (f32x4) per lane;
//   * the output projection writes [clip][feature][frame] with frames on lanes -> coalesced,
//     and the whole diffusion update is applied in registers (K10/K12 fused into K9).
//
// Block = 512 threads = 8 waves arranged WT (token groups of 32) x WN (feature groups of NT*32).
//   BT = 64  -> WT 2, WN 4 : block tile 64 tokens x (NT*128) features  (full-row / LayerNorm cfg)
//   BT = 128 -> WT 4, WN 2 : block tile 128 tokens x (NT*64) features  (wide cfg: QKV, FFN1)
// K is consumed in slabs of 64 through a double-buffered, XOR-swizzled LDS image filled from
// registers (global loads for slab k+1 are issued before the MFMAs of slab k and written to LDS
// after them; one barrier per slab).
#pragma once
#include "mst_common.h"

namespace mst {

// ------------------------------------------------------------------------------------------ loaders
// A loader maps (tile row, 16-byte K-chunk) -> 8 f16 of the activation operand.
// map(q, row, c): which (row, chunk) staging slot q handles (lets each loader pick the
// lane order that coalesces ITS global reads).

struct XRows {                       // X[token][k], f16 row-major, rows padded to the tile
    const f16* X; int ld;
    template <int ROWS> __device__ __forceinline__ static void map(int q, int& row, int& c) { row = q >> 3; c = q & 7; }
    __device__ __forceinline__ uint4 load(int tok0, int row, int kc) const {
        return *reinterpret_cast<const uint4*>(X + (size_t)(tok0 + row) * ld + kc * 8);
    }
};

struct XFrames {                     // rows = frames of the token stream (drop token 0 of every clip)
    const f16* X; int ld; int T, S, total; int cfg_rows;   // cfg_rows: row offset of the uncond half
    int BT;
    template <int ROWS> __device__ __forceinline__ static void map(int q, int& row, int& c) { row = q >> 3; c = q & 7; }
    __device__ __forceinline__ uint4 load(int tok0, int row, int kc) const {
        int half = row >= BT ? 1 : 0;
        int tok = tok0 + row - half * BT;
        if (tok >= total) tok = total - 1;
        int clip = tok / T, t = tok - clip * T;
        size_t r = (size_t)clip * S + 1 + t + (size_t)half * cfg_rows;
        return *reinterpret_cast<const uint4*>(X + r * ld + kc * 8);
    }
};

struct XInput {                      // X[token=(clip,t)][k=f] = x[clip][f][t] (float32, T contiguous)
    const float* x; int F, T, total;
    // consecutive staging slots walk consecutive tokens (frames) of one feature chunk: each of the
    // 8 scalar loads below is then a 256-byte coalesced read per wave.
    template <int ROWS> __device__ __forceinline__ static void map(int q, int& row, int& c) { row = q % ROWS; c = q / ROWS; }
    __device__ __forceinline__ uint4 load(int tok0, int row, int kc) const {
        int tok = tok0 + row;
        bool ok = tok < total;
        if (!ok) tok = total - 1;
        int clip = tok / T, t = tok - clip * T;
        const float* p = x + ((size_t)clip * F) * T + t;
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int f = kc * 8 + j;
            v[j] = (ok && f < F) ? (f16)p[(size_t)f * T] : (f16)0.0f;
        }
        return __builtin_bit_cast(uint4, v);
    }
};

// ------------------------------------------------------------------------------------------ mainloop
template <int BT, int NT, int NX>
struct Tile {
    static constexpr int WT = BT / 32;
    static constexpr int WN = 8 / WT;
    static constexpr int BF = WN * NT * 32;           // features per block
    static constexpr int XROWS = NX * BT;
    static constexpr int STAGE = (XROWS + BF) * 128;  // bytes per K-slab
    static constexpr int SMEM = 2 * STAGE;
    static constexpr int XCH = XROWS * 8, WCH = BF * 8;
    static constexpr int XPT = (XCH + 511) / 512, WPT = (WCH + 511) / 512;
};

// staging helpers (free functions on reference-to-array so the arrays stay in VGPRs)
template <class TL, class XL>
__device__ __forceinline__ void stage_gload(uint4 (&xr)[TL::XPT], uint4 (&wr)[TL::WPT], const XL& xl,
                                            const f16* __restrict__ W, int ldw, int tok0, int f0, int kt, int tid) {
#pragma unroll
    for (int p = 0; p < TL::XPT; p++) {
        int q = tid + p * 512;
        if (TL::XCH % 512 == 0 || q < TL::XCH) {
            int row, c;
            XL::template map<TL::XROWS>(q, row, c);
            xr[p] = xl.load(tok0, row, kt * 8 + c);
        }
    }
#pragma unroll
    for (int p = 0; p < TL::WPT; p++) {
        int q = tid + p * 512;
        if (TL::WCH % 512 == 0 || q < TL::WCH) {
            int row = q >> 3, c = q & 7;
            wr[p] = *reinterpret_cast<const uint4*>(W + (size_t)(f0 + row) * ldw + (kt * 8 + c) * 8);
        }
    }
}

template <class TL, class XL>
__device__ __forceinline__ void stage_sstore(const uint4 (&xr)[TL::XPT], const uint4 (&wr)[TL::WPT], char* xs, int tid) {
    char* ws = xs + TL::XROWS * 128;
#pragma unroll
    for (int p = 0; p < TL::XPT; p++) {
        int q = tid + p * 512;
        if (TL::XCH % 512 == 0 || q < TL::XCH) {
            int row, c;
            XL::template map<TL::XROWS>(q, row, c);
            *reinterpret_cast<uint4*>(xs + slab_off(row, c)) = xr[p];
        }
    }
#pragma unroll
    for (int p = 0; p < TL::WPT; p++) {
        int q = tid + p * 512;
        if (TL::WCH % 512 == 0 || q < TL::WCH) {
            int row = q >> 3, c = q & 7;
            *reinterpret_cast<uint4*>(ws + slab_off(row, c)) = wr[p];
        }
    }
}

template <int BT, int NT, int NX, class XL>
__device__ __forceinline__ void gemm_mainloop(char* smem, const XL& xl, const f16* __restrict__ W, int ldw,
                                              int tok0, int f0, int K, f32x16 (&acc)[NX][NT]) {
    using TL = Tile<BT, NT, NX>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave / TL::WN, wn = wave % TL::WN;
    uint4 xr[TL::XPT], wr[TL::WPT];
    const int KT = K >> 6;
    stage_gload<TL, XL>(xr, wr, xl, W, ldw, tok0, f0, 0, tid);
    stage_sstore<TL, XL>(xr, wr, smem, tid);
    __syncthreads();
    for (int kt = 0; kt < KT; kt++) {
        const int cur = kt & 1;
        const bool more = kt + 1 < KT;
        // slab kt+1: global -> registers now, registers -> LDS after this slab's MFMAs.
        // (the last iteration re-reads slab kt: harmless, keeps the loop body branch-free)
        stage_gload<TL, XL>(xr, wr, xl, W, ldw, tok0, f0, more ? kt + 1 : kt, tid);
        const char* xs = smem + cur * TL::STAGE;
        const char* ws = xs + TL::XROWS * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int c = ks * 2 + (lane >> 5);
            f16x8 xf[NX];
#pragma unroll
            for (int x = 0; x < NX; x++) {
                int row = x * BT + wt * 32 + (lane & 31);
                xf[x] = *reinterpret_cast<const f16x8*>(xs + slab_off(row, c));
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                int row = (wn * NT + nt) * 32 + (lane & 31);
                f16x8 wf = *reinterpret_cast<const f16x8*>(ws + slab_off(row, c));
#pragma unroll
                for (int x = 0; x < NX; x++) acc[x][nt] = mfma_f16(wf, xf[x], acc[x][nt]);
            }
        }
        if (more) stage_sstore<TL, XL>(xr, wr, smem + (cur ^ 1) * TL::STAGE, tid);
        __syncthreads();
    }
}

// per-lane coordinates inside a block tile
template <int BT, int NT>
struct LaneCoord {
    int tok_local;    // token of this lane within the block tile
    int fbase;        // first feature of this wave within the block tile
    int hh;           // lane >> 5
    int wn;
    __device__ __forceinline__ LaneCoord() {
        constexpr int WN = 8 / (BT / 32);
        int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int wt = wave / WN;
        wn = wave % WN;
        tok_local = wt * 32 + (lane & 31);
        hh = lane >> 5;
        fbase = wn * NT * 32;
    }
    // first of the 4 consecutive features held in registers 4*g .. 4*g+3 of tile nt
    __device__ __forceinline__ int feat(int nt, int g) const { return fbase + nt * 32 + 8 * g + 4 * hh; }
};

// ------------------------------------------------------------------------------------------ epilogues
// K4 (QKV) and K7 (FFN1 + exact GELU): + bias -> f16 row-major
struct EpiBiasF16 {
    const float* bias; f16* out; int ldo; int M; int gelu;
    template <int BT, int NT, int NX>
    __device__ __forceinline__ void run(f32x16 (&acc)[NX][NT], int tok0, int f0, char*) const {
        LaneCoord<BT, NT> lc;
        int tok = tok0 + lc.tok_local;
        if (tok >= M) return;
        f16* orow = out + (size_t)tok * ldo + f0;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                int f = lc.feat(nt, g);
                f32x4 b = *reinterpret_cast<const f32x4*>(bias + f0 + f);
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    v[i] = acc[0][nt][4 * g + i] + b[i];
                    if (gelu) v[i] = gelu_erf(v[i]);
                }
                *reinterpret_cast<uint2*>(orow + f) = pack4_f16(v[0], v[1], v[2], v[3]);
            }
    }
};

// K6 / K8: + bias + residual -> LayerNorm(eps 1e-5) -> fp32 stream and f16 operand copy.
// Needs the full 512-feature row in the block: BT = 64, NT = 4 (WN = 4 waves share a token).
struct EpiResidLN {
    const float* bias; const float* res; const float* gamma; const float* beta;
    float* out32; f16* out16; int M;
    template <int BT, int NT, int NX>
    __device__ __forceinline__ void run(f32x16 (&acc)[NX][NT], int tok0, int f0, char* smem) const {
        static_assert(BT == 64 && NT == 4, "LayerNorm epilogue needs the 64 x 512 tile");
        LaneCoord<BT, NT> lc;
        const int tok = tok0 + lc.tok_local;
        const bool ok = tok < M;
        const size_t rowoff = (size_t)(ok ? tok : 0) * MST_D;
        float* red = reinterpret_cast<float*>(smem);        // [2][4][64]
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                int f = lc.feat(nt, g);
                f32x4 b = *reinterpret_cast<const f32x4*>(bias + f);
                f32x4 r = *reinterpret_cast<const f32x4*>(res + rowoff + f);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float v = acc[0][nt][4 * g + i] + b[i] + r[i];
                    acc[0][nt][4 * g + i] = v;
                    s += v;
                }
            }
        s += __shfl_xor(s, 32);
        if (lc.hh == 0) red[lc.wn * 64 + lc.tok_local] = s;
        __syncthreads();
        float mean = (red[lc.tok_local] + red[64 + lc.tok_local] + red[128 + lc.tok_local] + red[192 + lc.tok_local]) * (1.0f / MST_D);
        float s2 = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float d = acc[0][nt][r] - mean;
                acc[0][nt][r] = d;
                s2 += d * d;
            }
        s2 += __shfl_xor(s2, 32);
        if (lc.hh == 0) red[256 + lc.wn * 64 + lc.tok_local] = s2;
        __syncthreads();
        float var = (red[256 + lc.tok_local] + red[320 + lc.tok_local] + red[384 + lc.tok_local] + red[448 + lc.tok_local]) * (1.0f / MST_D);
        float rstd = 1.0f / sqrtf(var + 1e-5f);
        if (!ok) return;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                int f = lc.feat(nt, g);
                f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + f);
                f32x4 be = *reinterpret_cast<const f32x4*>(beta + f);
                f32x4 y;
#pragma unroll
                for (int i = 0; i < 4; i++) y[i] = acc[0][nt][4 * g + i] * rstd * ga[i] + be[i];
                *reinterpret_cast<f32x4*>(out32 + rowoff + f) = y;
                *reinterpret_cast<uint2*>(out16 + rowoff + f) = pack4_f16(y[0], y[1], y[2], y[3]);
            }
    }
};

// K3: + bias + positional row (frames sit at positions 1..T) -> token stream rows clip*S + 1 + t
struct EpiEmbedIn {
    const float* bias; const float* pe; float* out32; f16* out16; int T, S, total;
    template <int BT, int NT, int NX>
    __device__ __forceinline__ void run(f32x16 (&acc)[NX][NT], int tok0, int f0, char*) const {
        LaneCoord<BT, NT> lc;
        int tok = tok0 + lc.tok_local;
        if (tok >= total) return;
        int clip = tok / T, t = tok - clip * T;
        size_t rowoff = ((size_t)clip * S + 1 + t) * MST_D;
        const float* perow = pe + (size_t)(t + 1) * MST_D;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                int f = f0 + lc.feat(nt, g);
                f32x4 b = *reinterpret_cast<const f32x4*>(bias + f);
                f32x4 p = *reinterpret_cast<const f32x4*>(perow + f);
                f32x4 y;
#pragma unroll
                for (int i = 0; i < 4; i++) y[i] = acc[0][nt][4 * g + i] + b[i] + p[i];
                *reinterpret_cast<f32x4*>(out32 + rowoff + f) = y;
                *reinterpret_cast<uint2*>(out16 + rowoff + f) = pack4_f16(y[0], y[1], y[2], y[3]);
            }
    }
};

// K9 + K10/K10'/K11/K12: output projection with the diffusion update applied in registers.
// MODE 0: write the model output (after the CFG blend) only.
// MODE 1: ancestral step (p_sample), MODE 2: DDIM step.
struct StepArgs {
    const float* tab; int nsteps; int t;          // schedule tables (device) and the diffusion index
    float eta;
    const float* mask; const float* motion;       // [B,F,1,T] or null
    const float* noise;                           // [B,F,1,T] or null (-> philox)
    const float* scale;                           // [B] guidance scale (cfg)
    const float* x;                               // x_t
    float* sample; float* xstart;                 // outputs (xstart may be null)
    unsigned long long seed; unsigned step;
    int mask_noise, clip, philox;
};

template <int MODE>
struct EpiEmbedOut {
    const float* bias; int F, T, total;           // total = clips * T
    float* out;                                   // MODE 0 destination
    StepArgs sa;
    template <int BT, int NT, int NX>
    __device__ __forceinline__ void run(f32x16 (&acc)[NX][NT], int tok0, int f0, char*) const {
        LaneCoord<BT, NT> lc;
        int tok = tok0 + lc.tok_local;
        if (tok >= total) return;
        int clip = tok / T, t = tok - clip * T;
        float gs = 0.f;
        if (NX == 2) gs = sa.scale[clip];
        StepCoef sc;
        if (MODE != 0) sc = step_coef(sa.tab, sa.nsteps, sa.t, sa.eta);
        const bool blend = sa.mask != nullptr && sa.motion != nullptr;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                int f = f0 + lc.feat(nt, g);
                if (f >= F) continue;
                float nrm[4] = {0.f, 0.f, 0.f, 0.f};
                if (MODE != 0 && sa.philox) philox_normal4((unsigned)t, (unsigned)(f >> 2), (unsigned)clip, sa.step, sa.seed, nrm);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    int fi = f + i;
                    if (fi >= F) break;
                    float b = bias[fi];
                    float mo = acc[0][nt][4 * g + i] + b;
                    if (NX == 2) {                       // model/cfg_sampler.py:43
                        float un = acc[NX - 1][nt][4 * g + i] + b;
                        mo = un + gs * (mo - un);
                    }
                    size_t idx = ((size_t)clip * F + fi) * T + t;
                    if (MODE == 0) { out[idx] = mo; continue; }
                    float m = 0.f, mot = 0.f;
                    if (sa.mask) m = sa.mask[idx];
                    if (blend) mot = sa.motion[idx];
                    float nz = sa.philox ? nrm[i] : (sa.noise ? sa.noise[idx] : 0.f);
                    float pred;
                    float nx = step_update<MODE == 2 ? 1 : 0>(sc, mo, sa.x[idx], nz, blend, m, mot,
                                                             sa.mask_noise && sa.mask, sa.clip, &pred);
                    sa.sample[idx] = nx;
                    if (sa.xstart) sa.xstart[idx] = pred;
                }
            }
    }
};

// ------------------------------------------------------------------------------------------ kernel
template <int BT, int NT, int NX, class XL, class EPI>
__global__ __launch_bounds__(512) void k_gemm(XL xl, const f16* __restrict__ W, int ldw, int K, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = Tile<BT, NT, NX>;
    const int tok0 = blockIdx.x * BT;
    const int f0 = blockIdx.y * TL::BF;
    f32x16 acc[NX][NT];
#pragma unroll
    for (int x = 0; x < NX; x++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[x][nt][r] = 0.f;
    gemm_mainloop<BT, NT, NX, XL>(smem, xl, W, ldw, tok0, f0, K, acc);
    epi.template run<BT, NT, NX>(acc, tok0, f0, smem);
}

}  // namespace mst
