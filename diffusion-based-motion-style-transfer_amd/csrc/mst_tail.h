// K6 + K7 + K8 fused: everything of an encoder layer behind the attention, one workgroup per 64-token tile:
//
//     x1 = LayerNorm1(x + att W_out^T + b_out)        (self-attention block's tail, mdm_forstyledataset.py:539-546 /
//     h  = GELU(x1 W1^T + b1)                           nn.TransformerEncoderLayer post-norm forward)
//     x2 = LayerNorm2(x1 + h W2^T + b2)
//
// Why one kernel.  As three launches (out-proj+LN, FFN1+GELU, FFN2+LN) a 64-clip step moves per layer ~180 MB through HBM
// that only exists to hand a tile from one launch to the next (`hid` written and read back: 51.6 MB; x1 written as the
// stream and read back twice), and every launch pays its own ring prologue and an epilogue in which all workgroups hit
// HBM at once while the matrix cores idle.  Here a tile's x1 and hidden activations never leave the CU except as the
// f16 operand copy the DMA ring re-reads (L2-hot) and the LayerNorm2 residual.
//
// What bounds it.  A 64-token tile needs all 2.5 MB of the three weight matrices; at B = 64 there are 197 tiles for
// 256 CUs, so every CU streams the full 2.5 MB through its L2->LDS path (~65 GB/s per CU, MI355X_MICROARCH "Indexed
// rows: gather into LDS"): ~40 us, against ~20 us of MFMA work.  The kernel is therefore organised around the weight
// stream: ONE linear stream of 80 slabs of [256 weight rows x 64 k] f16 (32 KB each), pre-packed per layer in
// consumption order and in LDS image order (k_pack_tail), fed through a 3-slot ring with two slabs in flight; the
// activation operand of a step is an 8 KB [64 tokens x 64 k] slab (att / x1 through a second small ring, the GELU output
// straight from its LDS image).  Waves: wave w owns weight rows [32 w, 32 w + 32) of every slab and both 32-token tiles.
//
//   phase P   16 steps: slab (ks, nh) = W_out rows [256 nh, +256), k [64 ks, +64)          acc[nh][m] += W . att^T
//   LN1       accumulators -> LDS transpose -> + b_out + x (requested before phase P) -> LayerNorm -> x1: fp32 rows stay in
//             registers (LayerNorm2's residual), the f16 operand copy goes to a global scratch for the activation ring
//   phase F   4 chunks of 256 hidden features, 16 steps each:
//               8 steps  slab = W1 rows of the chunk, k [64 ks, +64)                        acch[m] += W1 . x1^T
//               GELU(acch + b1) -> H image (LDS, 4 activation slabs)
//               8 steps  slab (ks2, nh) = W2 rows [256 nh, +256), k = chunk's [64 ks2, +64)  acc[nh][m] += W2 . H^T
//   LN2       + b2 + x1 (registers) -> LayerNorm -> the stream (hi/lo), in place for the next layer
#pragma once
#include "mst_common.h"
#include "mst_gemm_dma.h"

namespace mst {

struct TailCfg {
    static constexpr int BT = 64;                        // tokens per workgroup
    static constexpr int WROWS = 256, KD = 64;           // weight slab: 256 rows x 64 k
    static constexpr int WSLAB = WROWS * KD * 2;         // 32 KB
    static constexpr int ASLAB = BT * KD * 2;            // 8 KB activation slab
    static constexpr int NW = 3, NA = 3;                 // ring slots
    static constexpr int OFF_W = 0;
    static constexpr int OFF_A = OFF_W + NW * WSLAB;     // 96 KB
    static constexpr int OFF_H = OFF_A + NA * ASLAB;     // 120 KB
    static constexpr int OFF_B = OFF_H + 4 * ASLAB;      // 152 KB: b1 (4 KB), beyond the LayerNorm scratch
    static constexpr int SMEM = OFF_B + MST_FF * 4;      // 156 KB (LayerNorm scratch, 129 KB, overlays [0, 129 KB) between phases)
    static constexpr int P_STEPS = 16, F_STEPS = 64, SLABS = P_STEPS + F_STEPS;
    static constexpr size_t LAYER_BYTES = (size_t)SLABS * WSLAB;     // 2.5 MB per layer
};

// Pack one layer's three matrices ([out][in] f16, torch Linear layout) into the tail's weight stream.
// Slab image: 16-B chunk c of row r at byte r * 128 + ((c ^ ((r >> 1) & 7)) << 4)  (ring_off_rb<128>: conflict-free b128 reads).
__global__ __launch_bounds__(256) void k_pack_tail(const f16* __restrict__ w_out, const f16* __restrict__ w1,
                                                   const f16* __restrict__ w2, f16* __restrict__ dst) {
    const int total = TailCfg::SLABS * (TailCfg::WSLAB / 16);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int s = i / (TailCfg::WSLAB / 16), o = (i - s * (TailCfg::WSLAB / 16)) * 16;
        const int r = o >> 7, pc = (o & 127) >> 4, c = pc ^ ((r >> 1) & 7), k0 = c * 8;
        const f16* src;
        if (s < TailCfg::P_STEPS) {
            const int ks = s >> 1, nh = s & 1;
            src = w_out + (size_t)(256 * nh + r) * MST_D + 64 * ks + k0;
        } else {
            const int u = s - TailCfg::P_STEPS, hc = u >> 4, v = u & 15;
            if (v < 8) src = w1 + (size_t)(256 * hc + r) * MST_D + 64 * v + k0;
            else {
                const int ks2 = (v - 8) >> 1, nh = (v - 8) & 1;
                src = w2 + (size_t)(256 * nh + r) * MST_FF + 256 * hc + 64 * ks2 + k0;
            }
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst) + (size_t)s * TailCfg::WSLAB + o) = *reinterpret_cast<const uint4*>(src);
    }
}

struct TailLane {                 // accumulator -> (token, feature) map of this kernel's [nh][m] tiles
    int wave, hh, l31;
    __device__ __forceinline__ TailLane() {
        const int lane = threadIdx.x & 63;
        wave = threadIdx.x >> 6;
        hh = lane >> 5;
        l31 = lane & 31;
    }
    __device__ __forceinline__ int tok(int m) const { return 32 * m + l31; }
    __device__ __forceinline__ int feat(int n, int g) const { return 256 * n + 32 * wave + 8 * g + 4 * hh; }
};

// 4 consecutive 1-KiB pieces, contiguous in LDS and in global memory: one M0 write, the instruction's immediate offset
// advances both addresses (mst_gemm_dma.h, glds_group).
__device__ __forceinline__ void tail_glds4(unsigned voff, unsigned long long sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:2048\n\tglobal_load_lds_dwordx4 %2, %3 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void tail_glds1(unsigned voff, unsigned long long sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}


// LayerNorm of a 64 x 512 accumulator tile in ROW layout.  The accumulators (lane = token) are transposed through `scratch`
// (64 rows of 2064 B, conflict-free for 16-B accesses); afterwards wave w owns rows [8 w, 8 w + 8) and a lane holds features
// [4 lane, +4) and [256 + 4 lane, +4) of each: statistics are wave-level shuffles.  `ra` / `rb` carry the residual in and the
// normalised rows out -- in the same registers, so LayerNorm1's output stays on chip as LayerNorm2's residual.
struct TailRows { f32x4 a[8], b[8]; };

// `resid(r, xa, xb)` adds row 8 wave + r's residual to (xa, xb) = this lane's two 4-feature groups.
template <class Resid>
__device__ __forceinline__ void tail_layernorm(f32x16 (&acc)[1][2][2], const TailLane& lc, char* scratch, const float* __restrict__ bias,
                                               const float* __restrict__ gamma, const float* __restrict__ beta, TailRows& rv, Resid resid) {
    constexpr int LD = MST_D * 4 + 16;
#pragma unroll
    for (int m = 0; m < 2; m++) {
        char* trow = scratch + lc.tok(m) * LD;
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 v = {acc[0][m][n][4 * g], acc[0][m][n][4 * g + 1], acc[0][m][n][4 * g + 2], acc[0][m][n][4 * g + 3]};
                *reinterpret_cast<f32x4*>(trow + lc.feat(n, g) * 4) = v;
            }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fa = lane * 4, fb = 256 + lane * 4;
    const f32x4 ba = *reinterpret_cast<const f32x4*>(bias + fa), bb = *reinterpret_cast<const f32x4*>(bias + fb);
    float mean[8], rstd[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int row = wave * 8 + r;
        f32x4 xa = *reinterpret_cast<const f32x4*>(scratch + row * LD + fa * 4) + ba;
        f32x4 xb = *reinterpret_cast<const f32x4*>(scratch + row * LD + fb * 4) + bb;
        resid(r, xa, xb);
        rv.a[r] = xa;
        rv.b[r] = xb;
        const f32x4 t = xa + xb;
        mean[r] = wave_sum((t[0] + t[1]) + (t[2] + t[3])) * (1.0f / MST_D);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        rv.a[r] -= mean[r];
        rv.b[r] -= mean[r];
        const f32x4 q = rv.a[r] * rv.a[r] + rv.b[r] * rv.b[r];
        rstd[r] = ln_rstd(wave_sum((q[0] + q[1]) + (q[2] + q[3])));
    }
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + fa), gb = *reinterpret_cast<const f32x4*>(gamma + fb);
    const f32x4 ea = *reinterpret_cast<const f32x4*>(beta + fa), eb = *reinterpret_cast<const f32x4*>(beta + fb);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        rv.a[r] = rv.a[r] * (ga * rstd[r]) + ea;
        rv.b[r] = rv.b[r] * (gb * rstd[r]) + eb;
    }
}

__global__ __launch_bounds__(512) void k_layer_tail(const f16* __restrict__ att, const f16* __restrict__ wt,
                                                    const float* __restrict__ b_out, const float* __restrict__ g1, const float* __restrict__ be1,
                                                    const float* __restrict__ b1, const float* __restrict__ b2,
                                                    const float* __restrict__ g2, const float* __restrict__ be2,
                                                    f16* __restrict__ hx, f16* __restrict__ hl,
                                                    f16* __restrict__ x1h, int M) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = TailCfg;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, l31 = lane & 31;
    const int tok0 = blockIdx.x * C::BT;
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#if defined(TAIL_STAMP)                                 // diagnostic build (probes/tail_clock.hip): shader-clock / 100 MHz stamps per phase
#define TAIL_MARK(i) if (tid == 0) { g_tail_stamp[blockIdx.x][2 * (i)] = __builtin_amdgcn_s_memtime(); g_tail_stamp[blockIdx.x][2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); }
    unsigned long long tl_last = 0, tl_sum[3] = {0, 0, 0};   // FFN loop split: FFN1 steps / GELU + H image / FFN2 steps (shader cycles)
#define TAIL_LAP(j) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tl_sum[j] += now_ - tl_last; tl_last = now_; }
#else
#define TAIL_MARK(i)
#define TAIL_LAP(j)
#endif
    TAIL_MARK(0)

    // ---- DMA addressing.  Weight slab s: this wave's 4 pieces are bytes [4096 wave, +4096) of the slab, linear on both sides.
    const unsigned w_voff = (unsigned)lane * 16u;
    const char* const wbase = reinterpret_cast<const char*>(wt) + 4096 * wave;
    // Activation slab ks of a [M][512] f16 tensor: this wave's piece = rows [8 wave, +8); lane -> row, physical chunk.
    const int a_row = 8 * wave + (lane >> 3);
    int a_tok = tok0 + a_row;
    if (a_tok >= M) a_tok = M - 1;                                      // tail tile: clamp (rows beyond M are never stored)
    const unsigned a_voff = (unsigned)a_tok * (unsigned)(MST_D * 2) + (unsigned)(((lane & 7) ^ ((a_row >> 1) & 7)) << 4);

    auto issue_w = [&](int s) {                                          // weight slab s of the layer -> ring slot s % 3
#if defined(TAIL_NODMA)                                 // timing ablation: MFMA + LDS reads on whatever the ring holds
        return;
#endif
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + C::OFF_W + (s % C::NW) * C::WSLAB + 4096 * wave);
        tail_glds4(w_voff, (unsigned long long)(wbase + (size_t)s * C::WSLAB), dst);
    };
    auto issue_a = [&](const f16* X, int ks, int aslot) {               // activation slab ks of X -> A slot
#if defined(TAIL_NODMA)
        return;
#endif
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + C::OFF_A + aslot * C::ASLAB + 1024 * wave);
        tail_glds1(a_voff, (unsigned long long)(reinterpret_cast<const char*>(X) + ks * 128), dst);
    };
    // one slab against one activation slab: 4 k-steps of 16, A operand = weight rows of this wave, B = the two token tiles
    auto mma_slab = [&](const char* wslot, const char* aslab, f32x16& a0, f32x16& a1) {
#if defined(TAIL_NOMMA)                                 // timing ablation: DMA stream, waits and barriers only
        return;
#endif
#pragma unroll
        for (int k16 = 0; k16 < 4; k16++) {
            const int c = 2 * k16 + hh;
            f16x8 wf, x0, x1;
#if defined(TAIL_NOREAD)                                // timing ablation: MFMAs on register garbage, no LDS reads
            asm volatile("" : "=v"(wf), "=v"(x0), "=v"(x1));
#else
            wf = *reinterpret_cast<const f16x8*>(wslot + ring_off_rb<128>(32 * wave + l31, c));
            x0 = *reinterpret_cast<const f16x8*>(aslab + ring_off_rb<128>(l31, c));
            x1 = *reinterpret_cast<const f16x8*>(aslab + ring_off_rb<128>(32 + l31, c));
#endif
#if defined(TAIL_READONLY)                              // timing ablation: LDS reads only
            asm volatile("" :: "v"(wf), "v"(x0), "v"(x1));
#else
            a0 = mfma_f16(wf, x0, a0);
            a1 = mfma_f16(wf, x1, a1);
#endif
        }
    };

    // The two steps of a (k-slab, feature-half) pair -- out-proj and FFN2 -- multiply the SAME activation slab: its eight token
    // fragments are read once, by the first step, and stay in registers for the second (8 + 4 fragment reads per pair instead
    // of 12 + 12; fragment reads are one of the three ~0.5 kW consumers of DESIGN.md section 4 "Power").
#ifndef TAIL_XREUSE
#define TAIL_XREUSE 1
#endif
    f16x8 xk0[4], xk1[4];
    auto mma_first = [&](const char* wslot, const char* aslab, f32x16& a0, f32x16& a1) {
#if !TAIL_XREUSE || defined(TAIL_NOMMA) || defined(TAIL_NOREAD) || defined(TAIL_READONLY)
        mma_slab(wslot, aslab, a0, a1);
#else
#pragma unroll
        for (int k16 = 0; k16 < 4; k16++) {
            const int c = 2 * k16 + hh;
            const f16x8 wf = *reinterpret_cast<const f16x8*>(wslot + ring_off_rb<128>(32 * wave + l31, c));
            xk0[k16] = *reinterpret_cast<const f16x8*>(aslab + ring_off_rb<128>(l31, c));
            xk1[k16] = *reinterpret_cast<const f16x8*>(aslab + ring_off_rb<128>(32 + l31, c));
            a0 = mfma_f16(wf, xk0[k16], a0);
            a1 = mfma_f16(wf, xk1[k16], a1);
        }
#endif
    };
    auto mma_second = [&](const char* wslot, const char* aslab, f32x16& a0, f32x16& a1) {
#if !TAIL_XREUSE || defined(TAIL_NOMMA) || defined(TAIL_NOREAD) || defined(TAIL_READONLY)
        mma_slab(wslot, aslab, a0, a1);
#else
#pragma unroll
        for (int k16 = 0; k16 < 4; k16++) {
            const f16x8 wf = *reinterpret_cast<const f16x8*>(wslot + ring_off_rb<128>(32 * wave + l31, 2 * k16 + hh));
            a0 = mfma_f16(wf, xk0[k16], a0);
            a1 = mfma_f16(wf, xk1[k16], a1);
        }
#endif
    };

    // b1 (1024 floats) into LDS once: the GELU step then needs no global load inside the DMA-counted loop (an ordinary load
    // there makes hipcc wait vmcnt(0), draining the ring: cdna guide section 5, "three .s-level traps" (b))
    float* b1s = reinterpret_cast<float*>(smem + C::OFF_B);
    b1s[tid] = b1[tid];
    b1s[tid + 512] = b1[tid + 512];

    f32x16 acc[1][2][2];                    // [.][m][nh]: the LayerNorm epilogue's [1][MT][NT] shape (NT index = feature half)
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[0][m][n][r] = 0.f;
    };
    zero_acc();

    // =========================================================================================== phase P: out-proj
    // group of step s = weight slab s (+ the att slab ks = s / 2 when s is even); two groups in flight.
    // Every accumulator index below is a compile-time constant (runtime-indexed ext-vector arrays go to scratch).
    auto p_sync = [&](int s) {
        if (s + 1 < C::P_STEPS) wait_vmcnt<4>();           // all but the newest group (>= 4 pieces per wave) have landed
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                      // slab s visible to every wave; slot of slab s - 1 is free
        if (s + 2 < C::P_STEPS) {
            issue_w(s + 2);
            if (((s + 2) & 1) == 0) issue_a(att, (s + 2) >> 1, ((s + 2) >> 1) % C::NA);
        }
    };
    // LayerNorm1's residual (the stream rows of this tile, hi + lo) is requested NOW, as raw f16 pairs, and first touched after
    // the out-proj loop: its HBM latency and its 128 KB per tile hide behind the weight stream of phase P.
    const int row_lane = threadIdx.x & 63;
    const int fa = row_lane * 4, fb = 256 + row_lane * 4;
    uint2 rh_a[8], rl_a[8], rh_b[8], rl_b[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int tok = tok0 + 8 * wave + r;
        const size_t off = (size_t)(tok < M ? tok : M - 1) * MST_D;
        rh_a[r] = *reinterpret_cast<const uint2*>(hx + off + fa);
        rl_a[r] = *reinterpret_cast<const uint2*>(hl + off + fa);
        rh_b[r] = *reinterpret_cast<const uint2*>(hx + off + fb);
        rl_b[r] = *reinterpret_cast<const uint2*>(hl + off + fb);
    }
    issue_w(0); issue_a(att, 0, 0);
    issue_w(1);
#pragma unroll 1
    for (int ks = 0; ks < 8; ks++) {
        const char* aslab = smem + C::OFF_A + (ks % C::NA) * C::ASLAB;
        p_sync(2 * ks);
        mma_first(smem + C::OFF_W + ((2 * ks) % C::NW) * C::WSLAB, aslab, acc[0][0][0], acc[0][1][0]);
        p_sync(2 * ks + 1);
        mma_second(smem + C::OFF_W + ((2 * ks + 1) % C::NW) * C::WSLAB, aslab, acc[0][0][1], acc[0][1][1]);
    }
    __syncthreads();                                       // ring dead (no DMA in flight): LayerNorm scratch overlays it
    TAIL_MARK(1)

    const TailLane lm;
    TailRows rv;                                           // x1 = LayerNorm1 output; lives through phase F as LayerNorm2's residual
    tail_layernorm(acc, lm, smem, b_out, g1, be1, rv, [&](int r, f32x4& xa, f32x4& xb) {
        xa = add4_f16(rh_a[r], rl_a[r], xa);
        xb = add4_f16(rh_b[r], rl_b[r], xb);
    });
    // x1's f16 operand copy -> global scratch: phase F re-reads it through the activation ring (L2-hot; the LDS has no room
    // for a resident 64 KB image beside the weight ring).  The fp32 rows stay in `rv` as LayerNorm2's residual.
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int tok = tok0 + 8 * wave + r;
        if (tok < M) {
            const size_t off = (size_t)tok * MST_D;
            *reinterpret_cast<uint2*>(x1h + off + fa) = pack4_f16(rv.a[r][0], rv.a[r][1], rv.a[r][2], rv.a[r][3]);
            *reinterpret_cast<uint2*>(x1h + off + fb) = pack4_f16(rv.b[r][0], rv.b[r][1], rv.b[r][2], rv.b[r][3]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's x1 stores have reached L2 ...
    __syncthreads();                                       // ... and everybody's: the ring may re-read x1h, the scratch is dead
    TAIL_MARK(2)

    // =========================================================================================== phase F: FFN
    zero_acc();
    f32x16 acch0, acch1;
    // step u = 16 hc + v: v < 8 -> FFN1 k-slab v (activation: x1h slab v through the A ring), v >= 8 -> FFN2 (ks2, nh) (activation: H image)
    auto issue_f = [&](int u) {
        issue_w(C::P_STEPS + u);
        const int v = u & 15;
        if (v < 8) issue_a(x1h, v, ((u >> 4) * 8 + v) % C::NA);
    };
    auto f_sync = [&](int u) {
        if (u + 1 < C::F_STEPS) wait_vmcnt<4>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (u + 2 < C::F_STEPS) issue_f(u + 2);
    };
    issue_f(0);
    issue_f(1);
#if defined(TAIL_STAMP)
    tl_last = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (int hc = 0; hc < 4; hc++) {
        TAIL_LAP(2)
#pragma unroll
        for (int r = 0; r < 16; r++) { acch0[r] = 0.f; acch1[r] = 0.f; }
#pragma unroll 1
        for (int v = 0; v < 8; v++) {
            const int u = 16 * hc + v;
            f_sync(u);
            mma_slab(smem + C::OFF_W + ((C::P_STEPS + u) % C::NW) * C::WSLAB, smem + C::OFF_A + ((hc * 8 + v) % C::NA) * C::ASLAB, acch0, acch1);
        }
        TAIL_LAP(0)
        {
            // GELU(acch + b1) -> H image: hidden feature j = 32 wave + 8 g + 4 hh + i of the chunk lives in activation slab
            // j / 64 at k = j % 64.  The previous chunk's FFN2 steps finished 8 barriers ago; the barrier of the next step
            // publishes these writes.
            const float* bb = b1s + 256 * hc + 32 * wave + 4 * hh;
            char* hslab = smem + C::OFF_H + (wave >> 1) * C::ASLAB;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bb + 8 * g);
                const int c = 4 * (wave & 1) + g;
                *reinterpret_cast<uint2*>(hslab + ring_off_rb<128>(l31, c) + 8 * hh) =
                    pack4_f16(gelu_erf(acch0[4 * g] + bv[0]), gelu_erf(acch0[4 * g + 1] + bv[1]),
                              gelu_erf(acch0[4 * g + 2] + bv[2]), gelu_erf(acch0[4 * g + 3] + bv[3]));
                *reinterpret_cast<uint2*>(hslab + ring_off_rb<128>(32 + l31, c) + 8 * hh) =
                    pack4_f16(gelu_erf(acch1[4 * g] + bv[0]), gelu_erf(acch1[4 * g + 1] + bv[1]),
                              gelu_erf(acch1[4 * g + 2] + bv[2]), gelu_erf(acch1[4 * g + 3] + bv[3]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the raw s_barrier below waits for no counter
        }
        TAIL_LAP(1)
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
            const char* hsl = smem + C::OFF_H + q * C::ASLAB;
            const int u = 16 * hc + 8 + 2 * q;
            f_sync(u);
            mma_first(smem + C::OFF_W + ((C::P_STEPS + u) % C::NW) * C::WSLAB, hsl, acc[0][0][0], acc[0][1][0]);
            f_sync(u + 1);
            mma_second(smem + C::OFF_W + ((C::P_STEPS + u + 1) % C::NW) * C::WSLAB, hsl, acc[0][0][1], acc[0][1][1]);
        }
    }
    TAIL_LAP(2)
    __syncthreads();
    TAIL_MARK(3)
    {
        TailRows x2;                                       // residual = x1 (still in registers), result = the stream rows
        tail_layernorm(acc, lm, smem, b2, g2, be2, x2, [&](int r, f32x4& xa, f32x4& xb) { xa += rv.a[r]; xb += rv.b[r]; });
        rv = x2;
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int tok = tok0 + 8 * wave + r;
        if (tok < M) {
            const size_t off = (size_t)tok * MST_D;
            uint2 h, l;
            split4_f16(rv.a[r], h, l);
            *reinterpret_cast<uint2*>(hx + off + fa) = h;
            *reinterpret_cast<uint2*>(hl + off + fa) = l;
            split4_f16(rv.b[r], h, l);
            *reinterpret_cast<uint2*>(hx + off + fb) = h;
            *reinterpret_cast<uint2*>(hl + off + fb) = l;
        }
    }
    TAIL_MARK(4)
#if defined(TAIL_STAMP)
    if (tid == 0) { g_tail_stamp[blockIdx.x][10] = tl_sum[0]; g_tail_stamp[blockIdx.x][11] = tl_sum[1]; g_tail_stamp[blockIdx.x][12] = tl_sum[2]; }
#endif
#undef TAIL_LAP
#undef TAIL_MARK
}

}  // namespace mst
