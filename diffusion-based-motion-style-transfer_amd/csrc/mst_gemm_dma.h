// Hot GEMMs (QKV, out-proj+LN, FFN1+GELU, FFN2+LN) on an LDS-DMA ring.
//
// Why LDS-DMA: the first, register-staged loop (global -> VGPR -> ds_write, double buffer) measured 67-79 % SQ_WAIT_ANY and
// ~9 % MFMA busy on MI355X (profiles/r01_v0_*): with one 8-wave block per CU nothing hides the
// L2 latency of the next K-slab.  Here K is consumed in 32-deep slabs through a 4-slot LDS ring
// filled by `global_load_lds_dwordx4` (no VGPR staging, no ds_write); three slabs are always in
// flight behind a COUNTED `s_waitcnt vmcnt(N)` and a raw `s_barrier` (a `__syncthreads()` would
// drain the DMA queue: cdna guide section 5 "Pipelining across barriers").
//
// Slab image: [BT token rows | BF weight rows] x 64 B; 16-B chunk c of row r is stored at
// chunk c ^ ((r >> 2) & 3) (four 64-B rows share one 256-B bank row; the XOR makes every
// ds_read_b128 fragment read conflict-free).  A DMA wave-instruction writes 1 KiB = 16 rows
// lane-linearly, so the swizzle is applied on the per-lane SOURCE address (guide rule 21).
//
// Wave tile = MT x NT MFMA tiles (2 x 2 here): 4 fragment reads feed 4 MFMAs per k-step.
// Orientation: the MFMA A operand is the WEIGHT fragment, B the ACTIVATION fragment, so in the 32x32
// result a lane owns one token and its 16 registers run over features in groups of 4 consecutive ones.
#pragma once
#include "mst_common.h"

namespace mst {

template <int BT, int BF, int MT, int NT, int NS = 4, int NX = 1, int BK = 32, int XS = 1>
struct DTile {
    static_assert(BK == 32 || BK == 64, "slab depth");
    static constexpr int KDEPTH = BK;
    static constexpr int RB = BK * 2;                 // bytes per slab row: 64 (half a cache line) or 128 (a full line)
    static constexpr int RPP = 1024 / RB;             // tile rows per 1-KiB DMA piece
    static constexpr int CPR = RB / 16;               // 16-B chunks per row
    static constexpr int WT = BT / (32 * MT);       // wave rows (token direction)
    static constexpr int WN = 8 / WT;               // wave columns (feature direction)
    static_assert(WT * WN == 8 && WN * NT * 32 == BF, "8 waves must tile BT x BF");
    static constexpr int XGROUPS = XS * NX;         // NX = 2: a second token group (CFG: the uncond half); XS = 2: every group twice,
    static constexpr int XROWS = XGROUPS * BT;      // as the hi and the lo half of a split activation (same accumulators, weights staged once)
    static constexpr int XHI = NX * BT;             // rows [0, XHI): the hi (or only) tensor; [XHI, XROWS): the lo tensor
    static constexpr int ROWS = XROWS + BF;
    static constexpr int STAGE = ROWS * RB;         // bytes per slab
    static constexpr int NSTAGE = NS;               // ring slots: NS - 1 slabs in flight
    static constexpr int SMEM = NSTAGE * STAGE;
    static constexpr int INSTR = ROWS / RPP;        // 1-KiB DMA pieces per slab
    static constexpr int PER = (INSTR + 7) / 8;     // pieces per wave per slab (uniform; extras duplicate)
};

__device__ __forceinline__ int ring_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }
// 128-B rows (64-deep slabs): two rows per 256-B bank row, chunk c of row r at c ^ ((r >> 1) & 7)
template <int RB> __device__ __forceinline__ int ring_off_rb(int row, int c) {
    return RB == 64 ? ring_off(row, c) : row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
}

// LDS-DMA issue.  One `global_load_lds_dwordx4` moves 1 KiB (64 lanes x 16 B): LDS destination = M0 + imm + 16 * lane,
// global source = SGPR base + per-lane 32-bit offset + imm -- the SAME immediate is added on both sides (verified with
// csrc/probes/probe_glds.hip).  A wave owns PER CONSECUTIVE pieces of a slab, so up to four of them share one M0 write
// with imm = 0 / 1024 / 2048 / 3072 and the per-lane offset pre-compensated by -imm; the slab-to-slab advance is a
// scalar add on the two operand bases.  (First version: one piece per asm statement with M0 save/set/restore and a
// 64-bit VALU address add each -- ~150 issue cycles per piece, which is what bounded the slab loop at ~0.6 us.)
constexpr int kMaxPer = 12;             // most 1-KiB pieces one wave issues per slab
template <int G>
__device__ __forceinline__ void glds_group(const unsigned (&voff)[kMaxPer], const unsigned long long (&sb)[kMaxPer], int j0, unsigned lds_dst) {
    unsigned keep;
    if constexpr (G == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %6\n\tglobal_load_lds_dwordx4 %3, %7 offset:1024\n\t"
                     "global_load_lds_dwordx4 %4, %8 offset:2048\n\tglobal_load_lds_dwordx4 %5, %9 offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_dst), "v"(voff[j0]), "v"(voff[j0 + 1]), "v"(voff[j0 + 2]), "v"(voff[j0 + 3]),
                       "s"(sb[j0]), "s"(sb[j0 + 1]), "s"(sb[j0 + 2]), "s"(sb[j0 + 3]) : "memory");
    else if constexpr (G == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %5\n\tglobal_load_lds_dwordx4 %3, %6 offset:1024\n\t"
                     "global_load_lds_dwordx4 %4, %7 offset:2048\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_dst), "v"(voff[j0]), "v"(voff[j0 + 1]), "v"(voff[j0 + 2]),
                       "s"(sb[j0]), "s"(sb[j0 + 1]), "s"(sb[j0 + 2]) : "memory");
    else if constexpr (G == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %4\n\tglobal_load_lds_dwordx4 %3, %5 offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_dst), "v"(voff[j0]), "v"(voff[j0 + 1]), "s"(sb[j0]), "s"(sb[j0 + 1]) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_dst), "v"(voff[j0]), "s"(sb[j0]) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Per-wave DMA state of one GEMM: piece j of this wave (tile rows 16 * (start + j) ..) comes from operand
// `isx[j] ? X : W` at per-lane byte offset voff[j] (bias and immediate already folded in).
constexpr unsigned kDmaBias = 4096;      // keeps (offset - imm) non-negative; subtracted from the scalar bases
template <class TL>
struct DmaPlan {
    static_assert(TL::PER <= kMaxPer && TL::INSTR >= TL::PER, "piece plan");
    unsigned voff[kMaxPer];
    unsigned char kind[kMaxPer];            // 0: weight rows, 1: activation rows, 2: activation rows of the lo tensor
    int start;
    // rowbyte(r): byte offset of tile row r inside its operand (X rows for r < XROWS, else W rows)
    template <class F>
    __device__ __forceinline__ void init(int wave, int lane, F rowbyte) {
        start = wave * TL::PER;
        if (start > TL::INSTR - TL::PER) start = TL::INSTR - TL::PER;   // last wave re-issues a few of its neighbour's
#pragma unroll
        for (int j = 0; j < kMaxPer; j++) {
            if (j < TL::PER) {
                const int row = (start + j) * TL::RPP + lane / TL::CPR;
                const int c = TL::RB == 64 ? ((lane & 3) ^ ((row >> 2) & 3)) : ((lane & 7) ^ ((row >> 1) & 7));
                const int r0 = (start + j) * TL::RPP;                 // wave-uniform
                kind[j] = r0 >= TL::XROWS ? 0 : (r0 < TL::XHI ? 1 : 2);
                voff[j] = rowbyte(row) + c * 16 + kDmaBias - (j & 3) * 1024;
            } else { voff[j] = 0; kind[j] = 0; }
        }
    }
    // slab `ksrc` of the operands -> ring slot of slab-sequence position `kt`
    __device__ __forceinline__ void issue(unsigned smem_base, int kt, int ksrc, const char* xbase, const char* wbase,
                                          const char* xlobase = nullptr) const {
        const unsigned long long sx = (unsigned long long)(xbase + (size_t)ksrc * TL::RB - kDmaBias);
        const unsigned long long sw = (unsigned long long)(wbase + (size_t)ksrc * TL::RB - kDmaBias);
        const unsigned long long sl = (unsigned long long)((xlobase ? xlobase : xbase) + (size_t)ksrc * TL::RB - kDmaBias);
        unsigned long long sb[kMaxPer];
#pragma unroll
        for (int j = 0; j < kMaxPer; j++) sb[j] = kind[j] == 1 ? sx : (kind[j] == 2 ? sl : sw);
        const unsigned base = __builtin_amdgcn_readfirstlane(smem_base + (kt % TL::NSTAGE) * TL::STAGE + start * 1024);
        if constexpr (TL::PER >= 4) glds_group<4>(voff, sb, 0, base);
        else glds_group<TL::PER>(voff, sb, 0, base);
        if constexpr (TL::PER >= 8) glds_group<4>(voff, sb, 4, base + 4096);
        else if constexpr (TL::PER > 4) glds_group<TL::PER - 4>(voff, sb, 4, base + 4096);
        if constexpr (TL::PER >= 12) glds_group<4>(voff, sb, 8, base + 8192);
        else if constexpr (TL::PER > 8) glds_group<TL::PER - 8>(voff, sb, 8, base + 8192);
    }
};

// Row sources: which global row feeds tile row r of the activation operand.
// `Xlo` (optional, same layout as X): the activation is the pair hi + lo (f16(x) and f16(x - hi), ~22 significant bits) and the
// GEMM runs its K range twice -- W . hi + W . lo into the same accumulators -- so the operand rounding of THIS product
// disappears.  Used where it is nearly free and pays most: the pose embedding (the rounding of x_t perturbs everything
// behind it) and the output projection (the rounding of the last stream goes straight into the result): one 64-clip
// forward 4.8e-4 -> 3.6e-4, classifier-free guidance 7.3e-4 -> 5.5e-4 relative L2 in the oracle's rounding model.
// `Wlo` (optional, the weight matrix's layout): the weights are a hi + lo pair too -- one more pass of the K range, X . Wlo.  Precise
// mode only (mst_set_precise): with the activations split, what is left of the operand rounding is the weights' (oracle rounding
// model on weights with outlier statistics: 1.2e-3 all operands f16, 9.7e-4 activations exact, 4.0e-4 weights exact as well).
struct RowsDirect {                       // tile row r = matrix row tok0 + r
    const f16* X; int ld; const f16* Xlo = nullptr; const f16* Wlo = nullptr;
    __device__ __forceinline__ const char* base() const { return reinterpret_cast<const char*>(X); }
    __device__ __forceinline__ const char* base_lo() const { return reinterpret_cast<const char*>(Xlo); }
    __device__ __forceinline__ const char* base_wlo() const { return reinterpret_cast<const char*>(Wlo); }
    __device__ __forceinline__ unsigned rowbyte(int tok0, int r) const { return (unsigned)(tok0 + r) * (unsigned)ld * 2u; }
};
struct RowsFrames {                       // tile row r = frame (tok0 + r) of the token stream, conditioning token skipped;
    const f16* X; int ld; int T, S, total;   // rows >= BT (second group) come from the uncond half (+cfg_rows)
    int BT; size_t cfg_rows;
    int tok_off = 1;                          // tokens in front of the frames: 1 (conditioning token) or 2 (the motion encoder's mu / sigma queries)
    const f16* Xlo = nullptr; const f16* Wlo = nullptr;
    __device__ __forceinline__ const char* base() const { return reinterpret_cast<const char*>(X); }
    __device__ __forceinline__ const char* base_lo() const { return reinterpret_cast<const char*>(Xlo); }
    __device__ __forceinline__ const char* base_wlo() const { return reinterpret_cast<const char*>(Wlo); }
    __device__ __forceinline__ unsigned rowbyte(int tok0, int r) const {
        int half = r >= BT ? 1 : 0;
        int tok = tok0 + r - half * BT;
        if (tok >= total) tok = total - 1;
        int clip = tok / T, t = tok - clip * T;
        return (unsigned)((size_t)clip * S + tok_off + t + half * cfg_rows) * (unsigned)ld * 2u;
    }
};

// TL::XGROUPS = 2 NX (XS = 2): the activation is hi + lo and BOTH halves of a k-slab are staged beside ONE copy of the weight slab
// (the K-twice form below re-streams the weights: it is what small launches and tiles whose ring would not fit use).
template <class TL, int BT, int BF, int MT, int NT, int NX, class SRC>
__device__ __forceinline__ void gemm_mainloop_dma(char* smem, const SRC& xs, const f16* __restrict__ W, int ldw,
                                                  int tok0, int f0, int K, f32x16 (&acc)[NX][MT][NT]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave / TL::WN, wn = wave % TL::WN;
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    DmaPlan<TL> plan;
    constexpr bool SPLIT = TL::XGROUPS > NX;                 // hi and lo rows in the same slab
    plan.init(wave, lane, [&](int row) {
        return row < TL::XROWS ? xs.rowbyte(tok0, row < TL::XHI ? row : row - TL::XHI) : (unsigned)(f0 + row - TL::XROWS) * (unsigned)ldw * 2u;
    });
    const char* xb = xs.base();
    const char* xlo = xs.base_lo();
    const char* wlo = xs.base_wlo();
    const char* wb = reinterpret_cast<const char*>(W);
    // passes over the K range: X . W, then (K-twice form) Xlo . W, then (split weights) X . Wlo -- the weights are re-streamed (L2-hot).
    // In the SPLIT form a pass stages hi and lo rows together, so the weight-lo pass multiplies (X + Xlo) . Wlo: the extra term is 2^-22.
    const int KT1 = K / TL::KDEPTH;
    const int KTX = (xlo && !SPLIT) ? 2 * KT1 : KT1;         // end of the passes against W
    const int KT = wlo ? KTX + KT1 : KTX;
    constexpr int AHEAD = TL::NSTAGE - 1;                    // slabs in flight
    auto issue = [&](int kt) {
        if (kt >= KTX) plan.issue(smem_base, kt, kt - KTX, xb, wlo, SPLIT ? xlo : nullptr);
        else if (SPLIT) plan.issue(smem_base, kt, kt, xb, wb, xlo);
        else if (kt < KT1) plan.issue(smem_base, kt, kt, xb, wb);
        else plan.issue(smem_base, kt, kt - KT1, xlo, wb);
    };
#pragma unroll
    for (int s = 0; s < AHEAD; s++)
        if (s < KT) issue(s);

    for (int kt = 0; kt < KT; kt++) {
        const int rem = KT - 1 - kt;                        // slabs already issued beyond kt: min(rem, AHEAD - 1)
        if (AHEAD >= 2 && rem >= AHEAD - 1) wait_vmcnt<(AHEAD >= 2 ? AHEAD - 1 : 0) * TL::PER>();
        else if (AHEAD >= 3 && rem == 1) wait_vmcnt<TL::PER>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                       // slab kt landed for every wave; slot of slab kt-1 is free
        if (kt + AHEAD < KT) issue(kt + AHEAD);
        const char* st = smem + (kt % TL::NSTAGE) * TL::STAGE;
#pragma unroll
        for (int ks = 0; ks < TL::KDEPTH / 16; ks++) {
            const int c = ks * 2 + (lane >> 5);
            f16x8 xf[TL::XGROUPS][MT], wf[NT];
#pragma unroll
            for (int x = 0; x < TL::XGROUPS; x++)
#pragma unroll
                for (int m = 0; m < MT; m++)
                    xf[x][m] = *reinterpret_cast<const f16x8*>(st + ring_off_rb<TL::RB>(x * BT + (wt * MT + m) * 32 + (lane & 31), c));
#pragma unroll
            for (int n = 0; n < NT; n++) wf[n] = *reinterpret_cast<const f16x8*>(st + ring_off_rb<TL::RB>(TL::XROWS + (wn * NT + n) * 32 + (lane & 31), c));
#pragma unroll
            for (int x = 0; x < TL::XGROUPS; x++)            // group x >= NX = the lo half of group x - NX: same accumulators
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int n = 0; n < NT; n++) acc[x % NX][m][n] = mfma_f16(wf[n], xf[x][m], acc[x % NX][m][n]);
        }
    }
    __builtin_amdgcn_s_barrier();                           // every wave done reading: smem reusable
}

template <int BT, int BF, int MT, int NT>
struct DLane {
    int hh, wt, wn, l31;
    __device__ __forceinline__ DLane() {
        using TL = DTile<BT, BF, MT, NT>;
        int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        hh = lane >> 5;
        l31 = lane & 31;
        wt = wave / TL::WN;
        wn = wave % TL::WN;
    }
    __device__ __forceinline__ int tok(int m) const { return (wt * MT + m) * 32 + l31; }
    __device__ __forceinline__ int feat(int n, int g) const { return (wn * NT + n) * 32 + 8 * g + 4 * hh; }
};

// Epilogues.  Accumulators hold (lane = token, 4 consecutive features per register group); writing
// them straight to row-major global memory is a 64-rows-per-instruction scatter that measured ~28 us
// of fixed time per launch (address-processing bound, 2x write amplification).  Both epilogues
// therefore transpose through the LDS ring (free after the main loop) and touch global memory only
// with whole-row, fully coalesced 1-KiB wave accesses.

// + bias (+ erf GELU) -> f16 row-major   (K4 QKV: GELU = false, K7 FFN1: GELU = true)
template <bool GELU>
struct DEpiBiasF16 {
    const float* bias; f16* out; int ldo; int M;
    f16* out_lo = nullptr;                   // optional f16(v - hi) beside the result (precise tiny-clip path: the next GEMM's Xlo)
    __device__ __forceinline__ int rows() const { return M; }
    // rows transposed per pass: as many as fit ~68 KB (128 rows of 512 B, 64 rows of 1 KiB)
    template <int BT, int BF> static constexpr int pass_rows() { return (BT * (BF * 2 + 16) <= 69632) ? BT : (BF == 512 ? 64 : 128); }
    template <int BT, int BF> static constexpr int smem_bytes() { return pass_rows<BT, BF>() * (BF * 2 + 16); }
    template <int BT, int BF, int MT, int NT>
    __device__ __forceinline__ void run(f32x16 (&acc)[1][MT][NT], int tok0, int f0, char* smem) const {
        static_assert(BF == 128 || BF == 256 || BF == 512, "copy-out assumes 256-byte, 512-byte or 1-KiB tile rows");
        constexpr int LD = BF * 2 + 16;                       // bytes per tile row (+16: spreads banks, keeps 16-B alignment)
        constexpr int PR = pass_rows<BT, BF>();
        constexpr int PASSES = BT / PR;
        constexpr int LPR = BF / 8;                           // lanes per tile row (16 B each)
        constexpr int RPA = 64 / LPR;                         // tile rows covered by one 1-KiB wave access (4, 2 or 1)
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
                        int f = lc.feat(n, g);
                        f32x4 b = *reinterpret_cast<const f32x4*>(bias + f0 + f);
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            v[i] = acc[0][m][n][4 * g + i] + b[i];
                            if (GELU) v[i] = gelu_erf(v[i]);
                        }
                        uint2 hi2, lo2;
                        split4_f16(f32x4{v[0], v[1], v[2], v[3]}, hi2, lo2);
                        *reinterpret_cast<uint2*>(trow + f * 2) = hi2;
                        // lo half: straight from the accumulator layout (tiny launches only; no second transpose pass)
                        if (out_lo && tok0 + tl < M) *reinterpret_cast<uint2*>(out_lo + (size_t)(tok0 + tl) * ldo + f0 + f) = lo2;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < PR / (8 * RPA); p++) {
                int row = p * 8 * RPA + wave * RPA + sub;
                int tok = tok0 + pass * PR + row;
                uint4 v = *reinterpret_cast<const uint4*>(smem + row * LD + col);
                if (tok < M) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(out + (size_t)tok * ldo + f0) + col) = v;
            }
            if (pass + 1 < PASSES) __syncthreads();
        }
    }
};

// fp32 tile out (no bias): small launches whose LayerNorm runs as a separate row-wise kernel (k_ln_rows)
struct DEpiPlainF32 {
    float* out; int ldo; int M;
    __device__ __forceinline__ int rows() const { return M; }
    template <int BT, int BF> static constexpr int smem_bytes() { return 32 * (BF * 4 + 16); }
    template <int BT, int BF, int MT, int NT>
    __device__ __forceinline__ void run(f32x16 (&acc)[1][MT][NT], int tok0, int f0, char* smem) const {
        static_assert(BF == 128 || BF == 256, "512-byte or 1-KiB fp32 tile rows");
        constexpr int LD = BF * 4 + 16, PR = 32, PASSES = BT / PR;
        constexpr int LPR = BF / 4;                           // lanes per row (16 B = 4 floats each)
        constexpr int RPA = 64 / LPR;
        DLane<BT, BF, MT, NT> lc;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int sub = lane / LPR, col = (lane % LPR) * 16;
#pragma unroll
        for (int pass = 0; pass < PASSES; pass++) {
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const int tl = lc.tok(m);
                if (tl / PR != pass) continue;
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
                const f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * LD + col);
                if (tok < M) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out + (size_t)tok * ldo + f0) + col) = v;
            }
            if (pass + 1 < PASSES) __syncthreads();
        }
    }
};

// + bias + residual -> LayerNorm -> fp32 stream + f16 operand copy   (K6, K8); BF must be 512.
// After the transpose every wave owns whole rows: statistics are wave-level shuffles, no
// cross-wave exchange.
struct DEpiResidLN {
    const float* bias; const float* gamma; const float* beta;
    f16* hi; f16* lo; int M;                 // the stream (read as residual, rewritten in place): hi + lo pair
    const f16* rhi = nullptr; const f16* rlo = nullptr;     // residual source when it is not the output (fused layer tail)
    __device__ __forceinline__ int rows() const { return M; }
    template <int BT, int BF> static constexpr int smem_bytes() { return 64 * (MST_D * 4 + 16); }     // 64 rows per pass
    template <int BT, int BF, int MT, int NT>
    __device__ __forceinline__ void run(f32x16 (&acc)[1][MT][NT], int tok0, int f0, char* smem) const {
        static_assert(BF == MST_D, "LayerNorm needs the whole row in one block");
        run_map<BT, MT, NT>(acc, DLane<BT, BF, MT, NT>(), tok0, smem);
    }
    // LM: accumulator -> (token, feature) map of the calling kernel: tok(m), feat(n, g) = first of 4 consecutive features
    template <int BT, int MT, int NT, class LM>
    __device__ __forceinline__ void run_map(f32x16 (&acc)[1][MT][NT], const LM& lc, int tok0, char* smem) const {
        const f16* const rh = rhi ? rhi : hi;
        const f16* const rl = rlo ? rlo : lo;
        constexpr int LD = MST_D * 4 + 16;                    // 2064-B rows: conflict-free b128 writes and reads
        constexpr int PR = 64, PASSES = BT / PR;              // 128-token tiles (large launches) go through LDS in two halves
        const int tok0_tile = tok0;
#pragma unroll
        for (int pass = 0; pass < PASSES; pass++) {
        if (pass > 0) __syncthreads();                        // previous half consumed
        tok0 = tok0_tile + pass * PR;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const int tl = lc.tok(m);
            if (tl / PR != pass) continue;                    // wave-uniform
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
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        // lane owns features [4*lane, 4*lane+4) and [256 + 4*lane, ...): two 1-KiB accesses per row
        const int fa = lane * 4, fb = 256 + lane * 4;
        const f32x4 ba = *reinterpret_cast<const f32x4*>(bias + fa), bb = *reinterpret_cast<const f32x4*>(bias + fb);
        constexpr int RPW = PASSES > 1 ? 4 : 8;               // rows a wave handles at once (8 per pass; 128-row tiles keep 128
                                                              // accumulator registers live across the passes, so they go 4 + 4)
        for (int chunk = 0; chunk < (PR / 8) / RPW; chunk++) {
        const int row0 = wave * (PR / 8) + chunk * RPW;
        // all residual loads in flight first (they overlap the LDS reads below)
        f32x4 xa[RPW], xb[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            int tok = tok0 + row0 + r;
            size_t off = (size_t)(tok < M ? tok : 0) * MST_D;
            xa[r] = join4_f16(*reinterpret_cast<const uint2*>(rh + off + fa), *reinterpret_cast<const uint2*>(rl + off + fa));
            xb[r] = join4_f16(*reinterpret_cast<const uint2*>(rh + off + fb), *reinterpret_cast<const uint2*>(rl + off + fb));
        }
        // row sums for ALL rows first, then the shuffle ladders step by step across rows: RPW
        // independent ds_bpermutes per step instead of RPW serial 6-deep dependency chains
        float s[RPW], s2[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int row = row0 + r;
            const f32x4 ta = *reinterpret_cast<const f32x4*>(smem + row * LD + fa * 4);
            const f32x4 tb = *reinterpret_cast<const f32x4*>(smem + row * LD + fb * 4);
            s[r] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xa[r][i] = ta[i] + ba[i] + xa[r][i];
                xb[r][i] = tb[i] + bb[i] + xb[r][i];
                s[r] += xa[r][i] + xb[r][i];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < RPW; r++) s[r] += __shfl_xor(s[r], o);
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const float mean = s[r] * (1.0f / MST_D);
            s2[r] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xa[r][i] -= mean;
                xb[r][i] -= mean;
                s2[r] += xa[r][i] * xa[r][i] + xb[r][i] * xb[r][i];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < RPW; r++) s2[r] += __shfl_xor(s2[r], o);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + fa), gb = *reinterpret_cast<const f32x4*>(gamma + fb);
        const f32x4 ea = *reinterpret_cast<const f32x4*>(beta + fa), eb = *reinterpret_cast<const f32x4*>(beta + fb);
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int tok = tok0 + row0 + r;
            if (tok >= M) continue;
            const float rstd = ln_rstd(s2[r]);
            f32x4 ya, yb;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ya[i] = xa[r][i] * rstd * ga[i] + ea[i];
                yb[i] = xb[r][i] * rstd * gb[i] + eb[i];
            }
            const size_t off = (size_t)tok * MST_D;
            uint2 h, l;
            split4_f16(ya, h, l);
            *reinterpret_cast<uint2*>(hi + off + fa) = h;
            *reinterpret_cast<uint2*>(lo + off + fa) = l;
            split4_f16(yb, h, l);
            *reinterpret_cast<uint2*>(hi + off + fb) = h;
            *reinterpret_cast<uint2*>(lo + off + fb) = l;
        }
        }   // chunk
        }   // pass
    }
};

// K3: frames x pose-embedding + bias + positional row -> token stream rows clip*S + 1 + t (fp32 + f16).
// Same 64 x 512 tile and LDS transpose as the LayerNorm epilogue; `dup` > 0 also writes the rows of
// the CFG uncond half (identical frames, only the conditioning token differs).
// Conditioning token (token 0 of every clip) = timestep embedding + text projection + positional row 0, written by the tile that
// holds the clip's first frame -- the per-step k_cond_token launch folded into this epilogue.  Fields as k_cond_token's arguments.
struct CondTok {
    const float* temb = nullptr; const float* textproj = nullptr; const LoopDev* ld = nullptr;
    int uniform_row = 0, temb_mod = 1, tp_half = 0, tp_uncond = 0, joff = 0, rows = 0;
    const long long* tidx = nullptr;          // non-null (training calls, round 6): `temb` is the TABLE over every timestep, clip's row = tidx[clip % temb_mod]
};
struct DEpiEmbedIn {
    const float* bias; const float* pe; f16* hi; f16* lo; int T, S, total; size_t dup;
    int tok_off = 1;                          // frame t becomes token tok_off + t (positional row included)
    CondTok ct;                               // ct.temb != null: also write the conditioning tokens (tok_off == 1 callers)
    Drop pd = {0u, 0u, 1.0f};                 // pd.thr != 0 (training calls, round 6): PositionalEncoding's dropout on the assembled stream, element index =
                                              // offset in `hi` -- what k_dropout_stream did in a launch of its own behind this kernel
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
        finish<BT>(tok0, smem);
    }
    // everything behind the accumulator -> LDS rows hop (smem: BT rows of LD = 2064 bytes, fp32): + bias + positional row -> the stream's
    // hi / lo rows, whole-row stores; conditioning tokens.  Shared by the ring GEMM above and k_embed_in (mst_embed.h).
    template <int BT>
    __device__ __forceinline__ void finish(int tok0, char* smem) const {
        constexpr int LD = MST_D * 4 + 16;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int fa = lane * 4, fb = 256 + lane * 4;
        const f32x4 ba = *reinterpret_cast<const f32x4*>(bias + fa), bb = *reinterpret_cast<const f32x4*>(bias + fb);
        constexpr int RPW = BT / 8;
        // the positional rows of all RPW rows first (one memory latency, not RPW in series), then the rows themselves
        f32x4 pa[RPW], pb[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            int tok = tok0 + wave * RPW + r;
            if (tok >= total) tok = total - 1;
            const float* perow = pe + (size_t)(tok % T + tok_off) * MST_D;
            pa[r] = *reinterpret_cast<const f32x4*>(perow + fa);
            pb[r] = *reinterpret_cast<const f32x4*>(perow + fb);
        }
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int row = wave * RPW + r, tok = tok0 + row;
            if (tok >= total) continue;
            const int clip = tok / T, t = tok - clip * T;
            f32x4 xa = *reinterpret_cast<const f32x4*>(smem + row * LD + fa * 4);
            f32x4 xb = *reinterpret_cast<const f32x4*>(smem + row * LD + fb * 4);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xa[i] = xa[i] + ba[i] + pa[r][i];
                xb[i] = xb[i] + bb[i] + pb[r][i];
            }
            size_t off = ((size_t)clip * S + tok_off + t) * MST_D;
            if (pd.thr) {                                      // kernel-uniform
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    xa[i] *= drop_mul(pd, (uint32_t)(off + fa + i));
                    xb[i] *= drop_mul(pd, (uint32_t)(off + fb + i));
                }
            }
            uint2 ha, la, hb, lb;
            split4_f16(xa, ha, la);
            split4_f16(xb, hb, lb);
#pragma unroll
            for (int rep = 0; rep < 2; rep++) {
                *reinterpret_cast<uint2*>(hi + off + fa) = ha;
                *reinterpret_cast<uint2*>(lo + off + fa) = la;
                *reinterpret_cast<uint2*>(hi + off + fb) = hb;
                *reinterpret_cast<uint2*>(lo + off + fb) = lb;
                if (dup == 0) break;
                off += dup;
            }
        }
        if (ct.temb) {
            // clips whose frame 0 is a row of this tile; thread = feature (512 threads = MST_D)
            static_assert(MST_D == 512, "one thread per feature");
            const int f = threadIdx.x, x_clips = total / T;
            int uniform_row = ct.uniform_row;
            if (ct.ld) uniform_row = ct.ld->nrun - 1 - (ct.ld->jbase + ct.joff);      // sampling loop: the timestep row of step jbase + joff
            const int lim = tok0 + BT < total ? tok0 + BT : total;
            for (int c = (tok0 + T - 1) / T; c * T < lim; c++)
                for (int clip = c; clip < ct.rows; clip += x_clips) {                // the clip and, under CFG, its uncond twin
                    const int tr = ct.tidx ? (int)ct.tidx[clip % ct.temb_mod] : (uniform_row >= 0 ? uniform_row : clip % ct.temb_mod);
                    const int tp = (ct.tp_half > 0 && clip >= ct.tp_half) ? clip - ct.tp_half + ct.tp_uncond : clip;
                    float v = ct.temb[(size_t)tr * MST_D + f] + ct.textproj[(size_t)tp * MST_D + f] + pe[f];
                    const size_t o = (size_t)clip * S * MST_D + f;
                    if (pd.thr) v *= drop_mul(pd, (uint32_t)o);
                    const f16 h = (f16)v;
                    hi[o] = h;
                    lo[o] = (f16)(v - (float)h);
                }
        }
    }
};

// K9 + K10/K10'/K11/K12 on the ring: output projection with the diffusion update fused in.
// Accumulators (lane = frame, registers = features; CFG halves blended first, model/cfg_sampler.py:43) are
// transposed through LDS into [feature][frame] rows; then ALL 512 threads walk (feature, 4 consecutive
// frames) items with float4 accesses to x_t / mask / motion / noise / outputs -- the [clip][feature][frame]
// tensors are frame-contiguous, so a wave touches 256-B runs.  (First version: per-lane scalar gathers
// straight from the accumulator layout, all work in 4-5 of the 8 waves: 36 us even at batch 16.)
// MODE 0 model output only, 1 ancestral step, 2 DDIM step; NX = 2 is the CFG doubled batch.
template <int MODE>
struct DEpiEmbedOut {
    const float* bias; int F, T, total; float* out; StepArgs sa;
    // xt_next != null (sampling loop, T % 4 == 0): the updated clip is ALSO written as the f16 frame rows [token][kpad] the NEXT
    // step's pose-embedding GEMM stages -- bit for bit what k_frames_f16 would make of it -- so that step needs no transpose launch
    f16* xt_next = nullptr; int kpad = 0; f16* xt_next_lo = nullptr;      // lo: f16(x - hi), the second half of the pose embedding's hi + lo operand
    __device__ __forceinline__ int rows() const { return total; }
    template <int BT, int BF> static constexpr int smem_bytes() { return BF * (BT + 4) * 4; }

    __device__ __forceinline__ void one(const StepArgs& sa, const StepCoef& sc, float mo, size_t idx, float nz, bool blend, bool use_mask) const {
        if (MODE == 0) { out[idx] = mo; return; }
        const float mk = use_mask ? sa.mask[idx] : 0.f, mot = blend ? sa.motion[idx] : 0.f;
        float pred;
        const float nx = step_update<MODE == 2 ? 1 : 0>(sc, mo, sa.x[idx], nz, blend, mk, mot, sa.mask_noise && use_mask, sa.clip, &pred);
        sa.sample[idx] = nx;
        if (sa.xstart) sa.xstart[idx] = pred;
    }

    template <int BT, int BF, int MT, int NT, int NX = 1>
    __device__ __forceinline__ void run(f32x16 (&acc)[NX][MT][NT], int tok0, int f0, char* smem) const {
        constexpr int LDT = BT + 4;                          // floats per feature row of the tile
        DLane<BT, BF, MT, NT> lc;
        const StepArgs sa = step_resolve(this->sa);          // loop mode: tensors and step index come from device memory
        float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const int tl = lc.tok(m), tok = tok0 + tl;
            float gs = 0.f;
            if (NX == 2) gs = sa.scale[(tok < total ? tok : total - 1) / T];
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int f = f0 + lc.feat(n, r >> 2) + (r & 3);
                    float v = acc[0][m][n][r];
                    if (NX == 2) { const float u = acc[NX - 1][m][n][r]; v = u + gs * (v - u); }
                    if (f < F) tile[f * LDT + tl] = v;
                }
        }
        __syncthreads();
        finish<BT>(sa, tok0, f0, smem);
    }
    // everything behind the accumulator -> [feature][frame] tile hop (smem: F rows of BT + 4 floats): the diffusion update with float4
    // accesses along the frames, the next step's f16 frame rows.  Shared by the ring GEMM above and k_embed_out (mst_embed.h).
    template <int BT>
    __device__ __forceinline__ void finish(const StepArgs& sa, int tok0, int f0, char* smem) const {
        constexpr int LDT = BT + 4;
        float* tile = reinterpret_cast<float*>(smem);
        StepCoef sc;
        if (MODE != 0) sc = step_coef(sa.tab, sa.nsteps, sa.t, sa.eta);
        const bool blend = sa.mask != nullptr && sa.motion != nullptr, use_mask = sa.mask != nullptr;
        const bool use_noise = !sa.philox && sa.noise != nullptr;
        const bool vec = (T & 3) == 0;                       // 4 consecutive frames never straddle a clip
        constexpr int TG = BT / 4;
        if (vec && MODE != 0) {
            // Three items per thread and pass, loads first: one item at a time the loop is a chain of dependent global loads (x, the row
            // flag, then mask / motion) per iteration, nine iterations deep at F = 263 -- latency, not bandwidth.
            constexpr int U = 3;
            for (int it0 = threadIdx.x; it0 < F * TG; it0 += 512 * U) {
                bool ok[U];
                int fu[U], tgu[U], clipu[U], tu[U], rf[U];
                size_t idx[U];
                float bu[U];
                f32x4 xv[U], nz[U], mk[U], mot[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int it = it0 + 512 * u;
                    fu[u] = it / TG;
                    tgu[u] = it - fu[u] * TG;
                    const int tok = tok0 + tgu[u] * 4;
                    ok[u] = it < F * TG && tok < total;
                    clipu[u] = tok / T;
                    tu[u] = tok - clipu[u] * T;
                    idx[u] = ((size_t)clipu[u] * F + fu[u]) * T + tu[u];
                    xv[u] = nz[u] = mk[u] = mot[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    rf[u] = 2;                                          // 0: mask row all zeros, 1: all ones, 2: read it
                    bu[u] = 0.f;
                    if (ok[u]) {
                        xv[u] = *reinterpret_cast<const f32x4*>(sa.x + idx[u]);
                        if (sa.rowflag) rf[u] = sa.rowflag[clipu[u] * F + fu[u]];
                        bu[u] = bias[fu[u]];
                        if (use_noise) nz[u] = *reinterpret_cast<const f32x4*>(sa.noise + idx[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (!ok[u]) continue;
                    if (use_mask) {
                        if (rf[u] == 2) mk[u] = *reinterpret_cast<const f32x4*>(sa.mask + idx[u]);
                        else if (rf[u] == 1) mk[u] = f32x4{1.f, 1.f, 1.f, 1.f};
                    }
                    if (blend && rf[u] != 0) mot[u] = *reinterpret_cast<const f32x4*>(sa.motion + idx[u]);   // mask 0: motion * 0 is +-0 whatever it holds
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (!ok[u]) continue;
                    const f32x4 acc4 = *reinterpret_cast<const f32x4*>(tile + fu[u] * LDT + tgu[u] * 4);
                    f32x4 mo;
#pragma unroll
                    for (int j = 0; j < 4; j++) mo[j] = acc4[j] + bu[u];
                    if (sa.philox) {
                        float nrm[4];
                        philox_normal4((unsigned)(tu[u] >> 2), (unsigned)fu[u], (unsigned)clipu[u] + sa.clip0, sa.step, sa.seed, nrm);
#pragma unroll
                        for (int j = 0; j < 4; j++) nz[u][j] = nrm[j];
                    }
                    f32x4 nx, pred;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float p;
                        nx[j] = step_update<MODE == 2 ? 1 : 0>(sc, mo[j], xv[u][j], nz[u][j], blend, mk[u][j], mot[u][j], sa.mask_noise && use_mask, sa.clip, &p);
                        pred[j] = p;
                    }
                    *reinterpret_cast<f32x4*>(sa.sample + idx[u]) = nx;
                    if (sa.xstart) *reinterpret_cast<f32x4*>(sa.xstart + idx[u]) = pred;
                    if (xt_next) *reinterpret_cast<f32x4*>(tile + fu[u] * LDT + tgu[u] * 4) = nx;      // this item's own slot of the tile
                }
            }
        } else
        for (int it = threadIdx.x; it < F * TG; it += 512) {
            const int f = it / TG, tg = it - f * TG;
            const int tok = tok0 + tg * 4;
            if (tok >= total) continue;
            const f32x4 acc4 = *reinterpret_cast<const f32x4*>(tile + f * LDT + tg * 4);
            const float b = bias[f];
            if (vec) {                                                  // MODE 0: the plain output projection
                const int clip = tok / T, t = tok - clip * T;
                const size_t idx = ((size_t)clip * F + f) * T + t;
                f32x4 mo;
#pragma unroll
                for (int j = 0; j < 4; j++) mo[j] = acc4[j] + b;
                *reinterpret_cast<f32x4*>(out + idx) = mo;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int tk = tok + j;
                    if (tk >= total) break;
                    const int clip = tk / T, t = tk - clip * T;
                    const size_t idx = ((size_t)clip * F + f) * T + t;
                    float nz = use_noise ? sa.noise[idx] : 0.f;
                    if (MODE != 0 && sa.philox) {
                        float nrm[4];
                        philox_normal4((unsigned)(t >> 2), (unsigned)f, (unsigned)clip + sa.clip0, sa.step, sa.seed, nrm);
                        nz = nrm[t & 3];
                    }
                    one(sa, sc, acc4[j] + b, idx, nz, blend, use_mask);
                }
            }
        }
        if (MODE != 0 && vec) frames_next<BT>(tok0, f0, smem);
    }
    // second pass over the tile, now holding x_{t-1}: the NEXT step's f16 frame rows (hi + lo), [frame][kpad]
    template <int BT>
    __device__ __forceinline__ void frames_next(int tok0, int f0, char* smem) const {
        constexpr int LDT = BT + 4;
        const float* tile = reinterpret_cast<const float*>(smem);
        if (MODE != 0 && xt_next) {
            // lanes walk the frames of one 8-feature group (conflict-free LDS reads), each writing 16 bytes of its frame's row;
            // columns [F, kpad) are the zero padding of the GEMM's K
            __syncthreads();
            const int groups = kpad / 8;
            for (int it = threadIdx.x; it < BT * groups; it += 512) {
                const int tl = it % BT, fg = it / BT, tok = tok0 + tl;
                if (tok >= total) continue;
                f16x8 v, vl;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int f = f0 + 8 * fg + q;
                    const float xv = f < F ? tile[f * LDT + tl] : 0.f;
                    v[q] = (f16)xv;
                    vl[q] = (f16)(xv - (float)v[q]);
                }
                *reinterpret_cast<f16x8*>(xt_next + (size_t)tok * kpad + f0 + 8 * fg) = v;
                if (xt_next_lo) *reinterpret_cast<f16x8*>(xt_next_lo + (size_t)tok * kpad + f0 + 8 * fg) = vl;
            }
        }
    }
};

template <int BT, int BF, int MT, int NT, int NS, int NX, class SRC, class EPI, int BK = 32, int XS = 1>
__global__ __launch_bounds__(512) void k_gemm_dma(SRC xs, const f16* __restrict__ W, int ldw, int K, int xcd_ny, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = DTile<BT, BF, MT, NT, NS, NX, BK, XS>;
    int bx = blockIdx.x, by = blockIdx.y;
    if (xcd_ny > 0) {
        // XCD-aware order (1-D launch over ceil(nx/8)*8*ny ids): workgroups are dealt round-robin over the
        // 8 XCDs, so give every feature tile of one token tile the same id % 8: the token rows are then
        // fetched into ONE XCD's L2 instead of up to ny of them.  Speed only, never correctness.
        const int ny = xcd_ny, id = blockIdx.x, xcd = id & 7, s = id >> 3;
        bx = (s / ny) * 8 + xcd;
        by = s % ny;
        if (bx * BT >= epi.rows()) return;                        // padding ids of the last group of 8 token tiles
    }
    const int tok0 = bx * BT;
    const int f0 = by * BF;
    f32x16 acc[NX][MT][NT];
#pragma unroll
    for (int x = 0; x < NX; x++)
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[x][m][n][r] = 0.f;
    gemm_mainloop_dma<TL, BT, BF, MT, NT, NX, SRC>(smem, xs, W, ldw, tok0, f0, K, acc);
    if constexpr (NX == 1) epi.template run<BT, BF, MT, NT>(acc, tok0, f0, smem);
    else epi.template run<BT, BF, MT, NT, NX>(acc, tok0, f0, smem);
}

}  // namespace mst
