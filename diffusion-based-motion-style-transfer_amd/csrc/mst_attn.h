// K5: softmax(Q K^T / sqrt(128)) V for one (clip, head) per workgroup, S <= 224 tokens, no mask
// (the denoiser passes no key-padding mask, mdm_forstyledataset.py:622).
//
// The whole K and V of one head live in LDS (2 x S x 128 f16 <= 112 KB); each wave owns one 32-query
// tile and keeps the full score row block in registers, so softmax is exact two-pass (no online
// rescaling).  Both products keep the QUERY on the MFMA lane:
//     St[key][q] = K . Q^T        A = K fragment (ds_read_b128),  B = Q fragment (global -> regs)
//     Ot[d][q]   = V^T . P^T      A = V^T fragment (ds_read_b64_tr_b16), B = P straight from the
//                                 St accumulators (rows = keys sit in registers, cdna guide section 3
//                                 "accumulator tile as the next MFMA's operand")
// so row max / row sum are in-lane reductions plus one cross-half shuffle, 1/l is a per-lane
// scalar, and the output store is 8 bytes per lane (4 consecutive d of one query).
#pragma once
#include <type_traits>

#include "mst_common.h"
#include "mst_gemm_dma.h"   // ring_off, DmaPlan, wait_vmcnt

#ifndef MST_QA_OUT
// k_qkv_attention2's output stores.  0: 8 bytes per lane straight from the accumulator layout (rounds 2-4: partial 128-byte lines; written
// through they cost 25.9 -> 30.3 us per launch, round 3).  1: through the wave's own (dead) Q rows, then 16 bytes per lane = whole lines,
// plain.  2 (default since round 5): those whole-line stores write-through (`sc1`), which takes the 12.9 MB of att out of the dirty bytes
// the launch boundary waits for (profiles/r03_launch_boundary_probe.txt).  Same-box A/B of three library builds, three interleaved
// rounds (tools/r5_qa_out_ab.sh, profiles/r05_ab_attention_output_stores.txt): 104.85 / 104.48 / 105.42 clips/s for 0 / 1 / 2.
#define MST_QA_OUT 2
#endif
#ifndef QA_MARK              // probes/attn_clock.hip defines it (under MST_PROBE_BUILD) to stamp the phases; the product build has none
#define QA_MARK(i)
#endif

namespace mst {

// K image: 256-B rows, 16-B chunk ch of row `row` at chunk ch ^ (row & 15)  (ds_read_b128, T2)
__device__ __forceinline__ int k_off(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }
// V image: 256-B rows cut into four 64-B pieces (one 32-wide d tile each); piece p of key `key`
// at piece p ^ (key & 3): the 4 keys x 64 B a half-wave's transposed read touches cover all 64 banks.
__device__ __forceinline__ int v_off(int key, int d) { return key * 256 + ((((d >> 5) ^ (key & 3))) << 6) + ((d & 31) << 1); }
// The same image for a writer that stores 8 bytes (4 consecutive d) per lane with 16 consecutive keys per store group -- the fused
// QKV + attention kernel's accumulators.  In v_off those 16 rows meet on two bank pairs (8-way: the round-2 PMC profile's
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 18 % was exactly these 16 + 16 stores per wave); here the 8-byte slot inside the
// 64-byte piece is rotated by (key >> 1) & 7 as well, which puts the 16 rows on 16 different bank pairs.  A half-wave's
// transposed read takes the WHOLE piece of each of its four keys, so any per-key permutation of a piece's slots leaves it
// conflict-free.
__device__ __forceinline__ int v_off8(int key, int d) {
    return key * 256 + ((((d >> 5) ^ (key & 3))) << 6) + (((((d & 31) >> 2) ^ ((key >> 1) & 7))) << 3) + ((d & 3) << 1);
}

// qsplit = 1: grid.y = NKT and the workgroup computes only query tile blockIdx.y (small batches: rows * 4 workgroups
// would leave the chip idle; re-staging K and V seven times is cheap when only a few clips run).
template <int NKT>   // number of 32-key tiles = ceil(S / 32), 1..7
// out_lo != null: also f16(o - hi), so that the out-projection can multiply the attention output as hi + lo (RowsDirect::Xlo).
__global__ __launch_bounds__(512) void k_attention(const f16* __restrict__ qkv, f16* __restrict__ out, int S, int qsplit, f16* __restrict__ out_lo = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KEYS = NKT * 32;
    char* ks = smem;
    char* vs = smem + KEYS * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int clip = blockIdx.x / MST_H, head = blockIdx.x % MST_H;
    const f16* base = qkv + (size_t)clip * S * (3 * MST_D) + head * MST_HD;

    // ---- stage K and V (zero rows beyond S so P = 0 meets V = 0, never NaN)
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
    if (wave >= NKT) return;        // query tiles = key tiles; no barrier below
    int qtile = wave;
    if (qsplit) {
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

    // ---- scores St[key][q]
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

    // ---- softmax over keys (rows of St): registers, then the other half-wave
    const float scale = 0.08838834764831845f;   // 1/sqrt(128)
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = sc[kt][r] * scale;
            if (kt == NKT - 1) {
                int key = kt * 32 + mfma_row(r, lane);
                if (key >= S) v = -INFINITY;
            }
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

    // P as f16 B-operand fragments: k-step s2 of key tile kt = registers 8*s2 .. 8*s2+7
    f16x8 pf[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int j = 0; j < 8; j++) pf[kt][s2][j] = (f16)sc[kt][8 * s2 + j];

    // ---- Ot[d][q] = sum_key V[key][d] * P[q][key]
    // transposed read: lane L of 16-lane group g supplies row (L&15)>>2, columns 4*(L&3).. of a
    // 4-key x 16-d block and receives column (L&15) of the 4 keys.  Block for (kt, s2, half):
    // keys kt*32 + 16*s2 + 4*hh (+8 for fragment elements 4..7), d = dt*32 + 16*(g&1) ..
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
                int d = dt * 32 + d_lane;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(vs + v_off(key, d)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(vs + v_off(key + 8, d)));
                // whole-vector bit casts only: __builtin_bit_cast on an ext-vector ELEMENT (lo[j])
                // makes clang (ROCm 7.2) read element 0 for every j.
                const f16x4 lo_h = __builtin_bit_cast(f16x4, lo), hi_h = __builtin_bit_cast(f16x4, hi);
                f16x8 vf = __builtin_shufflevector(lo_h, hi_h, 0, 1, 2, 3, 4, 5, 6, 7);
                o = mfma_f16(vf, pf[kt][s2], o);
            }
        if (q_idx < S) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                int dd = dt * 32 + 8 * gq + 4 * hh;
                const f32x4 ov = {o[4 * gq] * inv_l, o[4 * gq + 1] * inv_l, o[4 * gq + 2] * inv_l, o[4 * gq + 3] * inv_l};
                uint2 h, l;
                split4_f16(ov, h, l);
                *reinterpret_cast<uint2*>(orow + dd) = h;
                if (out_lo) *reinterpret_cast<uint2*>(out_lo + (orow - out) + dd) = l;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// K4 + K5 fused: QKV projection of ONE head + attention, one workgroup per (clip, head).
//
// Why: as separate kernels the f16 `qkv` tensor (38.7 MB at batch 64) is written by one launch and read back by
// the next, and the QKV GEMM restages every weight tile once per 128-token block (228 MB of L2->LDS traffic per
// layer, the per-CU DMA rate being what bounds these loops).  Here a workgroup streams its head's 384 weight rows
// (q|k|v x 128) and the clip's S token rows through the DMA ring once (581 KB), wave w accumulating token tile w
// against all 12 feature tiles; then
//   * Q never leaves registers: the accumulator layout (lane = token, registers = features) IS the B-operand
//     fragment of St = K.Q^T in the permuted k order slot(hh, j) <-> d = 16 ks + 8 (j >> 2) + 4 hh + (j & 3)
//     (cdna guide section 3, "accumulator tile as the next MFMA's operand");
//   * K is written to its LDS image with the 4-element groups of every 16-d run swapped the same way, so the
//     A-operand read stays ONE ds_read_b128 at the usual chunk 2 ks + hh;
//   * V is written in natural order into the transposed-read image.
// The attention core that follows is k_attention's.  `qkv` is not materialised at all.
// ------------------------------------------------------------------------------------------------------------
constexpr int QA_BK = 64;     // slab depth; same-box A/B: 64-deep slabs (whole cache lines) 38.7 vs 39.6 us per launch for 32-deep
template <int NKT>
struct QATile {
    static constexpr int KDEPTH = QA_BK, RB = QA_BK * 2, RPP = 1024 / RB, CPR = RB / 16;   // slab depth (see DTile)
    static constexpr int XR = NKT * 32;                 // token rows staged per slab
    static constexpr int WR = 3 * MST_HD;               // 384 weight rows: q | k | v of this head
    static constexpr int ROWS = XR + WR;
    static constexpr int STAGE = ROWS * RB;
    static constexpr int NSTAGE = QA_BK == 64 ? 2 : 3;
    static constexpr int INSTR = ROWS / RPP;
    static constexpr int PER = (INSTR + 7) / 8;
    static constexpr int RING = NSTAGE * STAGE;
    static constexpr int XROWS = XR, XHI = XR;          // (no lo tensor here)
    static constexpr int KV = NKT * 32 * 256 * 2;       // K and V images afterwards (reuse the ring)
    static constexpr int OFF_BIAS = RING > KV ? RING : KV;   // q | k | v bias of the head (384 floats), staged once at kernel start
    static constexpr int SMEM = OFF_BIAS + 3 * MST_HD * 4;
};

template <int NKT>
__global__ __launch_bounds__(512) void k_qkv_attention(const f16* __restrict__ hx, const f16* __restrict__ w_in,
                                                       const float* __restrict__ b_in, f16* __restrict__ out, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = QATile<NKT>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    QA_MARK(0)
    // Workgroups are dealt round-robin over the 8 XCDs (id % 8).  The four heads of a clip read the SAME token rows, so give
    // them the same id % 8: the rows are then fetched into one XCD's L2 once instead of into four (PMC: 55 MB of HBM reads per
    // 64-clip launch with the plain (clip, head) = (id / 4, id % 4) order, 4x the algorithmic 13 MB).
    const int nclip = gridDim.x / MST_H, full = (nclip / 8) * 8 * MST_H;
    int clip, head;
    if ((int)blockIdx.x < full) {
        const int grp = blockIdx.x >> 5, within = blockIdx.x & 31;
        clip = grp * 8 + (within & 7);
        head = within >> 3;
    } else {
        const int r = blockIdx.x - full;
        clip = (nclip / 8) * 8 + r / MST_H;
        head = r % MST_H;
    }
    const int hh = lane >> 5, l31 = lane & 31;
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const f16* xbase = hx + (size_t)clip * S * MST_D;
    // The head's projection bias goes to LDS now: read from global memory in the epilogue, its ~1 us of load latency sat in
    // front of the K / V image stores with every wave waiting (the barriers of the projection loop publish these writes).
    float* bias_s = reinterpret_cast<float*>(smem + TL::OFF_BIAS);
    if (tid < 3 * MST_HD) bias_s[tid] = b_in[(tid >> 7) * MST_D + head * MST_HD + (tid & 127)];

    // ---- DMA plan: tile rows [0, XR) = the clip's tokens (clamped), [XR, XR+384) = q|k|v weight rows of the head
    auto rowbyte = [&](int row) {
        if (row < TL::XR) return (unsigned)(row < S ? row : S - 1) * (unsigned)(MST_D * 2);
        const int r = row - TL::XR;
        return (unsigned)((r >> 7) * MST_D + head * MST_HD + (r & 127)) * (unsigned)(MST_D * 2);
    };
    const char* xb = reinterpret_cast<const char*>(xbase);
    const char* wb = reinterpret_cast<const char*>(w_in);
    DmaPlan<TL> plan;
    plan.init(wave, lane, rowbyte);
    f32x16 acc[12];
#pragma unroll
    for (int n = 0; n < 12; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;

    constexpr int KT = MST_D / TL::KDEPTH, AHEAD = TL::NSTAGE - 1;
    const bool active = wave < NKT;
#pragma unroll
    for (int s = 0; s < AHEAD; s++) plan.issue(smem_base, s, s, xb, wb);
    // Balanced roles at S = 193..224 (7 token tiles, the headline shape).  A wave per token tile x all 12 feature tiles leaves
    // the 8th wave idle while the SIMDs that host two waves do 24 MFMA tiles per k-step and the fourth does 12.  Waves are
    // placed on SIMDs in pairs (w, w + 4) (MI355X_MICROARCH.md, LDS section: cyclic 0->2->1->3), so waves 4..6 hand the last
    // three feature tiles (v, d = 32..127) of their token tiles to wave 7: every SIMD then carries 21 tiles.  Placement is a
    // speed assumption only; any mapping computes the same numbers.
    constexpr bool BAL = NKT == 7;
    const int nfeat = (BAL && wave >= 4) ? 9 : 12;      // feature tiles this wave accumulates for its own token tile
    // The role is decided OUTSIDE the k loop (three instantiations of the same loop with identical waits, barriers and DMA
    // issue): a role branch inside the loop made hipcc merge the accumulator sets of all roles and spill at the 256-VGPR cap.
    // ROLE 12 / 9: own token tile x that many feature tiles; ROLE 0: DMA and barriers only; ROLE -1: the helper wave.
    auto run_loop = [&](auto role_tag) {
        constexpr int ROLE = decltype(role_tag)::value;
#pragma unroll 1
        for (int kt = 0; kt < KT; kt++) {
            if (AHEAD >= 2 && KT - 1 - kt >= AHEAD - 1) wait_vmcnt<(AHEAD - 1) * TL::PER>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + AHEAD < KT) plan.issue(smem_base, kt + AHEAD, kt + AHEAD, xb, wb);
            const char* st = smem + (kt % TL::NSTAGE) * TL::STAGE;
            if constexpr (ROLE == -1) {
                // helper wave: acc[3 j + i] = token tile 4 + j  x  feature tile 9 + i
#pragma unroll
                for (int ks = 0; ks < TL::KDEPTH / 16; ks++) {
                    const int c = ks * 2 + hh;
                    f16x8 wf[3], xf[3];
#pragma unroll
                    for (int i = 0; i < 3; i++) wf[i] = *reinterpret_cast<const f16x8*>(st + ring_off_rb<TL::RB>(TL::XR + (9 + i) * 32 + l31, c));
#pragma unroll
                    for (int j = 0; j < 3; j++) xf[j] = *reinterpret_cast<const f16x8*>(st + ring_off_rb<TL::RB>((4 + j) * 32 + l31, c));
#pragma unroll
                    for (int j = 0; j < 3; j++)
#pragma unroll
                        for (int i = 0; i < 3; i++) acc[3 * j + i] = mfma_f16(wf[i], xf[j], acc[3 * j + i]);
                }
            } else if constexpr (ROLE > 0) {
#pragma unroll
                for (int ks = 0; ks < TL::KDEPTH / 16; ks++) {
                    const int c = ks * 2 + hh;
                    const f16x8 xf = *reinterpret_cast<const f16x8*>(st + ring_off_rb<TL::RB>(wave * 32 + l31, c));
#pragma unroll
                    for (int n = 0; n < ROLE; n++) {
                        const f16x8 wf = *reinterpret_cast<const f16x8*>(st + ring_off_rb<TL::RB>(TL::XR + n * 32 + l31, c));
                        acc[n] = mfma_f16(wf, xf, acc[n]);
                    }
                }
            }
        }
    };
    if (BAL && wave == 7) run_loop(std::integral_constant<int, -1>());
    else if (!active) run_loop(std::integral_constant<int, 0>());
    else if (BAL && wave >= 4) run_loop(std::integral_constant<int, 9>());
    else run_loop(std::integral_constant<int, 12>());
    __builtin_amdgcn_s_barrier();                       // ring dead: reuse it for the K / V images
    QA_MARK(1)

    char* ks_img = smem;
    char* vs_img = smem + NKT * 32 * 256;
    f16x8 qf[8];
    const int tok = wave * 32 + l31;                    // this lane's token = query = key row
    if (active) {
        const float scale = 0.08838834764831845f;       // 1/sqrt(128), applied to q in fp32 before rounding
        const float* bq = bias_s;
        const float* bk = bias_s + MST_HD;
        const float* bv = bias_s + 2 * MST_HD;
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int d = n * 32 + 8 * g + 4 * hh;  // 4 consecutive d of this head held in registers 4g..4g+3
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bq + d);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bk + d);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(bv + d);
                // Q fragment: k-step 2n + (g >> 1), elements 4 (g & 1) .. +3
#pragma unroll
                for (int i = 0; i < 4; i++) qf[2 * n + (g >> 1)][4 * (g & 1) + i] = (f16)((acc[n][4 * g + i] + b0[i]) * scale);
                // K image: column = 32 n + 16 (g >> 1) + 8 hh + 4 (g & 1)  (the 4-groups of each 16-d run swapped)
                const int kcol = n * 32 + 16 * (g >> 1) + 8 * hh + 4 * (g & 1);
                *reinterpret_cast<uint2*>(ks_img + k_off(tok, kcol >> 3) + (kcol & 7) * 2) =
                    pack4_f16(acc[4 + n][4 * g] + b1[0], acc[4 + n][4 * g + 1] + b1[1], acc[4 + n][4 * g + 2] + b1[2], acc[4 + n][4 * g + 3] + b1[3]);
                // V image: natural order; key rows beyond S must be zero (P = 0 there, never 0 * garbage)
                if (8 + n < nfeat) {                            // balanced roles: d = 32..127 of token tiles 4..6 comes from wave 7
                    uint2 vv = pack4_f16(acc[8 + n][4 * g] + b2[0], acc[8 + n][4 * g + 1] + b2[1], acc[8 + n][4 * g + 2] + b2[2], acc[8 + n][4 * g + 3] + b2[3]);
                    if (tok >= S) vv = make_uint2(0u, 0u);
                    *reinterpret_cast<uint2*>(vs_img + v_off8(tok, d)) = vv;
                }
            }
    } else if (BAL && wave == 7) {
        const float* bv = bias_s + 2 * MST_HD;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int tk = (4 + j) * 32 + l31;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int d = (1 + i) * 32 + 8 * g + 4 * hh;
                    const f32x4 b2 = *reinterpret_cast<const f32x4*>(bv + d);
                    uint2 vv = pack4_f16(acc[3 * j + i][4 * g] + b2[0], acc[3 * j + i][4 * g + 1] + b2[1], acc[3 * j + i][4 * g + 2] + b2[2],
                                         acc[3 * j + i][4 * g + 3] + b2[3]);
                    if (tk >= S) vv = make_uint2(0u, 0u);
                    *reinterpret_cast<uint2*>(vs_img + v_off8(tk, d)) = vv;
                }
        }
    }
    __syncthreads();
    QA_MARK(2)
    if (!active) return;

    // ---- attention core (as k_attention, Q already in registers)
    f32x16 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; kt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) sc[kt][r] = 0.f;
        const int row = kt * 32 + l31;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            f16x8 kf = *reinterpret_cast<const f16x8*>(ks_img + k_off(row, 2 * s + hh));
            sc[kt] = mfma_f16(kf, qf[s], sc[kt]);
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = sc[kt][r];
            if (kt == NKT - 1) {
                int key = kt * 32 + mfma_row(r, lane);
                if (key >= S) v = -INFINITY;
            }
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
    QA_MARK(3)
    f16x8 pf[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int j = 0; j < 8; j++) pf[kt][s2][j] = (f16)sc[kt][8 * s2 + j];

    const int i16 = lane & 15, g16 = lane >> 4;
    const int key_lane = 4 * hh + (i16 >> 2);
    const int d_lane = 16 * (g16 & 1) + 4 * (i16 & 3);
    const int q_ld = tok < S ? tok : S - 1;
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
                int d = dt * 32 + d_lane;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vs_img + v_off8(key, d)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vs_img + v_off8(key + 8, d)));
                const f16x4 lo_h = __builtin_bit_cast(f16x4, lo), hi_h = __builtin_bit_cast(f16x4, hi);
                f16x8 vf = __builtin_shufflevector(lo_h, hi_h, 0, 1, 2, 3, 4, 5, 6, 7);
                o = mfma_f16(vf, pf[kt][s2], o);
            }
        if (tok < S) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                int dd = dt * 32 + 8 * gq + 4 * hh;
                *reinterpret_cast<uint2*>(orow + dd) =
                    pack4_f16(o[4 * gq] * inv_l, o[4 * gq + 1] * inv_l, o[4 * gq + 2] * inv_l, o[4 * gq + 3] * inv_l);
            }
        }
    }
    QA_MARK(4)
}

// ------------------------------------------------------------------------------------------------------------
// Round 3: the same fusion with the operand roles of the projection swapped.
//
// In k_qkv_attention a wave owns a 32-token tile and multiplies it with all 384 weight rows of the head, so the WEIGHTS are
// the operand every wave reads: 384 KB of weight slabs + 224 KB of token slabs go through the LDS-DMA ring (ten pieces per
// wave and slab in front of its MFMAs: in-kernel stamps, 17.9 us = 8.1 us of DMA + 10.4 us of MFMA, in series) and every wave
// reads every weight slab back (3 MB of ds_read per workgroup = the LDS read peak for ~10 us).  Here a wave owns 48 of the 384
// weight rows for ALL tokens of the clip:
//   * its weights are wave-private, so they never touch the LDS: W_in is pre-packed per (head, wave) as 48 fragments of 1 KB in
//     the 16x16x32 A-operand lane order (k_pack_qkv) and streamed L2 -> VGPR with hand-counted waits, as in the layer tail;
//   * only the token rows go through the ring (26 KB per 64-deep slab at S = 197: 4 pieces per wave instead of 10), and the LDS
//     reads drop to 8 x 208 KB;
//   * 16-token MFMA tiles: 197 tokens pad to 208, not 224 (7 % fewer MFMAs), and all 8 waves carry the same 39 tiles.
// q, k and v then all go to LDS images (a wave holds 48 features of every token, not a query tile): [K | V | Q | 16 pad rows],
// 16 NT16 rows each.  The attention core reads 32-row tiles; when NT16 is odd its last tile runs 16 rows into the next image:
// K's into V rows 0..15 (keys >= S, masked to -inf), V's into Q rows 0..15 (finite numbers times P = 0), Q's into the pad rows
// (queries >= S, never stored).
//
// vmcnt bookkeeping (everything a wave issues is hand-counted; P = DMA pieces per wave and slab; R(k)a / R(k)b = the three weight
// fragments of slab k's first / second k-step; D(k) = the DMA pieces of token slab k; completion is in issue order):
//     prologue   R(0)a R(0)b D(0) D(1) D(2)
//     slab kt    wait D(kt), barrier | wait R(kt)a, 39 MFMAs, load R(kt+1)a | wait R(kt)b, 39 MFMAs, load R(kt+1)b | issue D(kt+3)
// A wait names the number of operations issued BEHIND the awaited one.  D(kt+3) goes out at the END of slab kt so that the first
// wait that forces it (R(kt+2)a, loaded behind it) comes two slabs later.
// ------------------------------------------------------------------------------------------------------------
template <int NT16>
struct QA2Tile {
    static constexpr int KDEPTH = 64, RB = 128, RPP = 8, CPR = 8;      // 64-deep token slabs, 128-B rows (see DTile)
    static constexpr int ROWS = 16 * NT16, XROWS = ROWS, XHI = ROWS;    // every ring row is a token row
    static constexpr int STAGE = ROWS * RB, NSTAGE = 4, INSTR = ROWS / RPP, PER = (INSTR + 7) / 8, RING = NSTAGE * STAGE;
    static constexpr int KT = MST_D / KDEPTH;                           // 8 slabs
    static constexpr int NKT = (NT16 + 1) / 2;                          // 32-key tiles of the attention core
    static constexpr int IMG = ROWS * 256;
    static constexpr int OFF_K = 0, OFF_V = IMG, OFF_Q = 2 * IMG, OFF_BIAS = 3 * IMG;   // bias: 1.5 KB at the head of the 16 pad rows
    static constexpr int SMEM = 3 * IMG + 16 * 256;
    static constexpr int WAVE_FRAGS = 3 * (MST_D / 32);                 // 48 fragments of 1 KB per wave: [k32][16-row group]
    static constexpr size_t LAYER_BYTES = (size_t)MST_H * 8 * WAVE_FRAGS * 1024;        // = 3 x 512 x 512 f16
    static_assert(RING <= OFF_BIAS && SMEM <= 163840 && NT16 >= 1 && NT16 <= 13, "LDS map");
};

// W_in ([1536][512] f16, torch in_proj layout: q rows | k rows | v rows) -> per (head, wave) fragment streams.  Fragment
// (k32, i) of wave w, lane l = 8 consecutive k (32 k32 + 8 (l >> 4) ..) of row 48 w + 16 i + (l & 15) of the head's q|k|v rows.
__global__ __launch_bounds__(256) void k_pack_qkv(const f16* __restrict__ w_in, f16* __restrict__ dst) {
    constexpr int NF = 3 * (MST_D / 32);
    const int total = MST_H * 8 * NF * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, f = (i >> 6) % NF, wave = ((i >> 6) / NF) & 7, head = (i >> 6) / (NF * 8);
        const int k32 = f / 3, g = f % 3;
        const int r = 48 * wave + 16 * g + (lane & 15);                 // 0..383: q | k | v row of the head
        const int row = (r >> 7) * MST_D + head * MST_HD + (r & 127);
        reinterpret_cast<uint4*>(dst)[i] = *reinterpret_cast<const uint4*>(w_in + (size_t)row * MST_D + 32 * k32 + 8 * (lane >> 4));
    }
}

template <int N> __device__ __forceinline__ void qa_wwait3(u32x4& a, u32x4& b, u32x4& c) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}
__device__ __forceinline__ void qa_glds1(unsigned voff, unsigned long long sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

// The kernel's body as a function of (clip, head): k_qkv_attention2 = one call per workgroup; the resident-group trunk (mst_trunk.h) calls it
// once per layer with PERSIST = true: the workgroup's weight / bias prefetch is issued, THEN it waits for its clip's stream rows
// (group_wait), the attention output goes through the wave's own (dead) Q rows so that it leaves as 16-byte write-through stores in
// whole 128-byte lines, and every wave returns (no early exit: the caller's barriers want all eight).
template <int NT16, bool PERSIST>
__device__ __forceinline__ void qa2_body(char* smem, const f16* __restrict__ hx, const f16* __restrict__ wq, const float* __restrict__ b_in,
                                         f16* __restrict__ out, int S, int clip, int head, const GroupSync sync, bool wait_input, int wave_in) {
    using TL = QA2Tile<NT16>;
    constexpr int NKT = TL::NKT, P = TL::PER, KT = TL::KT;
    // (PERSIST: the lane index is recomputed in every phase and the wave index arrives in a scalar register, so that no per-lane
    // constant stays alive across the other phases' bodies)
    int lane, wave;
    if constexpr (PERSIST) { lane = lane_id_now(); wave = opaque_uniform(wave_in); }      // (opaque: nothing derived from it is hoisted out of the phase loop)
    else { lane = threadIdx.x & 63; wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
    const int tid = wave * 64 + lane;
    (void)tid;
    QA_MARK(0)
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const char* xb = reinterpret_cast<const char*>(hx + (size_t)clip * S * MST_D);
    // the head's q | k | v bias (3 x 512 B) -> LDS by two DMA pieces of wave 0: the oldest operations of its queue
    if (wave == 0) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int reg = 2 * p + (lane >> 5) < 2 ? 2 * p + (lane >> 5) : 2;
            qa_glds1((unsigned)((reg * MST_D + head * MST_HD) * 4 + (lane & 31) * 16), (unsigned long long)b_in,
                     __builtin_amdgcn_readfirstlane(smem_base + TL::OFF_BIAS + p * 1024));
        }
    }
    // weight stream of this wave: fragment n at byte 1024 n
    const unsigned long long wbase = (unsigned long long)(reinterpret_cast<const char*>(wq) + (size_t)(head * 8 + wave) * TL::WAVE_FRAGS * 1024);
    const unsigned wvoff = lane * 16;
    u32x4 q[6];
#define QA2_LOAD(slot, n) tail_wload<((n) & 3) * 1024>(q[slot], wvoff, wbase + (unsigned long long)(((n) & ~3) * 1024))
    QA2_LOAD(0, 0); QA2_LOAD(1, 1); QA2_LOAD(2, 2); QA2_LOAD(3, 3); QA2_LOAD(4, 4); QA2_LOAD(5, 5);
    DmaPlan<TL> plan;
    plan.init(wave, lane, [&](int row) { return (unsigned)(row < S ? row : S - 1) * (unsigned)(MST_D * 2); });
    if constexpr (PERSIST) {
        if (wait_input) group_wait(sync, wave);               // the clip's stream rows are complete (and this CU holds no stale copy of them)
    }
#pragma unroll
    for (int s = 0; s < 3; s++) plan.issue(smem_base, s, s, xb, xb);

    f32x4 acc[3][NT16];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int t = 0; t < NT16; t++) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // token fragment of tile t, k-step j of a slab: row 16 t + (lane & 15), 16-B chunk 4 j + (lane >> 4)
    const int r15 = lane & 15, q4 = lane >> 4;
    const int xoff0 = r15 * 128 + (((0 + q4) ^ (r15 >> 1)) << 4), xoff1 = r15 * 128 + (((4 + q4) ^ (r15 >> 1)) << 4);
    auto kstep = [&](const char* st, int xo, const u32x4& w0, const u32x4& w1, const u32x4& w2) {
        const f16x8 a0 = __builtin_bit_cast(f16x8, w0), a1 = __builtin_bit_cast(f16x8, w1), a2 = __builtin_bit_cast(f16x8, w2);
#pragma unroll
        for (int t = 0; t < NT16; t++) {
            const f16x8 xf = *reinterpret_cast<const f16x8*>(st + xo + t * 2048);
            acc[0][t] = mfma16(a0, xf, acc[0][t]);
            acc[1][t] = mfma16(a1, xf, acc[1][t]);
            acc[2][t] = mfma16(a2, xf, acc[2][t]);
        }
        // (Measured and rejected: pinning the issue order with sched_group_barrier so that four token fragments are read ahead of
        // their MFMAs -- hipcc keeps two in flight -- makes the projection 14.6 -> 13.7 us in the back-to-back probe and the 64-clip
        // loop 2 % SLOWER: this kernel 26.0 -> 26.5 us, the layer tail behind it 43.8 -> 44.8 us.  The package is at its power
        // limit; stalls removed without work removed are paid back in clock.)
    };
    auto slab = [&](auto kt_tag) {
        constexpr int kt = decltype(kt_tag)::value;
        constexpr int dprev = (kt >= 1 && kt + 2 < KT) ? P : 0;             // D(kt+2) went out at the end of slab kt-1
        constexpr int top = kt == 0 ? 2 * P : kt == 1 ? 2 * P + 6 : 12 + P * ((kt + 1 < KT ? 1 : 0) + (kt + 2 < KT ? 1 : 0));
        constexpr int w0 = kt == 0 ? 3 + 3 * P : 3 + dprev;
        constexpr int w1 = kt == 0 ? 3 + 3 * P : dprev + (kt + 1 < KT ? 3 : 0);
        wait_vmcnt<top>();
        __builtin_amdgcn_s_barrier();
        const char* st = smem + (kt % TL::NSTAGE) * TL::STAGE;
        qa_wwait3<w0>(q[0], q[1], q[2]);
        kstep(st, xoff0, q[0], q[1], q[2]);
        if constexpr (kt + 1 < KT) { QA2_LOAD(0, 6 * (kt + 1)); QA2_LOAD(1, 6 * (kt + 1) + 1); QA2_LOAD(2, 6 * (kt + 1) + 2); }
        qa_wwait3<w1>(q[3], q[4], q[5]);
        kstep(st, xoff1, q[3], q[4], q[5]);
        if constexpr (kt + 1 < KT) { QA2_LOAD(3, 6 * (kt + 1) + 3); QA2_LOAD(4, 6 * (kt + 1) + 4); QA2_LOAD(5, 6 * (kt + 1) + 5); }
        if constexpr (kt + 3 < KT) plan.issue(smem_base, kt + 3, kt + 3, xb, xb);
    };
    static_assert(KT == 8, "eight slabs, written out");
    slab(std::integral_constant<int, 0>()); slab(std::integral_constant<int, 1>()); slab(std::integral_constant<int, 2>());
    slab(std::integral_constant<int, 3>()); slab(std::integral_constant<int, 4>()); slab(std::integral_constant<int, 5>());
    slab(std::integral_constant<int, 6>()); slab(std::integral_constant<int, 7>());
#undef QA2_LOAD
    __builtin_amdgcn_s_barrier();                       // ring dead: the images take its place
    QA_MARK(1)

    char* ks_img = smem + TL::OFF_K;
    char* vs_img = smem + TL::OFF_V;
    char* qs_img = smem + TL::OFF_Q;
    {
        const float scale = 0.08838834764831845f;       // 1/sqrt(128), applied to q in fp32 before rounding
        const float* bias_s = reinterpret_cast<const float*>(smem + TL::OFF_BIAS);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int f0 = 48 * wave + 16 * i;          // wave-uniform: the 16-row group lies inside q, k or v
            const int region = f0 >> 7, d = (f0 & 127) + 4 * q4;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + region * MST_HD + d);
            if (region < 2) {
                // q (scaled) and k: TWO token tiles per store.  v_permlane16_swap trades the odd 16-lane rows of one tile's packed
                // values with the even rows of the next tile's (lane map: csrc/probes/probe_permlane.hip), after which a lane holds 8
                // consecutive features of ONE token: rows q4 = 0, 2 of tile t, rows 1, 3 of tile t + 1.  One ds_write_b128 per lane
                // instead of two ds_write_b64 -- and conflict-free: 8-byte stores of 16 consecutive rows meet two by two on the
                // 128-byte bank row of LDS writes whatever 16-byte-chunk swizzle the image has (all of this kernel's bank-conflict
                // cycles were these stores, profiles/r03_pmc_lds_by_phase.txt).
                char* img = region == 0 ? qs_img : ks_img;
                const float mul = region == 0 ? scale : 1.0f;
                const int ch = ((f0 & 127) >> 3) + (q4 >> 1);
#pragma unroll
                for (int t = 0; t + 1 < NT16; t += 2) {
                    const f32x4 va = (acc[i][t] + b4) * mul, vb = (acc[i][t + 1] + b4) * mul;
                    const uint2 pa = pack4_f16(va[0], va[1], va[2], va[3]), pb = pack4_f16(vb[0], vb[1], vb[2], vb[3]);
                    const auto r0 = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
                    *reinterpret_cast<uint4*>(img + k_off(16 * (t + (q4 & 1)) + r15, ch)) = uint4{r0[0], r1[0], r0[1], r1[1]};
                }
                if (NT16 & 1) {
                    const f32x4 v = (acc[i][NT16 - 1] + b4) * mul;
                    *reinterpret_cast<uint2*>(img + k_off(16 * (NT16 - 1) + r15, d >> 3) + (d & 7) * 2) = pack4_f16(v[0], v[1], v[2], v[3]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT16; t++) {
                    const f32x4 v = acc[i][t] + b4;
                    uint2 vv = pack4_f16(v[0], v[1], v[2], v[3]);
                    if (16 * t + r15 >= S) vv = make_uint2(0u, 0u);      // key rows beyond S must be zero (P = 0 there, never 0 * garbage)
                    *reinterpret_cast<uint2*>(vs_img + v_off8(16 * t + r15, d)) = vv;
                }
            }
        }
    }
    __syncthreads();
    QA_MARK(2)
    if constexpr (!PERSIST) {
        if (wave >= NKT) return;
    }
    if (wave < NKT) {

    // ---- attention core (k_attention's; q from its image, natural d order)
    const int hh = lane >> 5, l31 = lane & 31;
    const int tok = wave * 32 + l31;                    // this lane's query
    f16x8 qf[8];
#pragma unroll
    for (int s = 0; s < 8; s++) qf[s] = *reinterpret_cast<const f16x8*>(qs_img + k_off(tok, 2 * s + hh));
    f32x16 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; kt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) sc[kt][r] = 0.f;
        const int row = kt * 32 + l31;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            f16x8 kf = *reinterpret_cast<const f16x8*>(ks_img + k_off(row, 2 * s + hh));
            sc[kt] = mfma_f16(kf, qf[s], sc[kt]);
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = sc[kt][r];
            if (kt == NKT - 1) {
                int key = kt * 32 + mfma_row(r, lane);
                if (key >= S) v = -INFINITY;
            }
            sc[kt][r] = v;
            m = fmaxf(m, v);
        }
    if constexpr (PERSIST) m = xor32_max(m); else m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float p = __expf(sc[kt][r] - m);
            sc[kt][r] = p;
            l += p;
        }
    if constexpr (PERSIST) l = xor32_add(l); else l += __shfl_xor(l, 32);
    const float inv_l = 1.0f / l;
    QA_MARK(3)
    f16x8 pf[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int j = 0; j < 8; j++) pf[kt][s2][j] = (f16)sc[kt][8 * s2 + j];

    const int i16 = lane & 15, g16 = lane >> 4;
    const int key_lane = 4 * hh + (i16 >> 2);
    const int d_lane = 16 * (g16 & 1) + 4 * (i16 & 3);
    const int q_ld = tok < S ? tok : S - 1;
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
                int d = dt * 32 + d_lane;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vs_img + v_off8(key, d)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vs_img + v_off8(key + 8, d)));
                const f16x4 lo_h = __builtin_bit_cast(f16x4, lo), hi_h = __builtin_bit_cast(f16x4, hi);
                f16x8 vf = __builtin_shufflevector(lo_h, hi_h, 0, 1, 2, 3, 4, 5, 6, 7);
                o = mfma_f16(vf, pf[kt][s2], o);
            }
        if constexpr (PERSIST || MST_QA_OUT != 0) {
            // -> the wave's own Q rows (read into qf above, by this wave only; rows 208.. are pad rows), natural d order
#pragma unroll
            for (int gq = 0; gq < 4; gq++)
                *reinterpret_cast<uint2*>(qs_img + k_off(tok, dt * 4 + gq) + 8 * hh) =
                    pack4_f16(o[4 * gq] * inv_l, o[4 * gq + 1] * inv_l, o[4 * gq + 2] * inv_l, o[4 * gq + 3] * inv_l);
        } else if (tok < S) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                int dd = dt * 32 + 8 * gq + 4 * hh;
                *reinterpret_cast<uint2*>(orow + dd) =
                    pack4_f16(o[4 * gq] * inv_l, o[4 * gq + 1] * inv_l, o[4 * gq + 2] * inv_l, o[4 * gq + 3] * inv_l);
            }
        }
    }
    if constexpr (PERSIST || MST_QA_OUT != 0) {
        // the tile leaves as whole lines: one wave instruction = 4 query rows x 256 B (the head's 128 features), 16 B per lane, write-through
        asm volatile("" ::: "memory");                      // the read-back below is of another type than the stores above: no reordering across
        const int ch = lane & 15;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int row = wave * 32 + 4 * j + (lane >> 4);
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(qs_img + k_off(row, ch));
            f16* dst = out + ((size_t)clip * S + row) * MST_D + head * MST_HD + ch * 8;
            if (row < S) {
                if constexpr (PERSIST || MST_QA_OUT == 2) store16_sc1(dst, v);
                else *reinterpret_cast<u32x4_t*>(dst) = v;
            }
        }
    }
    QA_MARK(4)
    }
}

template <int NT16>
__global__ __launch_bounds__(512) void k_qkv_attention2(const f16* __restrict__ hx, const f16* __restrict__ wq,
                                                        const float* __restrict__ b_in, f16* __restrict__ out, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (clip, head) of this workgroup: the four heads of a clip share id % 8 = one XCD's L2 (see k_qkv_attention)
    const int nclip = gridDim.x / MST_H, full = (nclip / 8) * 8 * MST_H;
    int clip, head;
    if ((int)blockIdx.x < full) {
        const int grp = blockIdx.x >> 5, within = blockIdx.x & 31;
        clip = grp * 8 + (within & 7);
        head = within >> 3;
    } else {
        const int r = blockIdx.x - full;
        clip = (nclip / 8) * 8 + r / MST_H;
        head = r % MST_H;
    }
    qa2_body<NT16, false>(smem, hx, wq, b_in, out, S, clip, head, GroupSync{nullptr, 0u, nullptr}, false, 0);
}

}  // namespace mst
