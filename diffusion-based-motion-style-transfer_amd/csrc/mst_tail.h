// K6 + K7 + K8 fused: everything of an encoder layer behind the attention, one workgroup per 64-token tile:
//
//     x1 = LayerNorm1(x + att W_out^T + b_out)        (self-attention block's tail, mdm_forstyledataset.py:539-546 /
//     h  = GELU(x1 W1^T + b1)                           nn.TransformerEncoderLayer post-norm forward)
//     x2 = LayerNorm2(x1 + h W2^T + b2)
//
// Round 3 design: the weights never touch the LDS.  A 64-token tile needs all 2.5 MB of W_out | W1 | W2, and in this tiling a
// weight fragment is used by exactly ONE wave (wave w owns weight rows [32 w, 32 w + 32) of every 256-row block, for all 64
// tokens).  Round 2 pushed them through an LDS-DMA ring anyway: 75-90 GB/s per CU, 40 us of streaming per tile against 19 us of
// matrix work.  Measured on MI355X (csrc/probes/wstream.hip, 197 workgroups streaming the same layer): plain
// global_load_dwordx4 into VGPRs reads the same bytes at ~160 GB/s per CU with nothing else running and sustains 83 % of the
// MFMA issue rate with the token fragments coming from LDS.  So:
//
//   * W_out | W1 | W2 are pre-packed per layer (k_pack_tail) as EIGHT per-wave streams of 320 fragments of 1 KB, each the A
//     operand of v_mfma_f32_16x16x32_f16 for 16 weight rows x 32 k in lane order: one wave-instruction = one coalesced 1-KB load.
//     A wave keeps TailCfg::D fragments in flight (4 VGPRs each) behind hand-counted s_waitcnt vmcnt(D - 1); the stream runs
//     through the LayerNorm and GELU stages without draining, so no phase starts on a cold pipeline.
//   * The shared operand -- the tile's 64 token rows -- lives in LDS for the whole phase: att (64 KB, one LDS-DMA burst at kernel
//     start), x1 (64 KB f16 image written by LayerNorm1; no global round trip any more), the GELU output (32 KB per 256-feature
//     chunk, double-buffered: ONE barrier per chunk instead of one per 32-KB weight slab).
//   * 16x16x32 tiles instead of 32x32x16: same registers, same LDS bytes, +16 % in the probe at the clock the chip holds.
//   * LayerNorm1 runs in the accumulator layout (statistics: wave-level sums + a 4 KB exchange between the eight waves): its fp32
//     output stays in the accumulators as FFN2's initial value, so the residual of LayerNorm2 costs no registers during the FFN.
//
//   phase P   16 k-steps of 32: acc[nh][rb][tb] += W_out[256 nh + 32 w + 16 rb .., k] . att[16 tb .., k]^T            (4 fragments / step)
//   LN1       in the accumulators: + b_out + stream (hi / lo rows staged in the x1 image area during phase P) -> LayerNorm -> x1: f16 image, fp32 stays
//   phase F   4 chunks of 256 hidden features: FFN1 16 k-steps (2 fragments / step) -> GELU -> H image -> barrier -> FFN2 8 k-steps (4)
//   LN2       accumulators (x1 + FFN2) -> scratch -> rows (+ b2) -> LayerNorm -> the stream (hi/lo), in place for the next layer
#pragma once
#include "mst_common.h"
#include "mst_gemm_dma.h"

#ifndef MST_TAIL_OUT
#define MST_TAIL_OUT 0        // k_layer_tail's LayerNorm2 stores: 0 = 8 bytes per lane, plain; 1 = those write-through; 2 = 16 bytes per lane (a lane holds
                              // 8 consecutive features), plain; 3 = those write-through (tools/r5_tail_out_ab.sh)
#endif
#ifndef MST_TAIL_AWAIT_FLAT
#define MST_TAIL_AWAIT_FLAT 0
#endif
#ifndef TAIL_MARK            // probes/tail_clock.hip defines it (with MST_PROBE_BUILD) to stamp the phases; the product build has none
#define TAIL_MARK(i)
#endif

namespace mst {

struct TailCfg {
    static constexpr int BT = 64;                        // tokens per workgroup
    static constexpr int D = 16;                         // weight fragments in flight per wave (64 VGPRs; 128 KB per CU).  8 measures the same (43.2 vs 43.1 us)
    static constexpr int P_FRAG = 64, F1_FRAG = 32, F2_FRAG = 32, NFRAG = P_FRAG + 4 * (F1_FRAG + F2_FRAG);   // per wave
    static constexpr size_t WAVE_BYTES = (size_t)NFRAG * 1024, LAYER_BYTES = 8 * WAVE_BYTES;                    // 2.5 MB per layer
    static constexpr int LN_LD = MST_D * 4 + 16;         // LayerNorm scratch row: 512 fp32 + 16 B (conflict-free 16-B accesses)
    static constexpr int OFF_ATT = 0;                    // phase P: att image, 64 x 1 KB
    static constexpr int OFF_H = 0, HBUF = 32 * 1024;    // phase F: GELU output, 2 x (64 x 512 B)
    static constexpr int OFF_CNT = 68 * 1024;            // 4 arrival counters of the FFN chunks (above the att / H images, below OFF_TAB)
    // Phi(x) table of the GELU stage: entry i = {Phi(x_i), Phi(x_i+1) - Phi(x_i)}, x_i = -6 + i / 128, i < 1536 (+ one guard entry),
    // padded to 13 KB = 13 LDS-DMA pieces.  Linear interpolation error <= h^2 / 8 max|Phi''| = 1.9e-6, far below the f16 store (2^-11).
    static constexpr int GELU_N = 1536, GELU_TAB_BYTES = 13 * 1024;
    static constexpr int OFF_TAB = 69 * 1024;
    static constexpr int OFF_BO_TR = 82 * 1024, OFF_G1_TR = 88 * 1024, OFF_BE1_TR = 90 * 1024;   // TRAIN: b_out | LayerNorm1 weight | bias, 2 KB each, staged at kernel start
    static constexpr int OFF_EXCH_TR = 84 * 1024;        // TRAIN: LayerNorm1's 4 KB statistics exchange (the att image holds the residual's lo rows then)
    static constexpr int OFF_B1 = 92 * 1024;             // FFN1 bias (4 KB), staged at kernel start
    static constexpr int OFF_X1 = 96 * 1024;             // LayerNorm1 output, 64 x 1 KB; during phase P the residual rows (lo, then hi) are staged here
    static constexpr int SMEM = 160 * 1024;
    static_assert(64 * 1024 <= OFF_CNT && OFF_CNT + 16 <= OFF_TAB && OFF_TAB + GELU_TAB_BYTES <= OFF_B1 && 64 * LN_LD <= SMEM && 2 * HBUF <= OFF_CNT &&
                  OFF_TAB + GELU_TAB_BYTES <= OFF_BO_TR && OFF_BO_TR + 2048 <= OFF_EXCH_TR && OFF_EXCH_TR + 4096 <= OFF_G1_TR &&
                  OFF_G1_TR + 2048 <= OFF_BE1_TR && OFF_BE1_TR + 2048 <= OFF_B1, "LDS map");
    static_assert(P_FRAG % D == 0 && F1_FRAG % D == 0 && F2_FRAG % D == 0, "every phase starts on prefetch slot 0");
    static_assert(F1_FRAG == 32 && F2_FRAG == 32, "k_pack_tail's unit arithmetic");
};

// Pack one layer's three matrices ([out][in] f16, torch Linear layout) into the eight per-wave fragment streams.
// Fragment f of wave w, lane l (r = l & 15, kq = l >> 4) = 8 consecutive k of one weight row: the 16x16x32 A operand.
__global__ __launch_bounds__(256) void k_pack_tail(const f16* __restrict__ w_out, const f16* __restrict__ w1,
                                                   const f16* __restrict__ w2, f16* __restrict__ dst) {
    using C = TailCfg;
    const int total = 8 * C::NFRAG * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, f = (i >> 6) % C::NFRAG, wave = (i >> 6) / C::NFRAG;
        const int r = 32 * wave + (lane & 15), kq = 8 * (lane >> 4);
        const f16* src;
        if (f < C::P_FRAG) {
            const int k32 = f >> 2, nh = (f >> 1) & 1, rb = f & 1;
            src = w_out + (size_t)(256 * nh + 16 * rb + r) * MST_D + 32 * k32 + kq;
        } else {
            // FFN units of 32 fragments in consumption order: F1(0) F1(1) F2(0) F1(2) F2(1) F1(3) F2(2) F2(3) -- FFN2 of chunk hc runs
            // beside the GELU of chunk hc + 1, so FFN1 is one chunk ahead
            const int u = f - C::P_FRAG, unit = u >> 5, v = u & 31;
            const bool is2 = unit == 7 || (unit >= 2 && !(unit & 1));
            const int hc = unit == 7 ? 3 : is2 ? (unit >> 1) - 1 : unit == 0 ? 0 : (unit + 1) >> 1;
            if (!is2) {
                const int k32 = v >> 1, rb = v & 1;
                src = w1 + (size_t)(256 * hc + 16 * rb + r) * MST_D + 32 * k32 + kq;
            } else {
                const int k32 = v >> 2, nh = (v >> 1) & 1, rb = v & 1;
                src = w2 + (size_t)(256 * nh + 16 * rb + r) * MST_FF + 256 * hc + 32 * k32 + kq;
            }
        }
        reinterpret_cast<uint4*>(dst)[i] = *reinterpret_cast<const uint4*>(src);
    }
}

__device__ __forceinline__ void tail_fence() { asm volatile("" ::: "memory"); }
// add_half() is INLINE ASM (v_fma_mix_f32), and hipcc's hazard recognizer does not look inside inline asm: where it reads a register an
// MFMA has just written, nothing inserts the wait states the hardware needs between an XDL write and a VALU read of the same VGPR (it
// is not interlocked).  hipcc also moves MFMAs -- register-only instructions -- across asm barriers, so round 3's kernel was right only
// by the distance the scheduler happened to leave (found in round 4: the 48-token instantiation interleaved pass 1's last MFMAs with the
// lo-residual adds and came out 7 % wrong).  tail_acc_settle(): no instruction crosses, and every MFMA issued in front of it has written
// its result behind it (v_mfma_f32_16x16x32_f16 -> VALU read: 8 wait states by hipcc's own count; 10 here).
// tools/audit_asm_hazards.py checks the whole class on the compiled library (every inline-asm instruction against the wait states hipcc
// itself keeps behind an MFMA); `tools/audit_lib.sh --selftest` compiles this header with MST_AUDIT_SELFTEST_NO_SETTLE to prove that the
// audit reports the kernel without the settle.
__device__ __forceinline__ void tail_acc_settle() {
#ifndef MST_AUDIT_SELFTEST_NO_SETTLE
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 9" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
}
__device__ __forceinline__ void tail_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // LDS only: the stream stays in flight

// GELU(x) = x Phi(x), Phi by linear interpolation in the LDS table (TailCfg::OFF_TAB): 7 VALU ops + one 8-byte LDS read per value
// instead of the 12 ops + v_rcp + v_exp of gelu_erf.  The tail's FFN phase is bound by the SIMDs' VALU + MFMA issue slots
// (128 GELU values per lane and tile), so the instruction count is what this buys.  |x| > 6: Phi is 0 / 1 to 1e-9.
__device__ __forceinline__ float gelu_tab_lds(float x, const char* tab) {
    const float u = __builtin_amdgcn_fmed3f(fmaf(x, 128.0f, 768.0f), 0.0f, 1535.9999f);
    const float2 e = *reinterpret_cast<const float2*>(tab + ((unsigned)u << 3));
    return x * fmaf(e.y, __builtin_amdgcn_fractf(u), e.x);
}

// TRAIN instantiation (round 6: the training forward of a 64-clip call / the frozen motion encoder on the fused kernel).  Same tiling,
// same weight stream; what changes:
//   * dropout at the layer's three sites behind the attention (out-proj output, hidden, FFN2 output: counter-based keep masks, mst_train.h
//     `Drop`, the very counters the unfused epilogues and the backward kernels use), so the residuals can no longer ride in the
//     accumulators: LayerNorm1's residual is added behind the out-proj loop (hi rows staged during the loop as before, lo rows into the
//     dead att image behind it), LayerNorm2's is re-read from the x1 tape slot;
//   * the layer's tape slots are written from where the values are: z1 / x1 (hi + lo) from the accumulator layout, pre / hid from the
//     GELU stage (8 bytes per lane, inside the FFN2 passes: stores count in vmcnt IN ISSUE ORDER with the weight fragments, so every
//     hand-counted wait of the stream stays sufficient -- it waits for a few stores more than it needs), z2 and the layer output as rows;
//   * input and output streams are different buffers (tape slots l and l + 1).
// Dropout needs mst_train.h's Drop; the struct is repeated here as a POD so that this header does not depend on the training header.
struct TailDrop { uint32_t key, thr; float inv; };
__device__ __forceinline__ uint32_t tail_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
#if MST_TT_NOHASH          // A/B probe builds only (never the product): what the keep-mask hash costs inside the fused training tail
__device__ __forceinline__ float tail_drop_mul(const TailDrop& d, uint32_t idx) { return d.inv; }
#else
__device__ __forceinline__ float tail_drop_mul(const TailDrop& d, uint32_t idx) { return tail_mix32(idx * 0x9E3779B9u + d.key) >= d.thr ? d.inv : 0.f; }
#endif
// Tape stores.  Non-temporal stores (the slots are written once and read once, by the backward pass, behind a gigabyte of other traffic)
// were measured and LOSE: same box, three alternating runs of 50 fine-tune iterations, 10.03 / 10.09 / 10.13 ms with plain stores against
// 10.51 / 10.60 / 10.61 ms with `nt` ones (64-clip call 1.79 -> 2.01 ms, motion encoder 1.49 -> 1.72 ms; the chained calls beside them
// slower too).  MST_TT_NT=1 builds that variant.
#ifndef MST_TT_NT
#define MST_TT_NT 0
#endif
__device__ __forceinline__ void tape_store8(f16* dst, uint2 v) {
#if MST_TT_NT
    __builtin_nontemporal_store(__builtin_bit_cast(unsigned long long, v), reinterpret_cast<unsigned long long*>(dst));
#else
    *reinterpret_cast<uint2*>(dst) = v;
#endif
}
#if MST_TT_NOSTORE         // A/B probe builds only: the tape stores of the accumulator-layout stages dropped (wrong gradients; timing only)
#define TT_STORE(dst, v) do { } while (0)
#else
#define TT_STORE(dst, v) tape_store8(dst, v)
#endif
__device__ __forceinline__ uint32_t tail_keep_bit(const TailDrop& d, uint32_t idx) { return tail_mix32(idx * 0x9E3779B9u + d.key) >= d.thr ? 1u : 0u; }
struct TailTrain {
    const f16 *xin_h, *xin_l;                                     // the layer's input stream (tape slot l): LayerNorm1's residual
    f16 *z1h, *z1l, *x1h, *x1l, *pre, *hid, *z2h, *z2l;           // tape slots of the layer
    TailDrop d1, d2, d3;                                          // keep masks: out-proj output [tok][512], hidden [tok][1024], FFN2 output [tok][512]
};

__device__ __forceinline__ void tail_glds1(unsigned voff, unsigned long long sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

// NTB = 16-token blocks per tile (tile = 16 NTB tokens): 4 for launches that fill the chip or share it with other clip slices, 3 / 2 for
// a lone launch of fewer tokens.  A tile streams all 2.5 MB of the layer through its CU's L1 (64 B / clk: 41 k cycles) whatever its
// height; at 64 tokens its MFMA work is another 41 k cycles on every SIMD.  csrc/probes/tail_clock.hip, 4 334 tokens alone on the
// chip: 64-token tiles (68 workgroups) 37.5 us per launch, 48-token (91) 32.6 us, 32-token (136) 30.4 us.
// The kernel's body for the tile that starts at token row tok0 (rows >= M are clamped on load and never stored).  k_layer_tail = one call per
// workgroup; the resident-group trunk (mst_trunk.h) calls it once per layer with PERSIST = true: the tables, the FFN1 bias and the first D
// weight fragments are requested, THEN the workgroup waits for its clip's four attention heads (group_wait) and only then requests its
// att rows; LayerNorm2's rows leave as write-through stores.
template <int NTB, bool PERSIST, bool TRAIN = false>
__device__ __forceinline__ void tail_body(char* smem, const f16* __restrict__ att, const f16* __restrict__ wt,
                                          const float* __restrict__ b_out, const float* __restrict__ g1, const float* __restrict__ be1,
                                          const float* __restrict__ b1, const float* __restrict__ b2,
                                          const float* __restrict__ g2, const float* __restrict__ be2,
                                          f16* __restrict__ hx, f16* __restrict__ hl, const float* __restrict__ gelu_tab, int M, int tok0,
                                          const GroupSync sync, int wave_in, const TailTrain& tt = TailTrain{}) {
    using C = TailCfg;
    static_assert(!(TRAIN && PERSIST), "the training forward is one launch per layer");
    const f16* const rin_h = TRAIN ? tt.xin_h : hx;                // LayerNorm1's residual rows (TRAIN: hx / hl are the OUTPUT stream)
    const f16* const rin_l = TRAIN ? tt.xin_l : hl;
    // Fragments in flight per wave.  TRAIN: 8 -- the inference kernel measures the same at 8 and 16, and the 32 registers pay for the
    // training stages' addresses and masks: at 16 hipcc spilled five values, and every scratch reload is a vmcnt(0) that drains the
    // weight stream AND the tape stores in flight (found in the ISA: four such drains in the GELU(0) stage alone).
    constexpr int D = TRAIN ? 8 : C::D;
    static_assert(C::P_FRAG % D == 0 && C::F1_FRAG % D == 0 && C::F2_FRAG % D == 0, "every phase starts on prefetch slot 0");
    // (PERSIST: the lane index is recomputed in every phase and the wave index arrives in a scalar register, so that no per-lane
    // constant stays alive across the other phases' bodies)
    int lane, wave;
    if constexpr (PERSIST) { lane = lane_id_now(); wave = opaque_uniform(wave_in); }      // (opaque: nothing derived from it is hoisted out of the phase loop)
    else { lane = threadIdx.x & 63; wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
    const int tid = wave * 64 + lane;
    const int t16 = lane & 15, q4 = lane >> 4;                       // accumulator map: token 16 tb + t16, features .. + 4 q4 + i
    constexpr int BT = 16 * NTB, RPW = 2 * NTB;                      // tile rows; rows per wave in the row-wise stages (DMA, LayerNorm2)
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    TAIL_MARK(0)

    // ---- kernel-start burst, all LDS-DMA (counted in vmcnt with the weight fragments behind it, no register, no compiler-visible load):
    //  (1) the att image: token row r of the tile = 1 KB, 16-B chunk c at c ^ (r & 15) (conflict-free ds_read_b128 of the 16x16x32 B
    //      operand: a 16-lane group reads 16 rows at chunks {c, c ^ 1} -> 16 distinct slots of the 256-B bank row).  LDS-DMA writes
    //      lane-linearly, so the swizzle is on the per-lane SOURCE address; wave w fills rows [8 w, 8 w + 8);
    //  (2) the FFN1 bias (4 KB, waves 0..3) and the GELU stage's Phi table (13 KB): both are read from LDS;
    //  (3) NOT in this burst: LayerNorm1's residual, the stream rows of the tile.  LayerNorm1 runs in the ACCUMULATOR layout (no
    //      transposition through an fp32 scratch: that was 700 KB of LDS traffic per tile, 6 us), so the residual is needed as
    //      (token, 4 features) pieces per lane: the rows are staged in [OFF_X1, +64 KB) in the same image layout as att and read
    //      with 8-byte LDS loads.  hi + lo is 128 KB and the att image is alive until the loop ends, so the two halves take turns: lo
    //      is requested in front of the loop's first pass and added to the accumulators after its second, hi follows into the same
    //      64 KB during the last two and is added behind the loop; LayerNorm1's x1 image then overwrites hi slot by slot, each slot
    //      by the lane that read it.
    float* b1s = reinterpret_cast<float*>(smem + C::OFF_B1);
    const char* const gtab = smem + C::OFF_TAB;
    unsigned* const arrived = reinterpret_cast<unsigned*>(smem + C::OFF_CNT);      // [chunk]: waves whose GELU output of that chunk is in the H image
    if (tid < 4) arrived[tid] = 0;                                                  // published by the barriers in front of the out-proj loop
    // residual rows of the tile -> [OFF_X1, +64 KB) in the att / x1 image layout; wave w fills rows [8 w, 8 w + 8)
    auto stage_rows = [&](const f16* src, int dst_off = C::OFF_X1) {
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int r = RPW * wave + j;
            int tok = tok0 + r;
            if (tok >= M) tok = M - 1;                                // last tile: clamp (rows beyond M are never stored)
            const unsigned voff = (unsigned)tok * (unsigned)(MST_D * 2) + (unsigned)((lane ^ (r & 15)) << 4);
            tail_glds1(voff, (unsigned long long)src, __builtin_amdgcn_readfirstlane(smem_base + dst_off + r * 1024));
        }
    };
    constexpr int ROW_OPS = RPW;
    auto att_rows = [&]() {
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int r = RPW * wave + j;
            int tok = tok0 + r;
            if (tok >= M) tok = M - 1;
            const unsigned voff = (unsigned)tok * (unsigned)(MST_D * 2) + (unsigned)((lane ^ (r & 15)) << 4);
            tail_glds1(voff, (unsigned long long)att, __builtin_amdgcn_readfirstlane(smem_base + C::OFF_ATT + r * 1024));
        }
    };
    if constexpr (!PERSIST) att_rows();
    if (wave < 4) tail_glds1((unsigned)lane * 16u, (unsigned long long)(b1 + 256 * wave), __builtin_amdgcn_readfirstlane(smem_base + C::OFF_B1 + 1024 * wave));
    tail_glds1((unsigned)lane * 16u, (unsigned long long)(gelu_tab + 256 * wave), __builtin_amdgcn_readfirstlane(smem_base + C::OFF_TAB + 1024 * wave));
    if (wave < 5) tail_glds1((unsigned)lane * 16u, (unsigned long long)(gelu_tab + 256 * (8 + wave)), __builtin_amdgcn_readfirstlane(smem_base + C::OFF_TAB + 1024 * (8 + wave)));
    if constexpr (TRAIN) {
        // LayerNorm1's vectors -> LDS too (waves 0 .. 5, one 1-KB piece each): read with plain global loads in the LayerNorm1 stage, hipcc
        // waits vmcnt(0) for each of them -- with this kernel's tape stores in flight every such wait drained the store queue
        if (wave < 6) {
            const float* src = wave < 2 ? b_out : wave < 4 ? g1 : be1;
            const int off = wave < 2 ? C::OFF_BO_TR : wave < 4 ? C::OFF_G1_TR : C::OFF_BE1_TR;
            tail_glds1((unsigned)lane * 16u, (unsigned long long)(src + 256 * (wave & 1)), __builtin_amdgcn_readfirstlane(smem_base + off + 1024 * (wave & 1)));
        }
    }

    // ---- the weight stream of this wave
    const unsigned w_voff = (unsigned)lane * 16u;
    const char* wnext = reinterpret_cast<const char*>(wt) + (size_t)wave * C::WAVE_BYTES;       // next fragment to request (wave-uniform)
    u32x4 q[D];
    auto issue = [&](auto jc) {                                       // fragment -> prefetch slot j (compile-time)
        constexpr int j = decltype(jc)::value;
        if constexpr (j < D) tail_wload<(j & 3) * 1024>(q[j], w_voff, (unsigned long long)(wnext + (j >> 2) * 4096));
    };
#define TAIL_ISSUE(j) issue(std::integral_constant<int, (j)>())
#pragma unroll
    for (int j = 0; j < D; j++) {
        // (unrolled: j is a constant in every copy)
        switch (j) {
#define TAIL_CASE(J) case J: TAIL_ISSUE(J); break;
            TAIL_CASE(0) TAIL_CASE(1) TAIL_CASE(2) TAIL_CASE(3) TAIL_CASE(4) TAIL_CASE(5) TAIL_CASE(6) TAIL_CASE(7)
            TAIL_CASE(8) TAIL_CASE(9) TAIL_CASE(10) TAIL_CASE(11) TAIL_CASE(12) TAIL_CASE(13) TAIL_CASE(14) TAIL_CASE(15)
#undef TAIL_CASE
        }
    }
    wnext += D * 1024;

    // token fragments of k-step k32 from an image with ROWB-byte rows: lane -> token 16 tb + t16, chunk (4 k32 + q4) ^ t16
    const unsigned xlane1k = (unsigned)t16 * 1024u, xlane512 = (unsigned)t16 * 512u, xswz = (unsigned)((q4 ^ t16) << 4);
    auto xread = [&](const char* img, auto rowb, int k32, f16x8 (&x)[NTB]) {
        constexpr int ROWB = decltype(rowb)::value;
        const char* p = img + (ROWB == 1024 ? xlane1k : xlane512) + (((unsigned)k32 << 6) ^ xswz);
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) x[tb] = *reinterpret_cast<const f16x8*>(p + tb * 16 * ROWB);
    };
    using RB1K = std::integral_constant<int, 1024>;
    using RB512 = std::integral_constant<int, 512>;

    f32x4 acc[2][2][NTB];                   // [feature half nh][16-row block rb][16-token block tb]
    f32x4 acch[2][NTB];                     // FFN1 chunk: [rb][tb]
    f16x8 xs[2][NTB];                       // token fragments, double-buffered by k-step parity

    // One unrolled pass = D fragments (`more`: another pass of the same phase follows).  RA fragments per k-step (4: out-proj / FFN2, both feature halves; 2: FFN1), each against the
    // step's four token fragments.  LOAD = false: the stream's last pass (nothing left to request; the waits count down).
    // loadc: 1 = the pass requests the next D fragments; 0 = the stream's last pass (the waits count down); 1 + 2 E = as 1, and E other
    // vector-memory operations were issued by this wave right in front of the pass (younger than the D fragments in flight, older
    // than the ones it requests): every wait of the pass leaves them out.
    auto pass = [&](const char* img, auto rowb, auto rac, auto loadc, int k32base, bool more, auto side) {
        constexpr int RA = decltype(rac)::value;
        constexpr bool LOAD = (decltype(loadc)::value & 1) != 0;
        constexpr int EXTRA = decltype(loadc)::value >> 1;
        constexpr int STEPS = D / RA;
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            if (s + 1 < STEPS || more) xread(img, rowb, k32base + s + 1, xs[(s + 1) & 1]);   // one step ahead, across passes (STEPS is even); the phase's first by the caller
#pragma unroll
            for (int a = 0; a < RA; a++) {
                constexpr int dummy = 0; (void)dummy;
                const int j = s * RA + a;
                // static slot index: the switch folds after unrolling
                auto use = [&](auto jc) {
                    constexpr int J = decltype(jc)::value < D ? decltype(jc)::value : 0;      // (cases >= D are never taken)
                    if constexpr (LOAD) tail_wwait<D - 1 + EXTRA>(q[J]); else tail_wwait<D - 1 - J>(q[J]);
                    const f16x8 wf = __builtin_bit_cast(f16x8, q[J]);
                    if constexpr (RA == 4) {
                        constexpr int nh = (J >> 1) & 1, rb = J & 1;
#pragma unroll
                        for (int tb = 0; tb < NTB; tb++) acc[nh][rb][tb] = mfma16(wf, xs[(J / RA) & 1][tb], acc[nh][rb][tb]);
                    } else {
                        constexpr int rb = J & 1;
#pragma unroll
                        for (int tb = 0; tb < NTB; tb++) acch[rb][tb] = mfma16(wf, xs[(J / RA) & 1][tb], acch[rb][tb]);
                    }
                    if constexpr (LOAD) tail_wload<(J & 3) * 1024>(q[J], w_voff, (unsigned long long)(wnext + (J >> 2) * 4096));
                };
                switch (j) {
#define TAIL_CASE(J) case J: use(std::integral_constant<int, J>()); break;
                    TAIL_CASE(0) TAIL_CASE(1) TAIL_CASE(2) TAIL_CASE(3) TAIL_CASE(4) TAIL_CASE(5) TAIL_CASE(6) TAIL_CASE(7)
                    TAIL_CASE(8) TAIL_CASE(9) TAIL_CASE(10) TAIL_CASE(11) TAIL_CASE(12) TAIL_CASE(13) TAIL_CASE(14) TAIL_CASE(15)
#undef TAIL_CASE
                }
            }
            side(s);                                                   // VALU work riding in the matrix steps' shadow (the GELU of the next chunk)
        }
        if constexpr (LOAD) wnext += D * 1024;
    };
    using RA4 = std::integral_constant<int, 4>;
    using RA2 = std::integral_constant<int, 2>;
    using LD1 = std::integral_constant<int, 1>;
    using LD0 = std::integral_constant<int, 0>;

    // =========================================================================================== phase P: out-proj
    // accumulator slot (nh, rb, tb) of this lane in an image with 1-KB rows: token row 16 tb + t16, features 256 nh + 32 w + 16 rb + 4 q4 .. + 3
    // = 8-byte half (q4 & 1) of 16-byte chunk 32 nh + 4 w + 2 rb + (q4 >> 1), stored at chunk ^ (row & 15)
    auto slot1k = [&](int nh, int rb, int tb) {
        return (unsigned)((16 * tb + t16) * 1024 + (((32 * nh + 4 * wave + 2 * rb + (q4 >> 1)) ^ t16) << 4) + 8 * (q4 & 1));
    };
    // TRAIN: keep masks of sites 1 and 3 as bit sets, filled in the shadow of matrix steps (see phase P / the last FFN2 chunk)
    uint32_t keep1[2] = {0u, 0u}, keep3[2] = {0u, 0u};
    auto keep1_step = [&](int k) {                                     // accumulator slot k = (nh, rb, tb): bits 4 k .. 4 k + 3
        if constexpr (TRAIN) {
            const int nh = k >> 3, rb = (k >> 2) & 1, tb = k & 3;
            const uint32_t idx = (uint32_t)(tok0 + 16 * tb + t16) * (uint32_t)MST_D + (uint32_t)(256 * nh + 32 * wave + 16 * rb + 4 * q4);
            const uint32_t b = tail_keep_bit(tt.d1, idx) | (tail_keep_bit(tt.d1, idx + 1) << 1) | (tail_keep_bit(tt.d1, idx + 2) << 2) |
                               (tail_keep_bit(tt.d1, idx + 3) << 3);
            keep1[k >> 3] |= b << (4 * (k & 7));
        }
    };
    auto keep3_step = [&](int r) {                                     // LayerNorm2's row r of this wave: bits 8 r .. (features fa .. + 3 | fb .. + 3)
        if constexpr (TRAIN) {
            constexpr bool WIDE_ = MST_TAIL_OUT >= 2;
            const int fa_ = WIDE_ ? lane * 8 : lane * 4, fb_ = WIDE_ ? lane * 8 + 4 : 256 + lane * 4;
            const uint32_t base = (uint32_t)(tok0 + RPW * wave + r) * (uint32_t)MST_D;
            uint32_t b = 0u;
#pragma unroll
            for (int i = 0; i < 4; i++) b |= (tail_keep_bit(tt.d3, base + fa_ + i) << i) | (tail_keep_bit(tt.d3, base + fb_ + i) << (4 + i));
            keep3[r >> 2] |= b << (8 * (r & 3));
        }
    };
#pragma unroll
    for (int nh = 0; nh < 2; nh++)
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int tb = 0; tb < NTB; tb++) acc[nh][rb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PERSIST) {
        group_wait(sync, wave);                                        // the clip's four heads have written att (and this CU holds no stale copy)
        att_rows();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the att rows are the YOUNGEST operations here: tables and fragments landed long ago
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D + 1) : "memory");   // this wave's att rows have landed (behind them: 1 .. 3 table pieces and the D fragments)
    }
    tail_barrier();                                                    // ... and everybody's
    TAIL_MARK(1)
    {
        const char* img = smem + C::OFF_ATT;
        xread(img, RB1K(), 0, xs[0]);
        constexpr int NPP = C::P_FRAG / D;
        static_assert(TRAIN || NPP == 4, "passes 0 | 1: the lo rows land, 2 | 3: the hi rows");
        // Neither half of the residual is part of the kernel-start burst: there, 64 KB more in front of (or behind) the att image
        // delay the first MFMA by 1.5 us (measured both ways).  Requested in front of a pass, the rows travel beside the weight stream.
        if constexpr (TRAIN) {
            // the residual joins BEHIND the dropout of (acc + bias): nothing is added inside the loop.  What the loop's matrix steps DO
            // carry in their shadow is the keep mask of site 1 (out-proj output): k-step k = 4 pass + s hashes the four elements of
            // accumulator slot (nh, rb, tb) = (k >> 3, (k >> 2) & 1, k & 3) into bit 4 k + i of (keep1[0], keep1[1]) -- alone in front of
            // LayerNorm1 the 64 hashes per lane were ~5 us of exposed VALU time.
            constexpr int STP = D / 4;                                 // k-steps per pass
            static_assert(NTB == 4 && NPP * STP == 16, "mask bookkeeping: 16 k-steps = 16 accumulator slots");
#pragma unroll
            for (int pp = 0; pp < NPP; pp++) {
                if (pp == NPP / 2) {                                   // the hi rows travel beside the second half of the loop (as in the inference kernel)
                    stage_rows(rin_h);
                    pass(img, RB1K(), RA4(), std::integral_constant<int, 1 + 2 * ROW_OPS>(), pp * STP, pp + 1 < NPP, [&](int s_) { keep1_step(pp * STP + s_); });
                } else {
                    pass(img, RB1K(), RA4(), LD1(), pp * STP, pp + 1 < NPP, [&](int s_) { keep1_step(pp * STP + s_); });
                }
            }
        } else {
        stage_rows(rin_l);
        pass(img, RB1K(), RA4(), std::integral_constant<int, 1 + 2 * ROW_OPS>(), 0, true, [](int) {});
        pass(img, RB1K(), RA4(), LD1(), D / 4, true, [](int) {});
        // the lo rows are older than the fragments requested by pass 0, which pass 1 has consumed: landed
        tail_acc_settle();                                             // the adds below are inline asm reading MFMA results
        tail_barrier();
        {
            const char* stg = smem + C::OFF_X1;
#pragma unroll
            for (int nh = 0; nh < 2; nh++)
#pragma unroll
                for (int rb = 0; rb < 2; rb++)
#pragma unroll
                    for (int tb = 0; tb < NTB; tb++) {
                        const uint2 l = *reinterpret_cast<const uint2*>(stg + slot1k(nh, rb, tb));
                        f32x4& a = acc[nh][rb][tb];
                        a = f32x4{add_half<0>(l.x, a[0]), add_half<1>(l.x, a[1]), add_half<0>(l.y, a[2]), add_half<1>(l.y, a[3])};
                    }
        }
        tail_barrier();                                                // everybody has read lo: hi may overwrite it
        stage_rows(rin_h);
        pass(img, RB1K(), RA4(), std::integral_constant<int, 1 + 2 * ROW_OPS>(), 2 * (D / 4), true, [](int) {});
        pass(img, RB1K(), RA4(), LD1(), 3 * (D / 4), false, [](int) {});
        }
    }
    TAIL_MARK(2)

    // =========================================================================================== LayerNorm1 (accumulator layout)
    // A lane holds, for each of its four tokens (tb), 16 of the row's 512 values; a wave holds 64 (its 2 x 2 x 16 features).  Row statistics:
    // two-pass mean / M2 over the wave's 64 values (in-lane + the three other q4 groups), then the eight waves' (mean, M2) pairs
    // meet in LDS and are merged exactly (M2 = sum M2_w + 64 sum (mean_w - mean)^2).  The normalised row stays in the accumulators as
    // FFN2's initial value (the x1 residual of LayerNorm2) and goes to the x1 image as f16, the FFN1 operand.
    tail_fence();
    tail_barrier();                                                    // every wave's hi rows are in place (each waited for its own fragments behind them); att is dead
    {
        char* x1img = smem + C::OFF_X1;
        // [wave][token]: a 16-lane group writes / reads 128 contiguous bytes (no bank conflict).  TRAIN: the dead att image receives the
        // residual's lo rows, so the exchange moves into the free 10 KB between the Phi table and the FFN1 bias
        float2* exch = reinterpret_cast<float2*>(smem + (TRAIN ? C::OFF_EXCH_TR : C::OFF_ATT));
        float mw[NTB], m2[NTB];
        if constexpr (TRAIN) {
            stage_rows(rin_l, C::OFF_ATT);                             // behind the barrier above: nobody reads att any more
            tail_acc_settle();                                         // (the adds below are inline asm reading MFMA results)
#pragma unroll
            for (int nh = 0; nh < 2; nh++)
#pragma unroll
                for (int rb = 0; rb < 2; rb++) {
                    const int f = 256 * nh + 32 * wave + 16 * rb + 4 * q4;
                    const f32x4 bo = *reinterpret_cast<const f32x4*>(smem + C::OFF_BO_TR + f * 4);
#pragma unroll
                    for (int tb = 0; tb < NTB; tb++) {
                        const int k = 8 * nh + 4 * rb + tb;
                        const uint32_t kb = keep1[k >> 3] >> (4 * (k & 7));
                        const uint2 h = *reinterpret_cast<const uint2*>(x1img + slot1k(nh, rb, tb));
                        f32x4 a = acc[nh][rb][tb] + bo;
                        a = f32x4{(kb & 1u) ? a[0] * tt.d1.inv : 0.f, (kb & 2u) ? a[1] * tt.d1.inv : 0.f, (kb & 4u) ? a[2] * tt.d1.inv : 0.f,
                                  (kb & 8u) ? a[3] * tt.d1.inv : 0.f};
                        acc[nh][rb][tb] = f32x4{add_half<0>(h.x, a[0]), add_half<1>(h.x, a[1]), add_half<0>(h.y, a[2]), add_half<1>(h.y, a[3])};
                    }
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's lo rows (the youngest operations; the fragments in flight are older)
            tail_barrier();                                            // ... and everybody's
            const char* lo = smem + C::OFF_ATT;
#pragma unroll
            for (int nh = 0; nh < 2; nh++)
#pragma unroll
                for (int rb = 0; rb < 2; rb++)
#pragma unroll
                    for (int tb = 0; tb < NTB; tb++) {
                        const uint2 l = *reinterpret_cast<const uint2*>(lo + slot1k(nh, rb, tb));
                        f32x4& a = acc[nh][rb][tb];
                        a = f32x4{add_half<0>(l.x, a[0]), add_half<1>(l.x, a[1]), add_half<0>(l.y, a[2]), add_half<1>(l.y, a[3])};
                        const int tok = tok0 + 16 * tb + t16;
                        if (tok < M) {                                 // z1 = x + dropout(att W_out^T + b_out): LayerNorm1's input, hi / lo
                            const size_t o = (size_t)tok * MST_D + (256 * nh + 32 * wave + 16 * rb + 4 * q4);
                            uint2 zh, zl;
                            split4_f16(a, zh, zl);
                            TT_STORE(tt.z1h + o, zh);
                            TT_STORE(tt.z1l + o, zl);
                        }
                    }
        } else {
#pragma unroll
        for (int nh = 0; nh < 2; nh++)
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const f32x4 bo = *reinterpret_cast<const f32x4*>(b_out + 256 * nh + 32 * wave + 16 * rb + 4 * q4);
#pragma unroll
                for (int tb = 0; tb < NTB; tb++) {
                    const uint2 h = *reinterpret_cast<const uint2*>(x1img + slot1k(nh, rb, tb));
                    const f32x4 a = acc[nh][rb][tb] + bo;
                    acc[nh][rb][tb] = f32x4{add_half<0>(h.x, a[0]), add_half<1>(h.x, a[1]), add_half<0>(h.y, a[2]), add_half<1>(h.y, a[3])};
                }
            }
        }
        auto quad_sum = [](float v) {                                  // over the four q4 groups (lanes l, l ^ 16, l ^ 32, l ^ 48)
            if constexpr (PERSIST) return xor32_add(xor16_add(v));     // (the same two additions: a + b is commutative bit for bit)
            v += __shfl_xor(v, 16);
            return v + __shfl_xor(v, 32);
        };
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) {
            f32x4 t = (acc[0][0][tb] + acc[0][1][tb]) + (acc[1][0][tb] + acc[1][1][tb]);
            mw[tb] = quad_sum((t[0] + t[1]) + (t[2] + t[3])) * (1.0f / 64.0f);
        }
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) {
            f32x4 sq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nh = 0; nh < 2; nh++)
#pragma unroll
                for (int rb = 0; rb < 2; rb++) {
                    const f32x4 d = acc[nh][rb][tb] - mw[tb];
                    sq = __builtin_elementwise_fma(d, d, sq);
                }
            m2[tb] = quad_sum((sq[0] + sq[1]) + (sq[2] + sq[3]));
            if (q4 == 0) exch[wave * 64 + 16 * tb + t16] = make_float2(mw[tb], m2[tb]);
        }
        tail_barrier();
        TAIL_MARK(6)
        float mean[NTB], rstd[NTB];
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) {
            const float2* e = exch + 16 * tb + t16;
            const float2 e0 = e[0], e1 = e[64], e2 = e[128], e3 = e[192], e4 = e[256], e5 = e[320], e6 = e[384], e7 = e[448];   // (mean_w, M2_w) of waves 0 .. 7
            const float mu = (((e0.x + e1.x) + (e2.x + e3.x)) + ((e4.x + e5.x) + (e6.x + e7.x))) * 0.125f;
            const float d0 = e0.x - mu, d1 = e1.x - mu, d2 = e2.x - mu, d3 = e3.x - mu, d4 = e4.x - mu, d5 = e5.x - mu, d6 = e6.x - mu, d7 = e7.x - mu;
            const float within = ((e0.y + e1.y) + (e2.y + e3.y)) + ((e4.y + e5.y) + (e6.y + e7.y));
            // (every multiply-add spelled as an fma: under -ffp-contract=fast hipcc fuses `a * a + b * b` either way round, and which way has
            // differed between the tile-height instantiations of this function -- they must agree bit for bit)
            const float between = (fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3)) + (fmaf(d4, d4, d5 * d5) + fmaf(d6, d6, d7 * d7));
            mean[tb] = mu;
            rstd[tb] = ln_rstd(fmaf(64.0f, between, within));
        }
#pragma unroll
        for (int nh = 0; nh < 2; nh++)
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const int f = 256 * nh + 32 * wave + 16 * rb + 4 * q4;
                f32x4 g, be;
                if constexpr (TRAIN) {
                    g = *reinterpret_cast<const f32x4*>(smem + C::OFF_G1_TR + f * 4);
                    be = *reinterpret_cast<const f32x4*>(smem + C::OFF_BE1_TR + f * 4);
                } else {
                    g = *reinterpret_cast<const f32x4*>(g1 + f);
                    be = *reinterpret_cast<const f32x4*>(be1 + f);
                }
#pragma unroll
                for (int tb = 0; tb < NTB; tb++) {
                    const f32x4 y = __builtin_elementwise_fma(acc[nh][rb][tb] - mean[tb], g * rstd[tb], be);
                    if constexpr (TRAIN) {
                        // x1 -> the image (FFN1's operand) and its hi half to the tape (dW1's operand).  No lo half: LayerNorm2's residual is
                        // LayerNorm1 evaluated again, row-wise, from the z1 rows this workgroup has just stored.  FFN2 starts from 0.
                        const uint2 yh = pack4_f16(y[0], y[1], y[2], y[3]);
                        *reinterpret_cast<uint2*>(x1img + slot1k(nh, rb, tb)) = yh;
                        const int tok = tok0 + 16 * tb + t16;
                        if (tok < M) TT_STORE(tt.x1h + (size_t)tok * MST_D + f, yh);
                        acc[nh][rb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
                    } else {
                    acc[nh][rb][tb] = y;
                    *reinterpret_cast<uint2*>(x1img + slot1k(nh, rb, tb)) = pack4_f16(y[0], y[1], y[2], y[3]);
                    }
                }
            }
        TAIL_MARK(7)
        tail_barrier();                                                // the x1 image is complete; the exchange area is free (the H image overlays it)
        TAIL_MARK(8)
    }
    tail_fence();
    TAIL_MARK(3)

    // =========================================================================================== phase F: FFN
    // Order: F1(0) G(0) a0 | F1(1) w0 [F2(0) + G(1)] a1 | F1(2) w1 [F2(1) + G(2)] a2 | F1(3) w2 [F2(2) + G(3)] a3 | w3 F2(3).
    // The GELU of a chunk (128 values per lane and tile, ~12 VALU ops each: 1.8 us per chunk when it runs alone, with the matrix
    // pipe idle) rides in the FFN2 steps of the chunk before it: those accumulate into `acc`, the GELU reads the finished `acch`.
    // (Measured and not kept: a second acch set so that the GELU also spreads over the following FFN1 -- the phase is bound by the
    // SIMDs' VALU + MFMA issue slots, 20.4 us of busy time per wave however the ~2 300 GELU issue cycles per chunk are placed.)
    // a / w = a SPLIT barrier on an LDS counter per chunk: a wave announces its GELU output (a), runs the next chunk's FFN1 -- ~2 us
    // that need nobody else's data -- and only then waits for the other seven (w).  LDS operations of a wave are processed in
    // order, so the H stores are in place before the counter moves; whoever reads the counter as 8 reads the data behind it.
    {
        const char* x1img = smem + C::OFF_X1;
        auto ffn1 = [&](int hc) {                                      // acch = b1 + W1 . x1^T of chunk hc (the bias is the initial accumulator)
            const float* bb = b1s + 256 * hc + 32 * wave + 4 * q4;
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bb + 16 * rb);
#pragma unroll
                for (int tb = 0; tb < NTB; tb++) acch[rb][tb] = bv;
            }
            xread(x1img, RB1K(), 0, xs[0]);
#pragma unroll 1
            for (int ps = 0; ps < C::F1_FRAG / D; ps++) pass(x1img, RB1K(), RA2(), LD1(), ps * (D / 2), ps + 1 < C::F1_FRAG / D, [](int) {});
        };
        // GELU(acch[rb][tb]) -> H image of chunk hc: hidden feature 32 w + 16 rb + 4 q4 + i = 8-B half (q4 & 1) of chunk
        // 4 w + 2 rb + (q4 >> 1) of token row 16 tb + t16 (512-B rows, chunk ^ (row & 15)).  Buffer hc & 1: its previous readers (FFN2 of
        // chunk hc - 2) had all announced chunk hc - 1 before this wave got past w(hc - 1).
        auto gelu_group = [&](int hc, int rb, int tb) {
            const unsigned coff = (unsigned)(((4 * wave + 2 * rb + (q4 >> 1)) ^ t16) << 4) + 8u * (q4 & 1);
            const f32x4 v = acch[rb][tb];
            if constexpr (TRAIN) {
                // pre (the f16 the backward's GELU' reads) and hid = dropout(GELU(pre)) -> the tape; hid -> the H image
                const uint2 p16 = pack4_f16(v[0], v[1], v[2], v[3]);
                const f16x4 ph = __builtin_bit_cast(f16x4, p16);
                const int tok = tok0 + 16 * tb + t16;
                const uint32_t o = (uint32_t)tok * (uint32_t)MST_FF + (uint32_t)(256 * hc + 32 * wave + 16 * rb + 4 * q4);
                const uint2 h16 = pack4_f16(gelu_tab_lds((float)ph[0], gtab) * tail_drop_mul(tt.d2, o), gelu_tab_lds((float)ph[1], gtab) * tail_drop_mul(tt.d2, o + 1),
                                            gelu_tab_lds((float)ph[2], gtab) * tail_drop_mul(tt.d2, o + 2), gelu_tab_lds((float)ph[3], gtab) * tail_drop_mul(tt.d2, o + 3));
                *reinterpret_cast<uint2*>(smem + C::OFF_H + (hc & 1) * C::HBUF + (16 * tb + t16) * 512 + coff) = h16;
                if (tok < M) {
                    TT_STORE(tt.pre + o, p16);         // (hid is not stored: the backward regenerates it from pre, OpGeluBwd)
                }
            } else
            *reinterpret_cast<uint2*>(smem + C::OFF_H + (hc & 1) * C::HBUF + (16 * tb + t16) * 512 + coff) =
                pack4_f16(gelu_tab_lds(v[0], gtab), gelu_tab_lds(v[1], gtab), gelu_tab_lds(v[2], gtab), gelu_tab_lds(v[3], gtab));
        };
        auto announce = [&](int hc) {
            tail_fence();
            if (lane == 0) __hip_atomic_fetch_add(arrived + hc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            tail_fence();
        };
        // The poll is a ds_read_b32 written out: through a `volatile unsigned*` hipcc loses the address space (the pointer is a generic one
        // by then) and emits flat_load_dword + s_waitcnt vmcnt(0) lgkmcnt(0) -- every await DRAINED the weight stream (found in round 6 in
        // the ISA of the shipped kernel: four full drains per tile).  MST_TAIL_AWAIT_FLAT=1 brings the old poll back (A/B builds).
        auto await = [&](int hc) {
            tail_fence();
#if MST_TAIL_AWAIT_FLAT
            while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile unsigned*>(arrived + hc)) < 8u) __builtin_amdgcn_s_sleep(1);
#else
            const unsigned caddr = smem_base + C::OFF_CNT + 4u * (unsigned)hc;
            for (;;) {
                unsigned v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(caddr) : "memory");
                if (__builtin_amdgcn_readfirstlane(v) >= 8u) break;
                __builtin_amdgcn_s_sleep(1);
            }
#endif
            tail_fence();
        };
        static_assert(C::F2_FRAG / 4 == 8 && 2 * NTB <= 8, "FFN2 of a chunk = eight k-steps: at most one GELU group (rb, tb) per k-step");
        constexpr int NG = 2 * NTB;                                    // GELU groups per chunk: g -> (rb, tb) = (g / NTB, g % NTB)
        constexpr int NP2 = C::F2_FRAG / D, ST2 = D / 4;               // FFN2 passes per chunk, k-steps per pass
        ffn1(0);
        TAIL_MARK(12)
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int tb = 0; tb < NTB; tb++) gelu_group(0, rb, tb);
        announce(0);
        TAIL_MARK(13)
#pragma unroll 1
        for (int hc = 0; hc < 3; hc++) {
            ffn1(hc + 1);
            TAIL_MARK(14 + 3 * hc)
            await(hc);
            TAIL_MARK(15 + 3 * hc)
            const char* himg = smem + C::OFF_H + (hc & 1) * C::HBUF;
            xread(himg, RB512(), 0, xs[0]);
#pragma unroll
            for (int ps = 0; ps < NP2; ps++)
                pass(himg, RB512(), RA4(), LD1(), ps * ST2, ps + 1 < NP2, [&](int s) { const int g = ps * ST2 + s; if (g < NG) gelu_group(hc + 1, g / NTB, g % NTB); });
            announce(hc + 1);
            TAIL_MARK(16 + 3 * hc)
        }
        {
            await(3);
            TAIL_MARK(23)
            const char* himg = smem + C::OFF_H + C::HBUF;              // chunk 3
            xread(himg, RB512(), 0, xs[0]);
            // (TRAIN: the keep mask of site 3, the FFN2 output, one LayerNorm2 row per k-step -- this chunk's steps carry no GELU)
            static_assert(!TRAIN || (NP2 * ST2 == RPW), "one k-step of the last chunk per LayerNorm2 row of a wave");
#pragma unroll
            for (int ps = 0; ps + 1 < NP2; ps++) pass(himg, RB512(), RA4(), LD1(), ps * ST2, true, [&](int s_) { keep3_step(ps * ST2 + s_); });
            pass(himg, RB512(), RA4(), LD0(), (NP2 - 1) * ST2, false, [&](int s_) { keep3_step((NP2 - 1) * ST2 + s_); });
        }
    }
    tail_fence();
    tail_barrier();                                                    // everybody is done with the H / x1 images: the scratch overlays them
    TAIL_MARK(4)

    // =========================================================================================== LayerNorm2 -> the stream
    {
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) {
            char* trow = smem + (16 * tb + t16) * C::LN_LD;
#pragma unroll
            for (int nh = 0; nh < 2; nh++)
#pragma unroll
                for (int rb = 0; rb < 2; rb++)
                    *reinterpret_cast<f32x4*>(trow + (256 * nh + 32 * wave + 16 * rb + 4 * q4) * 4) = acc[nh][rb][tb];
        }
        tail_barrier();
        // MST_TAIL_OUT < 2: a lane holds features 4 l .. + 3 of both row halves (8-byte stores: 512 contiguous bytes per wave instruction);
        // >= 2: features 8 l .. + 7 (ONE 16-byte store per lane and stream half: a whole 1-KB row per wave instruction)
        constexpr bool WIDE = MST_TAIL_OUT >= 2;
        const int fa = WIDE ? lane * 8 : lane * 4, fb = WIDE ? lane * 8 + 4 : 256 + lane * 4;
        const f32x4 ba = *reinterpret_cast<const f32x4*>(b2 + fa), bb = *reinterpret_cast<const f32x4*>(b2 + fb);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(g2 + fa), gb = *reinterpret_cast<const f32x4*>(g2 + fb);
        const f32x4 ea = *reinterpret_cast<const f32x4*>(be2 + fa), eb = *reinterpret_cast<const f32x4*>(be2 + fb);
        f32x4 xa[RPW], xb[RPW];
        float mean[RPW], rstd[RPW];
        if constexpr (TRAIN) {
            // z2 = x1 + dropout(hid W2^T + b2).  x1 = LayerNorm1(z1) evaluated AGAIN here, row-wise, from the z1 rows (hi + lo) the
            // LayerNorm1 stage stored (every wave has since consumed weight fragments requested behind those stores, and operations complete
            // in issue order: the rows are in L2; this CU never read the lines, so its L1 holds no older copy) -- fp32, what the accumulators
            // held there to ~1e-7; the tape carries no lo half of x1.
            uint2 r1[RPW][4];
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                int tok = tok0 + RPW * wave + r;
                if (tok >= M) tok = M - 1;
                const size_t off = (size_t)tok * MST_D;
                r1[r][0] = *reinterpret_cast<const uint2*>(tt.z1h + off + fa);
                r1[r][1] = *reinterpret_cast<const uint2*>(tt.z1l + off + fa);
                r1[r][2] = *reinterpret_cast<const uint2*>(tt.z1h + off + fb);
                r1[r][3] = *reinterpret_cast<const uint2*>(tt.z1l + off + fb);
            }
            const f32x4 g1a = *reinterpret_cast<const f32x4*>(g1 + fa), g1b = *reinterpret_cast<const f32x4*>(g1 + fb);
            const f32x4 e1a = *reinterpret_cast<const f32x4*>(be1 + fa), e1b = *reinterpret_cast<const f32x4*>(be1 + fb);
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                const char* srow = smem + (RPW * wave + r) * C::LN_LD;
                const int tok = tok0 + RPW * wave + r;
                const uint32_t kb = keep3[r >> 2] >> (8 * (r & 3));
                f32x4 a = *reinterpret_cast<const f32x4*>(srow + fa * 4) + ba, b = *reinterpret_cast<const f32x4*>(srow + fb * 4) + bb;
                a = f32x4{(kb & 1u) ? a[0] * tt.d3.inv : 0.f, (kb & 2u) ? a[1] * tt.d3.inv : 0.f, (kb & 4u) ? a[2] * tt.d3.inv : 0.f, (kb & 8u) ? a[3] * tt.d3.inv : 0.f};
                b = f32x4{(kb & 16u) ? b[0] * tt.d3.inv : 0.f, (kb & 32u) ? b[1] * tt.d3.inv : 0.f, (kb & 64u) ? b[2] * tt.d3.inv : 0.f, (kb & 128u) ? b[3] * tt.d3.inv : 0.f};
                {
                    f32x4 za = join4_f16(r1[r][0], r1[r][1]), zb = join4_f16(r1[r][2], r1[r][3]);
                    const f32x4 t1 = za + zb;
                    const float m1 = wave_sum((t1[0] + t1[1]) + (t1[2] + t1[3])) * (1.0f / MST_D);
                    za -= m1;
                    zb -= m1;
                    const f32x4 q1 = {fmaf(za[0], za[0], zb[0] * zb[0]), fmaf(za[1], za[1], zb[1] * zb[1]), fmaf(za[2], za[2], zb[2] * zb[2]),
                                      fmaf(za[3], za[3], zb[3] * zb[3])};
                    const float rs1 = ln_rstd(wave_sum((q1[0] + q1[1]) + (q1[2] + q1[3])));
                    xa[r] = __builtin_elementwise_fma(za, g1a * rs1, e1a) + a;
                    xb[r] = __builtin_elementwise_fma(zb, g1b * rs1, e1b) + b;
                }
                if (tok < M) {
                    const size_t off = (size_t)tok * MST_D;
                    uint2 zh, zl;
                    split4_f16(xa[r], zh, zl);
                    tape_store8(tt.z2h + off + fa, zh);
                    tape_store8(tt.z2l + off + fa, zl);
                    split4_f16(xb[r], zh, zl);
                    tape_store8(tt.z2h + off + fb, zh);
                    tape_store8(tt.z2l + off + fb, zl);
                }
                const f32x4 t = xa[r] + xb[r];
                mean[r] = wave_sum((t[0] + t[1]) + (t[2] + t[3])) * (1.0f / MST_D);
            }
        } else {
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const char* srow = smem + (RPW * wave + r) * C::LN_LD;
            xa[r] = *reinterpret_cast<const f32x4*>(srow + fa * 4) + ba;
            xb[r] = *reinterpret_cast<const f32x4*>(srow + fb * 4) + bb;
            const f32x4 t = xa[r] + xb[r];
            mean[r] = wave_sum((t[0] + t[1]) + (t[2] + t[3])) * (1.0f / MST_D);
        }
        }
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            xa[r] -= mean[r];
            xb[r] -= mean[r];
            const f32x4 sq = {fmaf(xa[r][0], xa[r][0], xb[r][0] * xb[r][0]), fmaf(xa[r][1], xa[r][1], xb[r][1] * xb[r][1]),
                              fmaf(xa[r][2], xa[r][2], xb[r][2] * xb[r][2]), fmaf(xa[r][3], xa[r][3], xb[r][3] * xb[r][3])};
            rstd[r] = ln_rstd(wave_sum((sq[0] + sq[1]) + (sq[2] + sq[3])));
        }
        constexpr bool THROUGH = PERSIST || MST_TAIL_OUT == 1 || MST_TAIL_OUT == 3;      // write-through stores
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int tok = tok0 + RPW * wave + r;
            if (tok < M) {
                const size_t off = (size_t)tok * MST_D;
                uint2 ha, la, hb, lb;
                split4_f16(__builtin_elementwise_fma(xa[r], ga * rstd[r], ea), ha, la);
                split4_f16(__builtin_elementwise_fma(xb[r], gb * rstd[r], eb), hb, lb);
                if constexpr (WIDE) {
                    const u32x4_t h4 = {ha.x, ha.y, hb.x, hb.y}, l4 = {la.x, la.y, lb.x, lb.y};
                    if constexpr (THROUGH) { store16_sc1(hx + off + fa, h4); store16_sc1(hl + off + fa, l4); }
                    else {
                        *reinterpret_cast<u32x4_t*>(hx + off + fa) = h4;
                        *reinterpret_cast<u32x4_t*>(hl + off + fa) = l4;
                    }
                } else if constexpr (THROUGH) {
                    store8_sc1(hx + off + fa, ha); store8_sc1(hl + off + fa, la);
                    store8_sc1(hx + off + fb, hb); store8_sc1(hl + off + fb, lb);
                } else {
                    *reinterpret_cast<uint2*>(hx + off + fa) = ha;
                    *reinterpret_cast<uint2*>(hl + off + fa) = la;
                    *reinterpret_cast<uint2*>(hx + off + fb) = hb;
                    *reinterpret_cast<uint2*>(hl + off + fb) = lb;
                }
            }
        }
    }
    TAIL_MARK(5)
#undef TAIL_ISSUE
}

template <int NTB>
__global__ __launch_bounds__(512) void k_layer_tail(const f16* __restrict__ att, const f16* __restrict__ wt,
                                                    const float* __restrict__ b_out, const float* __restrict__ g1, const float* __restrict__ be1,
                                                    const float* __restrict__ b1, const float* __restrict__ b2,
                                                    const float* __restrict__ g2, const float* __restrict__ be2,
                                                    f16* __restrict__ hx, f16* __restrict__ hl, const float* __restrict__ gelu_tab, int M) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tail_body<NTB, false>(smem, att, wt, b_out, g1, be1, b1, b2, g2, be2, hx, hl, gelu_tab, M, blockIdx.x * (16 * NTB), GroupSync{nullptr, 0u, nullptr}, 0);
}

// The training forward's launch: tape slots of layer l in `tt`, the layer's output stream (slot l + 1) in hx / hl.
template <int NTB>
__global__ __launch_bounds__(512) void k_layer_tail_train(const f16* __restrict__ att, const f16* __restrict__ wt,
                                                          const float* __restrict__ b_out, const float* __restrict__ g1, const float* __restrict__ be1,
                                                          const float* __restrict__ b1, const float* __restrict__ b2,
                                                          const float* __restrict__ g2, const float* __restrict__ be2,
                                                          f16* __restrict__ hx, f16* __restrict__ hl, const float* __restrict__ gelu_tab, int M,
                                                          const TailTrain tt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tail_body<NTB, false, true>(smem, att, wt, b_out, g1, be1, b1, b2, g2, be2, hx, hl, gelu_tab, M, blockIdx.x * (16 * NTB), GroupSync{nullptr, 0u, nullptr}, 0, tt);
}

}  // namespace mst
