// Shared device helpers for the gfx950 denoising kernels (wave64, MFMA 32x32x16 f16).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// model constants (utils/model_util.py:160-167 hard-codes heads/ff; latent_dim default 512)
#define MST_D 512
#define MST_H 4
#define MST_HD 128
#define MST_FF 1024

// schedule table rows (include/mst_engine.h enum mst_table)
#define TAB_SQRT_AC 0
#define TAB_SQRT_1M_AC 1
#define TAB_COEF1 2
#define TAB_COEF2 3
#define TAB_LOGVAR 4
#define TAB_SQRT_RECIP_AC 5
#define TAB_SQRT_RECIPM1_AC 6
#define TAB_AC 7
#define TAB_AC_PREV 8
#define NTAB 9

__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// C/D fragment of the 32x32 MFMA: lane holds column (lane & 31); register r holds
// row (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ uint2 pack4_f16(float a, float b, float c, float d) {
    f16x4 v = {(f16)a, (f16)b, (f16)c, (f16)d};
    return __builtin_bit_cast(uint2, v);
}

// The residual stream is kept as an f16 PAIR: hi = f16(y) is the MFMA operand copy every GEMM reads
// anyway, lo = f16(y - hi) carries the next 11 mantissa bits, so hi + lo reproduces the fp32 value to
// ~2^-22 relative.  Versus fp32 + a separate f16 copy this removes 12.9 MB of HBM writes from each of
// the HBM-bound LayerNorm epilogues (64.5 -> 51.6 MB per launch at batch 64).
// acc + (f16 half of a packed pair), one v_fma_mix_f32 (the f16 operand is converted inside the FMA): HI selects the upper half.
template <int HI> __device__ __forceinline__ float add_half(unsigned packed, float acc) {
    float d;
    if constexpr (HI) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(acc));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(acc));
    return d;
}
template <int HI> __device__ __forceinline__ float sub_half(unsigned packed, float acc) {      // acc - half
    float d;
    if constexpr (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(acc));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(acc));
    return d;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// One weight fragment: issued and counted by hand.  hipcc would batch plain loads (all D at the end of an unrolled pass,
// vmcnt(0) at its head) and expose the L2 latency once per pass; it also cannot count loads it does not see, so NO
// compiler-visible global load or store may be issued between a fragment's load and its wait (cdna guide 5.7 item 1) -- the
// kernels that stream (mst_tail.h, mst_attn.h) keep theirs behind `tail_fence()`, where every outstanding fragment is OLDER than anything hipcc then counts.
template <int OFF> __device__ __forceinline__ void tail_wload(u32x4& d, unsigned voff, unsigned long long sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF));
}
template <int N> __device__ __forceinline__ void tail_wwait(u32x4& d) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "n"(N)); }
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// ---- hand-offs between the resident workgroups of ONE clip (mst_trunk.h: a persistent launch in which the four workgroups of a clip run
// head i of the fused QKV + attention phase, then token tile i of the layer tail, layer after layer).  cdna guide 6, Guideline 16, recipe R1:
// the payload is stored write-through (`sc1`), every storing wave drains (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, ONE lane
// adds to the clip's counter (agent scope); a consumer polls that ONE word (relaxed, `sc1`), has invalidated its L1 once, drains, meets
// its workgroup at a barrier and then loads.  The invalidate is issued at the END of the consumer's own previous phase, in front of the
// poll: legal here because the CU reads none of the handed-off lines between that point and the flag (its own stores do not allocate),
// and 0.4 us cheaper per hand-off (csrc/probes/group_chain.hip, profiles/r05_group_handoff_probe.txt: 2.9-3.9 us per hand-off under
// the traffic of all 64 groups, every 16-byte chunk tag-checked).  Spins are bounded: a give-up sets *err and lets the launch drain.
// Lane index recomputed where it is needed (two VALU instructions), never kept in a register across the phases of a resident launch:
// `threadIdx.x` lives in v0 from kernel entry and would stay allocated (or spilled) through every phase's body.
__device__ __forceinline__ int lane_id_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// x (+) its partner across the wave halves (lanes l, l ^ 32) / across neighbouring 16-lane rows (l, l ^ 16) for a COMMUTATIVE op, without a
// lane index: v_permlane32_swap(x, x) leaves {own half's value, other half's value} in its two results in an order that depends on the
// half -- irrelevant to a + b or max(a, b), whose result is the same bits either way (csrc/probes/probe_permlane.hip has the lane maps).
// __shfl_xor needs __lane_id(), which hipcc hoists to kernel entry and keeps in a VGPR through every phase of a resident launch.
__device__ __forceinline__ float xor32_add(float x) {
    const unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];          // (hipcc 7.2: __builtin_bit_cast of a vector ELEMENT reads element 0 for every index)
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float xor32_max(float x) {
    const unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return fmaxf(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
}
__device__ __forceinline__ float xor16_add(float x) {
    const unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];          // (hipcc 7.2: __builtin_bit_cast of a vector ELEMENT reads element 0 for every index)
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
// A wave-uniform value made opaque to the optimiser AND still known to be uniform: through a VGPR copy that an empty asm statement
// "modifies", then v_readfirstlane.  (An asm output constrained to "s" is treated as divergent by hipcc's uniformity analysis: address
// arithmetic on it moves to the VALU and inline-asm "s" operands derived from it silently become VGPRs.)  Used by the phases of a
// resident launch so that nothing derived from the wave index or a per-layer pointer is hoisted out of the phase loop and kept alive.
__device__ __forceinline__ int opaque_uniform(int x) {
    asm volatile("" : "+v"(x));
    return __builtin_amdgcn_readfirstlane(x);
}
template <class T> __device__ __forceinline__ T* opaque_uniform_ptr(T* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = (unsigned)opaque_uniform((int)(unsigned)u), hi = (unsigned)opaque_uniform((int)(unsigned)(u >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
struct GroupSync {
    unsigned* cnt;                 // the clip's arrival counter (a 128-byte line of its own)
    unsigned target;               // the value that says "every producer of my input has signalled"
    unsigned* err;                 // host-visible word: non-zero after a give-up
};
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ void group_wait(const GroupSync& g, int wave) {
    if (wave == 0 && lane_id_now() == 0) {
        unsigned spins = 0;
        while ((int)(__hip_atomic_load((gu32*)g.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g.target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 23)) { __hip_atomic_store((gu32*)g.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the poll's loads and the invalidate issued by group_signal() in front of them
    }
    __builtin_amdgcn_s_barrier();
}
// reset_at: the count that says "this was the clip's last arrival of the launch" -- whoever brings the counter there puts it back to 0 (nobody
// polls it any more: the launch's last phase has no consumer inside the launch), so every launch finds every counter at 0.
__device__ __forceinline__ void group_signal(unsigned* cnt, int wave, unsigned reset_at) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores; LDS reads are done
    __builtin_amdgcn_s_barrier();
    if (wave == 0 && lane_id_now() == 0) {
        const unsigned old = __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == reset_at) __hip_atomic_store((gu32*)cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // this CU's L1 copies of what its partners are rewriting: dropped before the next poll
    }
}
// 16- / 8-byte write-through stores (the trailing s_nop: cdna guide 5.7 item 1, the data registers may be reused right behind an asm store)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16_sc1(void* p, u32x4_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store8_sc1(void* p, uint2 v) {
    const u32x2_t w = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}

// y -> (hi, lo) f16 pairs with hi = f16(y), lo = f16(y - hi): the stream's storage format (~22 significant bits)
__device__ __forceinline__ void split4_f16(const f32x4& y, uint2& hi, uint2& lo) {
    f16x4 h = {(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
    hi = __builtin_bit_cast(uint2, h);
    f16x4 l = {(f16)sub_half<0>(hi.x, y[0]), (f16)sub_half<1>(hi.x, y[1]), (f16)sub_half<0>(hi.y, y[2]), (f16)sub_half<1>(hi.y, y[3])};
    lo = __builtin_bit_cast(uint2, l);
}
// base + hi + lo, two mixed-precision FMAs per element (no separate conversions)
__device__ __forceinline__ f32x4 add4_f16(uint2 hi, uint2 lo, const f32x4& base) {
    f32x4 y = {add_half<0>(hi.x, add_half<0>(lo.x, base[0])), add_half<1>(hi.x, add_half<1>(lo.x, base[1])),
               add_half<0>(hi.y, add_half<0>(lo.y, base[2])), add_half<1>(hi.y, add_half<1>(lo.y, base[3]))};
    return y;
}
// Sum over the 64 lanes of a wave, returned wave-uniform (an SGPR): four row_shr steps inside the 16-lane rows, row_bcast:15 /
// row_bcast:31 across them (six v_add_f32 with DPP operands; __shfl_xor is a ds_bpermute + add per step and lane).
__device__ __forceinline__ float wave_sum(float v) {
    auto step = [](float x, auto ctrl, auto rmask) {
        constexpr int C = decltype(ctrl)::value, R = decltype(rmask)::value;
        return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), C, R, 0xf, false));
    };
    using std::integral_constant;
    v = step(v, integral_constant<int, 0x111>(), integral_constant<int, 0xf>());      // row_shr:1
    v = step(v, integral_constant<int, 0x112>(), integral_constant<int, 0xf>());      // row_shr:2
    v = step(v, integral_constant<int, 0x114>(), integral_constant<int, 0xf>());      // row_shr:4
    v = step(v, integral_constant<int, 0x118>(), integral_constant<int, 0xf>());      // row_shr:8  -> lane 15 of a row = the row's sum
    v = step(v, integral_constant<int, 0x142>(), integral_constant<int, 0xa>());      // row_bcast:15 into rows 1, 3
    v = step(v, integral_constant<int, 0x143>(), integral_constant<int, 0xc>());      // row_bcast:31 into rows 2, 3 -> lane 63 = the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ f32x4 join4_f16(uint2 hi, uint2 lo) {
    const f16x4 h = __builtin_bit_cast(f16x4, hi), l = __builtin_bit_cast(f16x4, lo);
    f32x4 y = {(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
    return y;
}


// erf-form GELU (nn.TransformerEncoderLayer activation="gelu", mdm_forstyledataset.py:539-543).
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute): branch-free, ~14 VALU ops + one
// v_exp + one v_rcp (the hardware reciprocal, 1 ulp: a correctly rounded 1/x is a ten-instruction
// div_scale / fma / div_fixup sequence per element, a third of this function).  libm's erff inlines to ~50 ops with data-dependent branches per element, which
// measured ~15-20 us per FFN1 launch; its extra accuracy is invisible behind the f16 store (2^-11).
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float y = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(y, x);
}
// LayerNorm's 1 / sqrt(var + eps): v_rsq_f32 (1 ulp) instead of v_sqrt + the correctly rounded division sequence; every
// LayerNorm in the engine (fused tail, GEMM epilogue, training rows) goes through here so the paths agree bit for bit.
__device__ __forceinline__ float ln_rstd(float sum_sq) { return __builtin_amdgcn_rsqf(sum_sq * (1.0f / MST_D) + 1e-5f); }

// LayerNorm of one 512-wide row held by a wave: lane l has features 4 l .. 4 l + 3 (xa) and 256 + 4 l .. + 3 (xb).  Every multiply-add
// is spelled as an fma: under -ffp-contract=fast hipcc may fuse `a * a + b * b` either way round, and which way depends on the code
// around it -- the rows kernel and the GEMM prologue that share this function (mst_elem.h k_ln_rows, mst_small.h) must agree bit for bit.
__device__ __forceinline__ void ln_row_wave(f32x4& xa, f32x4& xb, const f32x4& ga, const f32x4& gb, const f32x4& ea, const f32x4& eb) {
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
        s2 += __builtin_fmaf(xa[i], xa[i], xb[i] * xb[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
    const float rstd = ln_rstd(s2);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        xa[i] = __builtin_fmaf(xa[i] * rstd, ga[i], ea[i]);
        xb[i] = __builtin_fmaf(xb[i] * rstd, gb[i], eb[i]);
    }
}

// GELU(x) = x Phi(x) written as max(x, 0) - |x| E(|x|) with E(a) = erfc(a / sqrt 2) / 2 = (t P(t) / 2) exp(-a^2 / 2), t = 1 / (1 + p a / sqrt 2)
// (the same Abramowitz-Stegun 7.1.26 polynomial as erf_as; 0.5 folded into its coefficients): 13 VALU ops instead of 18 -- no 1 + erf,
// no sign transfer, exp2 taken directly -- and no cancellation for x < 0.  The layer tail spends ~7 us per 64-token tile in this
// function (128 values per lane), which is why the instruction count matters.
__device__ __forceinline__ float gelu_erf(float x) {
    const float a = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, a, 1.0f));
    float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float m = a * 0.84932180028801904f;                    // sqrt(log2(e) / 2): exp(-a^2 / 2) = exp2(-m^2)
    const float e = __builtin_amdgcn_exp2f(-(m * m));
    return fmaf(-a, (p * t) * e, fmaxf(x, 0.0f));
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: counter-based normals for the in-kernel noise mode.
// Element (clip, f, t) of step `step` = component (t & 3) of the 4 normals generated from
// counter (t >> 2, f, clip, step) under key (seed_lo, seed_hi): one call serves 4 consecutive frames.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of v_mul_hi_u32 + v_mul_lo_u32: the same bits, half the
        // quarter-rate multiplies (the generator is ~2/3 of the noise draw's issue cycles)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void philox_normal4(uint32_t tq, uint32_t f, uint32_t clip, uint32_t step,
                                               uint64_t seed, float (&n)[4]) {
    uint32_t r[4];
    philox4x32_10(tq, f, clip, step, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    // u in (0, 1]: never log(0)
    float u0 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    float u1 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
    float u2 = ((float)(r[2] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    float u3 = (float)(r[3] >> 8) * (1.0f / 16777216.0f);
    // Box-Muller on the hardware transcendentals: v_log_f32 (log2, 1 ulp), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in
    // revolutions: u1 itself).  libm's logf / sqrtf / sincosf are ~170 instructions per four normals, half of this function and
    // ~6 us of VALU per denoise step; the in-kernel draw and `mst_philox_normal` share this code, so they stay bit-identical.
    const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));      // sqrt(-2 ln u0)
    const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
    n[0] = ra * __builtin_amdgcn_cosf(u1); n[1] = ra * __builtin_amdgcn_sinf(u1);
    n[2] = rb * __builtin_amdgcn_cosf(u3); n[3] = rb * __builtin_amdgcn_sinf(u3);
}

// ------------------------------------------------------------------------------------------
// The per-element diffusion update shared by the fused output-projection epilogue and the
// stand-alone kernel.  Restates gaussian_diffusion.py:341-349 (inpainting blend), :404-412
// (x0-hat -> posterior mean), :569-585 / inpainting_gaussian_diffusion.py:51-63 (ancestral step)
// and inpainting_gaussian_diffusion.py:157-177 (DDIM step) in the reference's operation order.
// ------------------------------------------------------------------------------------------
struct StepCoef {   // per-clip scalars gathered from the float32 tables at index t
    float c1, c2, sigma_ddpm;      // posterior_mean_coef1/2, nonzero * exp(0.5 * logvar)
    float srac, srm1ac;            // sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod
    float sq_abp, dir, sigma_ddim; // sqrt(abar_prev), sqrt(1 - abar_prev - sigma^2), nonzero * sigma
};

__device__ __forceinline__ StepCoef step_coef(const float* __restrict__ tab, int nsteps, int t, float eta) {
    StepCoef c;
    float nz = t != 0 ? 1.0f : 0.0f;
    c.c1 = tab[TAB_COEF1 * nsteps + t];
    c.c2 = tab[TAB_COEF2 * nsteps + t];
    c.sigma_ddpm = nz * expf(0.5f * tab[TAB_LOGVAR * nsteps + t]);
    c.srac = tab[TAB_SQRT_RECIP_AC * nsteps + t];
    c.srm1ac = tab[TAB_SQRT_RECIPM1_AC * nsteps + t];
    float ab = tab[TAB_AC * nsteps + t], abp = tab[TAB_AC_PREV * nsteps + t];
    float sigma = eta * sqrtf((1.0f - abp) / (1.0f - ab)) * sqrtf(1.0f - ab / abp);
    c.sq_abp = sqrtf(abp);
    c.dir = sqrtf(1.0f - abp - sigma * sigma);
    c.sigma_ddim = nz * sigma;
    return c;
}

// returns the next sample; x0-hat (after blend / clip) is written to *pred.
// MEAN: what the model predicts (reference :398-412, in the reference's order: the inpainting blend acts on the RAW model output, :341-349,
// the conversion to x0-hat follows): 0 = x_start (every model the factories build, utils/model_util.py:172), 1 = epsilon
// (_predict_xstart_from_eps, :426-431), 2 = x_{t-1} (_predict_xstart_from_xprev, :433-441; the posterior mean then IS the model output,
// :399-403).  A template parameter: the fused output-projection kernels instantiate 0 and compile to what they were.
template <int SAMPLER, int MEAN = 0>
__device__ __forceinline__ float step_update(const StepCoef& c, float model_out, float x, float noise,
                                             bool has_blend, float mask, float motion, bool mask_noise,
                                             bool clip, float* pred) {
    float out = model_out;
    if (has_blend) out = out * (1.0f - mask) + motion * mask;
    const float raw = out;
    if (MEAN == 1) out = c.srac * x - c.srm1ac * out;
    if (MEAN == 2) out = (1.0f / c.c1) * out - (c.c2 / c.c1) * x;
    if (clip) out = fminf(fmaxf(out, -1.0f), 1.0f);
    *pred = out;
    if (mask_noise) noise = noise * (1.0f - mask);
    if (SAMPLER == 0) {
        float mean = MEAN == 2 ? raw : c.c1 * out + c.c2 * x;
        return mean + c.sigma_ddpm * noise;
    } else {
        float eps = (c.srac * x - out) / c.srm1ac;
        float mean_pred = out * c.sq_abp + c.dir * eps;
        return mean_pred + c.sigma_ddim * noise;
    }
}

// Per-call arguments of a sampling loop, kept in DEVICE memory: a captured step graph reads its tensors and its position in
// the loop through this block, so one instantiated graph serves every later call with the same shapes (new clip tensors, new
// seed) and every replay (the step counter `jbase` is advanced on the device at the end of each replay).
struct LoopDev {
    float* x; const float* mask; const float* motion; const float* noise; const float* scale; float* xstart;
    unsigned long long seed; float eta; int t_start; int nrun; int jbase; int pad;
};

// arguments of the fused diffusion step (output-projection epilogue)
struct StepArgs {
    const float* tab; int nsteps; int t;          // schedule tables (device) and the diffusion index
    float eta;
    const float* mask; const float* motion;       // [B,F,1,T] or null
    const float* noise;                           // [B,F,1,T] or null (-> philox)
    const float* scale;                           // [B] guidance scale (cfg)
    const float* x;                               // x_t
    float* sample; float* xstart;                 // outputs (xstart may be null)
    unsigned long long seed; unsigned step; unsigned clip0;   // clip0: batch index of the slice's first clip (Philox counter)
    int mask_noise, clip, philox;
    // loop mode: the pointer fields above are PRESENCE flags (null / non-null) and are resolved in the kernel from *ld:
    // step j = ld->jbase + joff visits index t_start - j; tensors start `eo` elements into the caller's, per-step buffers
    // (noise, x0-hat dump) advance by `step_stride` elements per step; the slice's scales start at clip `clip0`.
    const LoopDev* ld; int joff; unsigned long long eo, step_stride;
    // per (clip, feature) row of the inpainting mask: 0 = all zeros, 1 = all ones, 2 = mixed (k_mask_rowflags, once per loop).
    // Rows flagged 0 / 1 skip the mask load, rows flagged 0 the motion load too: with the root_horizontal pattern (3 of 263
    // rows masked) the step kernel stops re-reading two full fp32 tensors every step (26 MB of 66 MB per 64-clip launch).
    const unsigned char* rowflag;
};

__device__ __forceinline__ StepArgs step_resolve(StepArgs sa) {
    if (!sa.ld) return sa;
    const LoopDev d = *sa.ld;
    const int j = d.jbase + sa.joff;
    sa.t = d.t_start - j;
    sa.eta = d.eta;
    sa.seed = d.seed;
    sa.step = (unsigned)j;
    sa.x = d.x + sa.eo;
    sa.sample = d.x + sa.eo;
    sa.mask = sa.mask ? d.mask + sa.eo : nullptr;
    sa.motion = sa.motion ? d.motion + sa.eo : nullptr;
    sa.noise = sa.noise ? d.noise + (size_t)j * sa.step_stride + sa.eo : nullptr;
    sa.xstart = sa.xstart ? d.xstart + (size_t)j * sa.step_stride + sa.eo : nullptr;
    sa.scale = sa.scale ? d.scale + sa.clip0 : nullptr;
    return sa;
}

namespace mst {
// ------------------------------------------------------------------------------------------------------------
// Dropout: a counter-based keep mask, regenerated (never stored) by the backward kernels.
// Element idx of a site is kept iff mix32(idx * phi + key) >= thr; kept values are multiplied by inv = 1/(1-p).
// thr = 0 keeps everything (eval / p = 0).  `key` mixes the call's seed, the layer and the site on the host.
// ------------------------------------------------------------------------------------------------------------
struct Drop { uint32_t key, thr; float inv; };
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float drop_mul(const Drop& d, uint32_t idx) {
    return mix32(idx * 0x9E3779B9u + d.key) >= d.thr ? d.inv : 0.f;
}
}  // namespace mst
