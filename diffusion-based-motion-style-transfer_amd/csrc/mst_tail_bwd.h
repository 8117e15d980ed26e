// The BACKWARD of everything of an encoder layer behind the attention, one workgroup per 64-token tile (round 6; the mirror of
// mst_tail.h's fused forward tail; gaussian_diffusion.py:1317-1399 back-propagates through nn.TransformerEncoderLayer's post-norm block,
// mdm_forstyledataset.py:539-546):
//
//     d hid = dbr2 W2            dbr2 = LayerNorm2-backward's branch gradient (k_ln_bwd in front of this launch), f16 [tok][512]
//     d pre = d hid * keep2 * GELU'(pre)                                   (pre from the tape)
//     g(x1) = d pre W1 + dz2                                               (dz2: LayerNorm2-backward's residual gradient, fp32, in `g`)
//     dz1   = LayerNorm1-backward(g(x1); z1 from the tape)  -> `g` in place (the residual gradient the QKV dgrad GEMM adds)
//     dbr1  = dz1 * keep1,   d att = dbr1 W_out
//
// Unfused these are three dgrad GEMM launches with row-wise epilogues + their round trips (FFN2 dgrad + GELU' 44 us, FFN1 dgrad +
// LayerNorm1-backward 54 us, out-proj dgrad 30 us at 12 608 rows: 128 of the layer's 268 us on the backward pass's dependent chain, twice
// per fine-tune iteration -- the 64-clip call and the frozen motion encoder).  Here the three products are the forward tail's three
// phases in another order on the SAME machinery: W2^T | W1^T | W_out^T ([in][out] copies the engine keeps for the dgrad GEMMs) packed per
// wave exactly as W1 | W2 | W_out are for the forward (k_pack_tail with the FFN units FIRST), streamed L2 -> VGPR behind hand-counted
// waits; dbr2 arrives as one LDS-DMA burst (the forward's att burst), d pre lives in the double-buffered H images, dbr1 in the image
// dbr2 leaves behind, d att leaves through the dead H images as whole rows.
//
// WG = false (a frozen stack: the motion encoder): nothing but dz1 and d att is written.  WG = true: dpre, hid = dropout(GELU(pre))
// (dW2's operand, what OpGeluBwd regenerates on the unfused path), dbr1 and the tile's [dgamma1 | dbeta1 | db_out] sums as well.
#pragma once
#include "mst_tail.h"
#include "mst_train.h"

#ifndef MST_TB_PROBE          // timing-only probe builds of tools/r6_tb_probe2.sh: 1 = no LayerNorm-stage global loads / stores, 2 = no out-proj phase, 3 = no pre loads / GELU'
#define MST_TB_PROBE 0
#endif
namespace mst {

struct TailBwdCfg {
    static constexpr int D = 8;                          // weight fragments in flight per wave
    static constexpr int OFF_H = 0, HBUF = 32 * 1024;    // d pre chunk images, 2 x (64 x 512 B); behind the stream: the d att image (64 x 1 KB)
    static constexpr int OFF_CNT = 68 * 1024;            // arrival counters of the four chunks
    static constexpr int OFF_G1 = 70 * 1024;             // LayerNorm1 weight (2 KB), staged at kernel start
    static constexpr int OFF_EXCH = 72 * 1024;           // [wave][token] float2 statistics exchange (4 KB)
    static constexpr int OFF_SUM = 76 * 1024;            // WG: [3][512] fp32 tile sums (6 KB)
    static constexpr int OFF_TAB = 82 * 1024;            // the forward tail's Phi(x) table (TailCfg::GELU_TAB_BYTES = 13 KB): GELU'(x) = Phi(x) + x phi(x)
    static constexpr int OFF_IMG = 96 * 1024;            // dbr2 image (64 x 1 KB), later the dbr1 image
    static constexpr int SMEM = 160 * 1024;
    static_assert(2 * HBUF <= OFF_CNT && OFF_CNT + 16 <= OFF_G1 && OFF_G1 + 2048 <= OFF_EXCH && OFF_EXCH + 4096 <= OFF_SUM &&
                  OFF_SUM + 6144 <= OFF_TAB && OFF_TAB + TailCfg::GELU_TAB_BYTES <= OFF_IMG && OFF_IMG + 64 * 1024 <= SMEM, "LDS map");
};

// W_out^T | W2^T | W1^T ([out][in] f16 as k_pack_tail takes them: w_outT is [512][512], w2T [1024][512], w1T [512][1024]) -> the eight
// per-wave streams of the backward tail: the eight FFN units first (F1(0) F1(1) F2(0) F1(2) F2(1) F1(3) F2(2) F2(3), as in the forward),
// then the 64 out-proj-shaped fragments.
__global__ __launch_bounds__(256) void k_pack_tail_bwd(const f16* __restrict__ w_outT, const f16* __restrict__ w2T,
                                                       const f16* __restrict__ w1T, f16* __restrict__ dst) {
    using C = TailCfg;
    const int total = 8 * C::NFRAG * 64, NF = 4 * (C::F1_FRAG + C::F2_FRAG);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, f = (i >> 6) % C::NFRAG, wave = (i >> 6) / C::NFRAG;
        const int r = 32 * wave + (lane & 15), kq = 8 * (lane >> 4);
        const f16* src;
        if (f >= NF) {
            const int u = f - NF, k32 = u >> 2, nh = (u >> 1) & 1, rb = u & 1;
            src = w_outT + (size_t)(256 * nh + 16 * rb + r) * MST_D + 32 * k32 + kq;
        } else {
            const int unit = f >> 5, v = f & 31;
            const bool is2 = unit == 7 || (unit >= 2 && !(unit & 1));
            const int hc = unit == 7 ? 3 : is2 ? (unit >> 1) - 1 : unit == 0 ? 0 : (unit + 1) >> 1;
            if (!is2) {                                   // "FFN1"-shaped: hidden rows of W2^T, k = the 512 stream features
                const int k32 = v >> 1, rb = v & 1;
                src = w2T + (size_t)(256 * hc + 16 * rb + r) * MST_D + 32 * k32 + kq;
            } else {                                      // "FFN2"-shaped: the 512 x1 features of W1^T, k = hidden chunk hc
                const int k32 = v >> 2, nh = (v >> 1) & 1, rb = v & 1;
                src = w1T + (size_t)(256 * nh + 16 * rb + r) * MST_FF + 256 * hc + 32 * k32 + kq;
            }
        }
        reinterpret_cast<uint4*>(dst)[i] = *reinterpret_cast<const uint4*>(src);
    }
}

struct TailBwdOut {                                       // WG only (null otherwise)
    f16 *dpre, *hid, *dbr1;                               // [M][1024], [M][1024], [M][512]: operands of dW1, dW2, dW_out
    float* part;                                          // [tiles][3][512]: the tile's dgamma1 | dbeta1 | db_out sums (k_ln_bwd_finish adds them in tile order)
};
// LN2 = true: LayerNorm2's backward (k_ln_bwd's row arithmetic, gaussian_diffusion.py's post-norm block differentiated) runs at the head of
// this launch instead of as a launch of its own in front of it: the tile's rows of the incoming gradient and of z2 come in as coalesced row
// loads, dz2 leaves for `g` as rows, and the dbr2 image is written straight into LDS (no HBM round trip for it).  Frozen stacks only
// (WG = false): with parameter gradients dbr2 must reach HBM for dW2 anyway and the three LayerNorm2 sums need a tile reduction.
struct TailBwdLn2 {
    const float* g_in;                                    // [M][512] fp32: gradient wrt the layer's output
    const f16 *z2h, *z2l;                                 // tape: LayerNorm2's input rows
    const float* g2;                                      // LayerNorm2 weight
    TailDrop d3;                                          // keep mask of the FFN2 output
};

// one 8-byte asm load (counted in vmcnt with the stream; hipcc must not wait for it)
__device__ __forceinline__ void tailb_load8(u32x2_t& d, unsigned voff, unsigned long long sbase) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void tailb_touch(u32x2_t& d) { asm volatile("" : "+v"(d)); }

template <bool WG, bool LN2 = false>
__global__ __launch_bounds__(512) void k_layer_tail_bwd(const f16* __restrict__ dbr2, const f16* __restrict__ wt, const f16* __restrict__ pre,
                                                        const f16* __restrict__ z1h, const f16* __restrict__ z1l,
                                                        const float* __restrict__ g1, float* __restrict__ g, f16* __restrict__ datt,
                                                        const TailBwdOut o, const TailDrop d1, const TailDrop d2, int M,
                                                        const float* __restrict__ gelu_tab, const TailBwdLn2 n2 = TailBwdLn2{}) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = TailCfg;
    using B = TailBwdCfg;
    constexpr int D = B::D, NTB = 4, RPW = 2 * NTB;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = wave * 64 + lane;
    const int t16 = lane & 15, q4 = lane >> 4;
    const int tok0 = blockIdx.x * 64;
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned* const arrived = reinterpret_cast<unsigned*>(smem + B::OFF_CNT);
    if (tid < 4) arrived[tid] = 0;

    // ---- kernel-start burst: the tile's dbr2 rows -> the image (row r = 1 KB, 16-B chunk c at c ^ (r & 15)), LayerNorm1's weight -> LDS
    if constexpr (!LN2) {
#pragma unroll
    for (int j = 0; j < RPW; j++) {
        const int r = RPW * wave + j;
        int tok = tok0 + r;
        if (tok >= M) tok = M - 1;
        const unsigned voff = (unsigned)tok * (unsigned)(MST_D * 2) + (unsigned)((lane ^ (r & 15)) << 4);
        tail_glds1(voff, (unsigned long long)dbr2, __builtin_amdgcn_readfirstlane(smem_base + B::OFF_IMG + r * 1024));
    }
    }
    if (wave < 2) tail_glds1((unsigned)lane * 16u, (unsigned long long)(g1 + 256 * wave), __builtin_amdgcn_readfirstlane(smem_base + B::OFF_G1 + 1024 * wave));
    tail_glds1((unsigned)lane * 16u, (unsigned long long)(gelu_tab + 256 * wave), __builtin_amdgcn_readfirstlane(smem_base + B::OFF_TAB + 1024 * wave));
    if (wave < 5) tail_glds1((unsigned)lane * 16u, (unsigned long long)(gelu_tab + 256 * (8 + wave)), __builtin_amdgcn_readfirstlane(smem_base + B::OFF_TAB + 1024 * (8 + wave)));

    // ---- the weight stream of this wave
    const unsigned w_voff = (unsigned)lane * 16u;
    const char* wnext = reinterpret_cast<const char*>(wt) + (size_t)wave * C::WAVE_BYTES;
    u32x4 q[D];
#define TB_ISSUE(J) tail_wload<((J) & 3) * 1024>(q[J], w_voff, (unsigned long long)(wnext + ((J) >> 2) * 4096))
    TB_ISSUE(0); TB_ISSUE(1); TB_ISSUE(2); TB_ISSUE(3); TB_ISSUE(4); TB_ISSUE(5); TB_ISSUE(6); TB_ISSUE(7);
#undef TB_ISSUE
    static_assert(D == 8, "written out");
    wnext += D * 1024;

    const unsigned xlane1k = (unsigned)t16 * 1024u, xlane512 = (unsigned)t16 * 512u, xswz = (unsigned)((q4 ^ t16) << 4);
    auto xread = [&](const char* img, auto rowb, int k32, f16x8 (&x)[NTB]) {
        constexpr int ROWB = decltype(rowb)::value;
        const char* p = img + (ROWB == 1024 ? xlane1k : xlane512) + (((unsigned)k32 << 6) ^ xswz);
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) x[tb] = *reinterpret_cast<const f16x8*>(p + tb * 16 * ROWB);
    };
    using RB1K = std::integral_constant<int, 1024>;
    using RB512 = std::integral_constant<int, 512>;
    f32x4 acc[2][2][NTB];
    f32x4 acch[2][NTB];
    f16x8 xs[2][NTB];
    // one pass = D fragments (see mst_tail.h `pass`): RA fragments per k-step; LOAD: the pass requests the next D fragments
    auto pass = [&](const char* img, auto rowb, auto rac, auto loadc, int k32base, bool more, auto side) {
        constexpr int RA = decltype(rac)::value;
        constexpr bool LOAD = decltype(loadc)::value != 0;
        constexpr int STEPS = D / RA;
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            if (s + 1 < STEPS || more) xread(img, rowb, k32base + s + 1, xs[(s + 1) & 1]);
#pragma unroll
            for (int a = 0; a < RA; a++) {
                const int j = s * RA + a;
                auto use = [&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    if constexpr (LOAD) tail_wwait<D - 1>(q[J]); else tail_wwait<D - 1 - J>(q[J]);
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
#define TB_CASE(J) case J: use(std::integral_constant<int, J>()); break;
                    TB_CASE(0) TB_CASE(1) TB_CASE(2) TB_CASE(3) TB_CASE(4) TB_CASE(5) TB_CASE(6) TB_CASE(7)
#undef TB_CASE
                }
            }
            side(s);
        }
        if constexpr (LOAD) wnext += D * 1024;
    };
    using RA4 = std::integral_constant<int, 4>;
    using RA2 = std::integral_constant<int, 2>;
    using LD1 = std::integral_constant<int, 1>;
    using LD0 = std::integral_constant<int, 0>;
    // accumulator slot (nh, rb, tb) of this lane in an image with 1-KB rows (see mst_tail.h)
    auto slot1k = [&](int nh, int rb, int tb) {
        return (unsigned)((16 * tb + t16) * 1024 + (((32 * nh + 4 * wave + 2 * rb + (q4 >> 1)) ^ t16) << 4) + 8 * (q4 & 1));
    };
#pragma unroll
    for (int nh = 0; nh < 2; nh++)
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int tb = 0; tb < NTB; tb++) acc[nh][rb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LN2) {
        // =========================================================================================== LayerNorm2 backward (row layout)
        // wave w: rows [8 w, 8 w + 8) of the tile; lane l: features 4 l .. + 3 and 256 + 4 l .. + 3 (k_ln_bwd's map).  The D weight fragments
        // requested above travel meanwhile (hipcc's waits for the loads below can only be stricter for them being older).
        const int fa = lane * 4, fb = 256 + lane * 4;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(n2.g2 + fa), gb = *reinterpret_cast<const f32x4*>(n2.g2 + fb);
        static_assert(!(WG && LN2), "LayerNorm2 rides in the frozen-stack launch only");
        char* const dimg = smem + B::OFF_IMG;
#pragma unroll
        for (int h4 = 0; h4 < RPW / 4; h4++) {
            uint2 zr[4][4];
            f32x4 gy[4][2];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int tok = tok0 + RPW * wave + 4 * h4 + i;
                if (tok >= M) tok = M - 1;
                const size_t off = (size_t)tok * MST_D;
                zr[i][0] = *reinterpret_cast<const uint2*>(n2.z2h + off + fa);
                zr[i][1] = *reinterpret_cast<const uint2*>(n2.z2l + off + fa);
                zr[i][2] = *reinterpret_cast<const uint2*>(n2.z2h + off + fb);
                zr[i][3] = *reinterpret_cast<const uint2*>(n2.z2l + off + fb);
                gy[i][0] = *reinterpret_cast<const f32x4*>(n2.g_in + off + fa);
                gy[i][1] = *reinterpret_cast<const f32x4*>(n2.g_in + off + fb);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = RPW * wave + 4 * h4 + i, tok = tok0 + r;
                f32x4 xa = join4_f16(zr[i][0], zr[i][1]), xb = join4_f16(zr[i][2], zr[i][3]);
                const f32x4 ya = gy[i][0], yb = gy[i][1];
                const f32x4 t = xa + xb;
                const float mean = wave_sum((t[0] + t[1]) + (t[2] + t[3])) * (1.0f / MST_D);
                xa -= mean;
                xb -= mean;
                const f32x4 sq = {fmaf(xa[0], xa[0], xb[0] * xb[0]), fmaf(xa[1], xa[1], xb[1] * xb[1]), fmaf(xa[2], xa[2], xb[2] * xb[2]),
                                  fmaf(xa[3], xa[3], xb[3] * xb[3])};
                const float rstd = ln_rstd(wave_sum((sq[0] + sq[1]) + (sq[2] + sq[3])));
                xa *= rstd;                                            // xhat
                xb *= rstd;
                const f32x4 aa = ga * ya, ab = gb * yb;
                const f32x4 u1 = aa + ab, u2 = aa * xa + ab * xb;
                const float c1 = wave_sum((u1[0] + u1[1]) + (u1[2] + u1[3])) * (1.0f / MST_D);
                const float c2 = wave_sum((u2[0] + u2[1]) + (u2[2] + u2[3])) * (1.0f / MST_D);
                const f32x4 za = (aa - c1 - xa * c2) * rstd, zb = (ab - c1 - xb * c2) * rstd;
                const uint32_t ia = (uint32_t)tok * (uint32_t)MST_D + (uint32_t)fa, ib = (uint32_t)tok * (uint32_t)MST_D + (uint32_t)fb;
                const f32x4 qa = {za[0] * tail_drop_mul(n2.d3, ia), za[1] * tail_drop_mul(n2.d3, ia + 1), za[2] * tail_drop_mul(n2.d3, ia + 2),
                                  za[3] * tail_drop_mul(n2.d3, ia + 3)};
                const f32x4 qb = {zb[0] * tail_drop_mul(n2.d3, ib), zb[1] * tail_drop_mul(n2.d3, ib + 1), zb[2] * tail_drop_mul(n2.d3, ib + 2),
                                  zb[3] * tail_drop_mul(n2.d3, ib + 3)};
                const uint2 pa = pack4_f16(qa[0], qa[1], qa[2], qa[3]), pb = pack4_f16(qb[0], qb[1], qb[2], qb[3]);
                // features fa .. + 3 = bytes 8 l .. of the row: chunk l / 2 (fb: 32 + l / 2), half (l & 1)
                char* dst = dimg + r * 1024 + (lane & 1) * 8;
                *reinterpret_cast<uint2*>(dst + (((lane >> 1) ^ (r & 15)) << 4)) = pa;
                *reinterpret_cast<uint2*>(dst + (((32 + (lane >> 1)) ^ (r & 15)) << 4)) = pb;
                if (tok < M) {
                    const size_t off = (size_t)tok * MST_D;
                    *reinterpret_cast<f32x4*>(g + off + fa) = za;
                    *reinterpret_cast<f32x4*>(g + off + fb) = zb;
                }
            }
        }
        tail_fence();
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");          // this wave's dbr2 rows (and its g1 / table pieces) have landed: behind them only the D fragments
    tail_barrier();                                                    // ... and everybody's (LN2: the image the waves have just written)

    // =========================================================================================== FFN backward, four hidden chunks
    // "F1"(hc): acch = W2^T[chunk hc] . dbr2^T = d hid of the chunk;  G'(hc): d pre = d hid * keep2 * GELU'(pre) -> H image (+ tape);
    // "F2"(hc): acc += W1^T[:, chunk hc] . d pre^T.  Order, split barriers and buffers exactly as in the forward tail's phase F.
    {
        const char* img = smem + B::OFF_IMG;
        u32x2_t pr[2 * NTB];                                           // the chunk's pre values of this lane: group (rb, tb) -> 4 f16
        const unsigned long long pre_b = (unsigned long long)pre;
        auto pre_issue = [&](int hc) {                                 // asm loads, older than the fragments the chunk's F1 passes request: landed when those are waited for
#pragma unroll
            for (int gi = 0; gi < 2 * NTB; gi++) {
                const int rb = gi / NTB, tb = gi % NTB;
                int tok = tok0 + 16 * tb + t16;
                if (tok >= M) tok = M - 1;
                tailb_load8(pr[gi], ((unsigned)tok * (unsigned)MST_FF + (unsigned)(256 * hc + 32 * wave + 16 * rb + 4 * q4)) * 2u, pre_b);
            }
        };
        auto f1 = [&](int hc) {
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int tb = 0; tb < NTB; tb++) acch[rb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#if MST_TB_PROBE != 3
            pre_issue(hc);
#endif
            xread(img, RB1K(), 0, xs[0]);
#pragma unroll 1
            for (int ps = 0; ps < C::F1_FRAG / D; ps++) pass(img, RB1K(), RA2(), LD1(), ps * (D / 2), ps + 1 < C::F1_FRAG / D, [](int) {});
        };
        auto gp_group = [&](int hc, int gi) {
            const int rb = gi / NTB, tb = gi % NTB;
            const unsigned coff = (unsigned)(((4 * wave + 2 * rb + (q4 >> 1)) ^ t16) << 4) + 8u * (q4 & 1);
            const f32x4 v = acch[rb][tb];
            const f16x4 ph = __builtin_bit_cast(f16x4, pr[gi]);
            const int tok = tok0 + 16 * tb + t16;
            const uint32_t oi = (uint32_t)tok * (uint32_t)MST_FF + (uint32_t)(256 * hc + 32 * wave + 16 * rb + 4 * q4);
            // GELU'(x) = Phi(x) + x phi(x): Phi by the forward tail's table (linear interpolation, 1.9e-6), phi by one exp2 -- a third of
            // gelu_grad's instructions (erf polynomial + exp + rcp), in a phase bound by the SIMDs' VALU + MFMA issue slots
            const char* const gtab = smem + B::OFF_TAB;
            float dp[4], hh[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float m = tail_drop_mul(d2, oi + i), x = (float)ph[i];
                const float u = __builtin_amdgcn_fmed3f(fmaf(x, 128.0f, 768.0f), 0.0f, 1535.9999f);
                const float2 e = *reinterpret_cast<const float2*>(gtab + ((unsigned)u << 3));
                const float Phi = fmaf(e.y, __builtin_amdgcn_fractf(u), e.x);
                const float phi = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
#if MST_TB_PROBE == 3
                dp[i] = v[i] * m;
                if constexpr (WG) hh[i] = v[i];
#else
                dp[i] = v[i] * m * fmaf(x, phi, Phi);
                if constexpr (WG) hh[i] = x * Phi * m;
#endif
            }
            const uint2 d16 = pack4_f16(dp[0], dp[1], dp[2], dp[3]);
            *reinterpret_cast<uint2*>(smem + B::OFF_H + (hc & 1) * B::HBUF + (16 * tb + t16) * 512 + coff) = d16;
            if constexpr (WG) {
                if (tok < M) {
                    *reinterpret_cast<uint2*>(o.dpre + oi) = d16;
                    *reinterpret_cast<uint2*>(o.hid + oi) = pack4_f16(hh[0], hh[1], hh[2], hh[3]);
                }
            }
        };
        auto announce = [&](int hc) {
            tail_fence();
            if (lane == 0) __hip_atomic_fetch_add(arrived + hc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            tail_fence();
        };
        auto await = [&](int hc) {
            tail_fence();
            const unsigned caddr = smem_base + B::OFF_CNT + 4u * (unsigned)hc;
            for (;;) {
                unsigned v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(caddr) : "memory");
                if (__builtin_amdgcn_readfirstlane(v) >= 8u) break;
                __builtin_amdgcn_s_sleep(1);
            }
            tail_fence();
        };
        constexpr int NG = 2 * NTB, NP2 = C::F2_FRAG / D, ST2 = D / 4;
        static_assert(NP2 * ST2 == NG, "one GELU' group per k-step of a chunk's F2");
        f1(0);
#pragma unroll
        for (int gi = 0; gi < NG; gi++) tailb_touch(pr[gi]);
#pragma unroll
        for (int gi = 0; gi < NG; gi++) gp_group(0, gi);
        announce(0);
#pragma unroll 1
        for (int hc = 0; hc < 3; hc++) {
            f1(hc + 1);
#pragma unroll
            for (int gi = 0; gi < NG; gi++) tailb_touch(pr[gi]);
            await(hc);
            const char* himg = smem + B::OFF_H + (hc & 1) * B::HBUF;
            xread(himg, RB512(), 0, xs[0]);
#pragma unroll
            for (int ps = 0; ps < NP2; ps++)
                pass(himg, RB512(), RA4(), LD1(), ps * ST2, ps + 1 < NP2, [&](int s) { gp_group(hc + 1, ps * ST2 + s); });
            announce(hc + 1);
        }
        {
            await(3);
            const char* himg = smem + B::OFF_H + B::HBUF;
            xread(himg, RB512(), 0, xs[0]);
            // (the stream PAUSES behind this chunk: fragments in flight across the LayerNorm stage cost 32 registers there, and hipcc then
            // spilled them -- a spilled register of a load still in flight is a corrupted register.  The out-proj phase starts its own D.)
#pragma unroll
            for (int ps = 0; ps + 1 < NP2; ps++) pass(himg, RB512(), RA4(), LD1(), ps * ST2, true, [](int) {});
            pass(himg, RB512(), RA4(), LD0(), (NP2 - 1) * ST2, false, [](int) {});
        }
    }
    tail_fence();
    tail_barrier();                                                    // everybody is done with the dbr2 image and the H images

    // =========================================================================================== LayerNorm1 backward (accumulator layout)
    // g(x1) = acc + dz2;  xhat = (z1 - mean) rstd (LayerNorm1's statistics again, from the z1 rows of the tape, merged exactly as in the
    // forward);  a = gamma g;  dz1 = rstd (a - mean(a) - xhat mean(a xhat)) -> `g` in place;  dbr1 = dz1 keep1 -> the image.
    {
        char* dimg = smem + B::OFF_IMG;
        float2* exch = reinterpret_cast<float2*>(smem + B::OFF_EXCH);
        const float* g1s = reinterpret_cast<const float*>(smem + B::OFF_G1);
        auto quad_sum = [](float v) {
            v += __shfl_xor(v, 16);
            return v + __shfl_xor(v, 32);
        };
        f32x4 zx[2][2][NTB];                                           // z1, then xhat
        char* const park = smem + B::OFF_H + tid * 16;                 // [8 slots][512 lanes][16 B] = the 64 KB of the H images
#pragma unroll
        for (int nh = 0; nh < 2; nh++)
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const int f = 256 * nh + 32 * wave + 16 * rb + 4 * q4;
#pragma unroll
                for (int tb = 0; tb < NTB; tb++) {
                    int tok = tok0 + 16 * tb + t16;
                    if (tok >= M) tok = M - 1;
                    const size_t off = (size_t)tok * MST_D + f;
#if MST_TB_PROBE == 1
                    const f32x4 r = {1.f, 2.f, 3.f, (float)off};
                    zx[nh][rb][tb] = f32x4{(float)tok, 1.f, (float)f, 2.f};
#else
                    const f32x4 r = *reinterpret_cast<const f32x4*>(g + off);
                    zx[nh][rb][tb] = join4_f16(*reinterpret_cast<const uint2*>(z1h + off), *reinterpret_cast<const uint2*>(z1l + off));
#endif
                    acc[nh][rb][tb] += r;
                    // g(x1) of the upper feature half waits in the dead H images (16 bytes per lane and slot, lane-linear: private to the
                    // lane, no barrier, no bank conflict): with acc AND xhat in registers the stage was 81 spilled dwords
                    if (nh == 1) *reinterpret_cast<f32x4*>(park + (rb * NTB + tb) * 8192) = acc[nh][rb][tb];
                }
                // (one feature half of loads at a time: hoisted together the 48 loads of the stage are 128 registers, and a spilled
                // register of a weight fragment still in flight is a corrupted register -- tools/audit_stream_isa.py)
                if (rb == 1) tail_fence();
            }
        float mw[NTB], m2[NTB];
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) {
            const f32x4 t = (zx[0][0][tb] + zx[0][1][tb]) + (zx[1][0][tb] + zx[1][1][tb]);
            mw[tb] = quad_sum((t[0] + t[1]) + (t[2] + t[3])) * (1.0f / 64.0f);
        }
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) {
            f32x4 sq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nh = 0; nh < 2; nh++)
#pragma unroll
                for (int rb = 0; rb < 2; rb++) {
                    const f32x4 dd = zx[nh][rb][tb] - mw[tb];
                    sq = __builtin_elementwise_fma(dd, dd, sq);
                }
            m2[tb] = quad_sum((sq[0] + sq[1]) + (sq[2] + sq[3]));
            if (q4 == 0) exch[wave * 64 + 16 * tb + t16] = make_float2(mw[tb], m2[tb]);
        }
        tail_barrier();
        float rstd[NTB];
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) {
            const float2* e = exch + 16 * tb + t16;
            const float2 e0 = e[0], e1 = e[64], e2 = e[128], e3 = e[192], e4 = e[256], e5 = e[320], e6 = e[384], e7 = e[448];
            const float mu = (((e0.x + e1.x) + (e2.x + e3.x)) + ((e4.x + e5.x) + (e6.x + e7.x))) * 0.125f;
            const float d0 = e0.x - mu, d1_ = e1.x - mu, d2_ = e2.x - mu, d3 = e3.x - mu, d4 = e4.x - mu, d5 = e5.x - mu, d6 = e6.x - mu, d7 = e7.x - mu;
            const float within = ((e0.y + e1.y) + (e2.y + e3.y)) + ((e4.y + e5.y) + (e6.y + e7.y));
            const float between = (fmaf(d0, d0, d1_ * d1_) + fmaf(d2_, d2_, d3 * d3)) + (fmaf(d4, d4, d5 * d5) + fmaf(d6, d6, d7 * d7));
            rstd[tb] = ln_rstd(fmaf(64.0f, between, within));
#pragma unroll
            for (int nh = 0; nh < 2; nh++)
#pragma unroll
                for (int rb = 0; rb < 2; rb++) zx[nh][rb][tb] = (zx[nh][rb][tb] - mu) * rstd[tb];      // xhat
        }
        tail_barrier();                                                // the exchange area is read: the second round may overwrite it
        // a = gamma g;  the rows' sums of a and a xhat: in-lane, the four q4 groups, then the eight waves through LDS
        float s1[NTB], s2[NTB];
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) { s1[tb] = 0.f; s2[tb] = 0.f; }
#pragma unroll
        for (int nh = 0; nh < 2; nh++)
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const f32x4 gm = *reinterpret_cast<const f32x4*>(g1s + 256 * nh + 32 * wave + 16 * rb + 4 * q4);
#pragma unroll
                for (int tb = 0; tb < NTB; tb++) {
                    const f32x4 gy = nh == 1 ? *reinterpret_cast<const f32x4*>(park + (rb * NTB + tb) * 8192) : acc[nh][rb][tb];
                    const f32x4 a = gy * gm;
                    const f32x4 ax = a * zx[nh][rb][tb];
                    s1[tb] += (a[0] + a[1]) + (a[2] + a[3]);
                    s2[tb] += (ax[0] + ax[1]) + (ax[2] + ax[3]);
                }
            }
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) {
            s1[tb] = quad_sum(s1[tb]);
            s2[tb] = quad_sum(s2[tb]);
            if (q4 == 0) exch[wave * 64 + 16 * tb + t16] = make_float2(s1[tb], s2[tb]);
        }
        tail_barrier();
        float c1[NTB], c2[NTB];
#pragma unroll
        for (int tb = 0; tb < NTB; tb++) {
            const float2* e = exch + 16 * tb + t16;
            const float2 e0 = e[0], e1 = e[64], e2 = e[128], e3 = e[192], e4 = e[256], e5 = e[320], e6 = e[384], e7 = e[448];
            c1[tb] = (((e0.x + e1.x) + (e2.x + e3.x)) + ((e4.x + e5.x) + (e6.x + e7.x))) * (1.0f / MST_D);
            c2[tb] = (((e0.y + e1.y) + (e2.y + e3.y)) + ((e4.y + e5.y) + (e6.y + e7.y))) * (1.0f / MST_D);
        }
        // column sums over the tile's 64 rows (WG): a lane's four tokens in-lane, the 16 token lanes by a shuffle ladder; every wave owns
        // its 64 features, so nothing crosses waves: lanes with t16 == 0 write the tile's sums
        float* tsum = reinterpret_cast<float*>(smem + B::OFF_SUM);
#pragma unroll
        for (int nh = 0; nh < 2; nh++)
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const int f = 256 * nh + 32 * wave + 16 * rb + 4 * q4;
                const f32x4 gm = *reinterpret_cast<const f32x4*>(g1s + f);
                f32x4 cg = {0.f, 0.f, 0.f, 0.f}, cb = cg, cc = cg;
#pragma unroll
                for (int tb = 0; tb < NTB; tb++) {
                    const int tok = tok0 + 16 * tb + t16;
                    const f32x4 gy = nh == 1 ? *reinterpret_cast<const f32x4*>(park + (rb * NTB + tb) * 8192) : acc[nh][rb][tb], xh = zx[nh][rb][tb];
                    const f32x4 a = gy * gm;
                    const f32x4 dz = (a - c1[tb] - xh * c2[tb]) * rstd[tb];
                    const uint32_t idx = (uint32_t)tok * (uint32_t)MST_D + (uint32_t)f;
                    const f32x4 db = {dz[0] * tail_drop_mul(d1, idx), dz[1] * tail_drop_mul(d1, idx + 1), dz[2] * tail_drop_mul(d1, idx + 2),
                                      dz[3] * tail_drop_mul(d1, idx + 3)};
                    *reinterpret_cast<uint2*>(dimg + slot1k(nh, rb, tb)) = pack4_f16(db[0], db[1], db[2], db[3]);
                    if (tok < M) {
#if MST_TB_PROBE != 1
                        *reinterpret_cast<f32x4*>(g + (size_t)tok * MST_D + f) = dz;
#endif
                        if constexpr (WG) {
                            cg += gy * xh;
                            cb += gy;
                            cc += db;
                        }
                    }
                }
                if constexpr (WG) {
#pragma unroll
                    for (int sh = 1; sh < 16; sh <<= 1)
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            cg[i] += __shfl_xor(cg[i], sh);
                            cb[i] += __shfl_xor(cb[i], sh);
                            cc[i] += __shfl_xor(cc[i], sh);
                        }
                    if (t16 == 0) {
                        *reinterpret_cast<f32x4*>(tsum + 0 * MST_D + f) = cg;
                        *reinterpret_cast<f32x4*>(tsum + 1 * MST_D + f) = cb;
                        *reinterpret_cast<f32x4*>(tsum + 2 * MST_D + f) = cc;
                    }
                }
#pragma unroll
                for (int tb = 0; tb < NTB; tb++) acc[nh][rb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    tail_fence();
    tail_barrier();                                                    // the dbr1 image (and the tile sums) are complete

    // =========================================================================================== d att = dbr1 W_out
#if MST_TB_PROBE != 2
    {
        const char* img = smem + B::OFF_IMG;
#define TB_ISSUE(J) tail_wload<((J) & 3) * 1024>(q[J], w_voff, (unsigned long long)(wnext + ((J) >> 2) * 4096))
        TB_ISSUE(0); TB_ISSUE(1); TB_ISSUE(2); TB_ISSUE(3); TB_ISSUE(4); TB_ISSUE(5); TB_ISSUE(6); TB_ISSUE(7);
#undef TB_ISSUE
        wnext += D * 1024;
        xread(img, RB1K(), 0, xs[0]);
        constexpr int NPP = C::P_FRAG / D;
#pragma unroll
        for (int pp = 0; pp + 1 < NPP; pp++) pass(img, RB1K(), RA4(), LD1(), pp * (D / 4), true, [](int) {});
        pass(img, RB1K(), RA4(), LD0(), (NPP - 1) * (D / 4), false, [](int) {});
    }
#endif
    tail_fence();
    {
        // -> the dead H images as a 64 x 1 KB image, then out as whole rows (16 bytes per lane); WG: the dbr1 rows and the tile sums too
        char* oimg = smem + B::OFF_H;
#pragma unroll
        for (int nh = 0; nh < 2; nh++)
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int tb = 0; tb < NTB; tb++) {
                    const f32x4 a = acc[nh][rb][tb];
                    *reinterpret_cast<uint2*>(oimg + slot1k(nh, rb, tb)) = pack4_f16(a[0], a[1], a[2], a[3]);
                }
        tail_barrier();
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int r = RPW * wave + j, tok = tok0 + r;
            if (tok < M) {
                const unsigned so = (unsigned)(r * 1024 + ((lane ^ (r & 15)) << 4));
                *reinterpret_cast<u32x4_t*>(datt + (size_t)tok * MST_D + lane * 8) = *reinterpret_cast<const u32x4_t*>(oimg + so);
                if constexpr (WG) *reinterpret_cast<u32x4_t*>(o.dbr1 + (size_t)tok * MST_D + lane * 8) = *reinterpret_cast<const u32x4_t*>(smem + B::OFF_IMG + so);
            }
        }
        if constexpr (WG) {
            const float* tsum = reinterpret_cast<const float*>(smem + B::OFF_SUM);
            for (int i = tid; i < 3 * MST_D; i += 512) o.part[(size_t)blockIdx.x * (3 * MST_D) + i] = tsum[i];
        }
    }
}

}  // namespace mst
