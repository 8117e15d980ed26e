// K3 and K9 (+ K10 .. K12) as latency kernels: the pose embedding (mdm_forstyledataset.py:425-449) and the output projection with the
// diffusion update behind it (:452-478; inpainting_gaussian_diffusion.py:25-64), one workgroup per 64-frame tile.
//
// Why not the ring GEMM (k_gemm_dma with DEpiEmbedIn / DEpiEmbedOut, round 1-3): a sampling loop runs as three clip slices whose launches
// are single rounds of workgroups, so a launch lasts as long as ONE workgroup does -- and the ring kernels took 18.6 / 29 us per launch
// at a third of the batch exactly as at the full batch (rocprofv3 kernel trace of the three-slice loop, round 4): 9 .. 16 slabs of
// DMA -> wait -> barrier -> MFMA in series for 0.7 % of the step's FLOPs each.  Here, as in the two big kernels:
//   * the tile's activation rows (hi + lo: both GEMMs multiply their activation as a split pair) land in LDS in ONE
//     LDS-DMA burst and stay for the whole K range: no slab loop, two barriers in front of the epilogue;
//   * the weights are wave-private -- wave w owns the 16-row blocks w, w + 8, w + 16, ... of the output features, for all 64 frames -- so
//     they stream L2 -> VGPR as pre-packed 1-KB fragments (k_pack_wave_blocks) behind hand-counted waits and never touch the LDS;
//   * the epilogues are the ring kernels' own (DEpiEmbedIn::finish, DEpiEmbedOut::finish): only the accumulator -> LDS hop differs
//     (16x16x32 accumulator layout), so the arithmetic behind the GEMM is the same code, bit for bit.
#pragma once
#include "mst_common.h"
#include "mst_gemm_dma.h"

#ifndef EMB_MARK            // diagnostic builds (tools/experiments/r4_embed_stamps.py) stamp the phases; the product build has none
#define EMB_MARK(i)
#endif

namespace mst {

// W [rows][K] f16 (row stride ldw, zero beyond nrows) -> eight per-wave streams: for k32 < KS, for bi < NBW: the 16x16x32 A-operand
// fragment (1 KB, lane order: lane l = row l & 15, k 8 (l >> 4) .. + 7) of block w + 8 bi.  Blocks beyond the matrix read as zeros.
__global__ __launch_bounds__(256) void k_pack_wave_blocks(const f16* __restrict__ W, int ldw, int nrows, int KS, int NBW, f16* __restrict__ dst) {
    const int total = 8 * KS * NBW * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, fi = i >> 6;
        const int bi = fi % NBW, k32 = (fi / NBW) % KS, w = fi / (NBW * KS);
        const int row = 16 * (w + 8 * bi) + (lane & 15), kq = 8 * (lane >> 4);
        uint4 v = {0u, 0u, 0u, 0u};
        if (row < nrows) v = *reinterpret_cast<const uint4*>(W + (size_t)row * ldw + 32 * k32 + kq);
        reinterpret_cast<uint4*>(dst)[i] = v;
    }
}

// Loads / stores through pointers that were themselves LOADED (the sampling loop's tensors come from the LoopDev block in device
// memory): hipcc cannot tell their address space and emits flat_load / flat_store (out-of-order return: a full vmcnt(0) lgkmcnt(0) wait
// per use, and no batching).  They are global memory: say so.
template <class T> __device__ __forceinline__ T gload(const void* p) {
    return *(const __attribute__((address_space(1))) T*)(unsigned long long)p;
}
template <class T> __device__ __forceinline__ void gstore(void* p, const T& v) {
    *(__attribute__((address_space(1))) T*)(unsigned long long)p = v;
}

struct EmbCfg {
    static constexpr int BT = 64;                         // frames per workgroup
    static constexpr int D = 16;                          // weight fragments in flight per wave: a wave streams 36 .. 64 KB alone, so the stream is
                                                          // latency-bound (16 KB per L2 round trip)
    static constexpr int SMEM = 160 * 1024;
};

// one 1-KB LDS-DMA piece with `nl` active lanes (16 B each)
__device__ __forceinline__ void emb_glds(unsigned voff, unsigned long long sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

// The weight stream of one wave over NFR fragments with D in flight, and the MFMA loop around it.  `body(jc, wf)` multiplies fragment
// j (compile-time index) -- the caller knows which (k-step, block) it is.  `before(j)` runs in front of fragment j's wait (fragment
// reads of the next k-step).  Every wait is a constant: vmcnt(D - 1) while the stream refills, counting down over the last D.
template <int NFR, int D, class Before, class Body>
__device__ __forceinline__ void emb_stream(const char* wsrc, unsigned w_voff, Before before, Body body) {
    static_assert(NFR >= D && D % 4 == 0, "stream shape");
    u32x4 q[D];
    auto issue = [&](auto jc) __attribute__((always_inline)) {      // fragment j -> slot j % D
        constexpr int j = decltype(jc)::value;
        tail_wload<(j & 3) * 1024>(q[j % D], w_voff, (unsigned long long)(wsrc + (j >> 2) * 4096));
    };
    auto step = [&](auto self, auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < NFR) {
            before(jc);
            if constexpr (j + D <= NFR) tail_wwait<D - 1>(q[j % D]); else tail_wwait<NFR - 1 - j>(q[j % D]);
            body(jc, __builtin_bit_cast(f16x8, q[j % D]));
            if constexpr (j + D < NFR) issue(std::integral_constant<int, j + D>());
            self(self, std::integral_constant<int, j + 1>());
        }
    };
    auto prime = [&](auto self, auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < D) { issue(jc); self(self, std::integral_constant<int, j + 1>()); }
    };
    prime(prime, std::integral_constant<int, 0>());
    step(step, std::integral_constant<int, 0>());
}

// The diffusion update of one thread's (feature, 4 consecutive frames) items of a tile, in three stages so that nothing waits in
// series behind the GEMM: noise() -- Philox + Box-Muller for every item, pure VALU work placed under the kernel-start DMA burst;
// load() -- x_t, bias and the mask row flags of ALL items issued at once (one memory latency; DEpiEmbedOut::finish takes three passes
// of dependent loads); apply() -- the update itself (step_update, the same function as everywhere), stores, the next step's frame rows.
// NU items per thread: item u = thread + 512 u, feature it / 16, frame group it % 16.
template <int MODE, int NU>
struct OutItems {
    f32x4 nz[NU], xv[NU], mk1, mot1;          // (mk1, mot1): mask / motion of the thread's FIRST item that needs them, prefetched (item u1)
    int u1;
    float bu[NU];
    int rf[NU];
    // A thread's items share their frame group: it = thread + 512 u -> feature f0 + 32 u, frame group thread % 16.  So clip, frame and
    // the element index are computed ONCE (the first version redid two integer divisions and a 64-bit multiply-add per item and stage:
    // ~150 instructions x 12 items x 3 stages, 9 us of integer arithmetic per workgroup by the in-kernel stamps); item u adds 32 T u.
    int f0, nvalid, clip, t;
    size_t idx0; unsigned istep;
    __device__ __forceinline__ void init(const DEpiEmbedOut<MODE>& epi, int tok0) {
        constexpr int TG = EmbCfg::BT / 4;
        f0 = threadIdx.x / TG;
        int tok = tok0 + (threadIdx.x % TG) * 4;
        nvalid = tok < epi.total ? (epi.F - f0 + 31) / 32 : 0;          // items u < nvalid exist (feature f0 + 32 u < F)
        if (nvalid > NU) nvalid = NU;
        if (nvalid == 0) { tok = 0; f0 = 0; }                            // (nothing to do: loads below still want a valid address)
        clip = tok / epi.T;
        t = tok - clip * epi.T;
        idx0 = ((size_t)clip * epi.F + f0) * epi.T + t;
        istep = 32u * (unsigned)epi.T;
    }
    __device__ __forceinline__ void noise1(const StepArgs& sa, int u) {  // one item's four normals (Philox4x32-10 + Box-Muller: ~900 issue cycles per wave)
        nz[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sa.philox && u < nvalid) {
            float nrm[4];
            philox_normal4((unsigned)(t >> 2), (unsigned)(f0 + 32 * u), (unsigned)clip + sa.clip0, sa.step, sa.seed, nrm);
            nz[u] = f32x4{nrm[0], nrm[1], nrm[2], nrm[3]};
        }
        // pin the arithmetic HERE (between the weight stream's asm statements of the k-step it was written into): left alone, hipcc
        // sinks it to its first use behind the GEMM loop, where it is 6.5 us in series (in-kernel stamps, round 4)
        asm volatile("" : "+v"(nz[u]));
    }
    // every item's x_t, bias, row flag (and injected noise), issued back to back without a branch: an item beyond the thread's last one
    // re-reads that one (its values are never used)
    __device__ __forceinline__ void load(const DEpiEmbedOut<MODE>& epi, const StepArgs& sa) {
        const bool use_noise = !sa.philox && sa.noise != nullptr;
        const int ulast = nvalid > 0 ? nvalid - 1 : 0;
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int uc = u < ulast ? u : ulast;
            const size_t idx = idx0 + (size_t)istep * uc;
            xv[u] = gload<f32x4>(sa.x + idx);
            rf[u] = sa.rowflag ? (int)gload<unsigned char>(sa.rowflag + clip * epi.F + f0 + 32 * uc) : 2;
            bu[u] = epi.bias[f0 + 32 * uc];
            if (use_noise) nz[u] = gload<f32x4>(sa.noise + idx);
        }
    }
    // behind load()'s barrier (the row flags have landed): mask and motion of the first item that needs them (rows flagged 0 read
    // neither tensor -- all but 3 of 263 with the root pattern: a thread has at most one such item there), in flight across the
    // accumulator scatter; further flagged items of a thread are loaded in apply()
    __device__ __forceinline__ void load_masked(const StepArgs& sa) {
        mk1 = mot1 = f32x4{0.f, 0.f, 0.f, 0.f};
        u1 = NU;
        if (sa.mask == nullptr) return;
#pragma unroll
        for (int u = NU - 1; u >= 0; u--)
            if (u < nvalid && rf[u] != 0) u1 = u;
        if (u1 < NU) {
            const size_t idx = idx0 + (size_t)istep * u1;
            mk1 = gload<f32x4>(sa.mask + idx);
            if (sa.motion) mot1 = gload<f32x4>(sa.motion + idx);
        }
    }
    __device__ __forceinline__ void apply(const DEpiEmbedOut<MODE>& epi, const StepArgs& sa, float* tile) {
        constexpr int LDT = EmbCfg::BT + 4, TG = EmbCfg::BT / 4;
        const StepCoef sc = step_coef(sa.tab, sa.nsteps, sa.t, sa.eta);
        const bool blend = sa.mask != nullptr && sa.motion != nullptr, use_mask = sa.mask != nullptr;
        float* trow = tile + f0 * LDT + (threadIdx.x % TG) * 4;
#pragma unroll
        for (int u = 0; u < NU; u++) {
            if (u >= nvalid) continue;
            const size_t idx = idx0 + (size_t)istep * u;
            f32x4 mk = {0.f, 0.f, 0.f, 0.f}, mot = {0.f, 0.f, 0.f, 0.f};
            if (use_mask && rf[u] != 0) {
                if (u == u1) { mk = mk1; mot = mot1; }
                else {
                    mk = gload<f32x4>(sa.mask + idx);
                    if (blend) mot = gload<f32x4>(sa.motion + idx);
                }
            }
            const f32x4 acc4 = *reinterpret_cast<const f32x4*>(trow + 32 * u * LDT);
            f32x4 nx, pred;
            if (use_mask && rf[u] == 0) {
                // the row's mask is all zeros: out (1 - 0) + motion 0 and noise (1 - 0) are the identity -- the same values without the
                // blend arithmetic (most rows: all but 3 of 263 with the root pattern)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float p;
                    nx[j] = step_update<MODE == 2 ? 1 : 0>(sc, acc4[j] + bu[u], xv[u][j], nz[u][j], false, 0.f, 0.f, false, sa.clip, &p);
                    pred[j] = p;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float p;
                    nx[j] = step_update<MODE == 2 ? 1 : 0>(sc, acc4[j] + bu[u], xv[u][j], nz[u][j], blend, mk[j], mot[j], sa.mask_noise && use_mask, sa.clip, &p);
                    pred[j] = p;
                }
            }
            gstore(sa.sample + idx, nx);
            if (sa.xstart) gstore(sa.xstart + idx, pred);
            if (epi.xt_next) *reinterpret_cast<f32x4*>(trow + 32 * u * LDT) = nx;
        }
    }
};

// ------------------------------------------------------------------------------------------------------------------ output projection
// NBW: 16-feature blocks per wave (F <= 128 NBW); NX = 2: classifier-free guidance (cond and uncond rows, blended in the accumulators).
// LDS: hi image [0, 64 K) | lo image [64 K, 128 K): frame row r = 1 KB, 16-B chunk c at c ^ (r & 15) (conflict-free ds_read_b128 of the
// 16x16x32 B operand, as the layer tail's att image); the epilogue's [feature][frame] tile overlays them.
// KSN > 0: the NEXT step's pose embedding in the same launch (the sampling loop's steps j and j + 1 meet in this tile: x_{t-1} is on chip,
// the 64 frames are the same rows of the token stream).  The updated clip goes from the [feature][frame] tile into two f16 images
// [frame][kpad] (hi, lo) BEHIND the tile, a second streamed GEMM (KSN k-steps, wave w owns output blocks w, w + 8, w + 16, w + 24) multiplies
// them with W_in, and DEpiEmbedIn::finish writes the stream rows and the conditioning tokens of step j + 1: one launch, one boundary
// and the 16 MB round trip of the f16 frame rows less per step.
template <int NBW, int MODE, int NX, int KSN = 0>
__global__ __launch_bounds__(512) void k_embed_out(RowsFrames xs, const f16* __restrict__ wpk, DEpiEmbedOut<MODE> epi,
                                                   const f16* __restrict__ wpk_in = nullptr, DEpiEmbedIn epi_in = DEpiEmbedIn{}) {
    static_assert(KSN == 0 || (NX == 1 && MODE != 0 && NBW <= 3), "the fused next-step embedding is for plain sampling steps");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = EmbCfg;
    constexpr int KS = MST_D / 32, NFR = KS * NBW, D = (NBW == 3 && NX == 2) ? 4 : 8;               // (8 fragments in flight: the registers go to the staged epilogue; 4 under CFG at three blocks per wave, where 8 spilled one register -- 58.8 -> 59.0 clips/s)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t16 = lane & 15, q4 = lane >> 4;
    const int tok0 = blockIdx.x * C::BT;
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    EMB_MARK(0)
    const StepArgs sa = step_resolve(epi.sa);
    const char* wsrc = reinterpret_cast<const char*>(wpk) + (size_t)wave * NFR * 1024;
    const unsigned w_voff = (unsigned)lane * 16u;
    f32x4 acc[NX][NBW][4];
#pragma unroll
    for (int x = 0; x < NX; x++)
#pragma unroll
        for (int bi = 0; bi < NBW; bi++)
#pragma unroll
            for (int tb = 0; tb < 4; tb++) acc[x][bi][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the staged epilogue (OutItems): frame counts that are multiples of 4 (a float4 never straddles a clip), a diffusion step; the noise
    // is drawn up front only where the registers allow (one accumulator set)
    constexpr int NU = 4 * NBW;                                        // ceil(128 NBW features x 16 frame groups / 512 threads)
    constexpr bool EARLY_NOISE = NX == 1;
    const bool staged = MODE != 0 && NBW <= 3 && (epi.T & 3) == 0;     // (4 blocks per wave = 16 items per thread: too many registers)
    OutItems<MODE == 0 ? 1 : MODE, MODE == 0 ? 1 : NU> items;
    if constexpr (MODE != 0) items.init(epi, tok0);

    const unsigned xlane = (unsigned)t16 * 1024u, xswz = (unsigned)((q4 ^ t16) << 4);
#pragma unroll
    for (int x = 0; x < NX; x++) {
        if (x > 0) __syncthreads();                                    // everybody has read the cond rows
        // the tile's rows, hi then lo: wave w fills rows [8 w, 8 w + 8) of both images (16 pieces)
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = 8 * wave + j;
                const unsigned voff = xs.rowbyte(tok0, r + x * xs.BT) + (unsigned)((lane ^ (r & 15)) << 4);
                emb_glds(voff, (unsigned long long)(h ? xs.base_lo() : xs.base()), __builtin_amdgcn_readfirstlane(smem_base + h * 65536 + r * 1024));
            }
        EMB_MARK(1)
        f16x8 xh[2][4], xl[2][4];
        auto xread = [&](int k32, int p) {
            const char* src = smem + xlane + (((unsigned)k32 << 6) ^ xswz);
#pragma unroll
            for (int tb = 0; tb < 4; tb++) {
                xh[p][tb] = *reinterpret_cast<const f16x8*>(src + tb * 16384);
                xl[p][tb] = *reinterpret_cast<const f16x8*>(src + 65536 + tb * 16384);
            }
        };
        emb_stream<NFR, D>(wsrc, w_voff,
            [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j == 0) {
                    // the 16 row pieces are older than the D fragments behind them
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
                    __syncthreads();
                    EMB_MARK(2)
                    xread(0, 0);
                }
                if constexpr (j % NBW == 0 && j / NBW + 1 < KS) xread(j / NBW + 1, (j / NBW + 1) & 1);      // one k-step ahead
                // the noise of item k32, drawn in k-step k32: VALU work beside the MFMAs of a loop that mostly waits for its weight stream
                if constexpr (MODE != 0 && EARLY_NOISE && j % NBW == 0 && j / NBW < NU) { if (staged) items.noise1(sa, j / NBW); }
            },
            [&](auto jc, f16x8 wf) {
                constexpr int j = decltype(jc)::value;
                constexpr int k32 = j / NBW, bi = j % NBW;
                if (16 * (wave + 8 * bi) >= epi.F) return;             // (wave-uniform) a block beyond the matrix: its fragment is zeros
#pragma unroll
                for (int tb = 0; tb < 4; tb++) {
                    acc[x][bi][tb] = mfma16(wf, xh[k32 & 1][tb], acc[x][bi][tb]);
                    acc[x][bi][tb] = mfma16(wf, xl[k32 & 1][tb], acc[x][bi][tb]);
                }
            });
    }
    EMB_MARK(3)
    if constexpr (MODE != 0) {
        if (staged) {
            if constexpr (EARLY_NOISE) items.load(epi, sa);            // in flight across the barrier and the scatter below
        }
    }
    EMB_MARK(7)
    __syncthreads();                                                   // the images are dead: the tile overlays them
    if constexpr (MODE != 0 && EARLY_NOISE) { if (staged) items.load_masked(sa); }
    // accumulators -> [feature][frame] tile (LDT floats per feature row); under CFG the blend u + s (c - u) on the way
    constexpr int LDT = C::BT + 4;
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int tb = 0; tb < 4; tb++) {
        const int tl = 16 * tb + t16, tok = tok0 + tl;
        float gs = 0.f;
        if (NX == 2) gs = sa.scale[(tok < epi.total ? tok : epi.total - 1) / epi.T];
#pragma unroll
        for (int bi = 0; bi < NBW; bi++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int f = 16 * (wave + 8 * bi) + 4 * q4 + i;
                float v = acc[0][bi][tb][i];
                if (NX == 2) { const float u = acc[NX - 1][bi][tb][i]; v = u + gs * (v - u); }
                if (f < epi.F) tile[f * LDT + tl] = v;
            }
    }
    __syncthreads();
    EMB_MARK(4)
    if constexpr (MODE != 0) {
        if (staged) {
            if constexpr (!EARLY_NOISE) {                              // two accumulator sets: only now are registers free
#pragma unroll
                for (int u = 0; u < NU; u++) items.noise1(sa, u);
                items.load(epi, sa);
                items.load_masked(sa);
            }
            items.apply(epi, sa, tile);
            EMB_MARK(5)
            if constexpr (KSN == 0) {
                epi.template frames_next<C::BT>(tok0, 0, smem);
                EMB_MARK(6)
                return;
            } else {
                // ---- step j + 1's pose embedding.  (The host sends only frame counts that are multiples of 4 here: `staged` holds.)
                constexpr int RS = KSN * 64 + 16, IMG = C::BT * RS, NFR2 = KSN * 4, D2 = NFR2 < 16 ? NFR2 / 4 * 4 : 16;
                const int off_img = (epi.F * LDT * 4 + 15) & ~15;      // behind the tile's F feature rows (the host checked that both images fit)
                __syncthreads();                                       // the tile holds all of x_{t-1}
                char* img = smem + off_img;
                for (int it = tid; it < C::BT * KSN * 4; it += 512) {  // (frame, 8 features) items: conflict-free tile reads, 16-byte image writes
                    const int tl = it % C::BT, fg = it / C::BT;
                    f16x8 v, vl;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int f = 8 * fg + q;
                        const float xv = f < epi.F ? tile[f * LDT + tl] : 0.f;      // columns [F, kpad): the zero padding of the GEMM's K
                        v[q] = (f16)xv;
                        vl[q] = (f16)(xv - (float)v[q]);
                    }
                    *reinterpret_cast<f16x8*>(img + tl * RS + fg * 16) = v;
                    *reinterpret_cast<f16x8*>(img + IMG + tl * RS + fg * 16) = vl;
                }
                __syncthreads();
                f32x4 acc2[4][4];
#pragma unroll
                for (int bi = 0; bi < 4; bi++)
#pragma unroll
                    for (int tb = 0; tb < 4; tb++) acc2[bi][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
                f16x8 yh[2][4], yl[2][4];
                auto yread = [&](int k32, int p) {
                    const char* src = img + (unsigned)t16 * RS + (unsigned)k32 * 64u + (unsigned)q4 * 16u;
#pragma unroll
                    for (int tb = 0; tb < 4; tb++) {
                        yh[p][tb] = *reinterpret_cast<const f16x8*>(src + tb * 16 * RS);
                        yl[p][tb] = *reinterpret_cast<const f16x8*>(src + IMG + tb * 16 * RS);
                    }
                };
                yread(0, 0);
                emb_stream<NFR2, D2>(reinterpret_cast<const char*>(wpk_in) + (size_t)wave * NFR2 * 1024, w_voff,
                    [&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        if constexpr (j % 4 == 0 && j / 4 + 1 < KSN) yread(j / 4 + 1, (j / 4 + 1) & 1);
                    },
                    [&](auto jc, f16x8 wf) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int k32 = j / 4, bi = j % 4;
#pragma unroll
                        for (int tb = 0; tb < 4; tb++) {
                            acc2[bi][tb] = mfma16(wf, yh[k32 & 1][tb], acc2[bi][tb]);
                            acc2[bi][tb] = mfma16(wf, yl[k32 & 1][tb], acc2[bi][tb]);
                        }
                    });
                __syncthreads();                                       // tile and images are dead: the fp32 rows overlay them
                constexpr int LD = MST_D * 4 + 16;
#pragma unroll
                for (int tb = 0; tb < 4; tb++) {
                    char* trow = smem + (16 * tb + t16) * LD;
#pragma unroll
                    for (int bi = 0; bi < 4; bi++) *reinterpret_cast<f32x4*>(trow + (16 * (wave + 8 * bi) + 4 * q4) * 4) = acc2[bi][tb];
                }
                __syncthreads();
                epi_in.template finish<C::BT>(tok0, smem);
                EMB_MARK(6)
                return;
            }
        }
    }
    epi.template finish<C::BT>(sa, tok0, 0, smem);
    EMB_MARK(6)
}

// ------------------------------------------------------------------------------------------------------------------ pose embedding
// KS = kpad / 32 k-steps.  LDS: hi image | lo image, frame row r at r * RS, RS = 2 kpad + 16 bytes (the 16 spread the banks: RS / 4 is
// 4 mod 16, so sixteen rows start on sixteen different multiples of 4 dwords -- conflict-free ds_read_b128); one DMA piece per row
// with kpad / 8 active lanes.  Wave w owns output blocks w, w + 8, w + 16, w + 24 (all 512 features: 4 blocks per wave).
template <int KS>
__global__ __launch_bounds__(512) void k_embed_in(const f16* __restrict__ xhi, const f16* __restrict__ xlo, int kpad,
                                                  const f16* __restrict__ wpk, DEpiEmbedIn epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = EmbCfg;
    constexpr int NBW = 4, NFR = KS * NBW, D = NFR < C::D ? NFR / 4 * 4 : C::D, RS = KS * 64 + 16, IMG = C::BT * RS;
    static_assert(2 * IMG <= C::SMEM && C::BT * (MST_D * 4 + 16) <= C::SMEM, "LDS map");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t16 = lane & 15, q4 = lane >> 4;
    const int tok0 = blockIdx.x * C::BT;
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const char* wsrc = reinterpret_cast<const char*>(wpk) + (size_t)wave * NFR * 1024;
    const unsigned w_voff = (unsigned)lane * 16u;
    f32x4 acc[NBW][4];
#pragma unroll
    for (int bi = 0; bi < NBW; bi++)
#pragma unroll
        for (int tb = 0; tb < 4; tb++) acc[bi][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // rows [8 w, 8 w + 8) of both images: one piece per row, lanes < kpad / 8 (a row is kpad f16 = kpad / 8 chunks of 16 B)
    if (lane < KS * 4) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = 8 * wave + j;
                int tok = tok0 + r;
                if (tok >= epi.total) tok = epi.total - 1;
                const unsigned voff = (unsigned)tok * (unsigned)(kpad * 2) + (unsigned)lane * 16u;
                emb_glds(voff, (unsigned long long)(h ? xlo : xhi), __builtin_amdgcn_readfirstlane(smem_base + h * IMG + r * RS));
            }
    }
    f16x8 xh[2][4], xl[2][4];
    auto xread = [&](int k32, int p) {
        const char* src = smem + (unsigned)t16 * RS + (unsigned)k32 * 64u + (unsigned)q4 * 16u;
#pragma unroll
        for (int tb = 0; tb < 4; tb++) {
            xh[p][tb] = *reinterpret_cast<const f16x8*>(src + tb * 16 * RS);
            xl[p][tb] = *reinterpret_cast<const f16x8*>(src + IMG + tb * 16 * RS);
        }
    };
    emb_stream<NFR, D>(wsrc, w_voff,
        [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j == 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");      // the row pieces are older than the D fragments
                __syncthreads();
                xread(0, 0);
            }
            if constexpr (j % NBW == 0 && j / NBW + 1 < KS) xread(j / NBW + 1, (j / NBW + 1) & 1);
        },
        [&](auto jc, f16x8 wf) {
            constexpr int j = decltype(jc)::value;
            constexpr int k32 = j / NBW, bi = j % NBW;
#pragma unroll
            for (int tb = 0; tb < 4; tb++) {
                acc[bi][tb] = mfma16(wf, xh[k32 & 1][tb], acc[bi][tb]);
                acc[bi][tb] = mfma16(wf, xl[k32 & 1][tb], acc[bi][tb]);
            }
        });
    __syncthreads();                                                   // the images are dead: the fp32 rows overlay them
    constexpr int LD = MST_D * 4 + 16;
#pragma unroll
    for (int tb = 0; tb < 4; tb++) {
        char* trow = smem + (16 * tb + t16) * LD;
#pragma unroll
        for (int bi = 0; bi < NBW; bi++) *reinterpret_cast<f32x4*>(trow + (16 * (wave + 8 * bi) + 4 * q4) * 4) = acc[bi][tb];
    }
    __syncthreads();
    epi.template finish<C::BT>(tok0, smem);
}

}  // namespace mst
