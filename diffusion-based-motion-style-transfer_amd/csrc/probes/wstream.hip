// Per-CU weight-stream rate, L2 -> VGPR (wave-private MFMA A fragments, no LDS hop) against the LDS-DMA ring the fused layer
// tail used in round 2.  Every workgroup (512 threads, one per CU) streams the SAME 2.5 MB "layer" -- 8 waves x 320 fragments of
// 1 KB, fragment-ordered so that one wave-instruction is one coalesced global_load_dwordx4 -- exactly what k_layer_tail's
// out-proj + FFN phases consume per 64-token tile.  Knobs: D = fragments in flight per wave (4 VGPRs each), MMA = two
// 32x32x16 MFMAs per fragment (the tail's arithmetic intensity), XLDS = the two token fragments of every k-step read from an
// LDS image (the shared activation operand).  Prints us per launch, the in-kernel clock and GB/s per CU.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 wstream.hip -o bin/wstream && bin/wstream [grid=197]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../mst_common.h"

__device__ unsigned long long g_stamp[1024][8][4];   // per wave: shader clock / 100 MHz clock at its start and end

// hipcc batches plain loads (all D at the loop's end, vmcnt(0) at its head), which exposes the whole L2 latency once per D
// fragments; the stream is therefore issued and counted by hand (cdna guide 5.7, form (ii)): load -> "=v", wait names it "+v".
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload(u32x4& d, const uint4* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p)); }
template <int N> __device__ __forceinline__ void wait_frag(u32x4& d) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "n"(N)); }

constexpr int TOTAL_FRAG = 2560;                 // 1-KB weight fragments per 64-token tile: 8 x (64 out-proj + 4 x (32 FFN1 + 32 FFN2))

typedef float f32x4v __attribute__((ext_vector_type(4)));

// WAVES x 64 threads; a wave consumes groups of RA weight fragments against ONE set of token fragments:
//   SHAPE 32: v_mfma_f32_32x32x16_f16, weight fragment = 32 rows x 16 k, token fragments = 2 x (32 tokens x 16 k), 2 RA MFMAs (32 cycles each)
//   SHAPE 16: v_mfma_f32_16x16x32_f16, weight fragment = 16 rows x 32 k, token fragments = 4 x (16 tokens x 32 k), 4 RA MFMAs (16 cycles each)
// RA = 1 (SHAPE 32) / 2 (SHAPE 16) at 8 waves is round 2's tile (32 weight rows x 64 tokens per wave); doubling RA is the reuse of the token
// fragments across the two feature halves of out-proj / FFN2, or a 64-row tile at 4 waves (one wave per SIMD, 512 registers).
template <int WAVES, int D, int RA, int SHAPE, bool XLDS>
__global__ __launch_bounds__(WAVES * 64) void k_wstream(const uint4* __restrict__ w, const uint4* __restrict__ xg, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NFRAG = TOTAL_FRAG / WAVES, NX = SHAPE == 32 ? 2 : 4;
    static_assert(D % RA == 0 && NFRAG % D == 0, "whole groups per unrolled pass");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (XLDS) {                                   // 64 KB activation image (64 tokens x 512 k f16), any content
        for (int i = tid; i < 4096; i += WAVES * 64) reinterpret_cast<uint4*>(smem)[i] = xg[i];
        __syncthreads();
    }
    unsigned long long t0 = 0, r0 = 0;
    if (lane == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    const uint4* p = w + (size_t)wave * NFRAG * 64 + lane;
    u32x4 q[D];
#pragma unroll
    for (int j = 0; j < D; j++) gload(q[j], p + j * 64);
    f32x16 a32[RA][2];
    f32x4v a16[RA][4];
#pragma unroll
    for (int a = 0; a < RA; a++) {
#pragma unroll
        for (int r = 0; r < 16; r++) { a32[a][0][r] = 0.f; a32[a][1][r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; t++) a16[a][t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    f16x8 xs[2][NX];                              // token fragments, double-buffered by group parity; no compiler-counted load beside the stream
#pragma unroll
    for (int t = 0; t < NX; t++)
#pragma unroll
        for (int i = 0; i < 8; i++) { xs[0][t][i] = (f16)(0.03f + 0.001f * (float)((lane + i + t) & 15)); xs[1][t][i] = (f16)(-0.05f + 0.002f * (float)((lane * 3 + i + t) & 7)); }
    auto xread = [&](int g, f16x8 (&x)[NX]) {     // token fragments of k-step g from the image (1-KB rows, XOR-swizzled 16-B chunks)
        if constexpr (SHAPE == 32) {
            const int l31 = lane & 31, c = (2 * (g & 31) + (lane >> 5)) ^ (l31 & 7);
#pragma unroll
            for (int t = 0; t < 2; t++) x[t] = *reinterpret_cast<const f16x8*>(smem + (32 * t + l31) * 1024 + (c << 4));
        } else {
            const int l15 = lane & 15, c = (4 * (g & 15) + (lane >> 4)) ^ (l15 & 7);
#pragma unroll
            for (int t = 0; t < 4; t++) x[t] = *reinterpret_cast<const f16x8*>(smem + (16 * t + l15) * 1024 + (c << 4));
        }
    };
    if (XLDS) xread(0, xs[0]);
    constexpr int GROUPS = D / RA;
    static_assert(GROUPS % 2 == 0 || GROUPS == 1, "token fragments are double-buffered by group parity");
#pragma unroll 1
    for (int f = 0; f < NFRAG; f += D) {
#pragma unroll
        for (int g = 0; g < GROUPS; g++) {
            constexpr int dummy = 0; (void)dummy;
            const int cur = GROUPS == 1 ? 0 : (g & 1);
            if (XLDS) xread(f / RA + g + 1, xs[GROUPS == 1 ? 0 : (cur ^ 1)]);       // one group ahead
#pragma unroll
            for (int a = 0; a < RA; a++) {
                const int j = g * RA + a;
                wait_frag<D - 1>(q[j]);
                const f16x8 wf = __builtin_bit_cast(f16x8, q[j]);
                if constexpr (SHAPE == 32) {
                    a32[a][0] = mfma_f16(wf, xs[cur][0], a32[a][0]);
                    a32[a][1] = mfma_f16(wf, xs[cur][1], a32[a][1]);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++) a16[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xs[cur][t], a16[a][t], 0, 0, 0);
                }
                gload(q[j], p + (f + D + j) * 64);        // the buffer is padded: the last prefetches read the pad
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < RA; a++) {
#pragma unroll
        for (int r = 0; r < 16; r++) s += a32[a][0][r] + a32[a][1][r];
#pragma unroll
        for (int t = 0; t < 4; t++) s += a16[a][t][0] + a16[a][t][1] + a16[a][t][2] + a16[a][t][3];
    }
#pragma unroll
    for (int j = 0; j < D; j++) s += (float)(q[j].x & 1);
    if (s == 12345.678f) out[tid] = s;            // keep everything live
    if (lane == 0) {
        g_stamp[blockIdx.x][wave][0] = t0; g_stamp[blockIdx.x][wave][1] = r0;
        g_stamp[blockIdx.x][wave][2] = __builtin_amdgcn_s_memtime(); g_stamp[blockIdx.x][wave][3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int WAVES, int D, int RA, int SHAPE, bool XLDS>
static void run(const char* name, int grid, const uint4* w, const uint4* xg, float* out) {
    auto kern = k_wstream<WAVES, D, RA, SHAPE, XLDS>;
    const int smem = XLDS ? 65536 : 0;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        const int iters = rep == 0 ? 200 : 6000;
        hipEventRecord(e0);
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), smem, 0, w, xg, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        ms /= iters;
    }
    static unsigned long long st[1024][8][4];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
    std::vector<double> us, ghz, first;            // workgroup = first wave's start -> last wave's end; `first` = its fastest wave
    for (int g = 0; g < grid; g++) {
        unsigned long long r0 = ~0ull, r1 = 0; double fw = 1e30, cyc = 0, rt = 0;
        for (int wv = 0; wv < WAVES; wv++) {
            r0 = std::min(r0, st[g][wv][1]); r1 = std::max(r1, st[g][wv][3]);
            fw = std::min(fw, (double)(st[g][wv][3] - st[g][wv][1]) * 0.01);
            cyc += (double)(st[g][wv][2] - st[g][wv][0]); rt += (double)(st[g][wv][3] - st[g][wv][1]);
        }
        if (rt <= 0) continue;
        us.push_back((double)(r1 - r0) * 0.01); ghz.push_back(cyc / rt * 0.1); first.push_back(fw);
    }
    std::sort(us.begin(), us.end()); std::sort(ghz.begin(), ghz.end()); std::sort(first.begin(), first.end());
    const double bytes = TOTAL_FRAG * 1024.0, clk = ghz[ghz.size() / 2];
    printf("%-34s grid %3d: %6.2f us/launch | workgroup median %6.2f us (max %6.2f; fastest wave %6.2f), clock %.3f GHz | %6.1f GB/s per CU, %5.2f TB/s chip | MFMA floor at that clock %.2f us\n",
           name, grid, ms * 1e3, us[us.size() / 2], us.back(), first[first.size() / 2], clk, bytes / us[us.size() / 2] * 1e-3,
           bytes * grid / us[us.size() / 2] * 1e-6, TOTAL_FRAG / 4 * 64.0 / (clk * 1e3));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 197;
    const size_t nw = (size_t)TOTAL_FRAG * 1024 + 64 * 1024;
    uint4 *w, *xg; float* out;
    hipMalloc(&w, nw); hipMalloc(&xg, 65536); hipMalloc(&out, 4096);
    std::vector<unsigned short> h(nw / 2);
    unsigned s = 4242; for (auto& e : h) { s = s * 1664525u + 1013904223u; e = 0x2800 | ((s >> 16) & 0x7FF) | ((s >> 3) & 0x8000); }   // +-[0.03, 0.06)
    hipMemcpy(w, h.data(), nw, hipMemcpyHostToDevice);
    hipMemcpy(xg, h.data(), 65536, hipMemcpyHostToDevice);
#define RUN(W, D, RA, SH, X) run<W, D, RA, SH, X>("waves=" #W " D=" #D " RA=" #RA " mfma" #SH " xlds=" #X, grid, w, xg, out)
    RUN(8, 8, 1, 32, false); RUN(8, 16, 1, 32, false);                                  // stream + MFMA, token fragments in registers
    RUN(8, 8, 1, 32, true); RUN(8, 16, 1, 32, true); RUN(8, 16, 2, 32, true);            // + token fragments from LDS; RA 2 = reuse across feature halves
    RUN(8, 8, 2, 16, false); RUN(8, 16, 2, 16, true); RUN(8, 16, 4, 16, true);           // 16x16x32 tiles, same wave tile
    RUN(4, 16, 2, 32, false); RUN(4, 16, 2, 32, true); RUN(4, 32, 4, 32, true); RUN(4, 32, 2, 32, true);   // one wave per SIMD, 64-row wave tiles
    RUN(4, 32, 4, 16, true); RUN(4, 32, 8, 16, true);
    return 0;
}
