// In-kernel clock and phase breakdown of the fused layer tail at the headline shape (12608 tokens = 197 workgroups): lane 0 of
// every wave stamps s_memtime (shader clock) / s_memrealtime (100 MHz) at the phase boundaries (mst_tail.h, TAIL_MARK).  The
// waves of a workgroup run decoupled between barriers, so a boundary's time is the LAST wave's stamp.  Diagnostic build only:
// the product library is compiled without MST_PROBE_BUILD and contains no stamp.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DTAIL_NTB=3] tail_clock.hip -o bin/tail_clock && bin/tail_clock [batch=64 [tokens]]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define MST_PROBE_BUILD
__device__ unsigned long long g_tail_stamp[1024][8][48];
#define TAIL_MARK(i) if ((threadIdx.x & 63) == 0) { g_tail_stamp[blockIdx.x][threadIdx.x >> 6][2 * (i)] = __builtin_amdgcn_s_memtime(); g_tail_stamp[blockIdx.x][threadIdx.x >> 6][2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); }
#ifndef TAIL_HEADER
#define TAIL_HEADER "../mst_tail.h"
#endif
#include TAIL_HEADER
using namespace mst;
#ifndef TAIL_NTB
#define TAIL_NTB 4           // -DTAIL_NTB=3 / 2: 48- / 32-token tiles
#endif
#ifdef TAIL_TRAIN            // -DTAIL_TRAIN: the training instantiation (dropout 0.1 at the three sites, every tape slot written)
#define KTAIL k_layer_tail_train<TAIL_NTB>
#else
#define KTAIL k_layer_tail<TAIL_NTB>
#endif

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, M = argc > 2 ? atoi(argv[2]) : B * 197;   // [batch] or [_ tokens]
    using C = TailCfg;
    const size_t nx = (size_t)(M + 64) * MST_D, nw = C::LAYER_BYTES / 2;
    f16 *att, *wt, *hx, *hl; float* v;
    hipMalloc(&att, nx * 2); hipMalloc(&hx, nx * 2); hipMalloc(&hl, nx * 2); hipMalloc(&wt, nw * 2);
    hipMalloc(&v, 4096 * 4);
    std::vector<unsigned short> h(std::max(nx, nw));
    unsigned s = 4242; for (auto& e : h) { s = s * 1664525u + 1013904223u; e = 0x2800 | ((s >> 16) & 0x7FF) | ((s >> 3) & 0x8000); }   // +-[0.03, 0.06)
    hipMemcpy(att, h.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(hx, h.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemset(hl, 0, nx * 2);
    hipMemcpy(wt, h.data(), nw * 2, hipMemcpyHostToDevice);
    std::vector<float> ones(4096, 1.0f);
    hipMemcpy(v, ones.data(), 4096 * 4, hipMemcpyHostToDevice);
#ifdef TAIL_TRAIN
    f16* tp[8];
    for (int i = 0; i < 8; i++) hipMalloc(&tp[i], (size_t)(M + 64) * MST_FF * 2);
    f16 *oh, *ol; hipMalloc(&oh, nx * 2); hipMalloc(&ol, nx * 2);
    const TailDrop dd{0x1234567u, (uint32_t)(0.1 * 4294967296.0), 1.0f / 0.9f};
    const TailTrain tt{hx, hl, tp[0], tp[1], tp[2], tp[3], tp[4], tp[5], tp[6], tp[7], dd, dd, dd};
#endif
    if (hipFuncSetAttribute((const void*)KTAIL, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM) != hipSuccess) { printf("LDS attribute failed\n"); return 1; }
    const int grid = (M + 16 * TAIL_NTB - 1) / (16 * TAIL_NTB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        const int iters = rep == 0 ? 100 : 5000;
        hipEventRecord(e0);
        for (int i = 0; i < iters; i++)
#ifdef TAIL_TRAIN
            hipLaunchKernelGGL(KTAIL, dim3(grid), dim3(512), C::SMEM, 0, att, wt, v, v, v, v, v, v, v, oh, ol, v, M, tt);
#else
            hipLaunchKernelGGL(KTAIL, dim3(grid), dim3(512), C::SMEM, 0, att, wt, v, v, v, v, v, v, v, hx, hl, v, M);
#endif
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long st[1024][8][48];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tail_stamp), sizeof(st));
        std::vector<double> ghz, tot, ph[5], ln[7], ff[5];
        for (int g = 0; g < grid; g++) {
            unsigned long long rt[12], r0 = ~0ull;
            double cyc = 0, rsum = 0;
            for (int i = 0; i < 12; i++) { rt[i] = 0; for (int wv = 0; wv < 8; wv++) rt[i] = std::max(rt[i], st[g][wv][2 * i + 1]); }
            for (int wv = 0; wv < 8; wv++) { r0 = std::min(r0, st[g][wv][1]); cyc += (double)(st[g][wv][10] - st[g][wv][0]); rsum += (double)(st[g][wv][11] - st[g][wv][1]); }
            if (rsum <= 0) continue;
            ghz.push_back(cyc / rsum * 0.1);
            tot.push_back((double)(rt[5] - r0) * 0.01);
            ph[0].push_back((double)(rt[1] - r0) * 0.01);
            for (int p = 1; p < 5; p++) ph[p].push_back((double)(rt[p + 1] - rt[p]) * 0.01);
            {   // phase F as each wave lives it (mean over the 8 waves): FFN1 passes alone, GELU(0) alone, the waits on the arrival counters, FFN2 with the next GELU inside, FFN2(3)
                double a[5] = {0, 0, 0, 0, 0};
                for (int wv = 0; wv < 8; wv++) {
                    auto R = [&](int i) { return (double)st[g][wv][2 * i + 1]; };
                    a[0] += R(12) - R(3);                  // F1(0)
                    a[1] += R(13) - R(12);                 // G(0) + announce
                    double prev = R(13);
                    for (int hc = 0; hc < 3; hc++) { a[0] += R(14 + 3 * hc) - prev; a[2] += R(15 + 3 * hc) - R(14 + 3 * hc); a[3] += R(16 + 3 * hc) - R(15 + 3 * hc); prev = R(16 + 3 * hc); }
                    a[2] += R(23) - prev;
                    a[4] += R(4) - R(23);
                }
                for (int j = 0; j < 5; j++) ff[j].push_back(a[j] * 0.01 / 8);
            }
            { const unsigned long long seq[5] = {rt[2], rt[6], rt[7], rt[8], rt[3]};
              for (int p = 0; p < 4; p++) ln[p].push_back(((double)seq[p + 1] - (double)seq[p]) * 0.01); }
        }
        auto med = [](std::vector<double>& x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
        printf("rep %d: %.2f us/launch over %d launches | in-kernel (median over %zu workgroups of the last launch, boundaries = last wave): clock %.3f GHz, "
               "workgroup %.2f us = att image %.2f + out-proj %.2f + LN1 %.2f + FFN %.2f + LN2/store %.2f us\n",
               rep, ms * 1e3 / iters, iters, ghz.size(), med(ghz), med(tot), med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]));
        printf("        LN1: barrier + bias + residual + wave statistics + barrier %.2f, merge + normalise + x1 image %.2f, barrier %.2f, rest %.2f us\n",
               med(ln[0]), med(ln[1]), med(ln[2]), med(ln[3]));
        printf("        FFN per wave (mean of 8): 4 x FFN1 %.2f, GELU(0) alone %.2f, 3 x [FFN2 with the next chunk's GELU inside] %.2f, FFN2(3) + final barrier %.2f, waits on the arrival counters %.2f us\n",
               med(ff[0]), med(ff[1]), med(ff[3]), med(ff[4]), med(ff[2]));
    }
    return 0;
}
