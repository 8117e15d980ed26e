// In-kernel clock and phase breakdown of the fused layer tail at the headline shape (12608 tokens = 197 workgroups):
// wave 0 of every workgroup stamps s_memtime / s_memrealtime at the phase boundaries (see attn_clock.hip).
// Build with -DTAIL_NODMA or -DTAIL_NOMMA for the two ablations (results are then garbage; timing only).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define TAIL_STAMP
__device__ unsigned long long g_tail_stamp[1024][16];
#ifndef TAIL_HEADER
#define TAIL_HEADER "../mst_tail.h"
#endif
#include TAIL_HEADER
using namespace mst;

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, M = B * 197;
    using C = TailCfg;
    const size_t nx = (size_t)(M + 64) * MST_D, nw = (size_t)(C::P_STEPS + C::F_STEPS) * C::WSLAB / 2;
    f16 *att, *wt, *hx, *hl; float* v;
    hipMalloc(&att, nx * 4); hipMalloc(&hx, nx * 2); hipMalloc(&hl, nx * 2); hipMalloc(&wt, nw * 2);
    hipMalloc(&v, 4096 * 4);
    std::vector<unsigned short> h(nx);
    unsigned s = 4242; for (auto& e : h) { s = s * 1664525u + 1013904223u; e = 0x2800 | ((s >> 16) & 0x7FF) | ((s >> 3) & 0x8000); }   // +-[0.03, 0.06)
    hipMemcpy(att, h.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(hx, h.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemset(hl, 0, nx * 2);
    hipMemcpy(wt, h.data(), nw * 2, hipMemcpyHostToDevice);
    std::vector<float> ones(4096, 1.0f);
    hipMemcpy(v, ones.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_layer_tail, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    const int grid = (M + C::BT - 1) / C::BT;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        const int iters = rep == 0 ? 100 : 20000;
        hipEventRecord(e0);
        for (int i = 0; i < iters; i++)
            hipLaunchKernelGGL(k_layer_tail, dim3(grid), dim3(512), C::SMEM, 0, att, wt, v, v, v, v, v, v, v, hx, hl, att + nx, M);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long st[1024][16];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tail_stamp), sizeof(st));
        std::vector<double> ghz, ph[4], tot, lap[3];
        for (int g = 0; g < grid; g++) {
            const double cyc = (double)(st[g][8] - st[g][0]), rt = (double)(st[g][9] - st[g][1]);
            if (rt <= 0) continue;
            ghz.push_back(cyc / rt * 0.1);
            tot.push_back(rt * 0.01);
            for (int j = 0; j < 3; j++) lap[j].push_back((double)st[g][10 + j] / (cyc / rt * 100.0));   // cycles -> us at this workgroup's clock
            for (int p = 0; p < 4; p++) ph[p].push_back((double)(st[g][2 * p + 3] - st[g][2 * p + 1]) * 0.01);
        }
        // dispatch skew and drain: first start -> each workgroup's start / end on the 100 MHz counter (one launch = the last one)
        {
            unsigned long long t0 = ~0ull, t1 = 0; std::vector<double> st0, en;
            const int G = (int)ghz.size();
            for (int g = 0; g < G; g++) { t0 = std::min(t0, st[g][1]); t1 = std::max(t1, st[g][9]); }
            for (int g = 0; g < G; g++) { st0.push_back((double)(st[g][1] - t0) * 0.01); en.push_back((double)(st[g][9] - t0) * 0.01); }
            std::sort(st0.begin(), st0.end()); std::sort(en.begin(), en.end());
            printf("        starts after the first workgroup's: median %.2f, last %.2f us; ends: first %.2f, median %.2f, last %.2f us\n",
                   st0[G / 2], st0[G - 1], en[0], en[G / 2], en[G - 1]);
        }
        auto med = [](std::vector<double>& x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
        printf("rep %d: %.2f us/launch over %d launches | in-kernel (median over %zu workgroups of the last launch): clock %.3f GHz, "
               "workgroup %.2f us = out-proj loop %.2f + LN1/x1h %.2f + FFN loop %.2f (FFN1 steps %.2f, GELU + H image %.2f, FFN2 steps %.2f; wave 0's view: the first FFN2 barrier of a chunk also waits for the slowest wave's GELU) + LN2/store %.2f us\n",
               rep, ms * 1e3 / iters, iters, ghz.size(), med(ghz), med(tot), med(ph[0]), med(ph[1]), med(ph[2]), med(lap[0]), med(lap[1]), med(lap[2]), med(ph[3]));
    }
    return 0;
}
