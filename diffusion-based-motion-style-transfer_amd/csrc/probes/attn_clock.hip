// In-kernel clock and phase breakdown of the fused QKV + attention kernel at the headline shape (64 clips x 197 tokens,
// 256 workgroups).  Wave 0 of every workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) at the phase
// boundaries; the ratio of the two deltas is the shader clock the kernel actually ran at, the deltas are the phase times.
// Run after ~2 s of back-to-back launches so the DVFS state is the sustained one.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define MST_PROBE_BUILD
__device__ unsigned long long g_qa_stamp[1024][10];
#define QA_MARK(i) if (threadIdx.x == 0) { g_qa_stamp[blockIdx.x][2 * (i)] = __builtin_amdgcn_s_memtime(); g_qa_stamp[blockIdx.x][2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); }
#ifndef ATTN_HEADER
#define ATTN_HEADER "../mst_attn.h"
#endif
#include ATTN_HEADER
using namespace mst;

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, S = 197, NKT = 7;
    const size_t nx = (size_t)B * S * MST_D;
    f16 *hx, *w, *out; float* b;
    hipMalloc(&hx, nx * 2); hipMalloc(&out, nx * 2); hipMalloc(&w, (size_t)3 * MST_D * MST_D * 2); hipMalloc(&b, 3 * MST_D * 4);
    std::vector<unsigned short> h(nx);
    unsigned s = 777; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = 0x3000 | ((s >> 16) & 0x7FF) | ((s >> 3) & 0x8000); }
    hipMemcpy(hx, h.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), (size_t)3 * MST_D * MST_D * 2, hipMemcpyHostToDevice);
    hipMemset(b, 0, 3 * MST_D * 4);
    // attn_clock [B] [1]: a second argument times round 2's kernel (weights through the ring) instead of the role-swapped one
    const bool old = argc > 2 && atoi(argv[2]) == 1;
    auto kern = old ? k_qkv_attention<NKT> : k_qkv_attention2<13>;
    const int SMEM = old ? QATile<NKT>::SMEM : QA2Tile<13>::SMEM;
    printf("%s\n", old ? "k_qkv_attention<7> (round 2)" : "k_qkv_attention2<13>");
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        const int iters = rep == 0 ? 200 : 40000;        // rep 0: cold; rep 1, 2: ~1.4 s each, sustained
        hipEventRecord(e0);
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, dim3(B * MST_H), dim3(512), SMEM, 0, hx, w, b, out, S);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long st[1024][10];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_qa_stamp), sizeof(st));
        std::vector<double> ghz, ph[4], tot;
        for (int g = 0; g < B * MST_H; g++) {
            const double cyc = (double)(st[g][8] - st[g][0]), rt = (double)(st[g][9] - st[g][1]);
            if (rt <= 0) continue;
            ghz.push_back(cyc / rt * 0.1);
            tot.push_back(rt * 0.01);
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
        auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("rep %d: %.2f us/launch over %d launches | in-kernel (median over %zu workgroups of the last launch): clock %.3f GHz, "
               "workgroup %.2f us = projection %.2f + images %.2f + scores/softmax %.2f + PV/store %.2f us\n",
               rep, ms * 1e3 / iters, iters, ghz.size(), med(ghz), med(tot), med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]));
    }
    return 0;
}
