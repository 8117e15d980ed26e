// Per-phase PMC counters of the two dominant kernels: the product headers' phase hooks (QA_MARK / TAIL_MARK, empty in the library)
// become early exits -- every thread returns at mark g_stop -- so a kernel can be launched "up to phase p" and the counter
// difference between two stops is that phase's share.  One dispatch per stop value, in a fixed order; run under
//   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- bin/phase_pmc
// and read the dispatches in order (tools/phase_pmc.py).  All threads of a workgroup pass the same marks in the same order, and no
// wave waits on an arrival counter in front of a mark its peers return at, so every stop value drains.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 phase_pmc.hip -o bin/phase_pmc
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define MST_PROBE_BUILD
__device__ int g_stop;
#define QA_MARK(i) if ((i) == g_stop) return;
#define TAIL_MARK(i) if ((i) == g_stop) return;
#include "../mst_attn.h"
#include "../mst_tail.h"
using namespace mst;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int B = 64, S = 197, M = B * S;
    using C = TailCfg;
    const size_t nx = (size_t)(M + 64) * MST_D, nw = std::max<size_t>(C::LAYER_BYTES / 2, (size_t)3 * MST_D * MST_D);
    f16 *att, *wt, *hx, *hl, *out; float* v;
    CK(hipMalloc(&att, nx * 2)); CK(hipMalloc(&hx, nx * 2)); CK(hipMalloc(&hl, nx * 2)); CK(hipMalloc(&out, nx * 2)); CK(hipMalloc(&wt, nw * 2));
    CK(hipMalloc(&v, 4096 * 4));
    std::vector<unsigned short> h(std::max(nx, nw));
    unsigned s = 4242; for (auto& e : h) { s = s * 1664525u + 1013904223u; e = 0x2800 | ((s >> 16) & 0x7FF) | ((s >> 3) & 0x8000); }
    CK(hipMemcpy(att, h.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(hx, h.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemset(hl, 0, nx * 2));
    CK(hipMemcpy(wt, h.data(), nw * 2, hipMemcpyHostToDevice));
    std::vector<float> ones(4096, 1.0f);
    CK(hipMemcpy(v, ones.data(), 4096 * 4, hipMemcpyHostToDevice));
    auto ka = k_qkv_attention2<13>;
    CK(hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, QA2Tile<13>::SMEM));
    CK(hipFuncSetAttribute((const void*)k_layer_tail<4>, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
    // attention: stop at 1 (projection done), 2 (images written), 3 (scores + softmax), 4 = end; each twice (the first dispatch of a value warms up)
    const int a_stops[] = {1, 1, 2, 2, 3, 3, 4, 4};
    for (int st : a_stops) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stop), &st, sizeof(int)));
        hipLaunchKernelGGL(ka, dim3(B * MST_H), dim3(512), QA2Tile<13>::SMEM, 0, hx, wt, v, out, S);
        CK(hipDeviceSynchronize());
    }
    // tail: 1 (att image landed), 2 (out-proj done), 3 (LayerNorm1 done), 12 (FFN1 of chunk 0), 13 (+ GELU of chunk 0), 4 (FFN done), 5 = end
    const int t_stops[] = {1, 1, 2, 2, 3, 3, 12, 12, 13, 13, 4, 4, 5, 5};
    for (int st : t_stops) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stop), &st, sizeof(int)));
        hipLaunchKernelGGL(k_layer_tail<4>, dim3((M + C::BT - 1) / C::BT), dim3(512), C::SMEM, 0, att, wt, v, v, v, v, v, v, v, hx, hl, v, M);
        CK(hipDeviceSynchronize());
    }
    printf("done: 8 attention dispatches (stops 1 1 2 2 3 3 4 4), 14 tail dispatches (stops 1 1 2 2 3 3 12 12 13 13 4 4 5 5)\n");
    return 0;
}
