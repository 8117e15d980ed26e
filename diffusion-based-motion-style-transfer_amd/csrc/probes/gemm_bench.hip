// Microbenchmark of the LDS-DMA ring main loop: tile shape x ring depth x K, epilogue-free, on the FFN-sized
// problem (M = 12608 tokens).  Prints us per launch and the per-slab slope, so tile decisions rest on numbers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../mst_gemm_dma.h"
using namespace mst;

struct DEpiNone {                       // keeps the accumulators alive, writes (almost) nothing
    float* sink; int M;
    __device__ __forceinline__ int rows() const { return M; }
    template <int BT, int BF> static constexpr int smem_bytes() { return 0; }
    template <int BT, int BF, int MT, int NT>
    __device__ __forceinline__ void run(f32x16 (&acc)[1][MT][NT], int, int, char*) const {
        float k = 0.f;
        for (int m = 0; m < MT; m++) for (int n = 0; n < NT; n++) for (int r = 0; r < 16; r++) k += acc[0][m][n][r];
        if (k == 123.456f) sink[0] = k;
    }
};

template <int BT, int BF, int MT, int NT, int NS, int BK = 32>
float run(const f16* X, const f16* W, float* sink, int M, int N, int K, int iters) {
    using TL = DTile<BT, BF, MT, NT, NS, 1, BK>;
    auto kern = k_gemm_dma<BT, BF, MT, NT, NS, 1, RowsDirect, DEpiNone, BK>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TL::SMEM);
    dim3 grid((M + BT - 1) / BT, N / BF);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kern, grid, dim3(512), TL::SMEM, 0, RowsDirect{X, K}, W, K, K, 0, DEpiNone{sink, M});
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, grid, dim3(512), TL::SMEM, 0, RowsDirect{X, K}, W, K, K, 0, DEpiNone{sink, M});
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

#define CASE(BT, BF, MT, NT, NS) CASEK(BT, BF, MT, NT, NS, 32)
#define CASEK(BT, BF, MT, NT, NS, BK)                                                                              \
    {                                                                                                              \
        printf("BK=%d ", BK);                                                                                      \
        float t1 = run<BT, BF, MT, NT, NS, BK>(X, W, sink, M, N, 512, 50), t2 = run<BT, BF, MT, NT, NS, BK>(X, W, sink, M, N, 2048, 50); \
        int blocks = ((M + BT - 1) / BT) * (N / BF);                                                               \
        double fl = 2.0 * M * N * 2048;                                                                            \
        printf("tile %3dx%3d wave %dx%d ring %d: blocks %4d  K=512 %7.1f us  K=2048 %7.1f us  -> %.3f us/slab, fixed %.1f us, %.0f TFLOP/s at K=2048\n", \
               BT, BF, MT, NT, NS, blocks, t1, t2, (t2 - t1) / 48.0, t1 - 16 * (t2 - t1) / 48.0, fl / t2 * 1e-6);     \
    }

int main() {
    const int M = 12608, N = 1024, Mp = 12800;
    f16 *X, *W; float* sink;
    hipMalloc(&X, (size_t)Mp * 2048 * 2); hipMalloc(&W, (size_t)N * 2048 * 2); hipMalloc(&sink, 64);
    std::vector<unsigned short> h((size_t)Mp * 2048);
    unsigned s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = 0x3000 | ((s >> 16) & 0x7FF) | ((s >> 3) & 0x8000); }  // random f16 in +-[0.125, 0.25)
    hipMemcpy(X, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * 2048 * 2, hipMemcpyHostToDevice);
    CASE(64, 512, 2, 2, 4)
    CASE(64, 512, 2, 2, 3)
    CASE(128, 256, 2, 2, 3)
    CASE(128, 256, 2, 2, 4)
    CASE(128, 512, 2, 4, 3)
    CASE(256, 256, 4, 2, 3)
    CASE(256, 256, 4, 2, 4)
    CASE(256, 256, 2, 4, 4)
    CASE(128, 128, 2, 1, 4)
    CASEK(64, 512, 2, 2, 2, 64)
    CASEK(128, 256, 2, 2, 2, 64)
    CASEK(128, 256, 2, 2, 3, 64)
    CASEK(256, 256, 4, 2, 2, 64)
    CASEK(128, 512, 2, 4, 2, 64)
    return 0;
}
