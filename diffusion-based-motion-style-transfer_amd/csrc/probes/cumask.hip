// Where do the workgroups of a CU-masked stream run?  hipExtStreamCreateWithCUMask with the first N bits / an interleaved pattern set; every
// workgroup records XCC_ID and HW_ID (SE, CU) and spins ~20 us so that a launch of 1024 workgroups has to spread over everything the mask allows.
// Then: two chains of short dependent kernels (a "side" chain of 8-workgroup kernels, a "main" chain of 256-workgroup kernels) alone, together on
// unmasked streams, and together on complementary masks -- does a reserved set of CUs keep the side chain at its solo pace?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 cumask.hip -o bin/cumask && bin/cumask
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_where(unsigned* out, int spin) {
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}
__global__ void k_spin(int spin) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}
static double chain_ms(hipStream_t s, int wgs, int spin, int n) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, s, spin);
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d\n", p.multiProcessorCount);
    unsigned* d;
    CK(hipMalloc(&d, 2 * 4096 * sizeof(unsigned)));
    std::vector<unsigned> h(2 * 4096);
    auto where = [&](const char* name, std::vector<uint32_t> mask) -> int {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: create -> %s\n", name, hipGetErrorString(e)); return 0; }
        hipLaunchKernelGGL(k_where, dim3(2048), dim3(256), 0, s, d, 2000);      // 100 MHz clock: 20 us
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, 2 * 2048 * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::set<unsigned> cus; int per_xcc[8] = {0};
        std::set<unsigned> cu_in_xcc[8];
        for (int i = 0; i < 2048; i++) {
            const unsigned xcc = h[2 * i] & 15, hw = h[2 * i + 1];
            const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;      // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
            cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
            if (xcc < 8) { per_xcc[xcc]++; cu_in_xcc[xcc].insert((se << 8) | (sh << 4) | cu); }
        }
        printf("%-34s distinct CUs used %3zu; per XCC (workgroups / distinct CUs):", name, cus.size());
        for (int x = 0; x < 8; x++) printf(" %d/%zu", per_xcc[x], cu_in_xcc[x].size());
        printf("\n");
        hipStreamDestroy(s);
        return 0;
    };
    const int words = (p.multiProcessorCount + 31) / 32;
    where("all bits", std::vector<uint32_t>(words, 0xFFFFFFFFu));
    { std::vector<uint32_t> m(words, 0); m[0] = 0xFFFFFFFFu; where("first 32 bits", m); }
    { std::vector<uint32_t> m(words, 0); m[0] = 0xFFu; where("first 8 bits", m); }
    { std::vector<uint32_t> m(words, 0x01010101u); where("every 8th bit", m); }
    { std::vector<uint32_t> m(words, 0xFFFFFFFFu); m[0] = 0; where("all but the first 32 bits", m); }
    { std::vector<uint32_t> m(words, 0xFEFEFEFEu); where("all but every 8th bit", m); }
    // chains
    hipStream_t a, b, ma, mb;
    CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
    std::vector<uint32_t> side(words, 0x01010101u), mainm(words, 0xFEFEFEFEu);
    CK(hipExtStreamCreateWithCUMask(&ma, words, mainm.data()));
    CK(hipExtStreamCreateWithCUMask(&mb, words, side.data()));
    const int N = 400;
    chain_ms(a, 8, 500, 20); chain_ms(b, 256, 3000, 20);
    printf("side chain alone (8 workgroups x 5 us, %d launches): %.2f ms\n", N, chain_ms(a, 8, 500, N));
    printf("main chain alone (256 workgroups x 30 us, %d launches): %.2f ms\n", N / 4, chain_ms(b, 256, 3000, N / 4));
    auto both = [&](hipStream_t sa, hipStream_t sb, const char* name) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, sa, 500);
            if (i % 4 == 0) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, sb, 3000);
        }
        hipStreamSynchronize(sa);
        const double ta = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        hipStreamSynchronize(sb);
        const double tb = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%s: side chain done at %.2f ms, main chain at %.2f ms\n", name, ta, tb);
    };
    both(a, b, "together, unmasked streams");
    both(mb, ma, "together, complementary masks (side: every 8th CU)");
    return 0;
}
