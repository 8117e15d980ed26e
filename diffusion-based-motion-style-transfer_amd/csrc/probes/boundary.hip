// Launch-boundary cost against the bytes the previous launch leaves dirty (MI355X_MICROARCH.md "launch overhead": 1.7-1.9 us
// + B / 6 TB/s).  A launch = 197 workgroups x 512 threads (the layer tail's shape): every workgroup idles ~20 us on the shader
// clock (so that launches of a stream are strictly serial and the stores sit at the END of the launch, as in k_layer_tail's
// LayerNorm2 and k_qkv_attention's PV store), then writes its share of B bytes with one of three store kinds and stamps the
// 100 MHz clock.  The gap = min(start stamp of launch i+1) - max(end stamp of launch i), median over 200 back-to-back launches.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 boundary.hip -o bin/boundary && bin/boundary
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: plain global_store_dwordx4 (write-back L2)   1: nontemporal (nt)   2: sc0 sc1 (write-through to memory)
template <int MODE>
__global__ __launch_bounds__(512) void k_write(u32x4* __restrict__ out, long long vec_per_wg, unsigned long long* __restrict__ stamp, int launch,
                                               long long spin_cycles) {
    const int tid = threadIdx.x;
    unsigned long long r0 = 0;
    if (tid == 0) r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin_cycles) __builtin_amdgcn_s_sleep(8);
    u32x4* p = out + (size_t)blockIdx.x * vec_per_wg;
    const u32x4 v{(unsigned)launch, (unsigned)tid, blockIdx.x, 7u};
    for (long long i = tid; i < vec_per_wg; i += 512) {
        if (MODE == 0) p[i] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, p + i);
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p + i), "v"(v) : "memory");
    }
    __syncthreads();
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[((size_t)launch * gridDim.x + blockIdx.x) * 2] = r0;
        stamp[((size_t)launch * gridDim.x + blockIdx.x) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int MODE>
static void run(const char* name, long long bytes, u32x4* buf, unsigned long long* dstamp, int grid) {
    const int N = 200;
    const long long vec_per_wg = bytes / 16 / grid;
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_write<MODE>, dim3(grid), dim3(512), 0, 0, buf, vec_per_wg, dstamp, i, (long long)40000);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> s((size_t)N * grid * 2);
    CK(hipMemcpy(s.data(), dstamp, s.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> gap, dur, spread;
    for (int i = 0; i + 1 < N; i++) {
        unsigned long long e = 0, b = ~0ull, e0 = ~0ull, b1 = 0;
        for (int g = 0; g < grid; g++) {
            e = std::max(e, s[((size_t)i * grid + g) * 2 + 1]);
            e0 = std::min(e0, s[((size_t)i * grid + g) * 2 + 1]);
            b = std::min(b, s[((size_t)(i + 1) * grid + g) * 2]);
            b1 = std::max(b1, s[((size_t)(i + 1) * grid + g) * 2]);
        }
        gap.push_back(((double)b - (double)e) * 0.01);
        spread.push_back(((double)e - (double)e0) * 0.01);
        dur.push_back(((double)e - (double)s[((size_t)i * grid) * 2]) * 0.01);
    }
    std::sort(gap.begin(), gap.end()); std::sort(dur.begin(), dur.end()); std::sort(spread.begin(), spread.end());
    printf("%-14s %6.1f MB dirty: gap last-end -> next-first-start %5.2f us (p10 %5.2f, p90 %5.2f); launch %5.1f us, ends spread over %4.2f us\n", name,
           bytes / 1e6, gap[gap.size() / 2], gap[gap.size() / 10], gap[gap.size() * 9 / 10], dur[dur.size() / 2], spread[spread.size() / 2]);
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 197;
    u32x4* buf; unsigned long long* dstamp;
    CK(hipMalloc(&buf, 128ll << 20));
    CK(hipMalloc(&dstamp, (size_t)200 * 256 * 16));
    CK(hipMemset(buf, 0, 128ll << 20));
    const long long sizes[] = {0, 197ll * 65536, 2 * 197ll * 65536, 4 * 197ll * 65536};   // 0, 12.9 (att), 25.8 (hx + hl), 51.6 MB
    for (int rep = 0; rep < 2; rep++)
        for (long long b : sizes) {
            run<0>("write-back", b, buf, dstamp, grid);
            if (b) run<1>("nontemporal", b, buf, dstamp, grid);
            if (b) run<2>("sc0 sc1", b, buf, dstamp, grid);
        }
    return 0;
}
