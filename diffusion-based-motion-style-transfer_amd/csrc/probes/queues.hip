// How many independent launch chains does the runtime keep in flight at once?  (Round 5: VERDICT r4 "the 4th clip slice halves
// throughput and nobody knows why".)  K chains on K streams, each chain L dependent launches of G workgroups x 512 threads with
// 160 KB of LDS (one workgroup per CU, the shape of the sampling loop's kernels) that idle `us` microseconds on the 100 MHz clock.
// G x K <= 256, so with perfect concurrency the wall time per chain step is flat in K.  The host enqueues round-robin over the
// chains, one launch each, exactly as mst_sample_loop does.  Stamps (first start / last end per launch) give, per K: wall time per
// step, the mean number of chains with a kernel running, and the mean gap between a chain's consecutive launches.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 queues.hip -o bin/queues && bin/queues [us=30] [G=40] [L=400] [prio=0|1] [blocked=0|1|2]
//   prio=1: streams alternate between the three stream priorities (separate hardware-queue pools in ROCclr)
//   blocked=1: one MORE stream whose queue holds a wait (hipStreamWaitEvent on an event recorded behind the chains' last launches) for
//              the whole run -- what the CALLER's stream is to mst_sample_loop: ordered behind the loop by ev_out, idle otherwise.
//   blocked=2: the same stream waits for chain 0's progress again and again (an event recorded on chain 0 after its first launch and
//              then every 8 launches): its queue is parked behind a barrier packet from the first launch to the last
//   GPU_MAX_HW_QUEUES=n in the environment sets ROCclr's queue count per priority (default 4)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void k_idle(unsigned long long* __restrict__ stamp, int slot, long long ticks) {
    extern __shared__ char smem[];
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) smem[0] = 1;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - r0) < ticks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) {
        atomicMin(&stamp[2 * slot], r0);
        atomicMax(&stamp[2 * slot + 1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

int main(int argc, char** argv) {
    const double us = argc > 1 ? atof(argv[1]) : 30.0;
    const int G = argc > 2 ? atoi(argv[2]) : 40;
    const int L = argc > 3 ? atoi(argv[3]) : 400;
    const int prio = argc > 4 ? atoi(argv[4]) : 0;
    const int blocked = argc > 5 ? atoi(argv[5]) : 0;
    const int KMAX = 8;
    CK(hipFuncSetAttribute((const void*)k_idle, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("idle %.1f us per launch, %d workgroups per launch, %d launches per chain, priorities %s (range %d..%d), GPU_MAX_HW_QUEUES=%s\n", us, G, L,
           prio ? "mixed" : "default", lo, hi, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)");
    hipStream_t st[KMAX];
    for (int k = 0; k < KMAX; k++) {
        if (prio) {
            const int p = (k % 3 == 0) ? 0 : (k % 3 == 1 ? hi : lo);
            CK(hipStreamCreateWithPriority(&st[k], hipStreamNonBlocking, p));
        } else CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
    }
    hipStream_t waiter;
    hipEvent_t ev_done;
    CK(hipStreamCreateWithFlags(&waiter, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
    if (blocked) printf("one more stream waits for the chains' end (mode %d)\n", blocked);
    unsigned long long* dstamp;
    const size_t nst = (size_t)KMAX * L * 2;
    CK(hipMalloc(&dstamp, nst * 8));
    std::vector<unsigned long long> init(nst), s(nst);
    for (size_t i = 0; i < nst; i += 2) { init[i] = ~0ull; init[i + 1] = 0; }
    for (int K = 1; K <= KMAX; K++) {
        if ((long long)K * G > 256) break;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemcpy(dstamp, init.data(), nst * 8, hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int j = 0; j < L; j++) {
                for (int k = 0; k < K; k++)
                    hipLaunchKernelGGL(k_idle, dim3(G), dim3(512), 160 * 1024, st[k], dstamp, k * L + j, (long long)(us * 100.0));
                if (blocked == 2 && j == 0) {                 // the waiter's queue is parked from here on ...
                    CK(hipEventRecord(ev_done, st[0]));
                    CK(hipStreamWaitEvent(waiter, ev_done, 0));
                }
                if (blocked == 2 && j > 0 && j % 8 == 0) {    // ... and re-parked every 8 steps (a caller that keeps waiting on the loop)
                    CK(hipEventRecord(ev_done, st[0]));
                    CK(hipStreamWaitEvent(waiter, ev_done, 0));
                }
            }
            if (blocked == 1) {
                CK(hipEventRecord(ev_done, st[0]));
                CK(hipStreamWaitEvent(waiter, ev_done, 0));
            }
            const auto t1 = std::chrono::steady_clock::now();
            CK(hipDeviceSynchronize());
            const auto t2 = std::chrono::steady_clock::now();
            if (rep == 0) continue;
            CK(hipMemcpy(s.data(), dstamp, nst * 8, hipMemcpyDeviceToHost));
            // concurrency: sum of kernel durations over all chains / span; gaps: start(j+1) - end(j) within a chain
            double busy = 0, gap = 0;
            unsigned long long first = ~0ull, last = 0;
            for (int k = 0; k < K; k++)
                for (int j = 0; j < L; j++) {
                    const unsigned long long a = s[2 * (k * L + j)], b = s[2 * (k * L + j) + 1];
                    busy += (double)(b - a) * 0.01;
                    first = std::min(first, a);
                    last = std::max(last, b);
                    if (j + 1 < L) gap += ((double)s[2 * (k * L + j + 1)] - (double)b) * 0.01;
                }
            const double span = (double)(last - first) * 0.01;
            printf("K=%d: wall %.1f us per chain step (host enqueue %.1f), device span %.1f us per step, mean chains running %.2f, mean gap inside a chain %.2f us\n", K,
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / L, std::chrono::duration<double, std::micro>(t1 - t0).count() / L, span / L, busy / span,
                   gap / ((double)K * (L - 1)));
        }
    }
    return 0;
}
