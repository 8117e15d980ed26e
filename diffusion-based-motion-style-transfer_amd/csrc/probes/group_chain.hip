// What would ONE persistent launch per denoise step cost in hand-offs?  (Round 5; docs/LAB_NOTES.md R5.2.)
//
// The proposed structure: 256 resident workgroups = 64 clips x 4 members.  Member i of a clip runs head i of the fused QKV + attention
// phase, then token tile i of the layer tail (13 blocks of 16 tokens per 197-token clip: 4 | 3 | 3 | 3), layer after layer; the only
// synchronisation is among the FOUR workgroups of a clip: a phase's output (att: 50 KB per head; the stream rows hx | hl: 128 / 96 KB
// per tile) is stored write-through, the storing waves drain, one lane adds to the clip's counter; consumers poll that counter,
// invalidate their L1 once and load (cdna guide 6, Guideline 16, recipe R1).
//
// This probe runs exactly that dependency structure and exactly those bytes (same buffers, same row ranges, LDS-DMA loads into the
// 160 KB LDS image, 8- and 16-byte stores in whole 128-byte lines), with the arithmetic of a phase replaced by an idle wait of the
// phase's measured duration (profiles/r04_phase_stamps.txt: attention workgroup 21.6 us, tail tile 39.6 us at 64 tokens, 33 at 48).
// What it measures is therefore what the structure ADDS: per layer, the time from a workgroup's last store to its partners' first use
// of the data, under the traffic of all 64 groups.  Every 16-byte chunk carries a tag (layer, row, chunk) that the consumer checks:
// the hand-off protocol is validated under load at the same time.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 group_chain.hip -o bin/group_chain
//   bin/group_chain [layers=8] [tA_us=21.6] [tT0_us=38.0] [tT_us=33.0] [spread=0|1] [bg=0|1] [early_inv=0|1]
//   spread=0: the members of a clip are blocks b, b+8, b+16, b+24 (one XCD under round-robin placement); 1: blocks 4c .. 4c+3 (four XCDs)
//   bg=1: every wave keeps eight 1-KB weight-fragment loads in flight during the idle wait (the real kernels' stream)
//   early_inv=1: the L1 invalidate is issued BEFORE the poll (legal here: the CU reads none of the handed-off lines in between)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int S = 197, D = 512, NCLIP = 64, ROWB = D * 2;          // f16 rows of 1 KB
constexpr int TILE0[5] = {0, 64, 112, 160, 208};

__device__ __forceinline__ unsigned tag_of(int buf, int layer, int row, int chunk) {
    return (unsigned)(buf * 0x9E3779B1u) ^ (unsigned)(layer * 0x85EBCA6Bu) ^ (unsigned)(row * 0xC2B2AE35u) ^ (unsigned)(chunk * 0x27D4EB2Fu);
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void store16_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store8_sc1(void* p, u32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }

struct Args {
    char *hx, *hl, *att;              // [NCLIP * S + pad][1 KB]
    unsigned* cnt;                    // [NCLIP][32] (one 128-B line per clip)
    const char* wts;                  // 4 MB of "weights" for the background stream
    unsigned long long* stamps;       // [256][layers][4]: wait A, wait T (100 MHz ticks), and errors
    unsigned* errors;
    int layers, spread, bg, early_inv;
    long long tA, tT0, tT;            // idle ticks (100 MHz)
};

__device__ __forceinline__ void idle_until(unsigned long long t_end, const Args& a, int wave, int lane, u32x4& sink) {
    if (a.bg) {
        const char* w = a.wts + (size_t)wave * 512 * 1024 + lane * 16;
        unsigned off = 0;
        while (__builtin_amdgcn_s_memrealtime() < t_end) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                u32x4 v;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(w + off) : "memory");
                off = (off + 1024) & (512 * 1024 - 1);
                asm volatile("s_waitcnt vmcnt(7)" : "+v"(v)::"memory");
                sink ^= v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        while (__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(8);
    }
}

// wait until the clip's counter reaches `target`; returns the 100 MHz ticks spent from entry to data-usable (after the invalidate + barrier)
__device__ __forceinline__ unsigned long long wait_clip(unsigned* cnt, unsigned target, int tid, int early_inv, unsigned* errors) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
        if (early_inv) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        unsigned spins = 0;
        while (__hip_atomic_load((__attribute__((address_space(1))) unsigned*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) { atomicAdd(errors + 1, 1u); break; }      // bounded: never hang the box
        }
        if (!early_inv) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return __builtin_amdgcn_s_memrealtime() - t0;
}
__device__ __forceinline__ void signal_clip(unsigned* cnt, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(512) void k_chain(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int clip, member;
    if (a.spread) { clip = blockIdx.x >> 2; member = blockIdx.x & 3; }
    else { const int grp = blockIdx.x >> 5, within = blockIdx.x & 31; clip = grp * 8 + (within & 7); member = within >> 3; }
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned* cnt = a.cnt + clip * 32;
    const size_t row0 = (size_t)clip * S;
    u32x4 sink = {0, 0, 0, 0};
    unsigned bad = 0;
    unsigned long long waitA = 0, waitT = 0;
    for (int l = 0; l < a.layers; l++) {
        // ------------------------------------------------------------------ phase A: head `member` of the clip
        if (l > 0) waitA += wait_clip(cnt, 8u * l, tid, a.early_inv, a.errors);
        {
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + a.tA;
            // the clip's stream rows (hi half: the projection's operand), 197 KB, in two bursts of <= 104 rows through the LDS image
            for (int half = 0; half < 2; half++) {
                const int r_lo = half * 104, r_hi = min(S, r_lo + 104);
                for (int r = r_lo + wave; r < r_hi; r += 8)
                    glds16(a.hx + (row0 + r) * ROWB + lane * 16, __builtin_amdgcn_readfirstlane(smem_base + (r - r_lo) * ROWB));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                for (int r = r_lo + wave; r < r_hi; r += 8) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (r - r_lo) * ROWB + lane * 16);
                    if (v[0] != tag_of(0, l, (int)row0 + r, lane)) bad++;
                }
                __syncthreads();
            }
            idle_until(t_end, a, wave, lane, sink);
            // att rows of the clip, this head's 128 columns = 256 B per row: a wave instruction writes 4 rows x 256 B (whole lines)
            for (int r = wave * 4 + (lane >> 4); r < S; r += 32) {
                const int chunk = member * 16 + (lane & 15);
                store16_sc1(a.att + (row0 + r) * ROWB + chunk * 16, u32x4{tag_of(2, l, (int)row0 + r, chunk), (unsigned)l, (unsigned)r, (unsigned)chunk});
            }
            signal_clip(cnt, tid);
        }
        // ------------------------------------------------------------------ phase T: token tile `member` of the clip
        waitT += wait_clip(cnt, 8u * l + 4u, tid, a.early_inv, a.errors);
        {
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + (member == 0 ? a.tT0 : a.tT);
            const int r_lo = TILE0[member], r_hi = min(S, TILE0[member + 1]);
            for (int r = r_lo + wave; r < r_hi; r += 8)
                glds16(a.att + (row0 + r) * ROWB + lane * 16, __builtin_amdgcn_readfirstlane(smem_base + (r - r_lo) * ROWB));
            // the residual rows (hi | lo): this workgroup's own output of the previous layer
            for (int r = r_lo + wave; r < r_hi; r += 8) {
                glds16(a.hx + (row0 + r) * ROWB + lane * 16, __builtin_amdgcn_readfirstlane(smem_base + 64 * 1024 + (r - r_lo) * ROWB));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int r = r_lo + wave; r < r_hi; r += 8) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (r - r_lo) * ROWB + lane * 16);
                if (v[0] != tag_of(2, l, (int)row0 + r, lane)) bad++;
                const u32x4 h = *reinterpret_cast<const u32x4*>(smem + 64 * 1024 + (r - r_lo) * ROWB + lane * 16);
                if (h[0] != tag_of(0, l, (int)row0 + r, lane)) bad++;
            }
            __syncthreads();
            idle_until(t_end, a, wave, lane, sink);
            // LayerNorm2's stores: hi and lo rows, 8 bytes per lane and half row (k_layer_tail's epilogue): the 16-B chunk's tag in its first dword
            for (int r = r_lo + wave; r < r_hi; r += 8) {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int c8 = half * 64 + lane;              // 8-byte piece of the row
                    const unsigned t0 = tag_of(0, l + 1, (int)row0 + r, c8 >> 1), t1 = tag_of(1, l + 1, (int)row0 + r, c8 >> 1);
                    store8_sc1(a.hx + (row0 + r) * ROWB + c8 * 8, (c8 & 1) ? u32x2{(unsigned)l, (unsigned)r} : u32x2{t0, 7u});
                    store8_sc1(a.hl + (row0 + r) * ROWB + c8 * 8, (c8 & 1) ? u32x2{(unsigned)l, (unsigned)r} : u32x2{t1, 9u});
                }
            }
            signal_clip(cnt, tid);
        }
    }
    if (tid == 0) {
        a.stamps[blockIdx.x * 2] = waitA;
        a.stamps[blockIdx.x * 2 + 1] = waitT;
    }
    if (bad) atomicAdd(a.errors, bad);
    if (sink[0] == 0x12345678u && sink[3] == 77u) a.errors[2] = 1;      // keep the background loads alive
}

__global__ void k_init(char* hx, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;                // one 16-B chunk each
    if (i >= rows * 64) return;
    const int r = i >> 6, c = i & 63;
    reinterpret_cast<u32x4*>(hx)[i] = u32x4{tag_of(0, 0, r, c), 7u, 0u, 0u};
}

int main(int argc, char** argv) {
    Args a{};
    a.layers = argc > 1 ? atoi(argv[1]) : 8;
    const double tA = argc > 2 ? atof(argv[2]) : 21.6, tT0 = argc > 3 ? atof(argv[3]) : 38.0, tT = argc > 4 ? atof(argv[4]) : 33.0;
    a.spread = argc > 5 ? atoi(argv[5]) : 0;
    a.bg = argc > 6 ? atoi(argv[6]) : 0;
    a.early_inv = argc > 7 ? atoi(argv[7]) : 0;
    a.tA = (long long)(tA * 100); a.tT0 = (long long)(tT0 * 100); a.tT = (long long)(tT * 100);
    const int rows = NCLIP * S + 64;
    CK(hipMalloc(&a.hx, (size_t)rows * ROWB)); CK(hipMalloc(&a.hl, (size_t)rows * ROWB)); CK(hipMalloc(&a.att, (size_t)rows * ROWB));
    CK(hipMalloc(&a.cnt, NCLIP * 128)); CK(hipMalloc(&a.stamps, 256 * 2 * 8)); CK(hipMalloc(&a.errors, 16));
    char* w; CK(hipMalloc(&w, 4 << 20)); CK(hipMemset(w, 1, 4 << 20)); a.wts = w;
    CK(hipMemset(a.errors, 0, 16));
    CK(hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 50;
    std::vector<float> ms;
    std::vector<unsigned long long> st(512);
    double wA = 0, wT = 0, wAmax = 0, wTmax = 0;
    for (int rep = 0; rep < reps; rep++) {
        hipLaunchKernelGGL(k_init, dim3((rows * 64 + 255) / 256), dim3(256), 0, 0, a.hx, rows);
        CK(hipMemsetAsync(a.cnt, 0, NCLIP * 128, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_chain, dim3(256), dim3(512), 160 * 1024, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 10) {
            ms.push_back(t);
            CK(hipMemcpy(st.data(), a.stamps, 512 * 8, hipMemcpyDeviceToHost));
            for (int b = 0; b < 256; b++) {
                wA += st[2 * b] * 0.01; wT += st[2 * b + 1] * 0.01;
                wAmax = std::max(wAmax, st[2 * b] * 0.01); wTmax = std::max(wTmax, st[2 * b + 1] * 0.01);
            }
        }
    }
    unsigned err[4]; CK(hipMemcpy(err, a.errors, 16, hipMemcpyDeviceToHost));
    std::sort(ms.begin(), ms.end());
    const double ideal = a.layers * (tA + tT0);
    const int n = (int)ms.size();
    printf("layers %d, tA %.1f, tT %.1f / %.1f us, members %s, background stream %d, early invalidate %d\n", a.layers, tA, tT0, tT, a.spread ? "4c..4c+3 (four XCDs)" : "b, b+8, b+16, b+24 (one XCD)", a.bg, a.early_inv);
    printf("  launch: median %.1f us (min %.1f, p90 %.1f); sum of the heavy member's phases %.1f us -> the structure adds %.1f us = %.2f us per hand-off\n", ms[n / 2] * 1e3, ms[0] * 1e3,
           ms[(int)(n * 0.9)] * 1e3, ideal, ms[n / 2] * 1e3 - ideal, (ms[n / 2] * 1e3 - ideal) / (2 * a.layers));
    printf("  waits per workgroup and launch (entry -> data usable, the partners' lag included): before attention mean %.1f us (max %.1f), before the tail mean %.1f (max %.1f)\n",
           wA / (256.0 * n), wAmax, wT / (256.0 * n), wTmax);
    printf("  tag errors %u, spin give-ups %u\n", err[0], err[1]);
    return err[0] || err[1];
}
