// What does the immediate offset of global_load_lds_dwordx4 apply to?  Global words hold their own index;
// each variant DMAs 1 KiB and we print where in LDS it landed and which global words it carried.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* g, unsigned* out, int variant) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4096];   // 16 KiB
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = 0xFFFFFFFFu;
  __syncthreads();
  unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
  const char* gp = (const char*)g + 8192 + l * 16;             // lane's 16 B at global byte 8192 + 16*lane
  unsigned voff = 8192 + l * 16;
  unsigned keep;
  if (variant == 0)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gp), "s"(lds_base) : "memory");
  else if (variant == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gp), "s"(lds_base) : "memory");
  else if (variant == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_base), "s"(g) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:2048\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_base), "s"(g) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
  unsigned *g, *o; hipMalloc(&g, 1 << 20); hipMalloc(&o, 16384);
  unsigned* h = (unsigned*)malloc(1 << 20); for (int i = 0; i < (1 << 18); i++) h[i] = i; hipMemcpy(g, h, 1 << 20, hipMemcpyHostToDevice);
  unsigned r[4096];
  for (int v = 0; v < 4; v++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, v); hipMemcpy(r, o, 16384, hipMemcpyDeviceToHost);
    int first = -1, last = -1; for (int i = 0; i < 4096; i++) if (r[i] != 0xFFFFFFFFu) { if (first < 0) first = i; last = i; }
    printf("variant %d: LDS words [%d..%d] written (byte %d..), first value = global word %u (byte %u), lane1 value word %u\n", v, first, last, first * 4, first >= 0 ? r[first] : 0, first >= 0 ? r[first] * 4 : 0, first >= 0 ? r[first + 4] : 0);
  }
  return 0;
}
