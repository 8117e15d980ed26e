// Lane map of v_permlane16_swap on gfx950 (exact integers): which rows (16-lane groups) of the two operands trade places.  Decides how two
// accumulator tiles are paired so that a lane ends up with 8 consecutive features of ONE token (a 16-byte LDS store instead of two 8-byte ones).
//   hipcc --offload-arch=gfx950 -O3 probe_permlane.hip -o bin/probe_permlane && bin/probe_permlane
#include <cstdio>
#include <hip/hip_runtime.h>
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 128; i++) printf("%u%c", h[i], (i % 16 == 15) ? '\n' : ' ');
    return 0;
}
