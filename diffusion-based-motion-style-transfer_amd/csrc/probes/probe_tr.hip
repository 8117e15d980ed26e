#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
// M[row][col] = row*64 + col, 16 rows x 64 cols, 128-B rows, no swizzle
__global__ void k(int* out, int mode) {
  __shared__ __attribute__((aligned(16))) f16 lds[16 * 64];
  int l = threadIdx.x;
  for (int i = l; i < 16 * 64; i += 64) lds[i] = (f16)(float)i;
  __syncthreads();
  int i16 = l & 15, g = l >> 4;
  int row, col;
  if (mode == 0) { row = (i16 >> 2) + 4 * (g >> 1); col = 16 * (g & 1) + 4 * (i16 & 3); }   // guide: lane 4q+p -> row q, cols 4p
  else { row = (i16 & 3) + 4 * (g >> 1); col = 16 * (g & 1) + 4 * (i16 >> 2); }             // alternative
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + row * 64 + col));
  f16x4 hv = __builtin_bit_cast(f16x4, v);  // NOT bit_cast of v[j]: clang reads element 0 for every j
  for (int j = 0; j < 4; j++) out[l * 4 + j] = (int)(float)hv[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * 4); int h[256];
  for (int mode = 0; mode < 2; mode++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (lane: (row,col) x4)\n", mode);
    for (int l = 0; l < 64; l++) { printf("L%02d:", l); for (int j = 0; j < 4; j++) printf(" (%d,%d)", h[l*4+j] / 64, h[l*4+j] % 64); printf("\n"); }
  }
  return 0;
}
