// Prints the lane permutation of v_permlane32_swap / v_permlane16_swap on gfx950 (used to design the
// transposed wave reduction in render.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned l = threadIdx.x;
  unsigned a = 100 + l, b = 200 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto r2 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + l] = r2[0]; out[192 + l] = r2[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
  for (int i = 0; i < 4; ++i) {
    printf("%s:", names[i]);
    for (int l = 0; l < 64; l += 1) printf(" %u", h[64 * i + l]);
    printf("\n");
  }
  return 0;
}
