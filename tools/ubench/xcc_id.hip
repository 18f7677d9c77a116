#include <hip/hip_runtime.h>
__global__ void k(unsigned* out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { out[2*blockIdx.x] = x; out[2*blockIdx.x+1] = hw; }
}
int main() {
  unsigned* d; hipMalloc(&d, 8*4096);
  hipLaunchKernelGGL(k, dim3(2048), dim3(64), 0, 0, d);
  unsigned h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int mism = 0; int hist[16] = {0};
  for (int b = 0; b < 2048; ++b) { unsigned xcc = h[2*b] & 0xf; hist[xcc]++; if (xcc != (b & 7)) mism++; }
  printf("mismatch vs bid&7: %d of 2048\n", mism);
  for (int i = 0; i < 16; ++i) printf("%d ", hist[i]); printf("\n");
  for (int b = 0; b < 24; ++b) printf("bid %d xcc_reg 0x%x hw_id 0x%x\n", b, h[2*b], h[2*b+1]);
  return 0;
}
