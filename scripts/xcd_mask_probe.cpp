// probe: which XCD do workgroups of a CU-masked stream land on?  (hipExtStreamCreateWithCUMask bit k -> XCC ?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = (int)(xcc & 0xf); out[2 * blockIdx.x + 1] = (int)((hwid >> 8) & 0xf); }
  // burn a little so that workgroups spread
  float x = threadIdx.x;
  for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
  if (x == 12345.f) out[0] = 0;
}
int main() {
  int* d; hipMalloc(&d, 2 * 512 * sizeof(int));
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<uint32_t> mask(8, 0);
    for (int k = 0; k < 256; ++k) {
      bool on = mode == 0 ? (k % 8 == 0) : mode == 1 ? (k < 32) : true;
      if (on) mask[k / 32] |= 1u << (k % 32);
    }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
    if (e != hipSuccess) { printf("mode %d: create failed %s\n", mode, hipGetErrorString(e)); continue; }
    hipMemsetAsync(d, 0xff, 2 * 512 * sizeof(int), s);
    hipLaunchKernelGGL(probe, dim3(512), dim3(256), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<int> h(1024);
    hipMemcpy(h.data(), d, 1024 * sizeof(int), hipMemcpyDeviceToHost);
    int hist[16] = {0}, cus[16] = {0};
    for (int i = 0; i < 512; ++i) { if (h[2 * i] >= 0 && h[2 * i] < 16) hist[h[2 * i]]++; if (h[2*i+1] >= 0 && h[2*i+1] < 16) cus[h[2*i+1]]++; }
    printf("mode %d (%s): workgroups per XCC:", mode, mode == 0 ? "bits k%8==0" : mode == 1 ? "bits 0..31" : "all");
    for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
    printf("   | per CU id (within SE):");
    for (int x = 0; x < 16; ++x) printf(" %d", cus[x]);
    printf("\n");
    hipStreamDestroy(s);
  }
  return 0;
}
