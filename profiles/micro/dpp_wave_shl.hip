#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double shl1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);   // wave_shl:1 : lane i <- lane i+1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(double* out) { double v = 100.0 + threadIdx.x; out[threadIdx.x] = shl1(v); }
int main() { double* d; hipMalloc(&d, 64 * 8); k<<<1, 64>>>(d); double h[64]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i += 7) printf("%d:%g ", i, h[i]); printf("63:%g\n", h[63]); return 0; }
