// fp64 issue rates on gfx950: v_fma_f64 / v_add_f64 against v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64.
//   hipcc -O3 --offload-arch=gfx950 -o fp64_rate fp64_rate.hip && ./fp64_rate
// Settles the question DESIGN.md left open in round 1: would the +-1 (Walsh) transforms of the element
// product run faster on the matrix pipe?  Independent accumulator chains, 4 waves per SIMD, every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024) k_rate(int iters, double* out, double seed)
{
    const int tid = threadIdx.x;
    double a = seed + tid * 1e-6, b = 1.0 - seed * 1e-3;
    if (MODE == 0 || MODE == 1) {
        double c[8];
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = seed * i;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
                else           asm volatile("v_add_f64 %0, %1, %0" : "+v"(c[i]) : "v"(a));
            }
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s += c[i];
        if (s == 1.2345e-300) out[0] = s;
    } else {
        double4_t c[4];
#pragma unroll
        for (int i = 0; i < 4; i++) c[i] = (double4_t){ seed, 0.0, 1.0, 2.0 };
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (MODE == 2) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
                else {
                    double d = c[i].x;
                    d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
                    c[i].x = d;
                }
            }
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) s += c[i].x + c[i].y + c[i].z + c[i].w;
        if (s == 1.2345e-300) out[0] = s;
    }
}

int main()
{
    double* d_out;
    CK(hipMalloc(&d_out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256, iters = 20000;
    struct { const char* name; void (*k)(int, double*, double); double flop_per_thread_iter; int instr; } t[4] = {
        { "v_fma_f64 (8 independent chains)", k_rate<0>, 8 * 2.0, 8 },
        { "v_add_f64 (8 independent chains)", k_rate<1>, 8 * 1.0, 8 },
        /* 16x16x4: 16*16*4*2 flop per wave-instruction = 32 per lane; 4x4x4 (4 blocks): 4*4*4*4*2 = 512 per wave = 8 per lane */
        { "v_mfma_f64_16x16x4_f64 (4 accumulators)", k_rate<2>, 4 * 32.0, 4 },
        { "v_mfma_f64_4x4x4_4b_f64 (4 accumulators)", k_rate<3>, 4 * 8.0, 4 },
    };
    for (auto& q : t) {
        float ms;
        q.k<<<grid, 1024>>>(iters, d_out, 0.5);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        q.k<<<grid, 1024>>>(iters, d_out, 0.5);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = (double)grid * 1024 * iters * q.flop_per_thread_iter;
        const double winstr = (double)grid * 16 * iters * q.instr;
        printf("%-44s %8.3f ms  %7.2f TFLOP/s  %6.2f ns per wave-instruction per SIMD (4 waves per SIMD)\n", q.name, ms,
               flop / (ms * 1e-3) / 1e12, ms * 1e6 / (winstr / (grid * 4.0)));
    }
    return 0;
}
