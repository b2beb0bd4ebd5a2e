// What does HBM deliver on this box for the access shapes the patch kernel uses?
//   hipcc -O3 --offload-arch=gfx950 -o stream_bw stream_bw.hip && ./stream_bw
// 1. read-only, 16 B per lane, fully coalesced          (upper bound for reads)
// 2. copy, 16 B per lane                                (read + write)
// 3. rows of 24 B per lane read as dwordx4 + dwordx2    (the node-row shape: u1, u2)
// 4. as 3 from two arrays, one 24 B row written per lane (u1, u2 -> un: the nodal update's traffic)
// 5. (round 4) the byte stream of hq_k_brick_het per node -- 152 B: five 24-byte rows read (u1, u2, n_t, the element's
//    c1 | c2 | beta, and the share of ring rows, padding and cap planes the counters show: 10.1 GB per step on c3h =
//    152 B per node) + 8 B of tables read, one 24-byte row written -- with NO arithmetic: the ceiling of that kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_read16(const double2* __restrict__ a, double* __restrict__ out, size_t n)
{
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double2 v = a[i];
        s += v.x + v.y;
    }
    if (s == 1.2345e-300) out[0] = s;
}
__global__ void k_copy16(const double2* __restrict__ a, double2* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_rows24(const double* __restrict__ a, double* __restrict__ out, size_t nrows)
{
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrows; i += (size_t)gridDim.x * blockDim.x) {
        const double* q = a + 3 * i;
        s += q[0] + q[1] + q[2];
    }
    if (s == 1.2345e-300) out[0] = s;
}
__global__ void k_update24(const double* __restrict__ u1, const double* __restrict__ u2, double* __restrict__ un, size_t nrows)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrows; i += (size_t)gridDim.x * blockDim.x) {
        const double* p = u1 + 3 * i; const double* q = u2 + 3 * i; double* o = un + 3 * i;
        o[0] = 2 * p[0] - q[0]; o[1] = 2 * p[1] - q[1]; o[2] = 2 * p[2] - q[2];
    }
}

__global__ void k_het_shape(const double* __restrict__ u1, const double* __restrict__ u2, const double* __restrict__ nt,
                            const double* __restrict__ cf, const double* __restrict__ rg, const double* __restrict__ tb,
                            double* __restrict__ un, size_t nrows)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrows; i += (size_t)gridDim.x * blockDim.x) {
        const double *p = u1 + 3 * i, *q = u2 + 3 * i, *m = nt + 3 * i, *c = cf + 3 * i, *r = rg + 3 * i;
        double* o = un + 3 * i;
        const double t = tb[i];
        o[0] = (m[1] * p[0] - m[2] * q[0] + c[0] * r[0]) * m[0] + t;
        o[1] = (m[1] * p[1] - m[2] * q[1] + c[1] * r[1]) * m[0] + t;
        o[2] = (m[1] * p[2] - m[2] * q[2] + c[2] * r[2]) * m[0] + t;
    }
}

int main()
{
    const size_t nrows = 64ull << 20;            // 64 Mi rows of 24 B = 1.5 GiB per array (the 64M box's node arrays)
    const size_t bytes = nrows * 24;
    double *a, *b, *c, *out, *d, *e, *f, *g;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 8));
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&e, bytes)); CK(hipMalloc(&f, bytes)); CK(hipMalloc(&g, nrows * 8));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
    CK(hipMemset(d, 0, bytes)); CK(hipMemset(e, 0, bytes)); CK(hipMemset(f, 0, bytes)); CK(hipMemset(g, 0, nrows * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int grid : { 256 * 8, 256 * 32 }) for (int threads : { 256, 1024 }) {
        float ms;
#define RUN(name, launch, moved)                                                                 \
        launch; launch; CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));                       \
        for (int r = 0; r < reps; r++) launch;                                                    \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
        printf("grid %5d x %4d  %-34s %7.3f ms  %6.0f GB/s\n", grid, threads, name, ms / reps, (double)(moved) / (ms / reps * 1e-3) / 1e9);
        RUN("read 16 B/lane", (k_read16<<<grid, threads>>>((const double2*)a, out, bytes / 16)), bytes)
        RUN("copy 16 B/lane (r+w)", (k_copy16<<<grid, threads>>>((const double2*)a, (double2*)b, bytes / 16)), 2 * bytes)
        RUN("read 24 B rows", (k_rows24<<<grid, threads>>>(a, out, nrows)), bytes)
        RUN("u1,u2 -> un, 24 B rows (2r+1w)", (k_update24<<<grid, threads>>>(a, b, c, nrows)), 3 * bytes)
        RUN("het shape: 5 rows + 8 B read, 1 row written", (k_het_shape<<<grid, threads>>>(a, b, d, e, f, g, c, nrows)), 6 * bytes + nrows * 8)
    }
    return 0;
}
