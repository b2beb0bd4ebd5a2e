// Micro-benchmark (round 3): what does a z-marching form of the assembled 27-point stencil + nodal update cost on a
// brick of uniformly refined nodes stored tile-major ([tile][z][y][x], TX x TY node tiles)?
//   hipcc -O3 --offload-arch=gfx950 -o march_stencil march_stencil.hip && ./march_stencil
// A workgroup owns one tile column over CZ planes: thread (x, y) loads its node of plane k+2 (u1, u2: needed for the
// update anyway) and, the first threads, one node of the plane's ring; w = u1 + beta (u1 - u2) of the planes k-1 .. k+2
// sits in a 4-slot LDS ring (one barrier per plane) or a 3-slot one (two barriers); the stencil of plane k reads 27 rows.
// Compared with what hq_k_patch_stencil moves (8x8x8 owned nodes + 488 halo rows gathered from Z-order): the ring of a
// plane is 148 rows for 512 owned nodes, and the owned rows are one contiguous 12 KB run per plane.
// Checked against a plain gather kernel on the same arrays (clamped neighbours at the brick's faces).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct coef { double P[6], Q[2], beta, m0, m1, m2; };
// (round 5) -DREAL=float: the STATE (u1, u2, un and the registers the loads land in) in floats, every sum in doubles -- the
// question behind libhq_solver_f32.so's 3 %: does the march get faster with half the bytes once TWO planes are in flight?
#ifndef REAL
#define REAL double
#endif
typedef REAL real;

template <int TX, int TY>
__device__ __host__ inline int64_t node_addr(int gx, int gy, int gz, int NX, int NY, int NZ)
{
    const int ntx = NX / TX;
    const int tx = gx / TX, ty = gy / TY;
    const int64_t tile = (int64_t)ty * ntx + tx;
    return (tile * NZ + gz) * (TX * TY) + (gy % TY) * TX + (gx % TX);
}

__device__ __forceinline__ void stencil27(const double* __restrict__ ctr, int px, int py, int pz_m, int pz_p, const coef& c, double f[3])
{
    // ctr: row of the node in the middle plane; planes below / above at row offsets pz_m / pz_p (ring slots)
#pragma unroll
    for (int dz = -1; dz <= 1; dz++) {
        const int zo = dz < 0 ? pz_m : (dz > 0 ? pz_p : 0);
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                const double* q = ctr + 3 * (px * dx + py * dy + zo);
                const double ux = q[0], uy = q[1], uz = q[2];
                const int ax = dx != 0, ay = dy != 0, az = dz != 0;
                f[0] = fma(c.P[ax + 2 * (ay + az)], ux, f[0]);
                f[1] = fma(c.P[ay + 2 * (ax + az)], uy, f[1]);
                f[2] = fma(c.P[az + 2 * (ax + ay)], uz, f[2]);
                if (dx && dy) { const double k = dx * dy > 0 ? c.Q[az] : -c.Q[az]; f[0] = fma(k, uy, f[0]); f[1] = fma(k, ux, f[1]); }
                if (dx && dz) { const double k = dx * dz > 0 ? c.Q[ay] : -c.Q[ay]; f[0] = fma(k, uz, f[0]); f[2] = fma(k, ux, f[2]); }
                if (dy && dz) { const double k = dy * dz > 0 ? c.Q[ax] : -c.Q[ax]; f[1] = fma(k, uz, f[1]); f[2] = fma(k, uy, f[2]); }
            }
    }
}

// reference: one thread per node, gathers from global
template <int TX, int TY>
__global__ void k_ref(const real* __restrict__ u1, const real* __restrict__ u2, real* __restrict__ un, int NX, int NY, int NZ, coef c)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)NX * NY * NZ) return;
    const int gx = i % NX, gy = (i / NX) % NY, gz = i / ((int64_t)NX * NY);
    double f[3] = { 0, 0, 0 };
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int x = min(max(gx + dx, 0), NX - 1), y = min(max(gy + dy, 0), NY - 1), z = min(max(gz + dz, 0), NZ - 1);
                const int64_t a = node_addr<TX, TY>(x, y, z, NX, NY, NZ);
                double w[3];
                for (int d = 0; d < 3; d++) w[d] = u1[3 * a + d] + c.beta * (u1[3 * a + d] - u2[3 * a + d]);
                const int ax = dx != 0, ay = dy != 0, az = dz != 0;
                f[0] = fma(c.P[ax + 2 * (ay + az)], w[0], f[0]);
                f[1] = fma(c.P[ay + 2 * (ax + az)], w[1], f[1]);
                f[2] = fma(c.P[az + 2 * (ax + ay)], w[2], f[2]);
                if (dx && dy) { const double k = dx * dy > 0 ? c.Q[az] : -c.Q[az]; f[0] = fma(k, w[1], f[0]); f[1] = fma(k, w[0], f[1]); }
                if (dx && dz) { const double k = dx * dz > 0 ? c.Q[ay] : -c.Q[ay]; f[0] = fma(k, w[2], f[0]); f[2] = fma(k, w[0], f[2]); }
                if (dy && dz) { const double k = dy * dz > 0 ? c.Q[ax] : -c.Q[ax]; f[1] = fma(k, w[2], f[1]); f[2] = fma(k, w[1], f[2]); }
            }
    const int64_t a = node_addr<TX, TY>(gx, gy, gz, NX, NY, NZ);
    for (int d = 0; d < 3; d++) un[3 * a + d] = (f[d] + c.m2 * u1[3 * a + d] - c.m1 * u2[3 * a + d]) / c.m0;
}

// marching kernel. SLOTS = 4: one barrier per plane; SLOTS = 3: two.
template <int TX, int TY, int SLOTS>
__global__ void __launch_bounds__(TX * TY)
k_march(const real* __restrict__ u1, const real* __restrict__ u2, real* __restrict__ un, int NX, int NY, int NZ, int CZ,
        int units_per_xcd, int nunits, coef c)
{
    constexpr int PX = 1, PY = TX + 2, PLANE = (TX + 2) * (TY + 2);
    constexpr int NRING = 2 * (TX + 2) + 2 * TY;
    __shared__ __align__(16) double s_w[3 * PLANE * SLOTS];
    const int unit = (int)(blockIdx.x & 7) * units_per_xcd + (int)(blockIdx.x >> 3);
    if (unit >= nunits) return;
    const int ntx = NX / TX, nty = NY / TY, ntiles = ntx * nty;
    // unit order inside an XCD's run: chunk-major, tile-minor; the XCD's tiles are a slab of tile rows
    const int tiles_per_xcd = ntiles / 8;
    const int xcd = unit / units_per_xcd, u = unit - xcd * units_per_xcd;
    const int chunk = u / tiles_per_xcd, tl = xcd * tiles_per_xcd + u % tiles_per_xcd;
    const int tx = tl % ntx, ty = tl / ntx;
    const int z0 = chunk * CZ, z1 = min(z0 + CZ, NZ);
    const int t = threadIdx.x, lx = t % TX, ly = t / TX;
    const int gx = tx * TX + lx, gy = ty * TY + ly;
    const int64_t tile_base = ((int64_t)ty * ntx + tx) * NZ * (TX * TY);
    const int myrow = (ly + 1) * PY + (lx + 1);
    // ring slot of this thread (t < NRING): its lattice offset and LDS row
    int rx = 0, ry = 0;
    if (t < TX + 2) { rx = t - 1; ry = -1; }
    else if (t < 2 * (TX + 2)) { rx = t - (TX + 2) - 1; ry = TY; }
    else if (t < 2 * (TX + 2) + TY) { rx = -1; ry = t - 2 * (TX + 2); }
    else if (t < NRING) { rx = TX; ry = t - 2 * (TX + 2) - TY; }
    const bool ring = t < NRING;
    const int rrow = (ry + 1) * PY + (rx + 1);
    const int rgx = min(max(tx * TX + rx, 0), NX - 1), rgy = min(max(ty * TY + ry, 0), NY - 1);
    const int rtile = (rgy / TY) * ntx + rgx / TX;
    const int64_t ring_base = (int64_t)rtile * NZ * (TX * TY) + (rgy % TY) * TX + (rgx % TX);

    real x1[3], x2[3], y1[3] = { 0, 0, 0 }, y2[3] = { 0, 0, 0 };
    double rsA[3], rsB[3];           // m2 u1 - m1 u2 of the planes k and k+1
    auto load = [&](int z) {
        const int zc = min(max(z, 0), NZ - 1);
        const int64_t a = tile_base + (int64_t)zc * (TX * TY) + t;
#pragma unroll
        for (int d = 0; d < 3; d++) { x1[d] = u1[3 * a + d]; x2[d] = u2[3 * a + d]; }
        if (ring) {
            const int64_t b = ring_base + (int64_t)zc * (TX * TY);
#pragma unroll
            for (int d = 0; d < 3; d++) { y1[d] = u1[3 * b + d]; y2[d] = u2[3 * b + d]; }
        }
    };
    auto put = [&](int z, double rs[3]) {
        double* img = s_w + 3 * PLANE * (((z % SLOTS) + SLOTS) % SLOTS);
#pragma unroll
        for (int d = 0; d < 3; d++) { img[3 * myrow + d] = x1[d] + c.beta * (x1[d] - x2[d]); rs[d] = c.m2 * x1[d] - c.m1 * x2[d]; }
        if (ring) {
#pragma unroll
            for (int d = 0; d < 3; d++) img[3 * rrow + d] = y1[d] + c.beta * (y1[d] - y2[d]);
        }
    };
    double dummy[3];
    load(z0 - 1); put(z0 - 1, dummy);
    load(z0); put(z0, rsA);
    load(z0 + 1); put(z0 + 1, rsB);
    for (int z = z0; z < z1; z++) {
        if (SLOTS == 4) load(z + 2);
        __syncthreads();
        const int s0 = ((z % SLOTS) + SLOTS) % SLOTS, sm = (s0 + SLOTS - 1) % SLOTS, sp = (s0 + 1) % SLOTS;
        double f[3] = { 0, 0, 0 };
        stencil27(s_w + 3 * (PLANE * s0 + myrow), PX, PY, PLANE * (sm - s0), PLANE * (sp - s0), c, f);
        const int64_t a = tile_base + (int64_t)z * (TX * TY) + t;
#pragma unroll
        for (int d = 0; d < 3; d++) un[3 * a + d] = (f[d] + rsA[d]) / c.m0;
#pragma unroll
        for (int d = 0; d < 3; d++) rsA[d] = rsB[d];
        if (SLOTS == 3) { load(z + 2); __syncthreads(); }
        put(z + 2, rsB);
    }
}


// marching kernel, plane sums: the 9 rows of the arriving plane are reduced ONCE to the in-plane sums the cube symmetry of
// S leaves distinct (22 numbers) and those feed the three output planes p-1, p, p+1: 27 LDS reads, ~82 fp64 operations
// per node instead of 81 / 153+; only the arriving plane has to be in LDS (2 slots, one barrier per plane).
template <int TX, int TY, int ABL>
__global__ void __launch_bounds__(TX * TY)
k_march2(const real* __restrict__ u1, const real* __restrict__ u2, real* __restrict__ un, int NX, int NY, int NZ, int CZ,
         int units_per_xcd, int nunits, coef c)
{
    constexpr int PY = TX + 2, PLANE = (TX + 2) * (TY + 2);
    constexpr int NRING = 2 * (TX + 2) + 2 * TY;
    __shared__ __align__(16) double s_w[3 * PLANE * 2];
    const int unit = (int)(blockIdx.x & 7) * units_per_xcd + (int)(blockIdx.x >> 3);
    if (unit >= nunits) return;
    const int ntx = NX / TX, nty = NY / TY, ntiles = ntx * nty;
    const int tiles_per_xcd = ntiles / 8;
    const int xcd = unit / units_per_xcd, u = unit - xcd * units_per_xcd;
    const int chunk = u / tiles_per_xcd, tl = xcd * tiles_per_xcd + u % tiles_per_xcd;
    const int tx = tl % ntx, ty = tl / ntx;
    const int z0 = chunk * CZ, z1 = min(z0 + CZ, NZ);
    const int t = threadIdx.x, lx = t % TX, ly = t / TX;
    const int64_t tile_base = ((int64_t)ty * ntx + tx) * NZ * (TX * TY);
    const int myrow = (ly + 1) * PY + (lx + 1);
    int rx = 0, ry = 0;
    if (t < TX + 2) { rx = t - 1; ry = -1; }
    else if (t < 2 * (TX + 2)) { rx = t - (TX + 2) - 1; ry = TY; }
    else if (t < 2 * (TX + 2) + TY) { rx = -1; ry = t - 2 * (TX + 2); }
    else if (t < NRING) { rx = TX; ry = t - 2 * (TX + 2) - TY; }
    const bool ring = t < NRING;
    const int rrow = (ry + 1) * PY + (rx + 1);
    const int rgx = min(max(tx * TX + rx, 0), NX - 1), rgy = min(max(ty * TY + ry, 0), NY - 1);
    const int rtile = (rgy / TY) * ntx + rgx / TX;
    const int64_t ring_base = (int64_t)rtile * NZ * (TX * TY) + (rgy % TY) * TX + (rgx % TX);

    real x1[3], x2[3], y1[3] = { 0, 0, 0 }, y2[3] = { 0, 0, 0 };
    double fA[3] = { 0, 0, 0 }, fB[3] = { 0, 0, 0 }, fC[3];
    auto load = [&](int z) {
        const int zc = min(max(z, 0), NZ - 1);
        const int64_t a = tile_base + (int64_t)zc * (TX * TY) + t;
#pragma unroll
        for (int d = 0; d < 3; d++) { x1[d] = u1[3 * a + d]; x2[d] = u2[3 * a + d]; }
        if (ring) {
            const int64_t b = ring_base + (int64_t)zc * (TX * TY);
#pragma unroll
            for (int d = 0; d < 3; d++) { y1[d] = u1[3 * b + d]; y2[d] = u2[3 * b + d]; }
        }
    };
    // image of plane z -> its slot; the node's own term m2 u1 - m1 u2 joins the accumulator of the plane (rs)
    auto put = [&](int z, double rs[3]) {
        double* img = s_w + 3 * PLANE * (z & 1);
#pragma unroll
        for (int d = 0; d < 3; d++) { img[3 * myrow + d] = x1[d] + c.beta * (x1[d] - x2[d]); rs[d] += c.m2 * x1[d] - c.m1 * x2[d]; }
        if (ring) {
#pragma unroll
            for (int d = 0; d < 3; d++) img[3 * rrow + d] = y1[d] + c.beta * (y1[d] - y2[d]);
        }
    };
    // planes z0-1 .. z1 arrive; plane p completes output plane p-1.  fA: accumulator of output p-1, fB: of output p
    // (seeded with the node's own term when its plane was put)
    double dummy[3] = { 0, 0, 0 };
    load(z0 - 1); put(z0 - 1, dummy);
    for (int p = z0 - 1; p <= z1; p++) {
        load(p + 1);
        __syncthreads();
        const double* q = s_w + 3 * (PLANE * (p & 1) + myrow);
        double g[3], m[3], U[3];
        if (ABL == 0) {
            double C[3], XM[3], XP[3], YM[3], YP[3], MM[3], PM[3], MP[3], PP[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                C[d] = q[d]; XM[d] = q[d - 3]; XP[d] = q[d + 3]; YM[d] = q[d - 3 * PY]; YP[d] = q[d + 3 * PY];
                MM[d] = q[d - 3 * PY - 3]; PM[d] = q[d - 3 * PY + 3]; MP[d] = q[d + 3 * PY - 3]; PP[d] = q[d + 3 * PY + 3];
            }
            double sx[3], sy[3], dg[3];
#pragma unroll
            for (int d = 0; d < 3; d++) { sx[d] = XM[d] + XP[d]; sy[d] = YM[d] + YP[d]; dg[d] = (MM[d] + PP[d]) + (PM[d] + MP[d]); }
            const double A_x = (PP[0] + MM[0]) - (PM[0] + MP[0]), A_y = (PP[1] + MM[1]) - (PM[1] + MP[1]);
            const double Bx0_z = XP[2] - XM[2], Bx0_x = XP[0] - XM[0], By0_z = YP[2] - YM[2], By0_y = YP[1] - YM[1];
            const double Bx1_z = (PP[2] - MP[2]) + (PM[2] - MM[2]), Bx1_x = (PP[0] - MP[0]) + (PM[0] - MM[0]);
            const double By1_z = (PP[2] - PM[2]) + (MP[2] - MM[2]), By1_y = (PP[1] - PM[1]) + (MP[1] - MM[1]);
            const double sxy_z = sx[2] + sy[2];
            // dz = 0
            m[0] = fma(c.P[0], C[0], fma(c.P[1], sx[0], fma(c.P[2], sy[0], fma(c.P[3], dg[0], c.Q[0] * A_y))));
            m[1] = fma(c.P[0], C[1], fma(c.P[1], sy[1], fma(c.P[2], sx[1], fma(c.P[3], dg[1], c.Q[0] * A_x))));
            m[2] = fma(c.P[0], C[2], fma(c.P[2], sxy_z, c.P[4] * dg[2]));
            // |dz| = 1, the part even in dz
            g[0] = fma(c.P[2], C[0], fma(c.P[3], sx[0], fma(c.P[4], sy[0], fma(c.P[5], dg[0], c.Q[1] * A_y))));
            g[1] = fma(c.P[2], C[1], fma(c.P[3], sy[1], fma(c.P[4], sx[1], fma(c.P[5], dg[1], c.Q[1] * A_x))));
            g[2] = fma(c.P[1], C[2], fma(c.P[3], sxy_z, c.P[5] * dg[2]));
            // the part odd in dz (sign: that of dz as seen from the output node)
            U[0] = fma(c.Q[0], Bx0_z, c.Q[1] * Bx1_z);
            U[1] = fma(c.Q[0], By0_z, c.Q[1] * By1_z);
            U[2] = fma(c.Q[0], Bx0_x, fma(c.Q[1], Bx1_x, fma(c.Q[0], By0_y, c.Q[1] * By1_y)));
        } else {
#pragma unroll
            for (int d = 0; d < 3; d++) { m[d] = q[d]; g[d] = 0; U[d] = 0; }
        }
        // plane p is at dz = +1 of output p-1, dz = 0 of output p, dz = -1 of output p+1
        if (p - 1 >= z0) {
            const int64_t a = tile_base + (int64_t)(p - 1) * (TX * TY) + t;
#pragma unroll
            for (int d = 0; d < 3; d++) un[3 * a + d] = (fA[d] + (g[d] + U[d])) / c.m0;
        }
#pragma unroll
        for (int d = 0; d < 3; d++) { fA[d] = fB[d] + m[d]; fB[d] = g[d] - U[d]; }
        put(p + 1, fB);                // plane p+1's own term joins ITS accumulator
    }
}

// Round 4: the plane-sum march with the loads TWO planes ahead (two register sets, the loop unrolled by two) and the plane
// sums component by component (9 LDS values live at a time: the registers the second set needs).  All loads
// unconditional (lanes without a ring node re-read their own row), so that the compiler can count them.
template <int TX, int TY, int BYCOMP>
__global__ void __launch_bounds__(TX * TY, 4)
k_march3(const real* __restrict__ u1, const real* __restrict__ u2, real* __restrict__ un, int NX, int NY, int NZ, int CZ,
         int units_per_xcd, int nunits, coef cin)
{
    constexpr int PY = TX + 2, PLANE = (TX + 2) * (TY + 2);
    constexpr int NRING = 2 * (TX + 2) + 2 * TY;
    __shared__ __align__(16) double s_w[3 * PLANE * 2];
    const int unit = (int)(blockIdx.x & 7) * units_per_xcd + (int)(blockIdx.x >> 3);
    if (unit >= nunits) return;
    auto uni = [](double v) {
        const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
        return __hiloint2double(hi, lo);
    };
    double P[6], Q[2];
#pragma unroll
    for (int i = 0; i < 6; i++) P[i] = uni(cin.P[i]);
#pragma unroll
    for (int i = 0; i < 2; i++) Q[i] = uni(cin.Q[i]);
    const double beta = cin.beta, m0i = uni(1.0 / cin.m0), m1 = cin.m1, m2 = cin.m2;
    const int ntx = NX / TX, nty = NY / TY, ntiles = ntx * nty;
    const int tiles_per_xcd = ntiles / 8;
    const int xcd = unit / units_per_xcd, u = unit - xcd * units_per_xcd;
    const int chunk = u / tiles_per_xcd, tl = xcd * tiles_per_xcd + u % tiles_per_xcd;
    const int tx = tl % ntx, ty = tl / ntx;
    const int z0 = chunk * CZ, z1 = min(z0 + CZ, NZ);
    const int t = threadIdx.x, lx = t % TX, ly = t / TX;
    const int64_t tile_base = ((int64_t)ty * ntx + tx) * NZ * (TX * TY);
    const int myrow = (ly + 1) * PY + (lx + 1);
    int rx = lx, ry = ly;
    if (t < TX + 2) { rx = t - 1; ry = -1; }
    else if (t < 2 * (TX + 2)) { rx = t - (TX + 2) - 1; ry = TY; }
    else if (t < 2 * (TX + 2) + TY) { rx = -1; ry = t - 2 * (TX + 2); }
    else if (t < NRING) { rx = TX; ry = t - 2 * (TX + 2) - TY; }
    const bool ring = t < NRING;
    const int rrow = ring ? (ry + 1) * PY + (rx + 1) : myrow;
    const int rgx = min(max(tx * TX + rx, 0), NX - 1), rgy = min(max(ty * TY + ry, 0), NY - 1);
    const int rtile = (rgy / TY) * ntx + rgx / TX;
    const int64_t ring_base = (int64_t)rtile * NZ * (TX * TY) + (rgy % TY) * TX + (rgx % TX);
    constexpr int ZS = TX * TY;

    real a1[3], a2[3], c1[3], c2[3];          // the own node's u1, u2: two sets, loaded two planes ahead
    real b1[3], b2[3];                        // the ring node's: one set, loaded one plane ahead (mostly L2 hits)
    double fA[3] = { 0, 0, 0 }, fB[3] = { 0, 0, 0 };
#define LOADOWN(z_, x1_, x2_)                                                                      \
    {                                                                                              \
        const int zc_ = min(max((z_), 0), NZ - 1);                                                 \
        const int64_t a_ = tile_base + (int64_t)zc_ * ZS + t;                                      \
        _Pragma("unroll") for (int d = 0; d < 3; d++) { x1_[d] = u1[3 * a_ + d]; x2_[d] = u2[3 * a_ + d]; } \
    }
#define LOADRING(z_)                                                                               \
    {                                                                                              \
        const int zc_ = min(max((z_), 0), NZ - 1);                                                 \
        const int64_t b_ = ring_base + (int64_t)zc_ * ZS;                                          \
        _Pragma("unroll") for (int d = 0; d < 3; d++) { b1[d] = u1[3 * b_ + d]; b2[d] = u2[3 * b_ + d]; } \
    }
#define PUTSET(z_, rs_, x1_, x2_)                                                        \
    {                                                                                              \
        double* img_ = s_w + 3 * PLANE * ((z_) & 1);                                               \
        _Pragma("unroll") for (int d = 0; d < 3; d++) {                                            \
            img_[3 * myrow + d] = x1_[d] + beta * (x1_[d] - x2_[d]);                               \
            rs_[d] += m2 * x1_[d] - m1 * x2_[d];                                                   \
        }                                                                                          \
        if (ring) { _Pragma("unroll") for (int d = 0; d < 3; d++) img_[3 * rrow + d] = b1[d] + beta * (b1[d] - b2[d]); } \
    }
#define STEP(p_, x1_, x2_)   /* consume plane p_ from LDS; PUT plane p_+1 from the given set; reload it with p_+3 */ \
    {                                                                                              \
        __syncthreads();                                                                           \
        const double* q = s_w + 3 * (PLANE * ((p_) & 1) + myrow);                                  \
        double m[3], g[3], U[3];                                                                   \
        if (BYCOMP) {                                                                              \
            double A_x, A_y;                                                                       \
            {                                                                                      \
                const double C = q[2], XM = q[2 - 3], XP = q[2 + 3], YM = q[2 - 3 * PY], YP = q[2 + 3 * PY];                      \
                const double MM = q[2 - 3 * PY - 3], PM = q[2 - 3 * PY + 3], MP = q[2 + 3 * PY - 3], PP = q[2 + 3 * PY + 3];      \
                const double sxy = (XM + XP) + (YM + YP), dg = (MM + PP) + (PM + MP);              \
                m[2] = fma(P[0], C, fma(P[2], sxy, P[4] * dg));                                    \
                g[2] = fma(P[1], C, fma(P[3], sxy, P[5] * dg));                                    \
                U[0] = fma(Q[0], XP - XM, Q[1] * ((PP - MP) + (PM - MM)));                         \
                U[1] = fma(Q[0], YP - YM, Q[1] * ((PP - PM) + (MP - MM)));                         \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            {                                                                                      \
                const double C = q[0], XM = q[0 - 3], XP = q[0 + 3], YM = q[0 - 3 * PY], YP = q[0 + 3 * PY];                      \
                const double MM = q[0 - 3 * PY - 3], PM = q[0 - 3 * PY + 3], MP = q[0 + 3 * PY - 3], PP = q[0 + 3 * PY + 3];      \
                const double sx = XM + XP, sy = YM + YP, dg = (MM + PP) + (PM + MP);               \
                A_x = (PP + MM) - (PM + MP);                                                       \
                m[0] = fma(P[0], C, fma(P[1], sx, fma(P[2], sy, P[3] * dg)));                      \
                g[0] = fma(P[2], C, fma(P[3], sx, fma(P[4], sy, P[5] * dg)));                      \
                U[2] = fma(Q[0], XP - XM, Q[1] * ((PP - MP) + (PM - MM)));                         \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            {                                                                                      \
                const double C = q[1], XM = q[1 - 3], XP = q[1 + 3], YM = q[1 - 3 * PY], YP = q[1 + 3 * PY];                      \
                const double MM = q[1 - 3 * PY - 3], PM = q[1 - 3 * PY + 3], MP = q[1 + 3 * PY - 3], PP = q[1 + 3 * PY + 3];      \
                const double sx = XM + XP, sy = YM + YP, dg = (MM + PP) + (PM + MP);               \
                A_y = (PP + MM) - (PM + MP);                                                       \
                m[1] = fma(P[0], C, fma(P[1], sy, fma(P[2], sx, fma(P[3], dg, Q[0] * A_x))));      \
                g[1] = fma(P[2], C, fma(P[3], sy, fma(P[4], sx, fma(P[5], dg, Q[1] * A_x))));      \
                U[2] += fma(Q[0], YP - YM, Q[1] * ((PP - PM) + (MP - MM)));                        \
            }                                                                                      \
            m[0] = fma(Q[0], A_y, m[0]);                                                           \
            g[0] = fma(Q[1], A_y, g[0]);                                                           \
        } else {                                                                                   \
            double C[3], XM[3], XP[3], YM[3], YP[3], MM[3], PM[3], MP[3], PP[3];                   \
            _Pragma("unroll") for (int d = 0; d < 3; d++) {                                        \
                C[d] = q[d]; XM[d] = q[d - 3]; XP[d] = q[d + 3]; YM[d] = q[d - 3 * PY]; YP[d] = q[d + 3 * PY];                   \
                MM[d] = q[d - 3 * PY - 3]; PM[d] = q[d - 3 * PY + 3]; MP[d] = q[d + 3 * PY - 3]; PP[d] = q[d + 3 * PY + 3];      \
            }                                                                                      \
            double sx[3], sy[3], dg[3];                                                            \
            _Pragma("unroll") for (int d = 0; d < 3; d++) { sx[d] = XM[d] + XP[d]; sy[d] = YM[d] + YP[d]; dg[d] = (MM[d] + PP[d]) + (PM[d] + MP[d]); } \
            const double A_x = (PP[0] + MM[0]) - (PM[0] + MP[0]), A_y = (PP[1] + MM[1]) - (PM[1] + MP[1]);                        \
            const double sxy_z = sx[2] + sy[2];                                                    \
            m[0] = fma(P[0], C[0], fma(P[1], sx[0], fma(P[2], sy[0], fma(P[3], dg[0], Q[0] * A_y))));  \
            m[1] = fma(P[0], C[1], fma(P[1], sy[1], fma(P[2], sx[1], fma(P[3], dg[1], Q[0] * A_x))));  \
            m[2] = fma(P[0], C[2], fma(P[2], sxy_z, P[4] * dg[2]));                                \
            g[0] = fma(P[2], C[0], fma(P[3], sx[0], fma(P[4], sy[0], fma(P[5], dg[0], Q[1] * A_y))));  \
            g[1] = fma(P[2], C[1], fma(P[3], sy[1], fma(P[4], sx[1], fma(P[5], dg[1], Q[1] * A_x))));  \
            g[2] = fma(P[1], C[2], fma(P[3], sxy_z, P[5] * dg[2]));                                \
            U[0] = fma(Q[0], XP[2] - XM[2], Q[1] * ((PP[2] - MP[2]) + (PM[2] - MM[2])));           \
            U[1] = fma(Q[0], YP[2] - YM[2], Q[1] * ((PP[2] - PM[2]) + (MP[2] - MM[2])));           \
            U[2] = fma(Q[0], XP[0] - XM[0], fma(Q[1], (PP[0] - MP[0]) + (PM[0] - MM[0]), fma(Q[0], YP[1] - YM[1], Q[1] * ((PP[1] - PM[1]) + (MP[1] - MM[1]))))); \
        }                                                                                          \
        {   /* unconditional (a store behind a branch makes the compiler drain every load in flight): the first,      \
             * incomplete result goes to plane z0 too and is overwritten by the right one two steps later */         \
            const int64_t a_ = tile_base + (int64_t)max((p_) - 1, z0) * ZS + t;                    \
            _Pragma("unroll") for (int d = 0; d < 3; d++) un[3 * a_ + d] = (fA[d] + (g[d] + U[d])) * m0i; \
        }                                                                                          \
        _Pragma("unroll") for (int d = 0; d < 3; d++) { fA[d] = fB[d] + m[d]; fB[d] = g[d] - U[d]; } \
        PUTSET((p_) + 1, fB, x1_, x2_)                                                             \
        LOADRING((p_) + 2)                                                                         \
        LOADOWN((p_) + 3, x1_, x2_)                                                                \
    }
    // planes z0-1 .. z1 are consumed; the own rows of plane p + 1 and p + 2 and the ring rows of plane p + 1 are in flight
    // when plane p is consumed
    double dummy[3] = { 0, 0, 0 };
    LOADOWN(z0 - 1, a1, a2)
    LOADRING(z0 - 1)
    PUTSET(z0 - 1, dummy, a1, a2)
    LOADOWN(z0, c1, c2)
    LOADRING(z0)
    LOADOWN(z0 + 1, a1, a2)
    for (int p = z0 - 1; p <= z1; p += 2) {     // (z1 - z0 + 2 planes: an even number for even chunks)
        STEP(p, c1, c2)                          // consumes plane p, PUTs plane p + 1 (c), requests ring p + 2 and own p + 3 (c)
        STEP(p + 1, a1, a2)
    }
#undef LOADOWN
#undef LOADRING
#undef PUTSET
#undef STEP
}

int main(int argc, char** argv)
{
    const int NX = 512, NY = 512, NZ = argc > 1 ? atoi(argv[1]) : 256;
    const int64_t N = (int64_t)NX * NY * NZ;
    real *u1, *u2, *un, *ur;
    const size_t RB = 3 * sizeof(real);
    CK(hipMalloc(&u1, N * RB)); CK(hipMalloc(&u2, N * RB)); CK(hipMalloc(&un, N * RB)); CK(hipMalloc(&ur, N * RB));
    {
        std::vector<real> h((size_t)N * 3);
        uint64_t s = 88172645463325252ull;
        for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (real)((double)(s >> 11) / 9007199254740992.0 - 0.5); }
        CK(hipMemcpy(u1, h.data(), N * RB, hipMemcpyHostToDevice));
        for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (real)((double)(s >> 11) / 9007199254740992.0 - 0.5); }
        CK(hipMemcpy(u2, h.data(), N * RB, hipMemcpyHostToDevice));
    }
    coef c;
    for (int i = 0; i < 6; i++) c.P[i] = 0.1 * (i + 1) - 0.3;
    c.Q[0] = 0.07; c.Q[1] = -0.05; c.beta = 0.02; c.m0 = 3.0; c.m1 = 0.9; c.m2 = 1.9;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    float ms;
#define RUN(TX_, TY_, SL_, CZ_)                                                                                       \
    {                                                                                                                  \
        const int ntiles = (NX / TX_) * (NY / TY_), nch = (NZ + CZ_ - 1) / CZ_, nunits = ntiles * nch, upx = nunits / 8; \
        k_ref<TX_, TY_><<<(unsigned)((N + 255) / 256), 256>>>(u1, u2, ur, NX, NY, NZ, c);                                \
        CK(hipMemset(un, 0, N * RB));                                                                                  \
        k_march<TX_, TY_, SL_><<<upx * 8, TX_ * TY_>>>(u1, u2, un, NX, NY, NZ, CZ_, upx, nunits, c);                     \
        CK(hipDeviceSynchronize());                                                                                    \
        {                                                                                                              \
            std::vector<real> a((size_t)1 << 22), b((size_t)1 << 22);                                                \
            double worst = 0;                                                                                          \
            for (int64_t off : { (int64_t)0, N * 3 / 2, N * 3 - ((int64_t)1 << 22) }) {                                \
                CK(hipMemcpy(a.data(), un + off, a.size() * sizeof(real), hipMemcpyDeviceToHost));                                \
                CK(hipMemcpy(b.data(), ur + off, b.size() * sizeof(real), hipMemcpyDeviceToHost));                                \
                for (size_t i = 0; i < a.size(); i++) worst = fmax(worst, fabs((double)a[i] - (double)b[i]));                          \
            }                                                                                                          \
            printf("tile %2dx%-2d slots %d chunk %3d  max|march - ref| = %.2e  ", TX_, TY_, SL_, CZ_, worst);          \
        }                                                                                                              \
        CK(hipEventRecord(e0));                                                                                        \
        for (int r = 0; r < reps; r++) k_march<TX_, TY_, SL_><<<upx * 8, TX_ * TY_>>>(u1, u2, un, NX, NY, NZ, CZ_, upx, nunits, c); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));                     \
        printf("%7.3f ms  (%.0f GB/s of the compulsory bytes; 67.6M-node box: %.3f ms)\n", ms / reps,              \
               3.0 * RB * N / (ms / reps * 1e-3) / 1e9, ms / reps * 67634433.0 / N);                                       \
    }
#define RUN2(TX_, TY_, CZ_, ABL_)                                                                                    \
    {                                                                                                                  \
        const int ntiles = (NX / TX_) * (NY / TY_), nch = (NZ + CZ_ - 1) / CZ_, nunits = ntiles * nch, upx = nunits / 8; \
        k_ref<TX_, TY_><<<(unsigned)((N + 255) / 256), 256>>>(u1, u2, ur, NX, NY, NZ, c);                                \
        CK(hipMemset(un, 0, N * RB));                                                                                  \
        k_march2<TX_, TY_, ABL_><<<upx * 8, TX_ * TY_>>>(u1, u2, un, NX, NY, NZ, CZ_, upx, nunits, c);                   \
        CK(hipDeviceSynchronize());                                                                                    \
        {                                                                                                              \
            std::vector<real> a((size_t)1 << 22), b((size_t)1 << 22);                                                \
            double worst = 0;                                                                                          \
            for (int64_t off : { (int64_t)0, N * 3 / 2, N * 3 - ((int64_t)1 << 22) }) {                                \
                CK(hipMemcpy(a.data(), un + off, a.size() * sizeof(real), hipMemcpyDeviceToHost));                                \
                CK(hipMemcpy(b.data(), ur + off, b.size() * sizeof(real), hipMemcpyDeviceToHost));                                \
                for (size_t i = 0; i < a.size(); i++) worst = fmax(worst, fabs((double)a[i] - (double)b[i]));                          \
            }                                                                                                          \
            printf("plane sums: tile %2dx%-2d chunk %3d abl %d  max|march - ref| = %.2e  ", TX_, TY_, CZ_, ABL_, worst); \
        }                                                                                                              \
        CK(hipEventRecord(e0));                                                                                        \
        for (int r = 0; r < reps; r++) k_march2<TX_, TY_, ABL_><<<upx * 8, TX_ * TY_>>>(u1, u2, un, NX, NY, NZ, CZ_, upx, nunits, c); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));                     \
        printf("%7.3f ms  (%.0f GB/s of the compulsory bytes; 67.6M-node box: %.3f ms)\n", ms / reps,              \
               3.0 * RB * N / (ms / reps * 1e-3) / 1e9, ms / reps * 67634433.0 / N);                                       \
    }
#define RUN3(TX_, TY_, CZ_, ABL_)                                                                                    \
    {                                                                                                                  \
        const int ntiles = (NX / TX_) * (NY / TY_), nch = (NZ + CZ_ - 1) / CZ_, nunits = ntiles * nch, upx = nunits / 8; \
        k_ref<TX_, TY_><<<(unsigned)((N + 255) / 256), 256>>>(u1, u2, ur, NX, NY, NZ, c);                                \
        CK(hipMemset(un, 0, N * RB));                                                                                  \
        k_march3<TX_, TY_, ABL_><<<upx * 8, TX_ * TY_>>>(u1, u2, un, NX, NY, NZ, CZ_, upx, nunits, c);                   \
        CK(hipDeviceSynchronize());                                                                                    \
        {                                                                                                              \
            std::vector<real> a((size_t)1 << 22), b((size_t)1 << 22);                                                \
            double worst = 0;                                                                                          \
            for (int64_t off : { (int64_t)0, N * 3 / 2, N * 3 - ((int64_t)1 << 22) }) {                                \
                CK(hipMemcpy(a.data(), un + off, a.size() * sizeof(real), hipMemcpyDeviceToHost));                                \
                CK(hipMemcpy(b.data(), ur + off, b.size() * sizeof(real), hipMemcpyDeviceToHost));                                \
                for (size_t i = 0; i < a.size(); i++) worst = fmax(worst, fabs((double)a[i] - (double)b[i]));                          \
            }                                                                                                          \
            printf("plane sums, loads two planes ahead (bycomp = abl): tile %2dx%-2d chunk %3d abl %d  max|march - ref| = %.2e  ", TX_, TY_, CZ_, ABL_, worst); \
        }                                                                                                              \
        CK(hipEventRecord(e0));                                                                                        \
        for (int r = 0; r < reps; r++) k_march3<TX_, TY_, ABL_><<<upx * 8, TX_ * TY_>>>(u1, u2, un, NX, NY, NZ, CZ_, upx, nunits, c); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));                     \
        printf("%7.3f ms  (%.0f GB/s of the compulsory bytes; 67.6M-node box: %.3f ms)\n", ms / reps,              \
               3.0 * RB * N / (ms / reps * 1e-3) / 1e9, ms / reps * 67634433.0 / N);                                       \
    }
    RUN3(64, 8, 32, 1)
    RUN3(64, 8, 64, 1)
    RUN3(64, 8, 32, 0)
    RUN3(64, 8, 64, 0)
    RUN2(64, 8, 32, 0)
    RUN2(64, 8, 64, 0)
    return 0;
    RUN2(64, 8, 32, 0)
    RUN2(64, 8, 64, 0)
    RUN2(64, 8, 256, 0)
    RUN2(64, 8, 32, 1)
    RUN2(32, 16, 32, 0)
    RUN2(32, 16, 64, 0)
    RUN2(32, 8, 32, 0)
    RUN2(32, 8, 64, 0)
    RUN2(64, 4, 32, 0)
    RUN2(64, 4, 64, 0)
    RUN2(64, 4, 64, 1)
    RUN2(128, 4, 64, 0)
    RUN2(128, 8, 64, 0)
    RUN(64, 8, 4, 32)
    RUN(64, 8, 4, 64)
    RUN(64, 8, 3, 32)
    RUN(64, 8, 3, 64)
    RUN(32, 16, 4, 32)
    RUN(32, 16, 3, 32)
    RUN(32, 16, 3, 64)
    RUN(32, 8, 4, 32)
    RUN(32, 8, 3, 64)
    RUN(64, 4, 4, 32)
    RUN(64, 4, 3, 64)
    return 0;
}
