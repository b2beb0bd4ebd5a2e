// Micro-benchmark (round 4): TWO time steps per pass of the z-marching plane-sum stencil (temporal blocking).
//   hipcc -O3 --offload-arch=gfx950 -o march_twostep march_twostep.hip && ./march_twostep
// One step per pass moves 72 B per node and step whatever the kernel does (read u(t), u(t-dt), write u(t+dt)); a pass
// that carries two steps reads u(t), u(t-dt) once and writes u(t+dt), u(t+2dt): 96 B per node and PAIR of steps, plus a
// second ring.  A workgroup's TX x TY threads are the stage-1 region (step 1 is computed on it, redundantly on its
// border), its inner (TX-2) x (TY-2) nodes the output tile (step 2); w(t) of the region and one more ring sits in LDS
// (2 slots), w(t+dt) of the region in a second pair of slots; stage 2 marches one plane behind stage 1 (two barriers per
// plane).  Layout: tile-major with the OUTPUT tiles ([tile][z][y][x]).
// Compared with two passes of the one-step kernel (march_stencil.hip: 0.98 ms per step on the 67M-node box).
// Checked against the plain gather kernel applied twice, at the nodes >= 3 away from the domain faces (the redundant
// border computation at a clamped face is not the reference's clamping; the engine takes those nodes from the shell).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct coef { double P[6], Q[2], beta, m0, m1, m2; };

template <int OX, int OY>
__device__ __host__ inline int64_t node_addr(int gx, int gy, int gz, int NX, int NY, int NZ)
{
    const int ntx = NX / OX;
    const int tx = gx / OX, ty = gy / OY;
    const int64_t tile = (int64_t)ty * ntx + tx;
    return (tile * NZ + gz) * (OX * OY) + (gy % OY) * OX + (gx % OX);
}

template <int OX, int OY>
__global__ void k_ref(const double* __restrict__ u1, const double* __restrict__ u2, double* __restrict__ un, int NX, int NY, int NZ, coef c)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)NX * NY * NZ) return;
    const int gx = i % NX, gy = (i / NX) % NY, gz = i / ((int64_t)NX * NY);
    double f[3] = { 0, 0, 0 };
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int x = min(max(gx + dx, 0), NX - 1), y = min(max(gy + dy, 0), NY - 1), z = min(max(gz + dz, 0), NZ - 1);
                const int64_t a = node_addr<OX, OY>(x, y, z, NX, NY, NZ);
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
    const int64_t a = node_addr<OX, OY>(gx, gy, gz, NX, NY, NZ);
    for (int d = 0; d < 3; d++) un[3 * a + d] = (f[d] + c.m2 * u1[3 * a + d] - c.m1 * u2[3 * a + d]) / c.m0;
}

// the plane sums of one plane as seen from a node: q = its row in an LDS plane image of pitch PY
template <int PY>
__device__ __forceinline__ void plane_sums(const double* __restrict__ q, const double* P, const double* Q, double m[3], double g[3], double U[3])
{
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
    m[0] = fma(P[0], C[0], fma(P[1], sx[0], fma(P[2], sy[0], fma(P[3], dg[0], Q[0] * A_y))));
    m[1] = fma(P[0], C[1], fma(P[1], sy[1], fma(P[2], sx[1], fma(P[3], dg[1], Q[0] * A_x))));
    m[2] = fma(P[0], C[2], fma(P[2], sxy_z, P[4] * dg[2]));
    g[0] = fma(P[2], C[0], fma(P[3], sx[0], fma(P[4], sy[0], fma(P[5], dg[0], Q[1] * A_y))));
    g[1] = fma(P[2], C[1], fma(P[3], sy[1], fma(P[4], sx[1], fma(P[5], dg[1], Q[1] * A_x))));
    g[2] = fma(P[1], C[2], fma(P[3], sxy_z, P[5] * dg[2]));
    U[0] = fma(Q[0], Bx0_z, Q[1] * Bx1_z);
    U[1] = fma(Q[0], By0_z, Q[1] * By1_z);
    U[2] = fma(Q[0], Bx0_x, fma(Q[1], Bx1_x, fma(Q[0], By0_y, Q[1] * By1_y)));
}

static __device__ __forceinline__ double uni(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// A = u(t), B = u(t - dt) -> C = u(t + dt), D = u(t + 2 dt).  TX x TY threads = stage-1 region; outputs (TX-2) x (TY-2).
template <int TX, int TY, int MINW, int ABL>
__global__ void __launch_bounds__(TX * TY, MINW)
k_ts(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, double* __restrict__ D, int NX, int NY,
     int NZ, int CZ, int units_per_xcd, int nunits, coef cin)
{
    constexpr int OX = TX - 2, OY = TY - 2;
    constexpr int P1Y = TX + 2, PLANE1 = (TX + 2) * (TY + 2);       // w(t): region + one ring
    constexpr int P2Y = TX, PLANE2 = TX * TY;                       // w(t + dt): region
    constexpr int NR2 = 2 * (TX + 2) + 2 * TY;
    __shared__ __align__(16) double s_w1[3 * PLANE1 * 2];
    __shared__ __align__(16) double s_w2[3 * PLANE2 * 2];
    const int unit = (int)(blockIdx.x & 7) * units_per_xcd + (int)(blockIdx.x >> 3);
    if (unit >= nunits) return;
    double P[6], Q[2];
#pragma unroll
    for (int i = 0; i < 6; i++) P[i] = uni(cin.P[i]);
#pragma unroll
    for (int i = 0; i < 2; i++) Q[i] = uni(cin.Q[i]);
    const double beta = cin.beta, m0i = uni(1.0 / cin.m0), m1 = cin.m1, m2 = cin.m2;
    const int ntx = NX / OX, nty = NY / OY, ntiles = ntx * nty;
    const int tiles_per_xcd = ntiles / 8;
    const int xcd = unit / units_per_xcd, u = unit - xcd * units_per_xcd;
    const int chunk = u / tiles_per_xcd, tl = xcd * tiles_per_xcd + u % tiles_per_xcd;
    const int tx = tl % ntx, ty = tl / ntx;
    const int z0 = chunk * CZ, z1 = min(z0 + CZ, NZ);
    const int t = threadIdx.x, lx = t % TX, ly = t / TX;
    const bool inner = lx >= 1 && lx <= OX && ly >= 1 && ly <= OY;
    // own node of the region (clamped at the domain faces)
    const int gx = min(max(tx * OX + lx - 1, 0), NX - 1), gy = min(max(ty * OY + ly - 1, 0), NY - 1);
    const int64_t own_base = node_addr<OX, OY>(gx, gy, 0, NX, NY, NZ);
    const int row1 = (ly + 1) * P1Y + (lx + 1), row2 = ly * P2Y + lx;
    // ring around the region: one node per thread t < NR2
    int rx = 0, ry = 0;
    if (t < TX + 2) { rx = t - 1; ry = -1; }
    else if (t < 2 * (TX + 2)) { rx = t - (TX + 2) - 1; ry = TY; }
    else if (t < 2 * (TX + 2) + TY) { rx = -1; ry = t - 2 * (TX + 2); }
    else if (t < NR2) { rx = TX; ry = t - 2 * (TX + 2) - TY; }
    const bool ring = t < NR2;
    const int rrow = (ry + 1) * P1Y + (rx + 1);
    const int rgx = min(max(tx * OX + rx - 1, 0), NX - 1), rgy = min(max(ty * OY + ry - 1, 0), NY - 1);
    const int64_t ring_base = node_addr<OX, OY>(rgx, rgy, 0, NX, NY, NZ);
    constexpr int ZS = OX * OY;

    double x1[3], x2[3], y1[3] = { 0, 0, 0 }, y2[3] = { 0, 0, 0 };
    double aP[3] = { 0, 0, 0 }, aQ[3] = { 0, 0, 0 };                 // u(t) of the own node at the planes p - 1, p
    double fA[3] = { 0, 0, 0 }, fB[3] = { 0, 0, 0 }, gA[3] = { 0, 0, 0 }, gB[3] = { 0, 0, 0 };
    auto load = [&](int z) {
        const int zc = min(max(z, 0), NZ - 1);
        const int64_t a = own_base + (int64_t)zc * ZS;
#pragma unroll
        for (int d = 0; d < 3; d++) { x1[d] = A[3 * a + d]; x2[d] = B[3 * a + d]; }
        if (ring) {
            const int64_t b = ring_base + (int64_t)zc * ZS;
#pragma unroll
            for (int d = 0; d < 3; d++) { y1[d] = A[3 * b + d]; y2[d] = B[3 * b + d]; }
        }
    };
    auto put = [&](int z, double rs[3]) {
        double* img = s_w1 + 3 * PLANE1 * (z & 1);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            img[3 * row1 + d] = x1[d] + beta * (x1[d] - x2[d]);
            rs[d] += m2 * x1[d] - m1 * x2[d];
            aP[d] = aQ[d]; aQ[d] = x1[d];
        }
        if (ring) {
#pragma unroll
            for (int d = 0; d < 3; d++) img[3 * rrow + d] = y1[d] + beta * (y1[d] - y2[d]);
        }
    };
    // planes z0 - 2 .. z1 + 1 arrive.  Stage 1: plane p completes u(t + dt) of plane p - 1; stage 2: w(t + dt) of plane
    // p - 1 completes u(t + 2 dt) of plane p - 2.
    double dummy[3] = { 0, 0, 0 };
    load(z0 - 2); put(z0 - 2, dummy);
    for (int p = z0 - 2; p <= z1 + 1; p++) {
        if (p <= z1) load(p + 1);
        __syncthreads();
        double m[3], g[3], U[3];
        if (ABL == 0) plane_sums<P1Y>(s_w1 + 3 * (PLANE1 * (p & 1) + row1), P, Q, m, g, U);
        else { const double* q = s_w1 + 3 * (PLANE1 * (p & 1) + row1); for (int d = 0; d < 3; d++) { m[d] = q[d]; g[d] = 0; U[d] = 0; } }
        // u(t + dt) of plane p - 1 (every thread of the region), kept as w(t + dt) for stage 2
        double un1[3];
#pragma unroll
        for (int d = 0; d < 3; d++) un1[d] = (fA[d] + (g[d] + U[d])) * m0i;
        if (inner && p - 1 >= z0 && p - 1 < z1) {
            const int64_t a = own_base + (int64_t)(p - 1) * ZS;
#pragma unroll
            for (int d = 0; d < 3; d++) C[3 * a + d] = un1[d];
        }
        {
            double* img2 = s_w2 + 3 * PLANE2 * ((p - 1) & 1);
#pragma unroll
            for (int d = 0; d < 3; d++) {
                img2[3 * row2 + d] = un1[d] + beta * (un1[d] - aP[d]);
                gB[d] += m2 * un1[d] - m1 * aP[d];                  // own term of output plane p - 1 of stage 2
            }
        }
#pragma unroll
        for (int d = 0; d < 3; d++) { fA[d] = fB[d] + m[d]; fB[d] = g[d] - U[d]; }
        __syncthreads();
        if (inner) {
            double m_[3], g_[3], U_[3];
            if (ABL == 0) plane_sums<P2Y>(s_w2 + 3 * (PLANE2 * ((p - 1) & 1) + row2), P, Q, m_, g_, U_);
            else { const double* q = s_w2 + 3 * (PLANE2 * ((p - 1) & 1) + row2); for (int d = 0; d < 3; d++) { m_[d] = q[d]; g_[d] = 0; U_[d] = 0; } }
            if (p - 2 >= z0 && p - 2 < z1) {
                const int64_t a = own_base + (int64_t)(p - 2) * ZS;
#pragma unroll
                for (int d = 0; d < 3; d++) D[3 * a + d] = (gA[d] + (g_[d] + U_[d])) * m0i;
            }
#pragma unroll
            for (int d = 0; d < 3; d++) { gA[d] = gB[d] + m_[d]; gB[d] = g_[d] - U_[d]; }
        }
        if (p <= z1) put(p + 1, fB);
    }
}

// Variant 2: the two stages on DIFFERENT waves of one workgroup (wave specialisation).  Waves 0 .. TY-1 are the stage-1
// region as above (loads, w(t) images, u(t + dt), store C) and hand w(t + dt) and the own term m2 u(t+dt) - m1 u(t) of
// every region node to LDS; waves TY .. 2 TY - 3 (one per inner row) march stage 2 one plane behind from those images
// (store D).  ONE barrier per plane; each role's registers hold only its own accumulators.
template <int TX, int TY, int ABL>
__global__ void __launch_bounds__(TX * TY + TX * (TY - 2), 4)
k_ts2(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, double* __restrict__ D, int NX, int NY,
      int NZ, int CZ, int units_per_xcd, int nunits, coef cin)
{
    constexpr int OX = TX - 2, OY = TY - 2;
    constexpr int P1Y = TX + 2, PLANE1 = (TX + 2) * (TY + 2);
    constexpr int P2Y = TX, PLANE2 = TX * TY;
    constexpr int NR2 = 2 * (TX + 2) + 2 * TY;
    constexpr int NS1 = TX * TY;
    __shared__ __align__(16) double s_w1[3 * PLANE1 * 2];
    __shared__ __align__(16) double s_w2[6 * PLANE2 * 2];           // per slot: w(t + dt) [3 PLANE2] | own term [3 PLANE2]
    const int unit = (int)(blockIdx.x & 7) * units_per_xcd + (int)(blockIdx.x >> 3);
    if (unit >= nunits) return;
    double P[6], Q[2];
#pragma unroll
    for (int i = 0; i < 6; i++) P[i] = uni(cin.P[i]);
#pragma unroll
    for (int i = 0; i < 2; i++) Q[i] = uni(cin.Q[i]);
    const double beta = cin.beta, m0i = uni(1.0 / cin.m0), m1 = cin.m1, m2 = cin.m2;
    const int ntx = NX / OX, nty = NY / OY, ntiles = ntx * nty;
    const int tiles_per_xcd = ntiles / 8;
    const int xcd = unit / units_per_xcd, u = unit - xcd * units_per_xcd;
    const int chunk = u / tiles_per_xcd, tl = xcd * tiles_per_xcd + u % tiles_per_xcd;
    const int tx = tl % ntx, ty = tl / ntx;
    const int z0 = chunk * CZ, z1 = min(z0 + CZ, NZ);
    const int t = threadIdx.x;
    const bool s1 = t < NS1;
    const int tt = s1 ? t : t - NS1;
    const int lx = tt % TX, ly = s1 ? tt / TX : 1 + tt / TX;
    const bool inner = lx >= 1 && lx <= OX && ly >= 1 && ly <= OY;
    const int gx = min(max(tx * OX + lx - 1, 0), NX - 1), gy = min(max(ty * OY + ly - 1, 0), NY - 1);
    const int64_t own_base = node_addr<OX, OY>(gx, gy, 0, NX, NY, NZ);
    const int row1 = (ly + 1) * P1Y + (lx + 1), row2 = ly * P2Y + lx;
    constexpr int ZS = OX * OY;
    if (s1) {
        int rx = 0, ry = 0;
        if (t < TX + 2) { rx = t - 1; ry = -1; }
        else if (t < 2 * (TX + 2)) { rx = t - (TX + 2) - 1; ry = TY; }
        else if (t < 2 * (TX + 2) + TY) { rx = -1; ry = t - 2 * (TX + 2); }
        else if (t < NR2) { rx = TX; ry = t - 2 * (TX + 2) - TY; }
        const bool ring = t < NR2;
        const int rrow = (ry + 1) * P1Y + (rx + 1);
        const int rgx = min(max(tx * OX + rx - 1, 0), NX - 1), rgy = min(max(ty * OY + ry - 1, 0), NY - 1);
        const int64_t ring_base = node_addr<OX, OY>(rgx, rgy, 0, NX, NY, NZ);
        double x1[3], x2[3], y1[3] = { 0, 0, 0 }, y2[3] = { 0, 0, 0 };
        double aP[3] = { 0, 0, 0 }, aQ[3] = { 0, 0, 0 };
        double fA[3] = { 0, 0, 0 }, fB[3] = { 0, 0, 0 };
        auto load = [&](int z) {
            const int zc = min(max(z, 0), NZ - 1);
            const int64_t a = own_base + (int64_t)zc * ZS;
#pragma unroll
            for (int d = 0; d < 3; d++) { x1[d] = A[3 * a + d]; x2[d] = B[3 * a + d]; }
            if (ring) {
                const int64_t b = ring_base + (int64_t)zc * ZS;
#pragma unroll
                for (int d = 0; d < 3; d++) { y1[d] = A[3 * b + d]; y2[d] = B[3 * b + d]; }
            }
        };
        auto put = [&](int z, double rs[3]) {
            double* img = s_w1 + 3 * PLANE1 * (z & 1);
#pragma unroll
            for (int d = 0; d < 3; d++) {
                img[3 * row1 + d] = x1[d] + beta * (x1[d] - x2[d]);
                rs[d] += m2 * x1[d] - m1 * x2[d];
                aP[d] = aQ[d]; aQ[d] = x1[d];
            }
            if (ring) {
#pragma unroll
                for (int d = 0; d < 3; d++) img[3 * rrow + d] = y1[d] + beta * (y1[d] - y2[d]);
            }
        };
        double dummy[3] = { 0, 0, 0 };
        load(z0 - 2); put(z0 - 2, dummy);
        for (int p = z0 - 2; p <= z1 + 2; p++) {
            if (p <= z1) load(p + 1);
            __syncthreads();
            if (p <= z1 + 1) {
                double m[3], g[3], U[3];
                if (ABL == 0) plane_sums<P1Y>(s_w1 + 3 * (PLANE1 * (p & 1) + row1), P, Q, m, g, U);
                else { const double* q = s_w1 + 3 * (PLANE1 * (p & 1) + row1); for (int d = 0; d < 3; d++) { m[d] = q[d]; g[d] = 0; U[d] = 0; } }
                double un1[3];
#pragma unroll
                for (int d = 0; d < 3; d++) un1[d] = (fA[d] + (g[d] + U[d])) * m0i;
                if (inner && p - 1 >= z0 && p - 1 < z1) {
                    const int64_t a = own_base + (int64_t)(p - 1) * ZS;
#pragma unroll
                    for (int d = 0; d < 3; d++) C[3 * a + d] = un1[d];
                }
                double* img2 = s_w2 + 6 * PLANE2 * ((p - 1) & 1);
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    img2[3 * row2 + d] = un1[d] + beta * (un1[d] - aP[d]);
                    img2[3 * PLANE2 + 3 * row2 + d] = m2 * un1[d] - m1 * aP[d];
                }
#pragma unroll
                for (int d = 0; d < 3; d++) { fA[d] = fB[d] + m[d]; fB[d] = g[d] - U[d]; }
                if (p <= z1) put(p + 1, fB);
            }
        }
    } else {
        double gA[3] = { 0, 0, 0 }, gB[3] = { 0, 0, 0 };
        for (int p = z0 - 2; p <= z1 + 2; p++) {
            __syncthreads();
            const int r = p - 2;                                     // the plane of w(t + dt) stage 1 wrote in the last iteration
            if (inner && r >= z0 - 1) {
                const double* img2 = s_w2 + 6 * PLANE2 * (r & 1);
                double m_[3], g_[3], U_[3];
                if (ABL == 0) plane_sums<P2Y>(img2 + 3 * row2, P, Q, m_, g_, U_);
                else { for (int d = 0; d < 3; d++) { m_[d] = img2[3 * row2 + d]; g_[d] = 0; U_[d] = 0; } }
#pragma unroll
                for (int d = 0; d < 3; d++) gB[d] += img2[3 * PLANE2 + 3 * row2 + d];
                if (r - 1 >= z0 && r - 1 < z1) {
                    const int64_t a = own_base + (int64_t)(r - 1) * ZS;
#pragma unroll
                    for (int d = 0; d < 3; d++) D[3 * a + d] = (gA[d] + (g_[d] + U_[d])) * m0i;
                }
#pragma unroll
                for (int d = 0; d < 3; d++) { gA[d] = gB[d] + m_[d]; gB[d] = g_[d] - U_[d]; }
            }
        }
    }
}

int main(int argc, char** argv)
{
    const int NZ = argc > 1 ? atoi(argv[1]) : 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    coef c;
    for (int i = 0; i < 6; i++) c.P[i] = 0.01 * (i + 1) - 0.03;
    c.Q[0] = 0.007; c.Q[1] = -0.005; c.beta = 0.02; c.m0 = 3.0; c.m1 = 0.9; c.m2 = 1.9;
    const int reps = 10;
    float ms;
#define RUNTS(TX_, TY_, MINW_, CZ_, ABL_, NTX_, NTY_)                                                                  \
    {                                                                                                                  \
        constexpr int OX = TX_ - 2, OY = TY_ - 2;                                                                      \
        const int NX = OX * NTX_, NY = OY * NTY_;                                                                      \
        const int64_t N = (int64_t)NX * NY * NZ;                                                                       \
        double *A, *B, *C, *D, *R1, *R2;                                                                               \
        CK(hipMalloc(&A, N * 24)); CK(hipMalloc(&B, N * 24)); CK(hipMalloc(&C, N * 24)); CK(hipMalloc(&D, N * 24));    \
        CK(hipMalloc(&R1, N * 24)); CK(hipMalloc(&R2, N * 24));                                                        \
        {                                                                                                              \
            std::vector<double> h((size_t)N * 3);                                                                      \
            uint64_t s = 88172645463325252ull;                                                                         \
            for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) / 9007199254740992.0 - 0.5; } \
            CK(hipMemcpy(A, h.data(), N * 24, hipMemcpyHostToDevice));                                                 \
            for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) / 9007199254740992.0 - 0.5; } \
            CK(hipMemcpy(B, h.data(), N * 24, hipMemcpyHostToDevice));                                                 \
        }                                                                                                              \
        const int ntiles = NTX_ * NTY_, nch = (NZ + CZ_ - 1) / CZ_, nunits = ntiles * nch, upx = (nunits + 7) / 8;     \
        k_ref<OX, OY><<<(unsigned)((N + 255) / 256), 256>>>(A, B, R1, NX, NY, NZ, c);                                   \
        k_ref<OX, OY><<<(unsigned)((N + 255) / 256), 256>>>(R1, A, R2, NX, NY, NZ, c);                                  \
        CK(hipMemset(C, 0, N * 24)); CK(hipMemset(D, 0, N * 24));                                                      \
        k_ts<TX_, TY_, MINW_, ABL_><<<upx * 8, TX_ * TY_>>>(A, B, C, D, NX, NY, NZ, CZ_, upx, nunits, c);                \
        CK(hipDeviceSynchronize());                                                                                    \
        {                                                                                                              \
            std::vector<double> hc((size_t)N * 3), hd((size_t)N * 3), r1((size_t)N * 3), r2((size_t)N * 3);            \
            CK(hipMemcpy(hc.data(), C, N * 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(hd.data(), D, N * 24, hipMemcpyDeviceToHost)); \
            CK(hipMemcpy(r1.data(), R1, N * 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), R2, N * 24, hipMemcpyDeviceToHost)); \
            double w1 = 0, w2 = 0, sc = 0;                                                                             \
            for (int z = 3; z < NZ - 3; z += 7)                                                                        \
                for (int y = 3; y < NY - 3; y++)                                                                       \
                    for (int x = 3; x < NX - 3; x++) {                                                                 \
                        const int64_t a = node_addr<OX, OY>(x, y, z, NX, NY, NZ);                                      \
                        for (int d = 0; d < 3; d++) {                                                                  \
                            w1 = fmax(w1, fabs(hc[3 * a + d] - r1[3 * a + d])); w2 = fmax(w2, fabs(hd[3 * a + d] - r2[3 * a + d])); \
                            sc = fmax(sc, fabs(r2[3 * a + d]));                                                        \
                        }                                                                                              \
                    }                                                                                                  \
            printf("two-step: region %2dx%-2d (outputs %dx%d, %d waves/SIMD min) chunk %3d abl %d  grid %dx%dx%d  max err step1 %.1e step2 %.1e (scale %.1e)  ", \
                   TX_, TY_, OX, OY, MINW_, CZ_, ABL_, NX, NY, NZ, w1, w2, sc);                                        \
        }                                                                                                              \
        CK(hipEventRecord(e0));                                                                                        \
        for (int r = 0; r < reps; r++) k_ts<TX_, TY_, MINW_, ABL_><<<upx * 8, TX_ * TY_>>>(A, B, C, D, NX, NY, NZ, CZ_, upx, nunits, c); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));                     \
        printf("%7.3f ms per PASS of two steps = %.3f ms per step on the 67.6M-node box (one-step kernel: 0.98)\n", ms / reps, \
               0.5 * ms / reps * 67634433.0 / N);                                                                      \
        hipFree(A); hipFree(B); hipFree(C); hipFree(D); hipFree(R1); hipFree(R2);                                      \
    }
#define RUNTS2(TX_, TY_, MINW_, CZ_, ABL_, NTX_, NTY_)                                                                  \
    {                                                                                                                  \
        constexpr int OX = TX_ - 2, OY = TY_ - 2;                                                                      \
        const int NX = OX * NTX_, NY = OY * NTY_;                                                                      \
        const int64_t N = (int64_t)NX * NY * NZ;                                                                       \
        double *A, *B, *C, *D, *R1, *R2;                                                                               \
        CK(hipMalloc(&A, N * 24)); CK(hipMalloc(&B, N * 24)); CK(hipMalloc(&C, N * 24)); CK(hipMalloc(&D, N * 24));    \
        CK(hipMalloc(&R1, N * 24)); CK(hipMalloc(&R2, N * 24));                                                        \
        {                                                                                                              \
            std::vector<double> h((size_t)N * 3);                                                                      \
            uint64_t s = 88172645463325252ull;                                                                         \
            for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) / 9007199254740992.0 - 0.5; } \
            CK(hipMemcpy(A, h.data(), N * 24, hipMemcpyHostToDevice));                                                 \
            for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) / 9007199254740992.0 - 0.5; } \
            CK(hipMemcpy(B, h.data(), N * 24, hipMemcpyHostToDevice));                                                 \
        }                                                                                                              \
        const int ntiles = NTX_ * NTY_, nch = (NZ + CZ_ - 1) / CZ_, nunits = ntiles * nch, upx = (nunits + 7) / 8;     \
        k_ref<OX, OY><<<(unsigned)((N + 255) / 256), 256>>>(A, B, R1, NX, NY, NZ, c);                                   \
        k_ref<OX, OY><<<(unsigned)((N + 255) / 256), 256>>>(R1, A, R2, NX, NY, NZ, c);                                  \
        CK(hipMemset(C, 0, N * 24)); CK(hipMemset(D, 0, N * 24));                                                      \
        k_ts2<TX_, TY_, ABL_><<<upx * 8, TX_ * TY_ + TX_ * (TY_ - 2)>>>(A, B, C, D, NX, NY, NZ, CZ_, upx, nunits, c);                \
        CK(hipDeviceSynchronize());                                                                                    \
        {                                                                                                              \
            std::vector<double> hc((size_t)N * 3), hd((size_t)N * 3), r1((size_t)N * 3), r2((size_t)N * 3);            \
            CK(hipMemcpy(hc.data(), C, N * 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(hd.data(), D, N * 24, hipMemcpyDeviceToHost)); \
            CK(hipMemcpy(r1.data(), R1, N * 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), R2, N * 24, hipMemcpyDeviceToHost)); \
            double w1 = 0, w2 = 0, sc = 0;                                                                             \
            for (int z = 3; z < NZ - 3; z += 7)                                                                        \
                for (int y = 3; y < NY - 3; y++)                                                                       \
                    for (int x = 3; x < NX - 3; x++) {                                                                 \
                        const int64_t a = node_addr<OX, OY>(x, y, z, NX, NY, NZ);                                      \
                        for (int d = 0; d < 3; d++) {                                                                  \
                            w1 = fmax(w1, fabs(hc[3 * a + d] - r1[3 * a + d])); w2 = fmax(w2, fabs(hd[3 * a + d] - r2[3 * a + d])); \
                            sc = fmax(sc, fabs(r2[3 * a + d]));                                                        \
                        }                                                                                              \
                    }                                                                                                  \
            printf("two-step, stages on different waves: region %2dx%-2d (outputs %dx%d, %d waves/SIMD min) chunk %3d abl %d  grid %dx%dx%d  max err step1 %.1e step2 %.1e (scale %.1e)  ", \
                   TX_, TY_, OX, OY, MINW_, CZ_, ABL_, NX, NY, NZ, w1, w2, sc);                                        \
        }                                                                                                              \
        CK(hipEventRecord(e0));                                                                                        \
        for (int r = 0; r < reps; r++) k_ts2<TX_, TY_, ABL_><<<upx * 8, TX_ * TY_ + TX_ * (TY_ - 2)>>>(A, B, C, D, NX, NY, NZ, CZ_, upx, nunits, c); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));                     \
        printf("%7.3f ms per PASS of two steps = %.3f ms per step on the 67.6M-node box (one-step kernel: 0.98)\n", ms / reps, \
               0.5 * ms / reps * 67634433.0 / N);                                                                      \
        hipFree(A); hipFree(B); hipFree(C); hipFree(D); hipFree(R1); hipFree(R2);                                      \
    }
    // grids of about 512 x 512 nodes: 62 x 8 = 496, 6 x 86 = 516; 10 x 51 = 510; 14 x 36 = 504
    RUNTS2(64, 8, 4, 32, 0, 8, 86)
    RUNTS2(64, 8, 4, 64, 0, 8, 86)
    RUNTS2(64, 8, 4, 128, 0, 8, 86)
    RUNTS2(64, 8, 4, 64, 1, 8, 86)
    RUNTS(64, 8, 2, 64, 0, 8, 86)
    RUNTS(64, 12, 3, 64, 0, 8, 51)
    RUNTS(64, 8, 2, 64, 1, 8, 86)
    RUNTS(64, 12, 3, 64, 1, 8, 51)
    return 0;
}
