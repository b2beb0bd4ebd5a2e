/*
 * hq_kernels.h -- device-side arithmetic shared by the gfx950 kernels.
 *
 * The element force of Hercules' explicit step, fused:
 *
 *     f_e = -(c1 K1 + c2 K2) (u1 + beta (u1 - u2)),      beta = c3/c1 = c4/c2
 *
 * i.e. compute_addforce_effective (stiffness.c:180-237) and damping_addforce
 * (damping.c:29-103) in ONE "effective" product, legal because
 * c3 = (b/dt) c1 and c4 = (b/dt) c2 (psolve.c:3387-3409).  The product is
 * evaluated through the reference's factorisation K = A D A^T
 * (aTransposeU / firstVector / au, stiffness.c:245-424) but organised as a
 * 3-stage radix-2 butterfly over the node-sign bits (A is the 8-point Walsh
 * transform of the trilinear hexahedron), 72 + 72 adds instead of 147 + 168.
 *
 * Mode index m = bit0:x bit1:y bit2:z.  Reference row r <-> m:
 * r1(z)=4 r2(y)=2 r3(x)=1 r4(yz)=6 r5(xz)=5 r6(xy)=3 r7(xyz)=7, r0 (rigid)=0.
 */
#ifndef HQ_KERNELS_H
#define HQ_KERNELS_H

#ifdef HQ_KERNEL_MATH_HOST_CHECK   /* tests/test_kernel_math_cpu.py: g++ compiles the arithmetic alone */
#include <cmath>
#define __device__
#define __host__
#define __forceinline__ inline
#else
#include <hip/hip_runtime.h>
#endif

/* forward butterfly: v[n] (n = node, bit d set = far side of axis d) ->
 * v[m] = sum_n prod_{d in m} sgn_d(n) v[n],  sgn = +1 on the far side. */
__host__ __device__ __forceinline__ void hq_wht_fwd(double v[8])
{
#pragma unroll
    for (int s = 1; s < 8; s <<= 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (!(i & s)) {
                double a = v[i], b = v[i | s];
                v[i] = a + b;
                v[i | s] = b - a;
            }
        }
    }
}

/* transposed butterfly: f[n] = sum_m prod_{d in m} sgn_d(n) g[m] */
__host__ __device__ __forceinline__ void hq_wht_inv(double v[8])
{
#pragma unroll
    for (int s = 1; s < 8; s <<= 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (!(i & s)) {
                double lo = v[i], hi = v[i | s];
                v[i] = lo - hi;
                v[i | s] = lo + hi;
            }
        }
    }
}

/* the transposed butterfly without its last (z) stage: v[0..3] = a, v[4..7] = b with f[n] = a[n] - b[n] on the near
 * z side and f[n + 4] = a[n] + b[n] on the far one -- for callers that sum over elements first (hq_k_brick_het: the
 * x / y reductions act on a and b, 6 values less to expand per element) */
__host__ __device__ __forceinline__ void hq_wht_inv_xy(double v[8])
{
#pragma unroll
    for (int s = 1; s < 4; s <<= 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (!(i & s)) {
                double lo = v[i], hi = v[i | s];
                v[i] = lo - hi;
                v[i | s] = lo + hi;
            }
        }
    }
}

/*
 * In: X,Y,Z = w[n][0..2] per node.  Out: X,Y,Z = f_e[n][0..2] (ZMODES: the z stage of the transposed butterfly is
 * left to the caller, see hq_wht_inv_xy).
 * c1, c2 as in e_t (psolve.h:196-198).
 */
template <bool ZMODES = false>
__host__ __device__ __forceinline__ void hq_element_force(double X[8], double Y[8], double Z[8],
                                                 double c1, double c2)
{
    /* stiffness.c:216-218 */
    const double a = -0.5625 * (c2 + 2.0 * c1);
    const double c = -0.5625 * c2;
    const double b = -0.5625 * c1;
    const double third = 1.0 / 3.0;
    const double b3 = b * third, c3 = c * third, ab3 = (a + b) * third;
    const double a2b9 = (a + 2.0 * b) * (1.0 / 9.0);

    hq_wht_fwd(X);
    hq_wht_fwd(Y);
    hq_wht_fwd(Z);

    /* D: firstVector (stiffness.c:291-319) in mode indices */
    const double sxy = b * (Y[1] + X[2]);                    /* shear xy */
    const double sxz = b * (Z[1] + X[4]);                    /* shear xz */
    const double syz = b * (Z[2] + Y[4]);                    /* shear yz */
    const double nx = a * X[1] + c * (Y[2] + Z[4]);          /* normal   */
    const double ny = a * Y[2] + c * (X[1] + Z[4]);
    const double nz = a * Z[4] + c * (X[1] + Y[2]);
    const double t = X[6] + Y[5] + Z[3];                     /* twist    */
    const double gx6 = b3 * (t + X[6]);
    const double gy5 = b3 * (t + Y[5]);
    const double gz3 = b3 * (t + Z[3]);
    const double gx5 = ab3 * X[5] + c3 * Y[6];
    const double gx3 = ab3 * X[3] + c3 * Z[6];
    const double gy6 = ab3 * Y[6] + c3 * X[5];
    const double gy3 = ab3 * Y[3] + c3 * Z[5];
    const double gz6 = ab3 * Z[6] + c3 * X[3];
    const double gz5 = ab3 * Z[5] + c3 * Y[3];

    X[0] = 0.0; X[1] = nx;  X[2] = sxy; X[3] = gx3; X[4] = sxz; X[5] = gx5; X[6] = gx6; X[7] = a2b9 * X[7];
    Y[0] = 0.0; Y[1] = sxy; Y[2] = ny;  Y[3] = gy3; Y[4] = syz; Y[5] = gy5; Y[6] = gy6; Y[7] = a2b9 * Y[7];
    Z[0] = 0.0; Z[1] = sxz; Z[2] = syz; Z[3] = gz3; Z[4] = nz;  Z[5] = gz5; Z[6] = gz6; Z[7] = a2b9 * Z[7];

    if (ZMODES) { hq_wht_inv_xy(X); hq_wht_inv_xy(Y); hq_wht_inv_xy(Z); }
    else { hq_wht_inv(X); hq_wht_inv(Y); hq_wht_inv(Z); }
}

/*
 * (c1, c2, beta) of an element from the three floats solver_init derived them from -- 12 bytes instead of 24, exactly:
 * mu_and_lambda (psolve.c:3236-3272) evaluates mu = rho Vs Vs in SINGLE precision (float operands), lambda as a
 * double difference, and solver_init (psolve.c:3387-3409) builds
 *     c1 = dt^2 h mu / 9,   c2 = dt^2 h lambda / 9,   c3 = b dt h mu / 9,   b = zeta bBase,   zeta = min(10 / Vs, threshold)
 * (10 / Vs again a float division), beta = c3 / c1 (hq_create).  Every operation below is the reference's, in its order
 * and its precision, as ONE IEEE operation each (no contraction: intrinsics on the device, an x86-64 host has no fused
 * form without -mfma); the divisions by 9 are Markstein's q = t r, q += fma(-9, q, t) r with r = RN(1/9).  hq_create
 * runs this very function on the host for every element of a unit and packs the unit only if the caller's eTable comes
 * out bit for bit; anything else keeps its 24 bytes.
 *   rho < 0: |rho| is the density and lambda = rho Vp Vp (the Poisson-ratio fix, psolve.c:3253-3263, whose Vp the caller's
 *   edata already holds); rho == 0: no element (all three come out 0).
 */
struct hq_mat_const {
    double A;                    /* dt^2 * h (psolve.c:3387: theDeltaTSquared * edgesize) */
    double h, dt;                /* edgesize (the float's value), theDeltaT                */
    double bbase, thr_damp, thr_vpvs;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define HQ_FMUL(a, b) __fmul_rn((a), (b))
#define HQ_DMUL(a, b) __dmul_rn((a), (b))
#define HQ_DSUB(a, b) __dsub_rn((a), (b))
#define HQ_FDIV(a, b) __fdiv_rn((a), (b))
#define HQ_DDIV(a, b) __ddiv_rn((a), (b))
#else
#define HQ_FMUL(a, b) ((float)((float)(a) * (float)(b)))
#define HQ_DMUL(a, b) ((double)(a) * (double)(b))
#define HQ_DSUB(a, b) ((double)(a) - (double)(b))
#define HQ_FDIV(a, b) ((float)((float)(a) / (float)(b)))
#define HQ_DDIV(a, b) ((double)(a) / (double)(b))
#endif

__host__ __device__ __forceinline__ double hq_div9(double t)
{
    const double r9 = 0.1111111111111111;            /* RN(1/9) */
    const double q = HQ_DMUL(t, r9);
    const double rem = fma(-9.0, q, t);              /* exact */
    return fma(rem, r9, q);
}

__host__ __device__ __forceinline__ void hq_material_coef(float rho_s, float Vs, float Vp, const hq_mat_const& K,
                                                          double* c1, double* c2, double* beta)
{
    const bool fixed = rho_s < 0.0f;
    const float rho = fixed ? -rho_s : rho_s;
    const double mu = (double)HQ_FMUL(HQ_FMUL(rho, Vs), Vs);
    const double P = (double)HQ_FMUL(HQ_FMUL(rho, Vp), Vp);
    const double two_mu = HQ_DMUL(2.0, mu);
    const double capped = HQ_DSUB(HQ_DMUL(HQ_DMUL(mu, K.thr_vpvs), K.thr_vpvs), two_mu);
    const double plain = HQ_DSUB(P, two_mu);
    const double lambda = fixed ? P : (((double)Vp > HQ_DMUL((double)Vs, K.thr_vpvs)) ? capped : plain);
    const double k1 = hq_div9(HQ_DMUL(K.A, mu));
    const double k2 = hq_div9(HQ_DMUL(K.A, lambda));
    double zeta = (double)HQ_FDIV(10.0f, Vs);
    if (zeta > K.thr_damp) zeta = K.thr_damp;
    const double b = HQ_DMUL(zeta, K.bbase);
    const double k3 = hq_div9(HQ_DMUL(HQ_DMUL(HQ_DMUL(b, K.dt), K.h), mu));
    const bool none = rho == 0.0f;
    *c1 = none ? 0.0 : k1;
    *c2 = none ? 0.0 : k2;
    *beta = (none || k1 == 0.0) ? 0.0 : HQ_DDIV(k3, k1);
}

#endif /* HQ_KERNELS_H */
