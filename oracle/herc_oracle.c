/*
 * herc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C99, fp64) of the explicit time-stepping hot path of
 * CMU-Quake/hercules `quake/forward`, written from the reference's behaviour
 * for use as the parity checker of the MI355X implementation and as the
 * "port" CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (hercules_amd/) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file
 * against (i) the station traces the reference ships in
 * examples/simple/expected-out/stations, (ii) full-field checkpoints and
 * station traces produced by the real reference binary (oracle/build_ref.sh ->
 * oracle/_ref/psolve) with tests/golden/make_golden.py, for the effective and
 * conventional stiffness methods, Rayleigh / mass / no damping, 1 and 8 ranks.
 *
 * Every function cites the reference lines (relative to /root/reference) it
 * follows.  Where the reference evaluates an expression in single precision
 * before widening (edata_t fields are float, psolve.h:95-97) the same casts are
 * kept so that eTable / nTable are reproduced to the last bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HO_API __attribute__((visibility("default")))

/* solver_float, psolve.h:60-64: the type of tm1 / tm2 / force (fvector_t, :102-104), of the n_t rows (:210-214) and of the
 * locals the reference declares with them -- float when the reference is built with -DSINGLE_PRECISION_SOLVER
 * (oracle/_ref/psolve_f32; this file then compiles to libherc_oracle_f32.so).  Everything the reference declares double
 * (e_t, K1 / K2, atu / firstVec, dashpot, the source table) stays double: the usual arithmetic conversions then round
 * where the reference's statements round. */
#ifdef SINGLE_PRECISION_SOLVER
typedef float ho_real;
#else
typedef double ho_real;
#endif
HO_API int32_t ho_real_bytes(void) { return (int32_t)sizeof(ho_real); }

/* quake_util.c:36 */
#define HO_UNDERFLOW_CAP 1e-20

enum { HO_DAMP_NONE = 0, HO_DAMP_RAYLEIGH = 1, HO_DAMP_MASS = 2 };
enum { HO_STIFF_EFFECTIVE = 0, HO_STIFF_CONVENTIONAL = 1 };

/* ------------------------------------------------------------------------ */
/* Z-order helpers                                                          */
/* ------------------------------------------------------------------------ */

/* spread the low 21 bits of v so that bit b lands on bit 3b */
static uint64_t ho_spread3(uint64_t v)
{
    v &= 0x1fffffULL;
    v = (v | (v << 32)) & 0x1f00000000ffffULL;
    v = (v | (v << 16)) & 0x1f0000ff0000ffULL;
    v = (v | (v << 8))  & 0x100f00f00f00f00fULL;
    v = (v | (v << 4))  & 0x10c30c30c30c30c3ULL;
    v = (v | (v << 2))  & 0x1249249249249249ULL;
    return v;
}

/* Z-value with x the least and z the most significant axis inside a bit
 * triplet: the child numbering k*4+j*2+i of octor.c:6444-6470 and the
 * tie-breaking order of octor_zcompare (octor.c:3034-3150: z, then y, then x). */
HO_API uint64_t ho_zvalue(uint32_t x, uint32_t y, uint32_t z)
{
    return ho_spread3(x) | (ho_spread3(y) << 1) | (ho_spread3(z) << 2);
}

typedef struct { uint64_t key; int32_t a, b, c; } ho_keyed_t;

static int ho_keyed_cmp(const void* p, const void* q)
{
    uint64_t a = ((const ho_keyed_t*)p)->key, b = ((const ho_keyed_t*)q)->key;
    return (a > b) - (a < b);
}

/*
 * Node sort key.  octor sorts the harbored nodes with octor_zcompare after
 * moving far-boundary nodes one tick inwards (farbound = farendp - 1,
 * octor.c:4142-4144, 6100-6106, 6166).  On a uniform mesh whose edge is 2^m
 * ticks the m low bits of every coordinate are all 0 (regular node) or all 1
 * (adjusted far node), so one extra low bit carries the same ordering: node
 * index i maps to 2i, the far node n to 2n-1.
 */
static uint32_t ho_node_coord_key(int32_t i, int32_t n)
{
    return (i == n) ? (uint32_t)(2 * n - 1) : (uint32_t)(2 * i);
}

/* ------------------------------------------------------------------------ */
/* Uniform box in octor order                                               */
/* ------------------------------------------------------------------------ */

/*
 * Restates what octor_extractmesh (octor.c:5268-6650) yields for a domain
 * refined uniformly to nx*ny*nz leaf octants: elements in octree pre-order
 * (= Z-order of their lower-left corner), nodes in Z-order of the adjusted
 * coordinates, lnid[c] with c = kk*4 + jj*2 + ii (x fastest).
 *
 * elem_ijk  [E][3]  lower-left corner of each element in element units
 * lnid      [E][8]
 * node_ijk  [N][3]  node coordinates in element units
 * returns 0, or -1 on allocation failure.
 */
HO_API int ho_uniform_mesh(int32_t nx, int32_t ny, int32_t nz,
                           int32_t* elem_ijk, int32_t* lnid, int32_t* node_ijk)
{
    int64_t E = (int64_t)nx * ny * nz;
    int64_t N = (int64_t)(nx + 1) * (ny + 1) * (nz + 1);
    ho_keyed_t* ek = (ho_keyed_t*)malloc(sizeof(ho_keyed_t) * (size_t)E);
    ho_keyed_t* nk = (ho_keyed_t*)malloc(sizeof(ho_keyed_t) * (size_t)N);
    int32_t* rank  = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    if (!ek || !nk || !rank) { free(ek); free(nk); free(rank); return -1; }

    int64_t t = 0;
    for (int32_t k = 0; k < nz; k++)
        for (int32_t j = 0; j < ny; j++)
            for (int32_t i = 0; i < nx; i++, t++) {
                ek[t].key = ho_zvalue((uint32_t)i, (uint32_t)j, (uint32_t)k);
                ek[t].a = i; ek[t].b = j; ek[t].c = k;
            }
    qsort(ek, (size_t)E, sizeof(ho_keyed_t), ho_keyed_cmp);

    t = 0;
    for (int32_t k = 0; k <= nz; k++)
        for (int32_t j = 0; j <= ny; j++)
            for (int32_t i = 0; i <= nx; i++, t++) {
                nk[t].key = ho_zvalue(ho_node_coord_key(i, nx), ho_node_coord_key(j, ny),
                                      ho_node_coord_key(k, nz));
                nk[t].a = i; nk[t].b = j; nk[t].c = k;
            }
    qsort(nk, (size_t)N, sizeof(ho_keyed_t), ho_keyed_cmp);

    for (int64_t n = 0; n < N; n++) {
        node_ijk[3 * n + 0] = nk[n].a;
        node_ijk[3 * n + 1] = nk[n].b;
        node_ijk[3 * n + 2] = nk[n].c;
        rank[((int64_t)nk[n].c * (ny + 1) + nk[n].b) * (nx + 1) + nk[n].a] = (int32_t)n;
    }
    for (int64_t e = 0; e < E; e++) {
        int32_t i = ek[e].a, j = ek[e].b, k = ek[e].c;
        elem_ijk[3 * e + 0] = i; elem_ijk[3 * e + 1] = j; elem_ijk[3 * e + 2] = k;
        for (int c = 0; c < 8; c++) {
            int32_t ii = i + (c & 1), jj = j + ((c >> 1) & 1), kk = k + ((c >> 2) & 1);
            lnid[8 * e + c] = rank[((int64_t)kk * (ny + 1) + jj) * (nx + 1) + ii];
        }
    }
    free(ek); free(nk); free(rank);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Reference-cube matrices                                                  */
/* ------------------------------------------------------------------------ */

/* sign of local node n along axis d: psolve.c:5451-5453 */
static double ho_sgn(int d, int n) { return ((n >> d) & 1) ? 1.0 : -1.0; }

/* closed-form integrals of trilinear shape-function gradient products on the
 * reference cube: psolve.c:2574-2578 */
static double ho_int_same(double xki, double xkj, double xli, double xlj, double xmi, double xmj)
{
    return 4.5 * xki * xkj * (1 + xli * xlj / 3) * (1 + xmi * xmj / 3) / 8;
}
static double ho_int_cross(double xki, double xlj, double xmi, double xmj)
{
    return 4.5 * xki * xlj * (1 + xmi * xmj / 3) / 8;
}

/*
 * K1 (with the diagonal "K3" merged in) and K2 as 8x8 blocks of 3x3:
 * compute_K, psolve.c:5446-5573.  Layout K[i][j][k][l] flattened row-major.
 */
HO_API void ho_compute_K(double* K1, double* K2)
{
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++)
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 3; l++) {
                    double* k1 = &K1[((i * 8 + j) * 3 + k) * 3 + l];
                    double* k2 = &K2[((i * 8 + j) * 3 + k) * 3 + l];
                    if (k == l) {
                        int k0 = k, k1a = (k + 1) % 3, k2a = (k + 2) % 3;
                        /* psolve.c:5506-5517 */
                        *k1 = ho_int_same(ho_sgn(k0, i), ho_sgn(k0, j), ho_sgn(k1a, i), ho_sgn(k1a, j),
                                          ho_sgn(k2a, i), ho_sgn(k2a, j));
                        *k2 = ho_int_same(ho_sgn(k0, j), ho_sgn(k0, i), ho_sgn(k1a, j), ho_sgn(k1a, i),
                                          ho_sgn(k2a, j), ho_sgn(k2a, i));
                        /* K3 diagonal, psolve.c:5466-5487, merged :5558-5570 */
                        double I1 = ho_int_same(ho_sgn(k0, i), ho_sgn(k0, j), ho_sgn(k1a, i), ho_sgn(k1a, j),
                                                ho_sgn(k2a, i), ho_sgn(k2a, j));
                        double I2 = ho_int_same(ho_sgn(k1a, i), ho_sgn(k1a, j), ho_sgn(k2a, i), ho_sgn(k2a, j),
                                                ho_sgn(k0, i), ho_sgn(k0, j));
                        double I3 = ho_int_same(ho_sgn(k2a, i), ho_sgn(k2a, j), ho_sgn(k0, i), ho_sgn(k0, j),
                                                ho_sgn(k1a, i), ho_sgn(k1a, j));
                        *k1 += I1 + I2 + I3;
                    } else {
                        int m = 3 - (k + l);
                        /* psolve.c:5522-5530 */
                        *k1 = ho_int_cross(ho_sgn(k, j), ho_sgn(l, i), ho_sgn(m, j), ho_sgn(m, i));
                        *k2 = ho_int_cross(ho_sgn(k, i), ho_sgn(l, j), ho_sgn(m, i), ho_sgn(m, j));
                    }
                }
}

/* ------------------------------------------------------------------------ */
/* solver_init: element and node constants                                  */
/* ------------------------------------------------------------------------ */

/* compute_setab, psolve.c:5813-5876 */
HO_API void ho_setab(double freq, int damping, double* aBase, double* bBase)
{
    const double PI = 3.14159265358979323846;
    *aBase = 0; *bBase = 0;
    if (damping == HO_DAMP_RAYLEIGH) {
        double w1 = 2 * PI * freq * .2, w2 = 2 * PI * freq * 1;
        double lw1 = log(w1), lw2 = log(w2);
        double sw1 = w1 * w1, sw2 = w2 * w2;
        double cw1 = w1 * w1 * w1, cw2 = w2 * w2 * w2;
        double numer = w1 * w2 *
            (-2 * sw1 * lw2 + 2 * sw1 * lw1 - 2 * w1 * w2 * lw2
             + 2 * w1 * w2 * lw1 + 3 * sw2 - 3 * sw1
             - 2 * sw2 * lw2 + 2 * sw2 * lw1);
        double denom = (cw1 - cw2 + 3 * sw2 * w1 - 3 * sw1 * w2);
        *aBase = numer / denom;
        numer = 3 * (2 * w1 * w2 * lw2 - 2 * w1 * w2 * lw1 + sw1 - sw2);
        *bBase = numer / denom;
    } else if (damping == HO_DAMP_MASS) {
        double w1 = 2 * PI * freq * .1, w2 = 2 * PI * freq * 8;
        double numer = 2 * w2 * w1 * log(w2 / w1);
        double denom = w2 - w1;
        *aBase = 1.3 * numer / denom;
        *bBase = 0;
    }
}

/*
 * Boundary class of an element from the six "touches domain face" bits:
 * compute_setflag, psolve.c:5629-5714 (later tests override earlier ones).
 * face bit 0/1/2 = x/y/z near end, bit 3/4/5 = x/y/z far end.
 */
static int ho_setflag(int face)
{
    int lx = face & 1, ly = (face >> 1) & 1, lz = (face >> 2) & 1;
    int ux = (face >> 3) & 1, uy = (face >> 4) & 1, uz = (face >> 5) & 1;
    int flag = 13;
    if (lx) flag = 12;
    if (ly) flag = 10;
    if (lz) flag = 4;
    if (ux) flag = 14;
    if (uy) flag = 16;
    if (uz) flag = 22;
    if (lx && ly) flag = 9;
    if (ux && ly) flag = 11;
    if (lx && uy) flag = 15;
    if (ux && uy) flag = 17;
    if (lx && lz) flag = 3;
    if (ux && lz) flag = 5;
    if (lx && uz) flag = 21;
    if (ux && uz) flag = 23;
    if (ly && lz) flag = 1;
    if (uy && lz) flag = 7;
    if (ly && uz) flag = 19;
    if (uy && uz) flag = 25;
    if (lx && ly && lz) flag = 0;
    if (ux && ly && lz) flag = 2;
    if (lx && uy && lz) flag = 6;
    if (ux && uy && lz) flag = 8;
    if (lx && ly && uz) flag = 18;
    if (ux && ly && uz) flag = 20;
    if (lx && uy && uz) flag = 24;
    if (ux && uy && uz) flag = 26;
    return flag;
}

/*
 * Which absorbing faces contribute to local node n of an element of boundary
 * class `flag`, as a bitmask (bit d = a face normal to axis d).  This is the
 * content of theIDBoundaryMatrix (psolve.c:5718-5746) derived from its
 * geometry: class = 9*cz + 3*cy + cx with c in {0 near, 1 interior, 2 far};
 * node n lies on the near (far) face of axis d when its sign bit is 0 (1).
 */
static int ho_boundary_bits(int flag, int n)
{
    int bits = 0;
    int c[3] = { flag % 3, (flag / 3) % 3, flag / 9 };
    for (int d = 0; d < 3; d++) {
        int far = (n >> d) & 1;
        if ((c[d] == 0 && !far) || (c[d] == 2 && far)) bits |= 1 << d;
    }
    return bits;
}

/* Lysmer dashpots: compute_setboundary, psolve.c:5752-5804 */
static void ho_setboundary(float size, float Vp, float Vs, float rho, int flag, int halfspace,
                           double dashpot[8][3])
{
    memset(dashpot, 0, sizeof(double) * 24);
    if (halfspace && flag < 9) flag += 9;              /* free surface at z = 0, :5762-5764 */
    double scale = rho * (size / 2) * (size / 2);      /* float product, :5766 */
    for (int n = 0; n < 8; n++) {
        int bits = ho_boundary_bits(flag, n);
        int nfaces = (bits & 1) + ((bits >> 1) & 1) + ((bits >> 2) & 1);
        for (int d = 0; d < 3; d++) {
            if (nfaces == 3)
                dashpot[n][d] = (Vp + 2 * Vs) * scale;
            else if (nfaces == 2)
                dashpot[n][d] = (Vs + ((bits & (1 << d)) ? Vp : Vs)) * scale;
            else if (nfaces == 1)
                dashpot[n][d] = ((bits & (1 << d)) ? Vp : Vs) * scale;
        }
    }
}

/*
 * eTable / nTable: solver_init, psolve.c:3360-3473, with mu_and_lambda
 * (:3236-3272) inlined.
 *
 * edata   [E][4] float : edgesize, Vp, Vs, rho  (Vp may be rewritten, :3252-3261)
 * face    [E]          : six face bits (see ho_setflag); ignored if !boundary
 * etable  [E][4]       : c1..c4          (e_t, psolve.h:196-198)
 * ntable  [N][7]       : mass_simple, mass2_minusaM[3], mass_minusaM[3] (n_t, psolve.h:210-214)
 * returns 0, or the 1-based index of the first element with negative lambda.
 */
HO_API int64_t ho_solver_init(int64_t E, int64_t N, const int32_t* lnid, float* edata,
                              const uint8_t* face, double dt, double freq, int damping,
                              double thr_damping, double thr_vpvs, int boundary, int halfspace,
                              double* etable, ho_real* ntable)
{
    double aBase, bBase;
    ho_setab(freq, damping, &aBase, &bBase);
    double dt2 = dt * dt;                               /* psolve.c:998 */
    memset(ntable, 0, sizeof(ho_real) * 7 * (size_t)N);

    for (int64_t e = 0; e < E; e++) {
        float* ed = &edata[4 * e];
        float h = ed[0], Vs = ed[2], rho = ed[3];
        double mu, lambda;

        mu = rho * Vs * Vs;                             /* float product, :3242 */
        if (ed[1] > (Vs * thr_vpvs))
            lambda = rho * Vs * Vs * thr_vpvs * thr_vpvs - 2 * mu;
        else
            lambda = rho * ed[1] * ed[1] - 2 * mu;
        if (lambda < 0) {
            if (Vs < 500)       ed[1] = 2.45 * Vs;
            else if (Vs < 1200) ed[1] = 2 * Vs;
            else                ed[1] = 1.87 * Vs;
            lambda = rho * ed[1] * ed[1];
        }
        if (lambda < 0) return e + 1;
        float Vp = ed[1];

        double* ep = &etable[4 * e];
        ep[0] = dt2 * h * mu / 9;                       /* :3387-3388 */
        ep[1] = dt2 * h * lambda / 9;

        double zeta = 10 / Vs;                          /* float division, :3397 */
        if (zeta > thr_damping) zeta = thr_damping;
        double a = zeta * aBase, b = zeta * bBase;
        ep[2] = b * dt * h * mu / 9;                    /* :3408-3409 */
        ep[3] = b * dt * h * lambda / 9;

        double dashpot[8][3];
        int flag = 13;
        if (boundary) {
            flag = ho_setflag(face[e]);
            if (flag != 13) ho_setboundary(h, Vp, Vs, rho, flag, halfspace, dashpot);
        }

        double mass = rho * h * h * h;                  /* float product, :3436 */
        double M = mass / 8;
        for (int j = 0; j < 8; j++) {
            ho_real* np = &ntable[7 * (int64_t)lnid[8 * e + j]];
            np[0] += M;
            for (int ax = 0; ax < 3; ax++) {            /* :3452-3469 */
                np[4 + ax] -= (dt * a * M);
                np[1 + ax] -= (dt * a * M);
                if (boundary && flag != 13) {
                    np[4 + ax] -= (dt * dashpot[j][ax]);
                    np[1 + ax] -= (dt * dashpot[j][ax]);
                }
                np[4 + ax] += M;
                np[1 + ax] += (M * 2);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Element force kernels                                                    */
/* ------------------------------------------------------------------------ */

/* vector_is_zero, quake_util.c:49-68: 1 if any of the 24 entries is "non-zero" */
static int ho_any_nonzero24(const ho_real* v)
{
    for (int i = 0; i < 24; i++)
        if (fabs(v[i]) > HO_UNDERFLOW_CAP) return 1;
    return 0;
}

/* Mode-sign table of the effective method: row r of the 8x8 +-1 transform used
 * by aTransposeU / au (stiffness.c:260-288, 388-413).  Rows 1..7 are the
 * products of the nodal sign coordinates (z, y, x, yz, xz, xy, xyz); row 0 is
 * the rigid mode (all +1; aTransposeU forces its result to 0). */
static double ho_mode_sign(int r, int n)
{
    double xs = ho_sgn(0, n), ys = ho_sgn(1, n), zs = ho_sgn(2, n);
    switch (r) {
    case 0: return 1.0;
    case 1: return zs;
    case 2: return ys;
    case 3: return xs;
    case 4: return ys * zs;
    case 5: return xs * zs;
    case 6: return xs * ys;
    default: return xs * ys * zs;
    }
}

static double HO_SIGN[8][8];
static int ho_sign_ready = 0;
static void ho_sign_init(void)
{
    if (ho_sign_ready) return;
    for (int r = 0; r < 8; r++)
        for (int n = 0; n < 8; n++) HO_SIGN[r][n] = ho_mode_sign(r, n);
    ho_sign_ready = 1;
}

/*
 * f_e = A D(a,c,b) A^T u for one element, u and f node-major [8][3]:
 * aTransposeU + firstVector + au, stiffness.c:245-424.  Sums run in the
 * reference's left-to-right order so results are bit-identical.
 */
static void ho_effective_elem(const double* u, double a, double c, double b, ho_real* f)
{
    double atu[3][8], fv[3][8];

    for (int d = 0; d < 3; d++) {
        atu[d][0] = 0;
        for (int r = 1; r < 8; r++) {
            double s = HO_SIGN[r][0] * u[d];
            for (int n = 1; n < 8; n++) s += HO_SIGN[r][n] * u[3 * n + d];
            atu[d][r] = s;
        }
    }
    const double* X = atu[0]; const double* Y = atu[1]; const double* Z = atu[2];

    /* firstVector, stiffness.c:291-319 */
    fv[0][0] = 0;
    fv[0][1] = b * (Z[3] + X[1]);
    fv[0][2] = b * (Y[3] + X[2]);
    fv[0][3] = a * X[3] + c * (Y[2] + Z[1]);
    fv[0][4] = b * (Y[5] + Z[6] + 2. * X[4]) / 3.;
    fv[0][5] = ((a + b) * X[5] + c * Y[4]) / 3.;
    fv[0][6] = ((a + b) * X[6] + c * Z[4]) / 3.;
    fv[0][7] = ((a + 2. * b) * X[7]) / 9.;

    fv[1][0] = 0;
    fv[1][1] = b * (Z[2] + Y[1]);
    fv[1][2] = a * Y[2] + c * (X[3] + Z[1]);
    fv[1][3] = b * (Y[3] + X[2]);
    fv[1][4] = ((a + b) * Y[4] + c * X[5]) / 3.;
    fv[1][5] = b * (X[4] + Z[6] + 2. * Y[5]) / 3.;
    fv[1][6] = ((a + b) * Y[6] + c * Z[5]) / 3.;
    fv[1][7] = (a + 2. * b) * Y[7] / 9.;

    fv[2][0] = 0;
    fv[2][1] = a * Z[1] + c * (X[3] + Y[2]);
    fv[2][2] = b * (Z[2] + Y[1]);
    fv[2][3] = b * (Z[3] + X[1]);
    fv[2][4] = ((a + b) * Z[4] + c * X[6]) / 3.;
    fv[2][5] = ((a + b) * Z[5] + c * Y[6]) / 3.;
    fv[2][6] = b * (X[4] + Y[5] + 2. * Z[6]) / 3.;
    fv[2][7] = (a + 2. * b) * Z[7] / 9.;

    /* au, stiffness.c:381-424: node n gets sum_r sign[r][n] * fv[r] */
    for (int d = 0; d < 3; d++)
        for (int n = 0; n < 8; n++) {
            double s = fv[d][0];
            for (int r = 1; r < 8; r++) s += HO_SIGN[r][n] * fv[d][r];
            f[3 * n + d] += s;
        }
}

/* lf += c * (M v), MultAddMatVec, quake_util.c:107-122 */
static void ho_mult_add(const double* M, const ho_real* v, double c, ho_real* out)
{
    ho_real t[3] = { 0, 0, 0 };                      /* fvector_t tmpV */
    for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) t[r] += M[3 * r + q] * v[q];
    for (int r = 0; r < 3; r++) out[r] += c * t[r];
}

/* compute_addforce_effective, stiffness.c:180-237 */
HO_API void ho_addforce_effective(int64_t E, const int32_t* lnid, const double* etable,
                                  const ho_real* tm1, ho_real* force, int zero_skip)
{
    ho_sign_init();
    for (int64_t e = 0; e < E; e++) {
        const int32_t* id = &lnid[8 * e];
        ho_real u[24], lf[24];                           /* fvector_t curDisp[8], localForce[8] */
        memset(lf, 0, sizeof lf);
        for (int i = 0; i < 8; i++)
            for (int d = 0; d < 3; d++) u[3 * i + d] = tm1[3 * (int64_t)id[i] + d];
        if (!zero_skip || ho_any_nonzero24(u)) {
            double c1 = etable[4 * e], c2 = etable[4 * e + 1];
            double a = -0.5625 * (c2 + 2 * c1);
            double c = -0.5625 * (c2);
            double b = -0.5625 * (c1);
            double ud[24];                               /* aTransposeU's double temp[24], stiffness.c:247-255 */
            for (int i = 0; i < 24; i++) ud[i] = u[i];
            ho_effective_elem(ud, a, c, b, lf);
        }
        for (int i = 0; i < 8; i++)
            for (int d = 0; d < 3; d++) force[3 * (int64_t)id[i] + d] += lf[3 * i + d];
    }
}

/* compute_addforce_conventional, stiffness.c:121-174 (per-node zero test,
 * vector_is_all_zero quake_util.c:79-96) */
HO_API void ho_addforce_conventional(int64_t E, const int32_t* lnid, const double* etable,
                                     const ho_real* tm1, const double* K1, const double* K2,
                                     ho_real* force, int zero_skip)
{
    for (int64_t e = 0; e < E; e++) {
        const int32_t* id = &lnid[8 * e];
        double c1 = etable[4 * e], c2 = etable[4 * e + 1];
        ho_real lf[24];
        memset(lf, 0, sizeof lf);
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++) {
                const ho_real* v = &tm1[3 * (int64_t)id[j]];
                int nz = (fabs(v[0]) > HO_UNDERFLOW_CAP) || (fabs(v[1]) > HO_UNDERFLOW_CAP) ||
                         (fabs(v[2]) > HO_UNDERFLOW_CAP);
                if (!zero_skip || nz) {
                    ho_mult_add(&K1[(i * 8 + j) * 9], v, -c1, &lf[3 * i]);
                    ho_mult_add(&K2[(i * 8 + j) * 9], v, -c2, &lf[3 * i]);
                }
            }
        for (int i = 0; i < 8; i++)
            for (int d = 0; d < 3; d++) force[3 * (int64_t)id[i] + d] += lf[3 * i + d];
    }
}

/* damping_addforce, damping.c:29-103 */
HO_API void ho_damping_addforce(int64_t E, const int32_t* lnid, const double* etable,
                                const ho_real* tm1, const ho_real* tm2, const double* K1,
                                const double* K2, ho_real* force, int zero_skip)
{
    for (int64_t e = 0; e < E; e++) {
        const int32_t* id = &lnid[8 * e];
        double c3 = etable[4 * e + 2], c4 = etable[4 * e + 3];
        ho_real dd[24], lf[24];                          /* fvector_t deltaDisp[8], localForce[8] */
        memset(lf, 0, sizeof lf);
        for (int i = 0; i < 8; i++)
            for (int d = 0; d < 3; d++)
                dd[3 * i + d] = tm1[3 * (int64_t)id[i] + d] - tm2[3 * (int64_t)id[i] + d];
        if (!zero_skip || ho_any_nonzero24(dd)) {
            for (int i = 0; i < 8; i++)
                for (int j = 0; j < 8; j++) {
                    ho_mult_add(&K1[(i * 8 + j) * 9], &dd[3 * j], -c3, &lf[3 * i]);
                    ho_mult_add(&K2[(i * 8 + j) * 9], &dd[3 * j], -c4, &lf[3 * i]);
                }
        }
        for (int i = 0; i < 8; i++)
            for (int d = 0; d < 3; d++) force[3 * (int64_t)id[i] + d] += lf[3 * i + d];
    }
}

/*
 * "Formulation B": the algebraically fused element force the GPU kernels use,
 * f_e = -(c1 K1 + c2 K2)(u1 + beta (u1 - u2)), beta = c3/c1 (= b/dt,
 * psolve.c:3387-3409), one effective product per element.  Not in the
 * reference; kept here as the like-for-like CPU baseline and
 * as a cross-check of the algebra against the two reference loops above.
 */
HO_API void ho_addforce_fused(int64_t E, const int32_t* lnid, const double* etable,
                              const ho_real* tm1, const ho_real* tm2, ho_real* force)
{
    ho_sign_init();
    for (int64_t e = 0; e < E; e++) {
        const int32_t* id = &lnid[8 * e];
        double c1 = etable[4 * e], c2 = etable[4 * e + 1];
        double beta = (c1 != 0.0) ? etable[4 * e + 2] / c1 : 0.0;
        double w[24];
        ho_real lf[24];
        memset(lf, 0, sizeof lf);
        for (int i = 0; i < 8; i++)
            for (int d = 0; d < 3; d++) {
                double u1 = tm1[3 * (int64_t)id[i] + d], u2 = tm2[3 * (int64_t)id[i] + d];
                w[3 * i + d] = u1 + beta * (u1 - u2);
            }
        ho_effective_elem(w, -0.5625 * (c2 + 2 * c1), -0.5625 * c2, -0.5625 * c1, lf);
        for (int i = 0; i < 8; i++)
            for (int d = 0; d < 3; d++) force[3 * (int64_t)id[i] + d] += lf[3 * i + d];
    }
}

/* ------------------------------------------------------------------------ */
/* Source, nodal update, hanging nodes                                      */
/* ------------------------------------------------------------------------ */

/* compute_addforce_s, psolve.c:5912-5928: assignment, not accumulation */
HO_API void ho_addforce_source(int32_t nloaded, const int32_t* loaded_lnid, const double* F,
                               double dt2, ho_real* force)
{
    for (int32_t i = 0; i < nloaded; i++)
        for (int d = 0; d < 3; d++)
            force[3 * (int64_t)loaded_lnid[i] + d] = F[3 * i + d] * dt2;
}

/* solver_compute_displacement, psolve.c:4072-4114 (tm3 optional) */
HO_API void ho_compute_displacement(int64_t N, const ho_real* ntable, const ho_real* tm1,
                                    ho_real* tm2, ho_real* force, ho_real* tm3)
{
    for (int64_t n = 0; n < N; n++) {
        const ho_real* np = &ntable[7 * n];
        for (int d = 0; d < 3; d++) {
            ho_real f = force[3 * n + d];                /* fvector_t nodalForce: a copy */
            f += np[1 + d] * tm1[3 * n + d] - np[4 + d] * tm2[3 * n + d];
            if (tm3) tm3[3 * n + d] = tm2[3 * n + d];
            tm2[3 * n + d] = f / np[0];
        }
    }
    memset(force, 0, sizeof(ho_real) * 3 * (size_t)N);
}

/*
 * compute_adjust, psolve.c:5936-6039.  Dangling node k has local id
 * dn_id[k], deps = dn_ptr[k+1]-dn_ptr[k] anchors dn_anchor[dn_ptr[k]..].
 * how = 0: DISTRIBUTION (value/deps added to every anchor), else ASSIGNMENT.
 */
HO_API void ho_compute_adjust(ho_real* table, int32_t items, int32_t how, int32_t ldnnum,
                              const int32_t* dn_id, const int32_t* dn_ptr,
                              const int32_t* dn_anchor)
{
    for (int32_t k = 0; k < ldnnum; k++) {
        ho_real* mine = table + (int64_t)dn_id[k] * items;
        uint32_t deps = (uint32_t)(dn_ptr[k + 1] - dn_ptr[k]);
        if (how == 0) {
            ho_real part[7];                             /* solver_float darray[7] */
            for (int t = 0; t < items; t++) part[t] = mine[t] / deps;
            for (int32_t p = dn_ptr[k]; p < dn_ptr[k + 1]; p++) {
                ho_real* anc = table + (int64_t)dn_anchor[p] * items;
                for (int t = 0; t < items; t++) anc[t] += part[t];
            }
        } else {
            for (int t = 0; t < items; t++) mine[t] = 0;
            for (int32_t p = dn_ptr[k]; p < dn_ptr[k + 1]; p++) {
                const ho_real* anc = table + (int64_t)dn_anchor[p] * items;
                for (int t = 0; t < items; t++) mine[t] += anc[t] / deps;
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* solver_run: one rank                                                     */
/* ------------------------------------------------------------------------ */

/*
 * March `nsteps` steps starting at `step0`, solver_run psolve.c:4241-4324:
 * swap tm1/tm2, [capture], source, stiffness, damping, [hanging-node force
 * distribution], update, [hanging-node displacement assignment].
 *
 * formulation: 0 = reference (stiff_method + conventional damping loop),
 *              1 = fused (formulation B).
 * forces  [nforce_steps][nloaded][3], rows indexed by absolute step
 *         (read_myForces, psolve.c:3651-3667); steps past the table apply none.
 * capture: if cap_n > 0, after the swap of step s (i.e. what stations /
 *         checkpoints see, psolve.c:4271-4280) tm1 of nodes cap_lnid[] is stored
 *         in cap_out[(s-step0)][cap_n][3].
 * On return tm1/tm2 hold the state *as the next loop iteration would find it
 * before its swap* (tm2 = newest), exactly like the reference's arrays.
 */
HO_API void ho_solver_run(int64_t E, int64_t N, const int32_t* lnid, const double* etable,
                          const ho_real* ntable, const double* K1, const double* K2,
                          ho_real* tm1, ho_real* tm2, ho_real* force, int32_t step0,
                          int32_t nsteps, double dt, int damping, int stiff_method,
                          int formulation, int zero_skip, int32_t nloaded,
                          const int32_t* loaded_lnid, const double* forces,
                          int32_t nforce_steps, int32_t cap_n, const int32_t* cap_lnid,
                          double* cap_out, int32_t ldnnum, const int32_t* dn_id,
                          const int32_t* dn_ptr, const int32_t* dn_anchor)
{
    double dt2 = dt * dt;
    ho_real *p1 = tm1, *p2 = tm2;
    for (int32_t s = step0; s < step0 + nsteps; s++) {
        ho_real* t = p2; p2 = p1; p1 = t;                    /* psolve.c:4271-4273 */
        for (int32_t c = 0; c < cap_n; c++)
            for (int d = 0; d < 3; d++)
                cap_out[((int64_t)(s - step0) * cap_n + c) * 3 + d] = p1[3 * (int64_t)cap_lnid[c] + d];
        if (nloaded > 0 && s < nforce_steps)
            ho_addforce_source(nloaded, loaded_lnid, &forces[(int64_t)s * nloaded * 3], dt2, force);
        if (formulation == 1) {
            ho_addforce_fused(E, lnid, etable, p1, p2, force);
        } else {
            if (stiff_method == HO_STIFF_EFFECTIVE)
                ho_addforce_effective(E, lnid, etable, p1, force, zero_skip);
            else
                ho_addforce_conventional(E, lnid, etable, p1, K1, K2, force, zero_skip);
            if (damping == HO_DAMP_RAYLEIGH || damping == HO_DAMP_MASS)   /* psolve.c:3991 */
                ho_damping_addforce(E, lnid, etable, p1, p2, K1, K2, force, zero_skip);
        }
        if (ldnnum > 0)                                     /* solver_adjust_forces, psolve.c:4299 */
            ho_compute_adjust(force, 3, 0, ldnnum, dn_id, dn_ptr, dn_anchor);
        ho_compute_displacement(N, ntable, p1, p2, force, NULL);
        if (ldnnum > 0)                                     /* solver_adjust_displacement, :4313 */
            ho_compute_adjust(p2, 3, 1, ldnnum, dn_id, dn_ptr, dn_anchor);
    }
    if (p1 != tm1) {            /* odd number of swaps: put the roles back into the caller's arrays */
        size_t bytes = sizeof(ho_real) * 3 * (size_t)N;
        ho_real* t = (ho_real*)malloc(bytes);
        memcpy(t, tm1, bytes); memcpy(tm1, tm2, bytes); memcpy(tm2, t, bytes);
        free(t);
    }
}
