#define _FILE_OFFSET_BITS 64
/*
 * hq_host.c -- C host side of the MI355X engine (include/hq_host.h).
 *
 * Builds, without the octree mesher or the etree database, exactly the arrays
 * the reference's solver_run() works on for a uniformly refined, horizontally
 * layered box, for one partition of `nranks`:
 *
 *   - elements in octree pre-order = Z-order (octor.c:6444-6470), block
 *     partition of the Z-ordered element list (octor.c:740-746, 4939-4944);
 *   - harbored nodes in Z-order of their far-boundary-adjusted coordinates
 *     (octor.c:6100-6106, 6166); owner = rank whose element interval contains
 *     the adjusted node (octor.c:5466-5475);
 *   - eTable (psolve.c:3387-3409) and nTable (psolve.c:3436-3471) with the
 *     reference's single-precision material arithmetic (edata_t is float);
 *   - an_sched messenger lists (psolve.c:4704-4863).
 *
 * nTable is evaluated node-by-node (each node gathers its <= 8 elements in
 * Z-order, the order the reference's element loop reaches them) so the build
 * parallelises over nodes, and every harbored copy holds the complete sums: no
 * initial mass exchange (psolve.c:3498-3507) is needed.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "hq_host.h"

#define HQH_PI 3.14159265358979323846

struct hqh_box {
    hqh_box_params p;
    int ax, ay, az;                 /* log2 of nx, ny, nz */
    uint64_t zmask;                 /* Z-value bits that in-domain elements can set */
    int64_t *zx, *zy, *zz;          /* the element index's bits from i, j, k: index = zx[i] | zy[j] | zz[k] (the compaction
                                       of the Z-value over zmask maps the three axes' bits to disjoint positions) */
    int64_t Eg, Ng;                 /* whole box */
    int64_t elo, ehi;               /* my element interval */
    int32_t lenum, nharbored, nowned;
    int32_t* lnid;                  /* [lenum][8] */
    int32_t* node_ijk;              /* [nharbored][3] */
    int32_t* node_xyz;              /* [nharbored][3] ticks */
    int32_t* owner;                 /* [nharbored] */
    int32_t* loc;                   /* [Ng] global grid index -> local id or -1 */
    double* etable;                 /* [lenum][4] */
    double* ntable;                 /* [nharbored][7] */
    /* element constants per material index (depth index x lateral class, mat_index) */
    int32_t ncls;                   /* lateral classes (1: the material depends on depth only) */
    float *k_vp, *k_vs, *k_rho;
    double *k_c1, *k_c2, *k_c3, *k_c4, *k_a, *k_M;
    /* schedule */
    int32_t nc, ns;
    hq_messenger *mc, *ms;
    int32_t *cmap, *smap;
    int64_t shared_nodes;
    float* layer_store;
    float* edata;                   /* [lenum][4] edgesize, Vp, Vs, rho as solver_init leaves edata_t (material that differs
                                       between elements only: hq_desc.edata lets hq_k_brick_het keep 12 bytes per element) */
    double bbase;                   /* Global.theBBase */
};

/* ------------------------------------------------------------------------ */
/* bit helpers                                                              */
/* ------------------------------------------------------------------------ */

static uint64_t spread3(uint64_t v)
{
    v &= 0x1fffffULL;
    v = (v | (v << 32)) & 0x1f00000000ffffULL;
    v = (v | (v << 16)) & 0x1f0000ff0000ffULL;
    v = (v | (v << 8))  & 0x100f00f00f00f00fULL;
    v = (v | (v << 4))  & 0x10c30c30c30c30c3ULL;
    v = (v | (v << 2))  & 0x1249249249249249ULL;
    return v;
}

static uint32_t compact3(uint64_t v)
{
    v &= 0x1249249249249249ULL;
    v = (v | (v >> 2))  & 0x10c30c30c30c30c3ULL;
    v = (v | (v >> 4))  & 0x100f00f00f00f00fULL;
    v = (v | (v >> 8))  & 0x1f0000ff0000ffULL;
    v = (v | (v >> 16)) & 0x1f00000000ffffULL;
    v = (v | (v >> 32)) & 0x1fffffULL;
    return (uint32_t)v;
}

static uint64_t zvalue(uint32_t x, uint32_t y, uint32_t z)
{
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}

/* software pext / pdep over the (sparse) element mask */
static uint64_t bits_extract(uint64_t v, uint64_t mask)
{
    uint64_t r = 0;
    int k = 0;
    for (uint64_t m = mask; m; m &= m - 1, k++)
        if (v & (m & -m)) r |= 1ULL << k;
    return r;
}

static uint64_t bits_deposit(uint64_t v, uint64_t mask)
{
    uint64_t r = 0;
    int k = 0;
    for (uint64_t m = mask; m; m &= m - 1, k++)
        if (v & (1ULL << k)) r |= (m & -m);
    return r;
}

static int ilog2_exact(int32_t n)
{
    int b = 0;
    if (n <= 0 || (n & (n - 1))) return -1;
    while ((1 << b) < n) b++;
    return b;
}

/* index of element (i,j,k) in the Z-ordered list of in-domain elements */
static int64_t elem_index(const hqh_box* b, int32_t i, int32_t j, int32_t k)
{
    if (b->zx) return b->zx[i] | b->zy[j] | b->zz[k];
    return (int64_t)bits_extract(zvalue((uint32_t)i, (uint32_t)j, (uint32_t)k), b->zmask);
}

/* BLOCK_OWNER, octor.c:741-746 */
static int32_t rank_of_elem(const hqh_box* b, int64_t idx)
{
    return (int32_t)((((idx + 1) * b->p.nranks) - 1) / b->Eg);
}

static int64_t grid_index(const hqh_box* b, int32_t i, int32_t j, int32_t k)
{
    return ((int64_t)k * (b->p.ny + 1) + j) * (b->p.nx + 1) + i;
}

static uint64_t node_key(const hqh_box* b, int32_t i, int32_t j, int32_t k)
{
    uint32_t dx = (i == b->p.nx) ? (uint32_t)(2 * i - 1) : (uint32_t)(2 * i);
    uint32_t dy = (j == b->p.ny) ? (uint32_t)(2 * j - 1) : (uint32_t)(2 * j);
    uint32_t dz = (k == b->p.nz) ? (uint32_t)(2 * k - 1) : (uint32_t)(2 * k);
    return zvalue(dx, dy, dz);
}

/* LSD radix sort of 64-bit keys, 11-bit digits, `bits` significant bits */
static int radix_sort_u64(uint64_t* a, int64_t n, int bits)
{
    uint64_t* t = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    if (!t) return -1;
    uint64_t* src = a;
    uint64_t* dst = t;
    int nth = 1;
#ifdef _OPENMP
    nth = n > (1 << 20) ? omp_get_max_threads() : 1;
    if (nth > 64) nth = 64;
#endif
    int64_t* hist = (int64_t*)malloc(sizeof(int64_t) * 2048 * (size_t)nth);
    if (!hist) { free(t); return -1; }
    for (int sh = 0; sh < bits; sh += 11) {
        /* per-thread histograms over contiguous blocks: a stable parallel counting pass */
        memset(hist, 0, sizeof(int64_t) * 2048 * (size_t)nth);
        /* slices dealt by a worksharing loop: independent of the team the runtime delivers */
#pragma omp parallel for schedule(static, 1) num_threads(nth)
        for (int tid = 0; tid < nth; tid++) {
            const int64_t lo = n * tid / nth, hi = n * (tid + 1) / nth;
            int64_t* h = hist + 2048 * (size_t)tid;
            for (int64_t i = lo; i < hi; i++) h[(src[i] >> sh) & 2047]++;
        }
        int64_t run = 0;
        for (int d = 0; d < 2048; d++)
            for (int q = 0; q < nth; q++) { int64_t c = hist[2048 * (size_t)q + d]; hist[2048 * (size_t)q + d] = run; run += c; }
#pragma omp parallel for schedule(static, 1) num_threads(nth)
        for (int tid = 0; tid < nth; tid++) {
            const int64_t lo = n * tid / nth, hi = n * (tid + 1) / nth;
            int64_t* h = hist + 2048 * (size_t)tid;
            for (int64_t i = lo; i < hi; i++) dst[h[(src[i] >> sh) & 2047]++] = src[i];
        }
        uint64_t* s = src; src = dst; dst = s;
    }
    free(hist);
    if (src != a) memcpy(a, src, sizeof(uint64_t) * (size_t)n);
    free(t);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* physics set-up                                                           */
/* ------------------------------------------------------------------------ */

/* compute_setab, psolve.c:5813-5876 */
static void rayleigh_base(double freq, int damping, double* aBase, double* bBase)
{
    *aBase = 0.0;
    *bBase = 0.0;
    if (damping == HQH_DAMP_RAYLEIGH) {
        double w1 = 2 * HQH_PI * freq * .2, w2 = 2 * HQH_PI * freq * 1;
        double l1 = log(w1), l2 = log(w2);
        double s1 = w1 * w1, s2 = w2 * w2;
        double q1 = w1 * w1 * w1, q2 = w2 * w2 * w2;
        double den = (q1 - q2 + 3 * s2 * w1 - 3 * s1 * w2);
        double num = w1 * w2 * (-2 * s1 * l2 + 2 * s1 * l1 - 2 * w1 * w2 * l2 + 2 * w1 * w2 * l1
                                + 3 * s2 - 3 * s1 - 2 * s2 * l2 + 2 * s2 * l1);
        *aBase = num / den;
        num = 3 * (2 * w1 * w2 * l2 - 2 * w1 * w2 * l1 + s1 - s2);
        *bBase = num / den;
    } else if (damping == HQH_DAMP_MASS) {
        double w1 = 2 * HQH_PI * freq * .1, w2 = 2 * HQH_PI * freq * 8;
        *aBase = 1.3 * (2 * w2 * w1 * log(w2 / w1)) / (w2 - w1);
    }
}

/*
 * Material classes (hqh_box_params.lateral_classes > 1): the element (ei, ej, ek) belongs to class
 * hash(ei, ej, ek) mod ncls and its Vp, Vs, rho are its layer's times a class factor in [1 - amp, 1 + amp] -- a mesh
 * whose material differs from element to element in all three directions, as solver_init sees it on any real CVM
 * (psolve.c:3360-3409 reads every element's own edata_t), while the table of distinct materials stays small.
 */
static inline int32_t lateral_class(const hqh_box* b, int32_t ei, int32_t ej, int32_t ek)
{
    if (b->ncls <= 1) return 0;
    ei += b->p.origin[0]; ej += b->p.origin[1]; ek += b->p.origin[2];
    uint32_t h = (uint32_t)ei * 0x9E3779B1u ^ ((uint32_t)ej * 0x85EBCA77u + 0x165667B1u) ^ ((uint32_t)ek * 0xC2B2AE3Du + 0x27D4EB2Fu);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return (int32_t)(h % (uint32_t)b->ncls);
}

static inline int64_t mat_index(const hqh_box* b, int32_t ei, int32_t ej, int32_t ek)
{
    return (int64_t)ek * b->ncls + lateral_class(b, ei, ej, ek);
}

static inline float lateral_factor(const hqh_box* b, int32_t cls)
{
    if (b->ncls <= 1) return 1.0f;
    return (float)(1.0 + b->p.lateral_amp * (2.0 * cls / (double)(b->ncls - 1) - 1.0));
}

/*
 * Element constants per material index (depth index k, lateral class):
 * mu_and_lambda (psolve.c:3236-3272) + psolve.c:3387-3409, 3436-3437.
 */
static int depth_constants(hqh_box* b)
{
    const hqh_box_params* p = &b->p;
    double aBase, bBase;
    rayleigh_base(p->freq, p->damping, &aBase, &bBase);
    double dt = p->deltaT, dt2 = dt * dt;
    float h = (float)p->h;
    for (int32_t k = 0; k < p->nz; k++) {
        double zc = (k + p->origin[2] + 0.5) * p->h;
        int L = 0;
        for (int l = 0; l < p->nlayers; l++)
            if (p->layer_ztop[l] <= zc) L = l;
      for (int32_t cls = 0; cls < b->ncls; cls++) {
        const float fac = lateral_factor(b, cls);
        float Vp = p->layer_vp[L] * fac, Vs = p->layer_vs[L] * fac, rho = p->layer_rho[L] * fac;
        const int64_t m = (int64_t)k * b->ncls + cls;
        double mu = rho * Vs * Vs;
        double lambda;
        if (Vp > (Vs * p->threshold_vpvs))
            lambda = rho * Vs * Vs * p->threshold_vpvs * p->threshold_vpvs - 2 * mu;
        else
            lambda = rho * Vp * Vp - 2 * mu;
        if (lambda < 0) {
            if (Vs < 500) Vp = 2.45 * Vs;
            else if (Vs < 1200) Vp = 2 * Vs;
            else Vp = 1.87 * Vs;
            lambda = rho * Vp * Vp;
        }
        if (lambda < 0) return -1;
        double zeta = 10 / Vs;
        if (zeta > p->threshold_damping) zeta = p->threshold_damping;
        double a = zeta * aBase, bb = zeta * bBase;
        b->k_vp[m] = Vp; b->k_vs[m] = Vs; b->k_rho[m] = rho;
        b->k_c1[m] = dt2 * h * mu / 9;
        b->k_c2[m] = dt2 * h * lambda / 9;
        b->k_c3[m] = bb * dt * h * mu / 9;
        b->k_c4[m] = bb * dt * h * lambda / 9;
        b->k_a[m] = a;
        double mass = rho * h * h * h;
        b->k_M[m] = mass / 8;
      }
    }
    return 0;
}

/*
 * Lysmer dashpot of the corner `corner` (bit d = far side of axis d) of an
 * element at (ei,ej,ek), per axis: compute_setflag + theIDBoundaryMatrix +
 * compute_setboundary (psolve.c:5629-5804).  An absorbing face normal to axis
 * d acts on the corner if the element touches that domain face and the corner
 * lies on it; with -DHALFSPACE the z = 0 face is free (:5762-5764).
 * Returns 0 if the element is interior (flag 13).
 */
static int corner_dashpot(const hqh_box* b, int32_t ei, int32_t ej, int32_t ek, int corner, double out[3])
{
    const hqh_box_params* p = &b->p;
    int e[3] = { ei, ej, ek }, n[3] = { p->nx, p->ny, p->nz };
    int touches = 0, bits = 0;
    for (int d = 0; d < 3; d++) {
        int near = (e[d] == 0), far = (e[d] == n[d] - 1);
        /* compute_setflag tests the far face after the near one, so an element that is
         * both (one element thick) is classed "far" (psolve.c:5636-5648) */
        if (near || far) touches = 1;
        int cls = far ? 2 : (near ? 0 : 1);
        if (d == 2 && p->halfspace && cls == 0) cls = 1;
        int cfar = (corner >> d) & 1;
        if ((cls == 0 && !cfar) || (cls == 2 && cfar)) bits |= 1 << d;
    }
    out[0] = out[1] = out[2] = 0.0;
    if (!touches) return 0;
    const int64_t mi = mat_index(b, ei, ej, ek);
    float size = (float)p->h, Vp = b->k_vp[mi], Vs = b->k_vs[mi], rho = b->k_rho[mi];
    double scale = rho * (size / 2) * (size / 2);
    int nf = (bits & 1) + ((bits >> 1) & 1) + ((bits >> 2) & 1);
    for (int d = 0; d < 3; d++) {
        if (nf == 3) out[d] = (Vp + 2 * Vs) * scale;
        else if (nf == 2) out[d] = (Vs + ((bits & (1 << d)) ? Vp : Vs)) * scale;
        else if (nf == 1) out[d] = ((bits & (1 << d)) ? Vp : Vs) * scale;
    }
    return 1;
}

/*
 * solver_float (psolve.h:60-64).  An n_t row is a sum of double terms into solver_float fields (psolve.c:3440-3471; the
 * hanging nodes' share, compute_adjust :5958-5990): built with -DSINGLE_PRECISION_SOLVER every `+=` there rounds to
 * float.  The tables stay double arrays in this library; with solver_float = 4 every update is rounded as the float field
 * rounds it -- (double) field + term -> float -- so that a row holds exactly the float build's values (a double sum or
 * quotient of two floats, rounded once more to float, IS the float operation: 53 >= 2 x 24 + 2 bits).  One rank: the
 * summation order of its element loop.  N ranks: every rank's elements apart and solver_init's three mass exchanges on
 * the partial rows -- nt_rank_rows / nt_parts_combine below.
 */
#define HQH_SF(f32, x) ((f32) ? (double)(float)(x) : (x))

static inline int hqh_sf_valid(int32_t solver_float) { return solver_float == 0 || solver_float == 4 || solver_float == 8; }

/* one element's share of its corner's row: psolve.c:3440-3471 */
static inline void nt_accumulate(double* np, int f32, double dt, double a, double M, int bnd, const double* dash)
{
    np[0] = HQH_SF(f32, np[0] + M);
    for (int ax = 0; ax < 3; ax++) {
        np[4 + ax] = HQH_SF(f32, np[4 + ax] - (dt * a * M));
        np[1 + ax] = HQH_SF(f32, np[1 + ax] - (dt * a * M));
        if (bnd) {
            np[4 + ax] = HQH_SF(f32, np[4 + ax] - (dt * dash[ax]));
            np[1 + ax] = HQH_SF(f32, np[1 + ax] - (dt * dash[ax]));
        }
        np[4 + ax] = HQH_SF(f32, np[4 + ax] + M);
        np[1 + ax] = HQH_SF(f32, np[1 + ax] + (M * 2));
    }
}

/* compute_adjust(nTable, 7, DISTRIBUTION), psolve.c:5958-5990: a hanging node's row, divided by its anchors' number, to every
 * anchor */
static void nt_distribute(double* ntable, int f32, int32_t ldnnum, const int32_t* dn_id, const int32_t* dn_ptr,
                          const int32_t* dn_anchor)
{
    for (int32_t k = 0; k < ldnnum; k++) {
        double part[7];
        uint32_t deps = (uint32_t)(dn_ptr[k + 1] - dn_ptr[k]);
        for (int q = 0; q < 7; q++) part[q] = HQH_SF(f32, ntable[7 * (int64_t)dn_id[k] + q] / deps);
        for (int32_t a = dn_ptr[k]; a < dn_ptr[k + 1]; a++)
            for (int q = 0; q < 7; q++) {
                double* v = &ntable[7 * (int64_t)dn_anchor[a] + q];
                *v = HQH_SF(f32, *v + part[q]);
            }
    }
}

/*
 * The float build on N ranks.  Every rank sums the rows over ITS elements (psolve.c:3440-3471, float fields), then
 * solver_init's three exchanges run on those partial rows (:3498-3507): (A) the sharers of a hanging node send theirs to
 * its owner, which adds them in the order of its messenger list (schedule_senddata, CONTRIBUTION: :5040-5060; the lists
 * in schedule_build's own order, hqh_share_list below); (B) the owner of a hanging node hands its row / deps to the anchors
 * (compute_adjust, its own dnodeTable in order); (C) the sharers of an anchored node send their rows -- their elements'
 * sums plus what (B) gave them -- to its owner, which adds them the same way.  In float every one of those additions
 * rounds, and a run is sensitive to WHICH roundings its rows carry (m2 - m1 = m0 holds only up to them): against the float
 * reference's own 8-rank checkpoints a field stepped on rows in this order is 1e-6 off, on rows in one rank's order 5e-5
 * (tests/test_gpu_single_precision.py).  So with solver_float = 4 a partition's rows follow this order; every harbored
 * copy of a node gets what its OWNER ends up with (the reference leaves a partial sum in the other copies, which the
 * sharing of the displacements makes irrelevant, psolve.c:4312-4315).
 * nt_rank_rows: rows[0..n) of ranks rk[0..n) -> out = the owner's row + the others' in the order of the owner's messenger
 * list (pos[rank] = the rank's place in it; schedule_build's order, see build_schedule).
 */
static void nt_rank_rows(int n, const int* rk, double (*rows)[7], int owner, const int* pos, double out[7])
{
    int ord[64], m = 0;
    for (int t = 0; t < 7; t++) out[t] = 0.0;
    for (int q = 0; q < n && q < 64; q++) {
        if (rk[q] == owner) { for (int t = 0; t < 7; t++) out[t] = rows[q][t]; continue; }
        int u = m++;
        while (u > 0 && pos[rk[ord[u - 1]]] > pos[rk[q]]) { ord[u] = ord[u - 1]; u--; }
        ord[u] = q;
    }
    for (int q = 0; q < m; q++) for (int t = 0; t < 7; t++) out[t] = HQH_SF(1, out[t] + rows[ord[q]][t]);
}

/* Every rank's messenger lists of one schedule, as places: the nodes in global (= every rank's local) order, an owned
 * node's sharers in its share list's order (hqh_share_list); a rank that turns up for the first time goes to the HEAD of
 * the owner's list.  first[o * P + r] (in: the order number of that first time, < 0 never) -> pos[o * P + r] = place of r
 * in o's list. */
static void nt_list_places(int P, const int64_t* first, int* pos)
{
    for (int o = 0; o < P; o++) {
        int seq[64], n = 0;
        for (int r = 0; r < P; r++) {
            pos[o * P + r] = 0;
            if (first[o * P + r] < 0) continue;
            int u = n++;
            while (u > 0 && first[o * P + seq[u - 1]] > first[o * P + r]) { seq[u] = seq[u - 1]; u--; }
            seq[u] = r;
        }
        for (int q = 0; q < n; q++) pos[o * P + seq[q]] = n - 1 - q;          /* the last one met is the first of the list */
    }
}

/*
 * The ORDER of a vertex's share list, and with it of the messenger lists (the reference's multi-rank runs come out bit for
 * bit only in it: tests/test_oracle_golden.py).  octor_extractmesh: every rank sends each vertex of its elements to the
 * ranks that hold one of the 8 pixels around it; the owner takes the messages in the order of its processor-controller
 * list and puts each sender at the HEAD of the vertex's list (octor.c:5700-5793) -- and that controller list has the
 * neighbours in the REVERSE of the order in which com_allocpctl met them (:2640-2741: the rank's leaves in order, around
 * each the 4 x 4 x 4 points at half-edge spacing from corner - edge / 2, z outermost, a new rank to the head).  So the share
 * list has the element-vertex sharers in the order the owner MET them; ahead of them, in descending rank, the ranks that
 * hold the vertex only as an anchor of a hanging node of theirs (:5800-5830, 5990-6050: a list of all ranks in ascending
 * order, each sender to the head).  schedule_build (psolve.c:4711-4795) then walks the owned nodes in local order and
 * each one's share list, a NEW messenger to the head of the schedule's list.
 * met[r] = when the owner met rank r (smaller = earlier; < 0: never); -> out[0..n): the sharers in share-list order.
 */
static int hqh_share_list(uint64_t all, uint64_t direct, int owner, int P, const int64_t* met, int* out)
{
    int n = 0;
    const uint64_t me = 1ull << owner, ind = all & ~direct & ~me, dir = direct & ~me;
    for (int r = P - 1; r >= 0; r--) if ((ind >> r) & 1) out[n++] = r;
    const int n0 = n;
    for (int r = 0; r < P; r++) {
        if (!((dir >> r) & 1)) continue;
        int u = n++;
        while (u > n0 && (met[out[u - 1]] < 0 || (met[r] >= 0 && met[out[u - 1]] > met[r]))) { out[u] = out[u - 1]; u--; }
        out[u] = r;
    }
    return n;
}

/* probe k = 0 .. 3 of com_allocpctl along one axis, for a leaf at l of edge s (finest-edge units): the cell that holds the
 * point l - s/2 + k s/2, or -1 out of bounds */
static inline int64_t hqh_probe_cell(int64_t l, int64_t s, int k, int64_t far)
{
    const int64_t p2 = 2 * l + (k - 1) * s;          /* twice the coordinate */
    if (p2 < 0 || p2 >= 2 * far) return -1;
    return p2 >> 1;
}

/* The same for the whole-mesh builders (hqh_octbox_create_levels, hqh_mesh_from_leaves: every rank builds the whole mesh and
 * cuts its part out): their element loops run over the elements in global order, so a node's row is summed in RUNS of
 * ranks; nt_parts_row closes a run when the rank changes (the row so far becomes a record, the row restarts at zero),
 * nt_parts_combine (octbox_cut, where the owners are known) replays the three exchanges on the records. */
typedef struct {
    int P;
    int64_t E, n, cap;
    int32_t *node, *rank, *cur;     /* records' node and rank; cur[node]: rank of the run its row is in, -1 none yet */
    double* row;                    /* records' rows [n][7] */
} nt_parts;

static void nt_parts_free(nt_parts* pp)
{
    if (!pp) return;
    free(pp->node); free(pp->rank); free(pp->cur); free(pp->row); free(pp);
}

static nt_parts* nt_parts_new(int64_t N, int P, int64_t E)
{
    nt_parts* pp = (nt_parts*)calloc(1, sizeof(nt_parts));
    if (!pp) return NULL;
    pp->P = P; pp->E = E;
    pp->cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
    if (!pp->cur) { nt_parts_free(pp); return NULL; }
    for (int64_t i = 0; i < N; i++) pp->cur[i] = -1;
    return pp;
}

static int nt_parts_push(nt_parts* pp, int32_t node, int32_t rank, const double* row)
{
    if (pp->n == pp->cap) {
        const int64_t cap = pp->cap ? 2 * pp->cap : 1 << 16;
        int32_t* a = (int32_t*)realloc(pp->node, sizeof(int32_t) * (size_t)cap);
        if (a) pp->node = a;
        int32_t* r = (int32_t*)realloc(pp->rank, sizeof(int32_t) * (size_t)cap);
        if (r) pp->rank = r;
        double* w = (double*)realloc(pp->row, sizeof(double) * 7 * (size_t)cap);
        if (w) pp->row = w;
        if (!a || !r || !w) return -1;
        pp->cap = cap;
    }
    pp->node[pp->n] = node; pp->rank[pp->n] = rank;
    for (int t = 0; t < 7; t++) pp->row[7 * pp->n + t] = row ? row[t] : 0.0;
    pp->n++;
    return 0;
}

/* ahead of adding element e's share to node n's row in `ntable` */
static inline int nt_parts_row(nt_parts* pp, double* ntable, int32_t n, int64_t e)
{
    const int r = (int)(((e + 1) * pp->P - 1) / pp->E);                /* octor.c:4939-4944 */
    if (pp->cur[n] == r) return 0;
    if (pp->cur[n] >= 0) {
        if (nt_parts_push(pp, n, pp->cur[n], &ntable[7 * (int64_t)n])) return -1;
        for (int t = 0; t < 7; t++) ntable[7 * (int64_t)n + t] = 0.0;
    }
    pp->cur[n] = r;
    return 0;
}

/* the runs still open, then exchanges (A), (B), (C) -> ntable[N][7] as every node's owner holds it */
static int nt_parts_combine(nt_parts* pp, int64_t N, double* ntable, const int32_t* gowner, int32_t ldnnum, const int32_t* dn_id,
                            const int32_t* dn_ptr, const int32_t* dn_anchor, const int* pos_an, const int* pos_dn)
{
    const int P = pp->P;
    for (int64_t n = 0; n < N; n++)
        if (pp->cur[n] >= 0 && nt_parts_push(pp, (int32_t)n, pp->cur[n], &ntable[7 * n])) return HQ_ERR_NOMEM;
    /* a rank that owns a hanging node holds its anchors (octor.c: indirect sharing) even where it has no element at them */
    for (int32_t k = 0; k < ldnnum; k++)
        for (int32_t a = dn_ptr[k]; a < dn_ptr[k + 1]; a++)
            if (nt_parts_push(pp, dn_anchor[a], gowner[dn_id[k]], NULL)) return HQ_ERR_NOMEM;
    const int64_t M = pp->n;
    int64_t* off = (int64_t*)calloc((size_t)N + 1, sizeof(int64_t));
    int64_t* ord = (int64_t*)malloc(sizeof(int64_t) * (size_t)(M ? M : 1));
    double* full = (double*)malloc(sizeof(double) * 7 * (size_t)(ldnnum ? ldnnum : 1));
    int rc = HQ_ERR_NOMEM;
    if (!off || !ord || !full) goto done;
    for (int64_t i = 0; i < M; i++) off[pp->node[i] + 1]++;
    for (int64_t n = 0; n < N; n++) off[n + 1] += off[n];
    {
        int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(N ? N : 1));
        if (!fill) goto done;
        memcpy(fill, off, sizeof(int64_t) * (size_t)N);
        for (int64_t i = 0; i < M; i++) ord[fill[pp->node[i]]++] = i;          /* stable: push order inside a node */
        free(fill);
    }
    /* inside a node: ascending rank (stable: the run of elements, pushed first, ahead of the empty records of the same rank) */
    for (int64_t n = 0; n < N; n++) {
        int64_t* o = ord + off[n];
        const int64_t c = off[n + 1] - off[n];
        for (int64_t i = 1; i < c; i++) {
            const int64_t v = o[i];
            int64_t j = i;
            while (j > 0 && pp->rank[o[j - 1]] > pp->rank[v]) { o[j] = o[j - 1]; j--; }
            o[j] = v;
        }
    }
    rc = HQ_OK;
    {
        /* several records of one (node, rank): the first holds the elements' run (or is empty too), the others are empty: dropped */
        for (int64_t n = 0; n < N; n++) {
            int last = -1;
            for (int64_t i = off[n]; i < off[n + 1]; i++) {
                const int r = pp->rank[ord[i]];
                if (r == last) pp->rank[ord[i]] = -1 - r;                      /* dropped (negative), order kept */
                else last = r;
            }
        }
#define NTP_GATHER(n_, rows_, rk_, cnt_)                                                               \
        { cnt_ = 0;                                                                                    \
          for (int64_t i_ = off[n_]; i_ < off[(n_) + 1]; i_++) {                                       \
              if (pp->rank[ord[i_]] < 0) continue;                                                     \
              if (cnt_ >= 64) { rc = HQ_ERR_STATE; goto done; }                                        \
              rk_[cnt_] = pp->rank[ord[i_]];                                                           \
              for (int t_ = 0; t_ < 7; t_++) rows_[cnt_][t_] = pp->row[7 * ord[i_] + t_];              \
              cnt_++; } }
        double rows[64][7];
        int rk[64], cnt;
        /* (A) */
        for (int32_t k = 0; k < ldnnum; k++) {
            const int32_t d = dn_id[k];
            NTP_GATHER(d, rows, rk, cnt)
            nt_rank_rows(cnt, rk, rows, gowner[d], pos_dn + gowner[d] * P, &full[7 * (int64_t)k]);
        }
        /* (B): every owner through its dnodeTable -- the global table's order restricted to its nodes */
        for (int32_t k = 0; k < ldnnum; k++) {
            const int r = gowner[dn_id[k]];
            const uint32_t deps = (uint32_t)(dn_ptr[k + 1] - dn_ptr[k]);
            double part[7];
            for (int t = 0; t < 7; t++) part[t] = HQH_SF(1, full[7 * (int64_t)k + t] / deps);
            for (int32_t a = dn_ptr[k]; a < dn_ptr[k + 1]; a++) {
                const int32_t an = dn_anchor[a];
                int64_t hit = -1;
                for (int64_t i = off[an]; i < off[an + 1]; i++) if (pp->rank[ord[i]] == r) { hit = ord[i]; break; }
                if (hit < 0) { rc = HQ_ERR_STATE; goto done; }
                for (int t = 0; t < 7; t++) pp->row[7 * hit + t] = HQH_SF(1, pp->row[7 * hit + t] + part[t]);
            }
        }
        /* (C); a hanging node's row is what (A) left (nothing is handed TO it; its owner's OTHER list ordered it) */
        for (int64_t n = 0; n < N; n++) {
            NTP_GATHER(n, rows, rk, cnt)
            nt_rank_rows(cnt, rk, rows, gowner[n], pos_an + gowner[n] * P, &ntable[7 * n]);
        }
        for (int32_t k = 0; k < ldnnum; k++) memcpy(&ntable[7 * (int64_t)dn_id[k]], &full[7 * (int64_t)k], 7 * sizeof(double));
#undef NTP_GATHER
    }
done:
    free(off); free(ord); free(full);
    return rc;
}

/* n_t of node (i,j,k): psolve.c:3440-3471 summed over its elements in Z-order (solver_float = 4 on partitions: every
 * rank's elements apart, then the owner's row + the sharers', see above) */
static void node_constants(const hqh_box* b, int32_t i, int32_t j, int32_t k, int owner, const int* places, double np[7])
{
    const hqh_box_params* p = &b->p;
    int64_t idx[8];
    int32_t ee[8][3];
    int cn[8], cnt = 0;
    for (int c = 0; c < 8; c++) {
        int32_t ei = i - 1 + (c & 1), ej = j - 1 + ((c >> 1) & 1), ek = k - 1 + ((c >> 2) & 1);
        if (ei < 0 || ej < 0 || ek < 0 || ei >= p->nx || ej >= p->ny || ek >= p->nz) continue;
        int64_t id = elem_index(b, ei, ej, ek);
        int pos = cnt++;
        while (pos > 0 && idx[pos - 1] > id) {
            idx[pos] = idx[pos - 1]; cn[pos] = cn[pos - 1];
            memcpy(ee[pos], ee[pos - 1], sizeof ee[0]);
            pos--;
        }
        idx[pos] = id;
        ee[pos][0] = ei; ee[pos][1] = ej; ee[pos][2] = ek;
        /* this node is corner (1-di, 1-dj, 1-dk) of that element */
        cn[pos] = (1 - (c & 1)) | ((1 - ((c >> 1) & 1)) << 1) | ((1 - ((c >> 2) & 1)) << 2);
    }
    double dt = p->deltaT;
    const int f32 = p->solver_float == 4, by_rank = f32 && p->nranks > 1;
    double rows[8][7];
    int rk[8], nr = 0;
    for (int t = 0; t < 7; t++) np[t] = 0.0;
    for (int q = 0; q < cnt; q++) {
        int32_t ek = ee[q][2];
        const int64_t mi = mat_index(b, ee[q][0], ee[q][1], ek);
        double M = b->k_M[mi], a = b->k_a[mi], dash[3];
        int bnd = corner_dashpot(b, ee[q][0], ee[q][1], ek, cn[q], dash);
        double* acc = np;
        if (by_rank) {                           /* (ranks hold runs of the Z-ordered elements: ascending with idx) */
            const int r = rank_of_elem(b, idx[q]);
            if (nr == 0 || rk[nr - 1] != r) { rk[nr] = r; for (int t = 0; t < 7; t++) rows[nr][t] = 0.0; nr++; }
            acc = rows[nr - 1];
        }
        nt_accumulate(acc, f32, dt, a, M, bnd, dash);
    }
    if (by_rank) nt_rank_rows(nr, rk, rows, owner, places + owner * p->nranks, np);
}

/* when did rank o meet rank q (hqh_share_list)?  met[o * P + q] = 64 x (index of o's first element with a probe point in
 * q's part) + the probe's number; only = -1: for every o (a pass over the whole box), else for that rank alone */
static int box_met(const hqh_box* b, int only, int64_t* met)
{
    const hqh_box_params* p = &b->p;
    const int P = p->nranks;
    for (int q = 0; q < P * P; q++) met[q] = -1;
    int nomem = 0;
#pragma omp parallel
    {
        int64_t* mine = (int64_t*)malloc(sizeof(int64_t) * (size_t)P * (size_t)P);
        if (!mine) {
#pragma omp atomic write
            nomem = 1;
        } else {
            for (int q = 0; q < P * P; q++) mine[q] = -1;
#pragma omp for schedule(static)
            for (int32_t k = 0; k < p->nz; k++)
                for (int32_t j = 0; j < p->ny; j++)
                    for (int32_t i = 0; i < p->nx; i++) {
                        const int64_t idx = elem_index(b, i, j, k);
                        const int o = rank_of_elem(b, idx);
                        if (only >= 0 && o != only) continue;
                        for (int kk = 0; kk < 4; kk++) {
                            const int64_t z = hqh_probe_cell(k, 1, kk, p->nz);
                            if (z < 0) continue;
                            for (int jj = 0; jj < 4; jj++) {
                                const int64_t y = hqh_probe_cell(j, 1, jj, p->ny);
                                if (y < 0) continue;
                                for (int ii = 0; ii < 4; ii++) {
                                    const int64_t x = hqh_probe_cell(i, 1, ii, p->nx);
                                    if (x < 0) continue;
                                    const int q = rank_of_elem(b, elem_index(b, (int32_t)x, (int32_t)y, (int32_t)z));
                                    if (q == o) continue;
                                    const int64_t when = 64 * idx + (kk * 4 + jj) * 4 + ii;
                                    int64_t* f = &mine[o * P + q];
                                    if (*f < 0 || when < *f) *f = when;
                                }
                            }
                        }
                    }
#pragma omp critical
            for (int q = 0; q < P * P; q++)
                if (mine[q] >= 0 && (met[q] < 0 || mine[q] < met[q])) met[q] = mine[q];
            free(mine);
        }
    }
    return nomem ? HQ_ERR_NOMEM : HQ_OK;
}

/* every rank's s-list of the box as places (nt_list_places): one pass over the whole grid for the node of smallest key that
 * an owner shares with each of the others (solver_float = 4 on partitions only) */
static int box_list_places(const hqh_box* b, int* pos)
{
    const hqh_box_params* p = &b->p;
    const int P = p->nranks;
    int64_t* first = (int64_t*)malloc(sizeof(int64_t) * (size_t)P * (size_t)P);
    int64_t* met = (int64_t*)malloc(sizeof(int64_t) * (size_t)P * (size_t)P);
    if (!first || !met || box_met(b, -1, met) != HQ_OK) { free(first); free(met); return HQ_ERR_NOMEM; }
    for (int q = 0; q < P * P; q++) first[q] = -1;
    int nomem = 0;
#pragma omp parallel
    {
        int64_t* mine = (int64_t*)malloc(sizeof(int64_t) * (size_t)P * (size_t)P);
        if (!mine) {
#pragma omp atomic write
            nomem = 1;
        } else {
            for (int q = 0; q < P * P; q++) mine[q] = -1;
#pragma omp for schedule(static)
            for (int32_t k = 0; k <= p->nz; k++)
                for (int32_t j = 0; j <= p->ny; j++)
                    for (int32_t i = 0; i <= p->nx; i++) {
                        int rk[8], n = 0;
                        for (int c = 0; c < 8; c++) {
                            const int32_t ei = i - 1 + (c & 1), ej = j - 1 + ((c >> 1) & 1), ek = k - 1 + ((c >> 2) & 1);
                            if (ei < 0 || ej < 0 || ek < 0 || ei >= p->nx || ej >= p->ny || ek >= p->nz) continue;
                            const int r = rank_of_elem(b, elem_index(b, ei, ej, ek));
                            int seen = 0;
                            for (int t = 0; t < n; t++) seen |= (rk[t] == r);
                            if (!seen) rk[n++] = r;
                        }
                        if (n < 2) continue;
                        const int32_t oi = i < p->nx ? i : p->nx - 1, oj = j < p->ny ? j : p->ny - 1, ok = k < p->nz ? k : p->nz - 1;
                        const int o = rank_of_elem(b, elem_index(b, oi, oj, ok));
                        const int64_t key = (int64_t)node_key(b, i, j, k);
                        uint64_t bits = 0;
                        int sl[8];
                        for (int t = 0; t < n; t++) bits |= 1ull << rk[t];
                        const int ns = hqh_share_list(bits, bits, o, P, met + o * P, sl);
                        for (int t = 0; t < ns; t++) {                   /* the node's share list: its place breaks the tie */
                            int64_t* f = &mine[o * P + sl[t]];
                            const int64_t when = 8 * key + t;
                            if (*f < 0 || when < *f) *f = when;
                        }
                    }
#pragma omp critical
            for (int q = 0; q < P * P; q++)
                if (mine[q] >= 0 && (first[q] < 0 || mine[q] < first[q])) first[q] = mine[q];
            free(mine);
        }
    }
    if (!nomem) nt_list_places(P, first, pos);
    free(first); free(met);
    return nomem ? HQ_ERR_NOMEM : HQ_OK;
}

/* HQH_VERBOSE=1: where the host side's time goes */
static double hqh_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void hqh_lap(double* t, const char* what)
{
    if (!getenv("HQH_VERBOSE")) return;
    const double n = hqh_now();
    fprintf(stderr, "  hq_host: %-32s %6.2f s\n", what, n - *t);
    *t = n;
}

/* ------------------------------------------------------------------------ */
/* box                                                                      */
/* ------------------------------------------------------------------------ */

void hqh_box_destroy(hqh_box* b)
{
    if (!b) return;
    free(b->lnid); free(b->node_ijk); free(b->node_xyz); free(b->owner); free(b->loc);
    free(b->etable); free(b->ntable);
    free(b->k_vp); free(b->k_vs); free(b->k_rho);
    free(b->k_c1); free(b->k_c2); free(b->k_c3); free(b->k_c4); free(b->k_a); free(b->k_M);
    free(b->mc); free(b->ms); free(b->cmap); free(b->smap);
    free(b->layer_store);
    free(b->zx); free(b->zy); free(b->zz);
    free(b->edata);
    free(b);
}

static int build_schedule(hqh_box* b);

int hqh_box_create(const hqh_box_params* p, hqh_box** out)
{
    if (!p || !out) return HQ_ERR_ARG;
    *out = NULL;
    int ax = ilog2_exact(p->nx), ay = ilog2_exact(p->ny), az = ilog2_exact(p->nz);
    if (ax < 0 || ay < 0 || az < 0 || ax > 10 || ay > 10 || az > 10) return HQ_ERR_ARG;
    if (p->nlayers < 1 || !p->layer_ztop || !p->layer_vp || !p->layer_vs || !p->layer_rho) return HQ_ERR_ARG;
    if (p->nranks < 1 || p->rank < 0 || p->rank >= p->nranks || p->h <= 0 || p->deltaT <= 0) return HQ_ERR_ARG;
    if (!hqh_sf_valid(p->solver_float) || p->nranks > 64) return HQ_ERR_ARG;       /* (a node's sharers are a 64-bit set, as in the octree boxes) */
    hqh_box* b = (hqh_box*)calloc(1, sizeof(hqh_box));
    if (!b) return HQ_ERR_NOMEM;
    b->p = *p;
    /* private copy of the layer tables */
    size_t nl = (size_t)p->nlayers;
    b->layer_store = (float*)malloc(nl * (3 * sizeof(float) + sizeof(double)));
    if (!b->layer_store) { hqh_box_destroy(b); return HQ_ERR_NOMEM; }
    {
        double* zt = (double*)b->layer_store;
        float* f = (float*)(zt + nl);
        memcpy(zt, p->layer_ztop, nl * sizeof(double));
        memcpy(f, p->layer_vp, nl * sizeof(float));
        memcpy(f + nl, p->layer_vs, nl * sizeof(float));
        memcpy(f + 2 * nl, p->layer_rho, nl * sizeof(float));
        b->p.layer_ztop = zt; b->p.layer_vp = f; b->p.layer_vs = f + nl; b->p.layer_rho = f + 2 * nl;
    }
    b->ax = ax; b->ay = ay; b->az = az;
    b->zmask = zvalue((uint32_t)p->nx - 1, (uint32_t)p->ny - 1, (uint32_t)p->nz - 1);
    b->zx = (int64_t*)malloc(sizeof(int64_t) * (size_t)p->nx);
    b->zy = (int64_t*)malloc(sizeof(int64_t) * (size_t)p->ny);
    b->zz = (int64_t*)malloc(sizeof(int64_t) * (size_t)p->nz);
    if (!b->zx || !b->zy || !b->zz) { free(b->zx); free(b->zy); free(b->zz); b->zx = b->zy = b->zz = NULL; hqh_box_destroy(b); return HQ_ERR_NOMEM; }
    for (int32_t i = 0; i < p->nx; i++) b->zx[i] = (int64_t)bits_extract(zvalue((uint32_t)i, 0, 0), b->zmask);
    for (int32_t j = 0; j < p->ny; j++) b->zy[j] = (int64_t)bits_extract(zvalue(0, (uint32_t)j, 0), b->zmask);
    for (int32_t k = 0; k < p->nz; k++) b->zz[k] = (int64_t)bits_extract(zvalue(0, 0, (uint32_t)k), b->zmask);
    b->Eg = (int64_t)p->nx * p->ny * p->nz;
    b->Ng = (int64_t)(p->nx + 1) * (p->ny + 1) * (p->nz + 1);
    if (b->Eg < p->nranks) { hqh_box_destroy(b); return HQ_ERR_ARG; }
    b->elo = (int64_t)p->rank * b->Eg / p->nranks;                       /* BLOCK_LOW  */
    b->ehi = (int64_t)(p->rank + 1) * b->Eg / p->nranks;                 /* BLOCK_HIGH + 1 */
    if (b->ehi - b->elo > 0x7fffffff / 8) { hqh_box_destroy(b); return HQ_ERR_ARG; }
    b->lenum = (int32_t)(b->ehi - b->elo);

    b->ncls = p->lateral_classes > 1 ? p->lateral_classes : 1;
    if (b->ncls > 1 && !(p->lateral_amp >= 0.0 && p->lateral_amp < 1.0)) { hqh_box_destroy(b); return HQ_ERR_ARG; }
    size_t nz = (size_t)p->nz * (size_t)b->ncls;
    b->k_vp = (float*)malloc(nz * sizeof(float)); b->k_vs = (float*)malloc(nz * sizeof(float));
    b->k_rho = (float*)malloc(nz * sizeof(float));
    b->k_c1 = (double*)malloc(nz * 8); b->k_c2 = (double*)malloc(nz * 8); b->k_c3 = (double*)malloc(nz * 8);
    b->k_c4 = (double*)malloc(nz * 8); b->k_a = (double*)malloc(nz * 8); b->k_M = (double*)malloc(nz * 8);
    if (!b->k_vp || !b->k_vs || !b->k_rho || !b->k_c1 || !b->k_c2 || !b->k_c3 || !b->k_c4 || !b->k_a || !b->k_M) {
        hqh_box_destroy(b); return HQ_ERR_NOMEM;
    }
    if (depth_constants(b) != 0) { hqh_box_destroy(b); return HQ_ERR_ARG; }

    double t_lap = hqh_now();
    /* harbored nodes: corners of my elements, marked on the global node grid */
    b->loc = (int32_t*)malloc(sizeof(int32_t) * (size_t)b->Ng);
    if (!b->loc) { hqh_box_destroy(b); return HQ_ERR_NOMEM; }
    memset(b->loc, 0xff, sizeof(int32_t) * (size_t)b->Ng);
    int64_t nh = 0;
    if (p->nranks == 1) {
        nh = b->Ng;
    } else {
        uint64_t z = bits_deposit((uint64_t)b->elo, b->zmask);
        for (int64_t e = b->elo; e < b->ehi; e++) {
            int32_t i = (int32_t)compact3(z), j = (int32_t)compact3(z >> 1), k = (int32_t)compact3(z >> 2);
            for (int c = 0; c < 8; c++) {
                int64_t g = grid_index(b, i + (c & 1), j + ((c >> 1) & 1), k + ((c >> 2) & 1));
                if (b->loc[g] < 0) { b->loc[g] = 0; nh++; }
            }
            z = ((z | ~b->zmask) + 1) & b->zmask;          /* next in-domain Z-value */
        }
    }
    hqh_lap(&t_lap, "box: mark harbored nodes");
    if (nh > 0x7fffffff / 8) { hqh_box_destroy(b); return HQ_ERR_ARG; }
    b->nharbored = (int32_t)nh;
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)nh);
    if (!keys) { hqh_box_destroy(b); return HQ_ERR_NOMEM; }
    {
        int64_t t = 0;
        for (int32_t k = 0; k <= p->nz; k++)
            for (int32_t j = 0; j <= p->ny; j++)
                for (int32_t i = 0; i <= p->nx; i++)
                    if (p->nranks == 1 || b->loc[grid_index(b, i, j, k)] == 0) keys[t++] = node_key(b, i, j, k);
    }
    hqh_lap(&t_lap, "box: node keys");
    int kbits = 3 * ((ax > ay ? (ax > az ? ax : az) : (ay > az ? ay : az)) + 2);
    if (radix_sort_u64(keys, nh, kbits) != 0) { free(keys); hqh_box_destroy(b); return HQ_ERR_NOMEM; }
    hqh_lap(&t_lap, "box: sort");

    b->node_ijk = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)nh);
    b->node_xyz = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)nh);
    b->owner = (int32_t*)malloc(sizeof(int32_t) * (size_t)nh);
    b->ntable = (double*)malloc(sizeof(double) * 7 * (size_t)nh);
    b->lnid = (int32_t*)malloc(sizeof(int32_t) * 8 * (size_t)(b->lenum ? b->lenum : 1));
    b->etable = (double*)malloc(sizeof(double) * 4 * (size_t)(b->lenum ? b->lenum : 1));
    if (!b->node_ijk || !b->node_xyz || !b->owner || !b->ntable || !b->lnid || !b->etable) {
        free(keys); hqh_box_destroy(b); return HQ_ERR_NOMEM;
    }
    /* tick coordinates as octor would give them: root edge 2^30 ticks over the longest axis */
    int amax = ax > ay ? (ax > az ? ax : az) : (ay > az ? ay : az);
    int32_t edge_ticks = (int32_t)1 << (30 - amax);
    int32_t nown = 0;
    int* places = NULL;              /* solver_float = 4 on a partition: every rank's messenger list, for the order of the sums */
    if (p->solver_float == 4 && p->nranks > 1) {
        places = (int*)malloc(sizeof(int) * (size_t)p->nranks * (size_t)p->nranks);
        if (!places || box_list_places(b, places) != HQ_OK) { free(places); free(keys); hqh_box_destroy(b); return HQ_ERR_NOMEM; }
    }
#pragma omp parallel for schedule(static) reduction(+ : nown)
    for (int64_t n = 0; n < nh; n++) {
        uint32_t d[3] = { compact3(keys[n]), compact3(keys[n] >> 1), compact3(keys[n] >> 2) };
        int32_t lim[3] = { p->nx, p->ny, p->nz }, c[3];
        for (int q = 0; q < 3; q++) c[q] = (d[q] & 1) ? lim[q] : (int32_t)(d[q] >> 1);
        for (int q = 0; q < 3; q++) {
            b->node_ijk[3 * n + q] = c[q];
            b->node_xyz[3 * n + q] = c[q] * edge_ticks;
        }
        b->loc[grid_index(b, c[0], c[1], c[2])] = (int32_t)n;
        int32_t oi = c[0] < p->nx ? c[0] : p->nx - 1, oj = c[1] < p->ny ? c[1] : p->ny - 1,
                ok = c[2] < p->nz ? c[2] : p->nz - 1;
        b->owner[n] = rank_of_elem(b, elem_index(b, oi, oj, ok));
        if (b->owner[n] == p->rank) nown++;
        node_constants(b, c[0], c[1], c[2], b->owner[n], places, &b->ntable[7 * n]);
    }
    b->nowned = nown;
    free(keys);
    free(places);
    hqh_lap(&t_lap, "box: node tables");

    /* elements */
#pragma omp parallel
    {
        int tid = 0, nt = 1;
#ifdef _OPENMP
        tid = omp_get_thread_num();
        nt = omp_get_num_threads();
#endif
        int64_t lo = b->elo + (int64_t)b->lenum * tid / nt, hi = b->elo + (int64_t)b->lenum * (tid + 1) / nt;
        uint64_t z = bits_deposit((uint64_t)lo, b->zmask);
        for (int64_t e = lo; e < hi; e++) {
            int32_t i = (int32_t)compact3(z), j = (int32_t)compact3(z >> 1), k = (int32_t)compact3(z >> 2);
            int64_t le = e - b->elo;
            for (int c = 0; c < 8; c++)
                b->lnid[8 * le + c] = b->loc[grid_index(b, i + (c & 1), j + ((c >> 1) & 1), k + ((c >> 2) & 1))];
            double* ep = &b->etable[4 * le];
            const int64_t mi = mat_index(b, i, j, k);
            ep[0] = b->k_c1[mi]; ep[1] = b->k_c2[mi]; ep[2] = b->k_c3[mi]; ep[3] = b->k_c4[mi];
            z = ((z | ~b->zmask) + 1) & b->zmask;
        }
    }
    if (b->ncls > 1 || p->nlayers > 1) {
        double aBase;
        rayleigh_base(p->freq, p->damping, &aBase, &b->bbase);
        b->edata = (float*)malloc(sizeof(float) * 4 * (size_t)(b->lenum ? b->lenum : 1));
        if (!b->edata) { hqh_box_destroy(b); return HQ_ERR_NOMEM; }
        hqh_box_material(b, NULL);                       /* fills b->edata */
    }
    hqh_lap(&t_lap, "box: element tables");
    int rc = build_schedule(b);
    hqh_lap(&t_lap, "box: schedule");
    if (rc != HQ_OK) { hqh_box_destroy(b); return rc; }
    *out = b;
    return HQ_OK;
}

int hqh_ntable_to_float(const double* ntable, int64_t rows, float* out)
{
    if (!ntable || !out || rows < 0) return HQ_ERR_ARG;
    for (int64_t i = 0; i < 7 * rows; i++) out[i] = (float)ntable[i];
    return HQ_OK;
}

/* edata_t of this partition's elements as solver_init reads them (psolve.c:3372-3385): out[lenum][3] = Vp, Vs, rho */
int hqh_box_material(const hqh_box* b, float* out)
{
    if (!b || (!out && !b->edata)) return HQ_ERR_ARG;
    uint64_t z = bits_deposit((uint64_t)b->elo, b->zmask);
    for (int64_t e = b->elo; e < b->ehi; e++) {
        int32_t i = (int32_t)compact3(z), j = (int32_t)compact3(z >> 1), k = (int32_t)compact3(z >> 2);
        const int64_t mi = mat_index(b, i, j, k);
        if (out) {
            float* o = out + 3 * (e - b->elo);
            o[0] = b->k_vp[mi]; o[1] = b->k_vs[mi]; o[2] = b->k_rho[mi];
        } else {                                         /* the box's own edata_t rows */
            float* o = b->edata + 4 * (e - b->elo);
            o[0] = (float)b->p.h; o[1] = b->k_vp[mi]; o[2] = b->k_vs[mi]; o[3] = b->k_rho[mi];
        }
        z = ((z | ~b->zmask) + 1) & b->zmask;
    }
    return HQ_OK;
}

/* schedule_build, psolve.c:4704-4863 (anchored nodes only: uniform meshes have no hanging nodes) */
static int build_schedule(hqh_box* b)
{
    const hqh_box_params* p = &b->p;
    int P = p->nranks, me = p->rank;
    if (P == 1) return HQ_OK;
    int64_t* ccount = (int64_t*)calloc((size_t)P, sizeof(int64_t));
    int64_t* scount = (int64_t*)calloc((size_t)P, sizeof(int64_t));
    /* the order of the lists: schedule_build walks the nodes in local order and an owned node's share list (hqh_share_list)
     * and puts a NEW messenger at the HEAD of its list (psolve.c:4736-4745, 4776-4785) -- the reverse of the order of first
     * encounter.  schedule_senddata adds what arrives messenger by messenger in that order (:5035-5073): with it the
     * oracle's multi-rank runs are bit-identical to the reference's per-rank checkpoints (tests/test_oracle_golden.py) */
    int* cseq = (int*)malloc(sizeof(int) * (size_t)P);
    int* sseq = (int*)malloc(sizeof(int) * (size_t)P);
    int64_t* met = (int64_t*)malloc(sizeof(int64_t) * (size_t)P * (size_t)P);
    int ncseq = 0, nsseq = 0;
    if (!ccount || !scount || !cseq || !sseq || !met || box_met(b, me, met) != HQ_OK) {
        free(ccount); free(scount); free(cseq); free(sseq); free(met); return HQ_ERR_NOMEM;
    }
    for (int pass = 0; pass < 2; pass++) {
        int64_t *cfill = NULL, *sfill = NULL;
        if (pass == 1) {
            cfill = (int64_t*)calloc((size_t)P, sizeof(int64_t));
            sfill = (int64_t*)calloc((size_t)P, sizeof(int64_t));
            int64_t ct = 0, st = 0;
            for (int r = 0; r < P; r++) { ct += ccount[r]; st += scount[r]; b->nc += ccount[r] > 0; b->ns += scount[r] > 0; }
            b->cmap = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ct ? ct : 1));
            b->smap = (int32_t*)malloc(sizeof(int32_t) * (size_t)(st ? st : 1));
            b->mc = (hq_messenger*)calloc((size_t)(b->nc ? b->nc : 1), sizeof(hq_messenger));
            b->ms = (hq_messenger*)calloc((size_t)(b->ns ? b->ns : 1), sizeof(hq_messenger));
            if (!cfill || !sfill || !b->cmap || !b->smap || !b->mc || !b->ms) {
                free(cfill); free(sfill); free(ccount); free(scount); free(cseq); free(sseq); free(met); return HQ_ERR_NOMEM;
            }
            int64_t co = 0, so = 0;
            int ic = 0, is = 0;
            for (int q = ncseq - 1; q >= 0; q--) {
                const int r = cseq[q];
                b->mc[ic].procid = r; b->mc[ic].nodecount = (int32_t)ccount[r]; b->mc[ic].mapping = b->cmap + co; cfill[r] = co; co += ccount[r]; ic++;
            }
            for (int q = nsseq - 1; q >= 0; q--) {
                const int r = sseq[q];
                b->ms[is].procid = r; b->ms[is].nodecount = (int32_t)scount[r]; b->ms[is].mapping = b->smap + so; sfill[r] = so; so += scount[r]; is++;
            }
        }
        for (int32_t n = 0; n < b->nharbored; n++) {
            int32_t i = b->node_ijk[3 * n], j = b->node_ijk[3 * n + 1], k = b->node_ijk[3 * n + 2];
            if (b->owner[n] != me) {
                if (pass == 0) { if (!ccount[b->owner[n]]++) cseq[ncseq++] = b->owner[n]; b->shared_nodes++; }
                else b->cmap[cfill[b->owner[n]]++] = n;
                continue;
            }
            int sh[8], nsh = 0;
            for (int c = 0; c < 8; c++) {
                int32_t ei = i - 1 + (c & 1), ej = j - 1 + ((c >> 1) & 1), ek = k - 1 + ((c >> 2) & 1);
                if (ei < 0 || ej < 0 || ek < 0 || ei >= p->nx || ej >= p->ny || ek >= p->nz) continue;
                int r = rank_of_elem(b, elem_index(b, ei, ej, ek));
                if (r == me) continue;
                int seen = 0;
                for (int t = 0; t < nsh; t++) seen |= (sh[t] == r);
                if (!seen) sh[nsh++] = r;
            }
            if (pass == 0 && nsh) b->shared_nodes++;
            {                                                   /* the node's share list (hqh_share_list) */
                uint64_t bits = 0;
                for (int t = 0; t < nsh; t++) bits |= 1ull << sh[t];
                nsh = hqh_share_list(bits, bits, me, P, met + (size_t)me * P, sh);
            }
            for (int t = 0; t < nsh; t++) {
                if (pass == 0) { if (!scount[sh[t]]++) sseq[nsseq++] = sh[t]; }
                else b->smap[sfill[sh[t]]++] = n;
            }
        }
        free(cfill); free(sfill);
    }
    free(ccount); free(scount); free(cseq); free(sseq); free(met);
    return HQ_OK;
}

int hqh_box_get_info(const hqh_box* b, hqh_box_info* info)
{
    if (!b || !info) return HQ_ERR_ARG;
    info->total_elements = b->Eg; info->total_nodes = b->Ng;
    info->lenum = b->lenum; info->nharbored = b->nharbored; info->nowned = b->nowned;
    int nb = 0;
    for (int r = 0; r < b->p.nranks; r++) {
        int hit = 0;
        for (int i = 0; i < b->nc; i++) hit |= (b->mc[i].procid == r);
        for (int i = 0; i < b->ns; i++) hit |= (b->ms[i].procid == r);
        nb += hit;
    }
    info->nneighbors = nb;
    info->shared_nodes = b->shared_nodes;
    return HQ_OK;
}

int hqh_box_desc(const hqh_box* b, hq_desc* d)
{
    if (!b || !d) return HQ_ERR_ARG;
    memset(d, 0, sizeof *d);
    d->lenum = b->lenum; d->nharbored = b->nharbored; d->ldnnum = 0;
    d->lnid = b->lnid; d->node_xyz = b->node_xyz;
    d->eTable = b->etable; d->nTable = b->ntable;
    d->an_sched.c_count = b->nc; d->an_sched.first_c = b->mc;
    d->an_sched.s_count = b->ns; d->an_sched.first_s = b->ms;
    d->deltaT = b->p.deltaT; d->rank = b->p.rank; d->nranks = b->p.nranks;
    d->variant = HQ_VARIANT_AUTO;
    if (b->edata) {
        d->edata = b->edata; d->mat_bbase = b->bbase;
        d->mat_threshold_damping = b->p.threshold_damping; d->mat_threshold_vpvs = b->p.threshold_vpvs;
    }
    return HQ_OK;
}

const int32_t* hqh_box_lnid(const hqh_box* b) { return b ? b->lnid : NULL; }
const int32_t* hqh_box_node_ijk(const hqh_box* b) { return b ? b->node_ijk : NULL; }
const double* hqh_box_etable(const hqh_box* b) { return b ? b->etable : NULL; }
const double* hqh_box_ntable(const hqh_box* b) { return b ? b->ntable : NULL; }
const int32_t* hqh_box_owner(const hqh_box* b) { return b ? b->owner : NULL; }

/* ------------------------------------------------------------------------ */
/* source and stations                                                      */
/* ------------------------------------------------------------------------ */

static int locate(const hqh_box* b, double x, double y, double z, int32_t e[3], double off[3])
{
    double c[3] = { x, y, z };
    int32_t n[3] = { b->p.nx, b->p.ny, b->p.nz };
    for (int d = 0; d < 3; d++) {
        if (c[d] < 0 || c[d] > n[d] * b->p.h) return -1;
        e[d] = (int32_t)floor(c[d] / b->p.h);
        if (e[d] >= n[d]) e[d] = n[d] - 1;
        off[d] = c[d] - (e[d] + 0.5) * b->p.h;         /* offset from the element centre */
    }
    return 0;
}

static int local_element(const hqh_box* b, const int32_t e[3], int32_t lnid[8])
{
    int64_t idx = elem_index(b, e[0], e[1], e[2]);
    if (idx < b->elo || idx >= b->ehi) return 0;
    memcpy(lnid, &b->lnid[8 * (idx - b->elo)], sizeof(int32_t) * 8);
    return 1;
}

int hqh_point_source(const hqh_box* b, double x, double y, double z, double strike, double dip, double rake,
                     int32_t* nloaded, int32_t lnid[8], double pattern[24])
{
    if (!b || !nloaded || !lnid || !pattern) return HQ_ERR_ARG;
    int32_t e[3];
    double off[3];
    if (locate(b, x, y, z, e, off) != 0) return HQ_ERR_ARG;
    *nloaded = local_element(b, e, lnid) ? 8 : 0;
    double s = strike / 180.0 * HQH_PI, d = dip / 180.0 * HQH_PI, r = rake / 180.0 * HQH_PI;
    /* fault normal and slip vectors, moment tensor n t^T + t n^T (quakesource.c:445-458) */
    double nv[3] = { -sin(s) * sin(d), cos(s) * sin(d), -cos(d) };
    double tv[3] = { cos(r) * sin(HQH_PI / 2 - s) + sin(r) * sin(s) * cos(d),
                     cos(r) * sin(s) - sin(r) * cos(s) * cos(d), -sin(r) * sin(d) };
    double h = b->p.h, h3 = h * h * h;
    for (int n = 0; n < 8; n++) {
        double sg[3] = { (n & 1) ? 1.0 : -1.0, (n & 2) ? 1.0 : -1.0, (n & 4) ? 1.0 : -1.0 };
        double w[3] = { h + 2 * sg[0] * off[0], h + 2 * sg[1] * off[1], h + 2 * sg[2] * off[2] };
        /* gradient of the trilinear shape function of corner n at the source point */
        double g[3] = { (2 * sg[0]) * w[1] * w[2] / (8 * h3), (2 * sg[1]) * w[2] * w[0] / (8 * h3),
                        (2 * sg[2]) * w[0] * w[1] / (8 * h3) };
        for (int a = 0; a < 3; a++) {
            double f = 0.0;
            for (int c = 0; c < 3; c++) f += (nv[a] * tv[c] + nv[c] * tv[a]) * g[c];
            pattern[3 * n + a] = f;
        }
    }
    return HQ_OK;
}

int hqh_stations(const hqh_box* b, int32_t n, const double* xyz, int32_t* ids, double* phi, int32_t* mine)
{
    if (!b || n < 0 || (n && (!xyz || !ids || !phi || !mine))) return HQ_ERR_ARG;
    for (int32_t s = 0; s < n; s++) {
        int32_t e[3];
        double off[3];
        if (locate(b, xyz[3 * s], xyz[3 * s + 1], xyz[3 * s + 2], e, off) != 0) return HQ_ERR_ARG;
        mine[s] = local_element(b, e, &ids[8 * s]);
        double lc[3] = { 2 * off[0] / b->p.h, 2 * off[1] / b->p.h, 2 * off[2] / b->p.h };
        for (int c = 0; c < 8; c++)
            phi[8 * s + c] = (1 + ((c & 1) ? 1 : -1) * lc[0]) * (1 + ((c & 2) ? 1 : -1) * lc[1]) *
                             (1 + ((c & 4) ? 1 : -1) * lc[2]) / 8;
    }
    return HQ_OK;
}

/* ------------------------------------------------------------------------ */
/* solver_run                                                               */
/* ------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------ */
/* reference file formats                                                   */
/* ------------------------------------------------------------------------ */

int hqh_forcefile_info(const char* path, int32_t* nloaded, int32_t* nsteps, int32_t* lnid, int32_t lnid_cap)
{
    if (!path || !nloaded || !nsteps) return HQ_ERR_ARG;
    FILE* fp = fopen(path, "rb");
    if (!fp) return HQ_ERR_ARG;
    int32_t n = 0;
    int rc = HQ_OK;
    if (fread(&n, sizeof n, 1, fp) != 1 || n < 0) rc = HQ_ERR_ARG;
    if (rc == HQ_OK && lnid) {
        if (lnid_cap < n || (n && fread(lnid, sizeof(int32_t), (size_t)n, fp) != (size_t)n)) rc = HQ_ERR_ARG;
    }
    if (rc == HQ_OK) {
        fseeko(fp, 0, SEEK_END);
        off_t payload = ftello(fp) - (off_t)sizeof(int32_t) * (1 + (off_t)n);
        *nloaded = n;
        *nsteps = n ? (int32_t)(payload / ((off_t)n * 24)) : 0;
    }
    fclose(fp);
    return rc;
}

int hqh_forcefile_read(const char* path, int32_t step0, int32_t nsteps, double* F)
{
    if (!path || step0 < 0 || nsteps < 0 || (nsteps && !F)) return HQ_ERR_ARG;
    FILE* fp = fopen(path, "rb");
    if (!fp) return HQ_ERR_ARG;
    int32_t n = 0;
    if (fread(&n, sizeof n, 1, fp) != 1 || n < 0) { fclose(fp); return HQ_ERR_ARG; }
    /* read_myForces, psolve.c:3657-3664 */
    off_t where = (off_t)sizeof(int32_t) + (off_t)n * sizeof(int32_t) + (off_t)n * step0 * sizeof(double) * 3;
    size_t want = (size_t)n * 3 * (size_t)nsteps, got = 0;
    if (fseeko(fp, where, SEEK_SET) == 0) got = fread(F, sizeof(double), want, fp);
    for (size_t i = got; i < want; i++) F[i] = 0.0;
    fclose(fp);
    return HQ_OK;
}

int hqh_forcefile_write(const char* path, int32_t nloaded, const int32_t* lnid, int32_t nsteps, const double* F)
{
    if (!path || nloaded < 0 || nsteps < 0 || (nloaded && !lnid) || (nloaded && nsteps && !F)) return HQ_ERR_ARG;
    FILE* fp = fopen(path, "wb");
    if (!fp) return HQ_ERR_ARG;
    size_t ok = fwrite(&nloaded, sizeof nloaded, 1, fp);
    if (nloaded) ok += fwrite(lnid, sizeof(int32_t), (size_t)nloaded, fp) == (size_t)nloaded;
    size_t cnt = (size_t)nloaded * 3 * (size_t)nsteps;
    if (cnt) ok += fwrite(F, sizeof(double), cnt, fp) == cnt;
    int bad = fclose(fp) != 0 || ok != (size_t)(1 + (nloaded ? 1 : 0) + (cnt ? 1 : 0));
    return bad ? HQ_ERR_ARG : HQ_OK;
}

int hqh_checkpoint_write(hq_ctx* ctx, const char* path, int32_t step, int32_t rank, int32_t nranks,
                         int32_t nharbored, int32_t nharboredmax)
{
    if (!ctx || !path || rank < 0 || rank >= nranks || nharbored < 0 || nharbored > nharboredmax) return HQ_ERR_ARG;
    size_t n3 = (size_t)nharbored * 3;
    double* tm1 = (double*)malloc(sizeof(double) * (n3 ? n3 : 1));
    double* tm2 = (double*)malloc(sizeof(double) * (n3 ? n3 : 1));
    if (!tm1 || !tm2) { free(tm1); free(tm2); return HQ_ERR_NOMEM; }
    int rc = hq_download(ctx, tm1, tm2);
    FILE* fp = NULL;
    if (rc == HQ_OK && rank == 0) {                       /* io_checkpoint.c:63-74 */
        fp = fopen(path, "wb");
        int hdr[3] = { nranks, step, nharboredmax };
        if (!fp || fwrite(hdr, sizeof(int), 3, fp) != 3) rc = HQ_ERR_ARG;
        if (fp) fclose(fp);
    }
    if (rc == HQ_OK) {
        fp = fopen(path, "rb+");
        if (!fp) rc = HQ_ERR_ARG;
    }
    if (rc == HQ_OK) {                                    /* io_checkpoint.c:93-118: older field first */
        off_t off = (off_t)(3 * sizeof(int)) + (off_t)2 * rank * nharboredmax * 24;
        if (fseeko(fp, off, SEEK_SET) != 0 || fwrite(tm2, 24, (size_t)nharbored, fp) != (size_t)nharbored) rc = HQ_ERR_ARG;
        off += (off_t)nharbored * 24;
        if (rc == HQ_OK && (fseeko(fp, off, SEEK_SET) != 0 || fwrite(tm1, 24, (size_t)nharbored, fp) != (size_t)nharbored))
            rc = HQ_ERR_ARG;
        if (fclose(fp) != 0) rc = HQ_ERR_ARG;
    }
    free(tm1); free(tm2);
    return rc;
}

int hqh_checkpoint_read(hq_ctx* ctx, const char* path, int32_t rank, int32_t nranks, int32_t nharbored,
                        int32_t* step)
{
    if (!ctx || !path || rank < 0 || rank >= nranks || nharbored < 0) return HQ_ERR_ARG;
    FILE* fp = fopen(path, "rb");
    if (!fp) return HQ_ERR_ARG;
    int hdr[3];
    if (fread(hdr, sizeof(int), 3, fp) != 3 || hdr[0] != nranks || nharbored > hdr[2]) { fclose(fp); return HQ_ERR_ARG; }
    size_t n3 = (size_t)nharbored * 3;
    double* older = (double*)malloc(sizeof(double) * (n3 ? n3 : 1));
    double* newer = (double*)malloc(sizeof(double) * (n3 ? n3 : 1));
    int rc = (older && newer) ? HQ_OK : HQ_ERR_NOMEM;
    if (rc == HQ_OK) {                                    /* io_checkpoint.c:205-222 */
        off_t off = (off_t)(3 * sizeof(int)) + (off_t)2 * rank * hdr[2] * 24;
        if (fseeko(fp, off, SEEK_SET) != 0 || fread(older, 24, (size_t)nharbored, fp) != (size_t)nharbored) rc = HQ_ERR_ARG;
        off += (off_t)nharbored * 24;
        if (rc == HQ_OK && (fseeko(fp, off, SEEK_SET) != 0 || fread(newer, 24, (size_t)nharbored, fp) != (size_t)nharbored))
            rc = HQ_ERR_ARG;
    }
    fclose(fp);
    /* the reference loads the older field into tm1 and swaps at the top of the loop
     * (psolve.c:4271-4273); the engine takes the post-swap view directly */
    if (rc == HQ_OK) rc = hq_upload(ctx, newer, older, hdr[1]);
    if (rc == HQ_OK && step) *step = hdr[1];
    free(older); free(newer);
    return rc;
}

int hqh_station_format(char* buf, int32_t cap, double time, const double disp[3])
{
    if (!buf || cap < 64 || !disp) return HQ_ERR_ARG;
    snprintf(buf, (size_t)cap, "\n%10.6f % 8e % 8e % 8e", time, disp[0], disp[1], disp[2]);
    return HQ_OK;
}

int hqh_station_format_derivs(char* buf, int32_t cap, double time, const double* vals, int32_t derivs)
{
    if (!buf || cap < 64 + 48 * derivs || !vals || derivs < 0 || derivs > 2) return HQ_ERR_ARG;
    int n = snprintf(buf, (size_t)cap, "\n%10.6f % 8e % 8e % 8e", time, vals[0], vals[1], vals[2]);
    for (int k = 1; k <= derivs; k++)
        n += snprintf(buf + n, (size_t)cap - (size_t)n, " % 8e % 8e % 8e", vals[3 * k], vals[3 * k + 1], vals[3 * k + 2]);
    return HQ_OK;
}

/* out_hdr_t as gcc lays it out (psolve.h:120-186): char[29], 3 x int8, ufid[16] at 32, int64 at 48,
 * 2 x int32 at 56, 4 x int8 at 64, 5 doubles at 72, int64 at 112, 2 x int32 at 120, int64 at 128. */
#define HQH_OUT_HDR_BYTES 136

int hqh_wavefield_create(const char* path, const hqh_wavefield_info* w, int32_t quantity)
{
    if (!path || !w || (quantity != 1 && quantity != 2) || w->output_rate < 1) return HQ_ERR_ARG;
    unsigned char h[HQH_OUT_HDR_BYTES];
    memset(h, 0, sizeof h);
    snprintf((char*)h, 29, "Hercules 4D output v%03u", 0u);      /* HERCULES_4D_FORMAT_VERSION 0, output.c:104 */
    h[29] = 0;                                                   /* format_version */
    h[30] = (unsigned char)-1;                                   /* endiannes: never filled in, output.c:536 */
    h[31] = (unsigned char)-1;                                   /* platform_id */
    for (int i = 0; i < 4; i++) { int32_t r = (int32_t)random(); memcpy(h + 32 + 4 * i, &r, 4); }   /* generate_fileid */
    int32_t i32; int64_t i64;
    i64 = w->total_nodes; memcpy(h + 48, &i64, 8);
    i32 = (w->total_time_steps - 1) / w->output_rate + 1; memcpy(h + 56, &i32, 4);   /* get_output_time_step_count */
    i32 = 3; memcpy(h + 60, &i32, 4);
    h[64] = 8; h[65] = 2; h[66] = 1; h[67] = (unsigned char)quantity;   /* double, FLOAT64, FLOAT_CLASS */
    const double d[5] = { w->domain_x, w->domain_y, w->domain_z, w->mesh_ticksize, w->delta_t };
    memcpy(h + 72, d, 40);
    i64 = w->total_elements; memcpy(h + 112, &i64, 8);
    i32 = w->output_rate; memcpy(h + 120, &i32, 4);
    i32 = w->total_time_steps; memcpy(h + 124, &i32, 4);
    i64 = (int64_t)time(NULL); memcpy(h + 128, &i64, 8);
    FILE* fp = fopen(path, "w+");
    if (!fp) return HQ_ERR_ARG;
    int ok = fwrite(h, sizeof h, 1, fp) == 1;
    ok = (fclose(fp) == 0) && ok;
    return ok ? HQ_OK : HQ_ERR_ARG;
}

int hqh_wavefield_write(const char* path, int64_t total_nodes, int32_t quantity, int32_t out_step, int64_t base_gnid,
                        int32_t first_owned, int32_t count, const double* tm1, const double* tm2, double dt)
{
    if (!path || !tm1 || (quantity == 2 && !tm2) || (quantity != 1 && quantity != 2) || out_step < 0 || count < 0 ||
        base_gnid < 0 || base_gnid + count > total_nodes)
        return HQ_ERR_ARG;
    FILE* fp = fopen(path, "r+");
    if (!fp) return HQ_ERR_ARG;
    /* compute_current_offset, output.c:1224-1229: header + stride * step + 24 * first global id */
    const off_t off = (off_t)HQH_OUT_HDR_BYTES + (off_t)24 * total_nodes * out_step + (off_t)24 * base_gnid;
    int ok = fseeko(fp, off, SEEK_SET) == 0;
    if (ok && quantity == 1) {
        ok = fwrite(tm1 + 3 * (size_t)first_owned, 24, (size_t)count, fp) == (size_t)count;
    } else if (ok) {
        for (int32_t i = 0; i < count && ok; i++) {
            const double* a = tm1 + 3 * (size_t)(first_owned + i);
            const double* b = tm2 + 3 * (size_t)(first_owned + i);
            const double v[3] = { (a[0] - b[0]) / dt, (a[1] - b[1]) / dt, (a[2] - b[2]) / dt };
            ok = fwrite(v, 24, 1, fp) == 1;
        }
    }
    ok = (fclose(fp) == 0) && ok;
    return ok ? HQ_OK : HQ_ERR_ARG;
}

int hqh_station_header(char* buf, int32_t cap, int32_t derivs)
{
    if (!buf || cap < 160 || derivs < 0 || derivs > 2) return HQ_ERR_ARG;
    int n = snprintf(buf, (size_t)cap, "#  Time(s)         X|(m)         Y-(m)         Z.(m)");
    if (derivs >= 1) n += snprintf(buf + n, (size_t)cap - (size_t)n, "       X|(m/s)       Y-(m/s)       Z.(m/s)");
    if (derivs == 2) n += snprintf(buf + n, (size_t)cap - (size_t)n, "      X|(m/s2)      Y-(m/s2)      Z.(m/s2)");
    return HQ_OK;
}

/* interpolate_station_displacements, psolve.c:6705-6787: the displacement sum over the 8 nodes;
 * for the velocity the same accumulator has phi * tm2 taken off node by node and is divided by dt;
 * for the acceleration phi * tm2 comes off once more and phi * tm3 is added, over dt^2. */
int hqh_station_kinematics(const double* phi, const double* tm1, const double* tm2, const double* tm3,
                           double dt, int32_t derivs, double* vals)
{
    if (!phi || !tm1 || !vals || derivs < 0 || derivs > 2 || (derivs >= 1 && !tm2) || (derivs == 2 && !tm3))
        return HQ_ERR_ARG;
    double d[3] = { 0.0, 0.0, 0.0 };
    for (int c = 0; c < 8; c++)
        for (int a = 0; a < 3; a++) d[a] += phi[c] * tm1[3 * c + a];
    for (int a = 0; a < 3; a++) vals[a] = d[a];
    if (derivs >= 1) {
        for (int c = 0; c < 8; c++)
            for (int a = 0; a < 3; a++) d[a] -= phi[c] * tm2[3 * c + a];
        for (int a = 0; a < 3; a++) vals[3 + a] = d[a] / dt;
    }
    if (derivs == 2) {
        const double dt2 = dt * dt;                              /* Param.theDeltaTSquared, psolve.c:998 */
        for (int c = 0; c < 8; c++)
            for (int a = 0; a < 3; a++) { d[a] -= phi[c] * tm2[3 * c + a]; d[a] += phi[c] * tm3[3 * c + a]; }
        for (int a = 0; a < 3; a++) vals[6 + a] = d[a] / dt2;
    }
    return HQ_OK;
}

void hqh_source_table(const hqh_run_params* rp, double dt, int32_t step0, int32_t nsteps, double* F)
{
    for (int32_t s = 0; s < nsteps; s++) {
        double t = (step0 + s) * dt;
        double g = (rp->rise_time > 0 && t < rp->rise_time) ? 0.5 * (1.0 - cos(HQH_PI * t / rp->rise_time)) : 1.0;
        for (int32_t i = 0; i < rp->nloaded * 3; i++)
            F[(int64_t)s * rp->nloaded * 3 + i] = rp->moment * g * rp->pattern[i];
    }
}

/*
 * solver_run, psolve.c:4241-4324.  Per step the reference does: swap, outputs
 * (stations read tm1), read source forces, physics + communication.  Here the
 * force table is uploaded one window at a time and the steps between two
 * output steps are enqueued back to back.
 */
int hqh_solver_run(hq_ctx* ctx, const hqh_box* b, const hqh_run_params* rp, int32_t step0, int32_t nsteps)
{
    if (!b) return HQ_ERR_ARG;
    return hqh_solver_run_on(ctx, b->p.deltaT, b->nharbored, rp, step0, nsteps);
}

int hqh_solver_run_on(hq_ctx* ctx, double deltaT, int32_t nharb, const hqh_run_params* rp, int32_t step0, int32_t nsteps)
{
    if (!ctx || !rp || nsteps < 0 || deltaT <= 0 || nharb < 0) return HQ_ERR_ARG;
    int32_t win = rp->source_window > 0 ? rp->source_window : 256;
    double* F = NULL;
    double *u = NULL, *disp = NULL;
    if (rp->nloaded > 0) {
        F = (double*)malloc(sizeof(double) * 3 * (size_t)rp->nloaded * (size_t)win);
        if (!F) return HQ_ERR_NOMEM;
    }
    if (rp->checkpoint_rate > 0 && rp->checkpoint_dir != NULL) {
        /* checkpoint.out<N> holds one stripe per rank behind a common header (io_checkpoint.c:29-127): a
         * partition cannot write it alone with a groupsize-1 header.  Partitioned callers call
         * hqh_checkpoint_write themselves (rank, nranks, nharboredmax, a barrier after rank 0 created the file). */
        hq_info inf;
        if (hq_get_info(ctx, &inf) != HQ_OK) { free(F); return HQ_ERR_ARG; }
        if (inf.nranks > 1) { free(F); return HQ_ERR_STATE; }
    }
    if (rp->nstations > 0 && rp->station_rate > 0 && rp->station_fn) {
        if (rp->station_derivs < 0 || rp->station_derivs > 2) { free(F); return HQ_ERR_ARG; }
        u = (double*)malloc(sizeof(double) * 24 * 3 * (size_t)rp->nstations);          /* tm1 | tm2 | tm3 rows */
        disp = (double*)malloc(sizeof(double) * 3 * (size_t)(1 + rp->station_derivs) * (size_t)rp->nstations);
        if (!u || !disp) { free(F); free(u); free(disp); return HQ_ERR_NOMEM; }
    }
    /* output planes */
    double *pu = NULL, *pbuf = NULL;
    FILE** pfp = NULL;
    int64_t npp = 0;
    int rc = HQ_OK;
    if (rp->nplanes > 0 && rp->plane_rate > 0 && rp->plane_dir) {
        for (int32_t i = 0; i < rp->nplanes; i++) npp += rp->plane_npoints[i];
        pu = (double*)malloc(sizeof(double) * 24 * (size_t)(npp ? npp : 1));
        pbuf = (double*)calloc((size_t)(npp ? npp : 1) * 3, sizeof(double));
        pfp = (FILE**)calloc((size_t)rp->nplanes, sizeof(FILE*));
        if (!pu || !pbuf || !pfp) rc = HQ_ERR_NOMEM;
        for (int32_t i = 0; i < rp->nplanes && rc == HQ_OK; i++) {
            char path[1200];
            snprintf(path, sizeof path, "%s/planedisplacements.%d", rp->plane_dir, i);
            pfp[i] = fopen(path, step0 > 0 ? "ab" : "wb");     /* a restart continues the file */
            if (!pfp[i]) rc = HQ_ERR_ARG;
        }
    }
    /* 4D wavefield files: whole-field download at the (rare) output steps */
    const int do_wave = rp->wavefield_rate > 0 && (rp->wavefield_disp_file || rp->wavefield_vel_file);
    double *w1 = NULL, *w2 = NULL;
    if (do_wave && rc == HQ_OK) {
        w1 = (double*)malloc(sizeof(double) * 3 * (size_t)(nharb ? nharb : 1));
        w2 = (double*)malloc(sizeof(double) * 3 * (size_t)(nharb ? nharb : 1));
        if (!w1 || !w2) rc = HQ_ERR_NOMEM;
    }
    int32_t step = step0, end = step0 + nsteps, win_end = step0;
    int ckpt_number = 0;                                         /* CheckpointNumber, io_checkpoint.c:38,126 */
    const int do_ckpt = rp->checkpoint_rate > 0 && rp->checkpoint_dir != NULL;
    while (step < end && rc == HQ_OK) {
        if (do_ckpt && step != step0 && step % rp->checkpoint_rate == 0) {   /* solver_write_checkpoint, :4277 */
            char path[1200];
            snprintf(path, sizeof path, "%s/checkpoint.out%d", rp->checkpoint_dir, ckpt_number);
            rc = hqh_checkpoint_write(ctx, path, step, 0, 1, nharb, nharb);
            if (rc != HQ_OK) break;
            ckpt_number = (ckpt_number + 1) % 2;
        }
        if (do_wave && step % rp->wavefield_rate == 0) {         /* solver_output_wavefield, :4278 */
            rc = hq_download(ctx, w1, rp->wavefield_vel_file ? w2 : NULL);
            const int32_t cnt = rp->wavefield_count > 0 ? rp->wavefield_count : nharb;
            const int32_t first = rp->wavefield_count > 0 ? rp->wavefield_first_owned : 0;
            const int64_t gbase = rp->wavefield_count > 0 ? rp->wavefield_base_gnid : 0;
            const int64_t total = rp->wavefield_total_nodes > 0 ? rp->wavefield_total_nodes : nharb;
            if (rc == HQ_OK && rp->wavefield_disp_file)
                rc = hqh_wavefield_write(rp->wavefield_disp_file, total, 1, step / rp->wavefield_rate, gbase, first,
                                         cnt, w1, NULL, deltaT);
            if (rc == HQ_OK && rp->wavefield_vel_file)
                rc = hqh_wavefield_write(rp->wavefield_vel_file, total, 2, step / rp->wavefield_rate, gbase, first,
                                         cnt, w1, w2, deltaT);
            if (rc != HQ_OK) break;
        }
        if (pfp && step % rp->plane_rate == 0) {                 /* solver_output_planes, :4279 */
            rc = hq_gather(ctx, (int32_t)(npp * 8), rp->plane_ids, pu, NULL);
            if (rc != HQ_OK) break;
            int64_t off = 0;
            for (int32_t i = 0; i < rp->nplanes && rc == HQ_OK; i++) {
                for (int64_t s = off; s < off + rp->plane_npoints[i]; s++) {
                    if (rp->plane_mine && !rp->plane_mine[s]) continue;
                    for (int d = 0; d < 3; d++) {                /* Old_planes_print, io_planes.c:176-200 */
                        double acc = 0.0;
                        for (int c = 0; c < 8; c++) acc += rp->plane_phi[8 * s + c] * pu[(8 * s + c) * 3 + d];
                        pbuf[3 * s + d] = acc;
                    }
                }
                size_t n = 3 * (size_t)rp->plane_npoints[i];
                if (fwrite(pbuf + 3 * off, sizeof(double), n, pfp[i]) != n) rc = HQ_ERR_ARG;
                off += rp->plane_npoints[i];
            }
            if (rc != HQ_OK) break;
        }
        if (u && step % rp->station_rate == 0) {                 /* solver_output_stations, :4280 */
            const int dv = rp->station_derivs;
            const size_t blk = 24 * (size_t)rp->nstations;
            rc = dv == 2 ? hq_gather3(ctx, rp->nstations * 8, rp->station_ids, u, u + blk, u + 2 * blk)
                         : hq_gather(ctx, rp->nstations * 8, rp->station_ids, u, dv ? u + blk : NULL);
            if (rc != HQ_OK) break;
            for (int32_t s = 0; s < rp->nstations; s++)
                hqh_station_kinematics(rp->station_phi + 8 * s, u + 24 * s, u + blk + 24 * s, u + 2 * blk + 24 * s,
                                       deltaT, dv, disp + 3 * (size_t)(1 + dv) * s);
            rp->station_fn(rp->station_user, step, rp->nstations, disp);
        }
        if (F && step >= win_end) {                              /* solver_read_source_forces, :4282 */
            int32_t n = end - step < win ? end - step : win;
            if (rp->force_file) {
                rc = hqh_forcefile_read(rp->force_file, step, n, F);
                if (rc != HQ_OK) break;
            } else {
                hqh_source_table(rp, deltaT, step, n, F);
            }
            rc = hq_set_source(ctx, rp->nloaded, rp->loaded_lnid, step, n, F);
            if (rc != HQ_OK) break;
            win_end = step + n;
        }
        int32_t next = end;
        if (F && win_end < next) next = win_end;
        if (u) {
            int32_t ns = (step / rp->station_rate + 1) * rp->station_rate;
            if (ns < next) next = ns;
        }
        if (pfp) {
            int32_t ns = (step / rp->plane_rate + 1) * rp->plane_rate;
            if (ns < next) next = ns;
        }
        if (do_ckpt) {
            int32_t ns = (step / rp->checkpoint_rate + 1) * rp->checkpoint_rate;
            if (ns < next) next = ns;
        }
        if (do_wave) {
            int32_t ns = (step / rp->wavefield_rate + 1) * rp->wavefield_rate;
            if (ns < next) next = ns;
        }
        rc = hq_run(ctx, next - step);
        step = next;
    }
    if (rc == HQ_OK) rc = hq_sync(ctx);
    if (pfp) for (int32_t i = 0; i < rp->nplanes; i++) if (pfp[i]) fclose(pfp[i]);
    free(F); free(u); free(disp); free(pu); free(pbuf); free(pfp); free(w1); free(w2);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* output planes: geometry                                                  */
/* ------------------------------------------------------------------------ */


/* compute_global_coords (geometrics.c:33-70) with rake = 0, over the grid of
 * Old_output_planes_construct_strips (io_planes.c:489-520) */
int hqh_plane_points(const hqh_plane* pl, double* xyz)
{
    if (!pl || !xyz || pl->n_strike < 0 || pl->n_dip < 0) return HQ_ERR_ARG;
    const double d = pl->dip_deg * HQH_PI / 180, l = 0.0 * HQH_PI / 180, p = pl->strike_deg * HQH_PI / 180;
    for (int32_t i = 0; i < pl->n_strike; i++)
        for (int32_t j = 0; j < pl->n_dip; j++) {
            const double x = i * pl->step_strike, y = j * pl->step_dip, z = 0;
            double* o = xyz + 3 * ((int64_t)i * pl->n_dip + j);
            o[0] = (cos(p) * cos(l) + sin(p) * cos(d) * sin(l)) * x - (-cos(p) * sin(l) + sin(p) * cos(d) * cos(l)) * y -
                   (-sin(p) * sin(d)) * z + pl->origin[0];
            o[1] = (sin(p) * cos(l) - cos(p) * cos(d) * sin(l)) * x - (-sin(p) * sin(l) - cos(p) * cos(d) * cos(l)) * y -
                   (cos(p) * sin(d)) * z + pl->origin[1];
            o[2] = -sin(d) * sin(l) * x + sin(d) * cos(l) * y + cos(d) * z + pl->origin[2];
        }
    return HQ_OK;
}

/* compute_domain_coords_linearinterp (geometrics.c:178-244): Newton iteration on the bilinear
 * map of the four corners; csi follows latitude and scales to len_x, eta longitude / len_y */
int hqh_domain_coords(double lon, double lat, const double lon_corners[4], const double lat_corners[4],
                      double len_x, double len_y, double* x, double* y)
{
    if (!lon_corners || !lat_corners || !x || !y) return HQ_ERR_ARG;
    const double X = lat, Y = lon;
    const double *Xi = lat_corners, *Yi = lon_corners;
    const double Ax = 4 * X - (Xi[0] + Xi[1] + Xi[2] + Xi[3]), Ay = 4 * Y - (Yi[0] + Yi[1] + Yi[2] + Yi[3]);
    const double Bx = -Xi[0] + Xi[1] + Xi[2] - Xi[3], By = -Yi[0] + Yi[1] + Yi[2] - Yi[3];
    const double Cx = -Xi[0] - Xi[1] + Xi[2] + Xi[3], Cy = -Yi[0] - Yi[1] + Yi[2] + Yi[3];
    const double Dx = Xi[0] - Xi[1] + Xi[2] - Xi[3], Dy = Yi[0] - Yi[1] + Yi[2] - Yi[3];
    double xn0 = 0, xn1 = 0, res = 1e10;
    for (int it = 0; res > 1e-6; it++) {
        if (it > 100) return HQ_ERR_ARG;                     /* corners do not span a quadrilateral */
        const double m00 = Bx + Dx * xn1, m01 = Cx + Dx * xn0, m10 = By + Dy * xn1, m11 = Cy + Dy * xn0;
        const double f0 = -Ax + Bx * xn0 + Cx * xn1 + Dx * xn0 * xn1;
        const double f1 = -Ay + By * xn0 + Cy * xn1 + Dy * xn0 * xn1;
        const double det = m00 * m11 - m10 * m01;
        const double d0 = -(f0 * m11 - f1 * m01) / det, d1 = -(f1 * m00 - f0 * m10) / det;
        res = fabs(f0) + fabs(f1);
        xn0 += d0; xn1 += d1;
    }
    *x = .5 * (xn0 + 1) * len_x;
    *y = .5 * (xn1 + 1) * len_y;
    return HQ_OK;
}

/* ------------------------------------------------------------------------ */
/* two-level layered box (hanging nodes)                                    */
/* ------------------------------------------------------------------------ */

#define HQH_MAXLEVELS 8

struct hqh_octbox {
    hqh_octlevels_params p;         /* (the two-level parameters are translated into these) */
    int32_t layers[HQH_MAXLEVELS];  /* element layers per level */
    int32_t lay0[HQH_MAXLEVELS + 1];/* index of every level's first element layer; [nlevels] = total */
    float *vp, *vs, *rho;           /* [total element layers], from the top */
    int32_t ztop[HQH_MAXLEVELS + 1];/* top plane of every level's slab, finest-edge units; [nlevels] = bottom */
    int32_t far_q[3];               /* domain extent in finest-edge units */
    int64_t E, N;
    int32_t ldnnum;
    int32_t *lnid, *node_xyz, *dn_id, *dn_ptr, *dn_anchor;
    double *etable, *ntable;
    /* partitions only */
    int32_t *owner, *gid;
    hq_messenger *mc[2], *ms[2];    /* [0] anchored-node schedule, [1] dangling-node schedule */
    int32_t nc[2], ns[2];
    int32_t *cmap[2], *smap[2];
    float* edata;                   /* [E][4] as solver_init leaves edata_t (hqh_mesh_from_leaves); NULL: not kept */
    double bbase, thr_damp, thr_vpvs;
};

void hqh_octbox_destroy(hqh_octbox* b)
{
    if (!b) return;
    free(b->lnid); free(b->node_xyz); free(b->dn_id); free(b->dn_ptr); free(b->dn_anchor);
    free(b->etable); free(b->ntable); free(b->owner); free(b->gid);
    free(b->vp); free(b->vs); free(b->rho);
    for (int s = 0; s < 2; s++) { free(b->mc[s]); free(b->ms[s]); free(b->cmap[s]); free(b->smap[s]); }
    free(b->edata);
    free(b);
}

/* level of the slab that contains plane z < bottom (finest units) */
static int octbox_level_at(const hqh_octbox* b, int32_t z)
{
    int L = 0;
    while (L + 1 < b->p.nlevels && z >= b->ztop[L + 1]) L++;
    return L;
}

/* Cut rank `me`'s part out of the whole box in `b` (see hq_host.h); `ek` = the sorted Z-values
 * of the element corners (element id = position). */
static int octbox_cut(hqh_octbox* b, const uint64_t* ek, int me, int P, nt_parts* parts)
{
    const int32_t nx = b->far_q[0], ny = b->far_q[1], nzt = b->far_q[2], NL = b->p.nlevels;
    const int64_t E = b->E, N = b->N;
    int rc = HQ_ERR_NOMEM;
    /* slab boxes: leaf of every cell of every level's slab -> element id; meshes from leaves
     * (NL == 0): the leaf containing a point is the last one whose corner precedes it in Z-order */
    int32_t* cell[HQH_MAXLEVELS];
    for (int L = 0; L < HQH_MAXLEVELS; L++) cell[L] = NULL;
    uint64_t* harb = (uint64_t*)calloc((size_t)N, sizeof(uint64_t));
    int32_t* gowner = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int32_t* g2l = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    uint8_t* hang = (uint8_t*)calloc((size_t)N, 1);
    uint64_t* vharb = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(N ? N : 1));   /* the ranks that have the node as an element vertex */
    int64_t* met = (int64_t*)malloc(sizeof(int64_t) * (size_t)P * (size_t)P);       /* when rank o met rank q (hqh_share_list) */
    int32_t *lnid = NULL, *xyz = NULL, *own = NULL, *gid = NULL, *dn_id = NULL, *dn_ptr = NULL, *dn_anchor = NULL;
    double *et = NULL, *nt = NULL;
    if (!harb || !gowner || !g2l || !hang || !vharb || !met) goto done;
    for (int L = 0; L < NL; L++) {
        int64_t n = (int64_t)(nx >> L) * (ny >> L) * b->layers[L];
        cell[L] = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
        if (!cell[L]) goto done;
    }
#define HQH_ERANK(e) ((int)((((int64_t)(e) + 1) * P - 1) / E))     /* octor.c:4939-4944 */
#define HQH_CELL(L, i, j, k) cell[L][((int64_t)(((k) - b->ztop[L]) >> (L)) * (ny >> (L)) + ((j) >> (L))) * (nx >> (L)) + ((i) >> (L))]
    for (int64_t e = 0; e < E; e++) {
        if (NL > 0) {
            int32_t i = (int32_t)compact3(ek[e]), j = (int32_t)compact3(ek[e] >> 1), k = (int32_t)compact3(ek[e] >> 2);
            HQH_CELL(octbox_level_at(b, k), i, j, k) = (int32_t)e;
        }
        const uint64_t bit = 1ull << HQH_ERANK(e);
        for (int c = 0; c < 8; c++) harb[b->lnid[8 * e + c]] |= bit;             /* element vertices */
    }
    memcpy(vharb, harb, sizeof(uint64_t) * (size_t)N);
    /* the leaf that holds the cell (x, y, z), finest-edge units */
#define HQH_LEAF_AT(x_, y_, z_, e_)                                                                                \
    if (NL > 0) e_ = HQH_CELL(octbox_level_at(b, (int32_t)(z_)), (int32_t)(x_), (int32_t)(y_), (int32_t)(z_));    \
    else {                                                                                                         \
        const uint64_t key_ = zvalue((uint32_t)(x_), (uint32_t)(y_), (uint32_t)(z_));                              \
        int64_t lo_ = 0, hi_ = E - 1;                                                                              \
        while (lo_ < hi_) { int64_t m_ = (lo_ + hi_ + 1) / 2; if (ek[m_] <= key_) lo_ = m_; else hi_ = m_ - 1; }   \
        e_ = lo_;                                                                                                  \
    }
    for (int64_t n = 0; n < N; n++) {                                            /* owners */
        const int32_t* c = &b->node_xyz[3 * n];
        int32_t ax = c[0] < nx ? c[0] : nx - 1, ay = c[1] < ny ? c[1] : ny - 1, az = c[2] < nzt ? c[2] : nzt - 1;
        int64_t e;
        HQH_LEAF_AT(ax, ay, az, e)
        gowner[n] = HQH_ERANK(e);
        harb[n] |= 1ull << gowner[n];
    }
    /* when every rank met its neighbours (com_allocpctl's scan, see hqh_share_list): only a leaf with a vertex that another
     * rank's leaf has too can see another rank's pixel -- on a 2:1 mesh every leaf at an interface has such a vertex */
    for (int q = 0; q < P * P; q++) met[q] = -1;
    for (int64_t e = 0; e < E; e++) {
        const int o = HQH_ERANK(e);
        if (!parts && o != me) continue;                                          /* (all ranks' lists: the float rows only) */
        int edge = 0;
        for (int c = 0; c < 8; c++) edge |= (vharb[b->lnid[8 * e + c]] & ~(1ull << o)) != 0;
        if (!edge) continue;
        const int32_t* c0 = &b->node_xyz[3 * (int64_t)b->lnid[8 * e]];
        const int64_t sz = b->node_xyz[3 * (int64_t)b->lnid[8 * e + 7]] - c0[0];
        for (int kk = 0; kk < 4; kk++) {
            const int64_t z = hqh_probe_cell(c0[2], sz, kk, nzt);
            if (z < 0) continue;
            for (int jj = 0; jj < 4; jj++) {
                const int64_t y = hqh_probe_cell(c0[1], sz, jj, ny);
                if (y < 0) continue;
                for (int ii = 0; ii < 4; ii++) {
                    const int64_t x = hqh_probe_cell(c0[0], sz, ii, nx);
                    if (x < 0) continue;
                    int64_t e2;
                    HQH_LEAF_AT(x, y, z, e2)
                    const int q = HQH_ERANK(e2);
                    if (q != o && met[o * P + q] < 0) met[o * P + q] = 64 * e + (kk * 4 + jj) * 4 + ii;
                }
            }
        }
    }
#undef HQH_LEAF_AT
#undef HQH_CELL
    for (int32_t k = 0; k < b->ldnnum; k++) {                                    /* indirect sharing */
        hang[b->dn_id[k]] = 1;
        const uint64_t bit = 1ull << gowner[b->dn_id[k]];
        for (int32_t a = b->dn_ptr[k]; a < b->dn_ptr[k + 1]; a++) harb[b->dn_anchor[a]] |= bit;
    }
    if (parts) {                     /* solver_float = 4: the whole mesh's rows in the N-rank build's order (nt_parts_combine) */
        int64_t* first = (int64_t*)malloc(sizeof(int64_t) * 2 * (size_t)P * (size_t)P);
        int* pos = (int*)malloc(sizeof(int) * 2 * (size_t)P * (size_t)P);
        int prc = HQ_ERR_NOMEM;
        if (first && pos) {
            for (int q = 0; q < 2 * P * P; q++) first[q] = -1;
            int64_t seen = 0;
            for (int64_t n = 0; n < N; n++) {                                     /* every rank's schedule_build at once */
                const int o = gowner[n], sdn = hang[n];
                int sl[64];
                const int ns = hqh_share_list(harb[n], vharb[n], o, P, met + o * P, sl);
                for (int t = 0; t < ns; t++)
                    if (first[((int64_t)sdn * P + o) * P + sl[t]] < 0) first[((int64_t)sdn * P + o) * P + sl[t]] = seen++;
            }
            nt_list_places(P, first, pos);
            nt_list_places(P, first + (int64_t)P * P, pos + P * P);
            prc = nt_parts_combine(parts, N, b->ntable, gowner, b->ldnnum, b->dn_id, b->dn_ptr, b->dn_anchor, pos, pos + P * P);
        }
        free(first); free(pos);
        if (prc != HQ_OK) { rc = prc; goto done; }
    }
    {
        const uint64_t mebit = 1ull << me;
        int64_t nh = 0;
        for (int64_t n = 0; n < N; n++) g2l[n] = (harb[n] & mebit) ? (int32_t)nh++ : -1;
        const int64_t elo = (int64_t)me * E / P, ehi = (int64_t)(me + 1) * E / P, ne = ehi - elo;
        int32_t ndn = 0, nan = 0;
        for (int32_t k = 0; k < b->ldnnum; k++)
            if (gowner[b->dn_id[k]] == me) { ndn++; nan += b->dn_ptr[k + 1] - b->dn_ptr[k]; }
        lnid = (int32_t*)malloc(sizeof(int32_t) * 8 * (size_t)(ne ? ne : 1));
        et = (double*)malloc(sizeof(double) * 4 * (size_t)(ne ? ne : 1));
        xyz = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(nh ? nh : 1));
        own = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nh ? nh : 1));
        gid = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nh ? nh : 1));
        nt = (double*)malloc(sizeof(double) * 7 * (size_t)(nh ? nh : 1));
        dn_id = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ndn ? ndn : 1));
        dn_ptr = (int32_t*)malloc(sizeof(int32_t) * ((size_t)ndn + 1));
        dn_anchor = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nan ? nan : 1));
        if (!lnid || !et || !xyz || !own || !gid || !nt || !dn_id || !dn_ptr || !dn_anchor) goto done;
        for (int64_t e = elo; e < ehi; e++) {
            for (int c = 0; c < 8; c++) lnid[8 * (e - elo) + c] = g2l[b->lnid[8 * e + c]];
            memcpy(et + 4 * (e - elo), b->etable + 4 * e, 4 * sizeof(double));
        }
        for (int64_t n = 0; n < N; n++) {
            const int32_t l = g2l[n];
            if (l < 0) continue;
            memcpy(xyz + 3 * (int64_t)l, b->node_xyz + 3 * n, 3 * sizeof(int32_t));
            memcpy(nt + 7 * (int64_t)l, b->ntable + 7 * n, 7 * sizeof(double));
            own[l] = gowner[n];
            gid[l] = (int32_t)n;
        }
        {   /* dnodeTable: the hanging nodes this rank owns, in (local = global) node order */
            int32_t kd = 0, na = 0;
            dn_ptr[0] = 0;
            for (int32_t k = 0; k < b->ldnnum; k++) {
                if (gowner[b->dn_id[k]] != me) continue;
                dn_id[kd] = g2l[b->dn_id[k]];
                for (int32_t a = b->dn_ptr[k]; a < b->dn_ptr[k + 1]; a++) dn_anchor[na++] = g2l[b->dn_anchor[a]];
                dn_ptr[++kd] = na;
            }
        }
        /* schedule_build (psolve.c:4704-4863): a node someone else owns goes on the c-list of its
         * owner, a node I own on the s-list of every other rank that harbors it; hanging nodes in
         * dn_sched, the rest in an_sched */
        for (int s = 0; s < 2; s++) {
            int64_t ccount[64], scount[64], cfill[64], sfill[64];
            int cseq[64], sseq[64], ncseq = 0, nsseq = 0;     /* ranks in the order of first encounter */
            memset(ccount, 0, sizeof ccount); memset(scount, 0, sizeof scount);
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 1) {
                    int64_t ct = 0, st = 0;
                    for (int r = 0; r < P; r++) { ct += ccount[r]; st += scount[r]; b->nc[s] += ccount[r] > 0; b->ns[s] += scount[r] > 0; }
                    b->cmap[s] = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ct ? ct : 1));
                    b->smap[s] = (int32_t*)malloc(sizeof(int32_t) * (size_t)(st ? st : 1));
                    b->mc[s] = (hq_messenger*)calloc((size_t)(b->nc[s] ? b->nc[s] : 1), sizeof(hq_messenger));
                    b->ms[s] = (hq_messenger*)calloc((size_t)(b->ns[s] ? b->ns[s] : 1), sizeof(hq_messenger));
                    if (!b->cmap[s] || !b->smap[s] || !b->mc[s] || !b->ms[s]) goto done;
                    int64_t co = 0, so = 0;
                    int ic = 0, is = 0;
                    for (int q = ncseq - 1; q >= 0; q--) {   /* a new messenger goes to the HEAD of its list (build_schedule) */
                        const int r = cseq[q];
                        b->mc[s][ic].procid = r; b->mc[s][ic].nodecount = (int32_t)ccount[r]; b->mc[s][ic].mapping = b->cmap[s] + co; cfill[r] = co; co += ccount[r]; ic++;
                    }
                    for (int q = nsseq - 1; q >= 0; q--) {
                        const int r = sseq[q];
                        b->ms[s][is].procid = r; b->ms[s][is].nodecount = (int32_t)scount[r]; b->ms[s][is].mapping = b->smap[s] + so; sfill[r] = so; so += scount[r]; is++;
                    }
                }
                for (int64_t l = 0; l < nh; l++) {
                    const int64_t g = gid[l];
                    if ((int)hang[g] != s) continue;
                    if (own[l] != me) {
                        if (pass == 0) { if (!ccount[own[l]]++) cseq[ncseq++] = own[l]; } else b->cmap[s][cfill[own[l]]++] = (int32_t)l;
                        continue;
                    }
                    int sl[64];                                  /* the node's share list, in its order */
                    const int nsl = hqh_share_list(harb[g], vharb[g], me, P, met + (size_t)me * P, sl);
                    for (int t = 0; t < nsl; t++) {
                        const int r = sl[t];
                        if (pass == 0) { if (!scount[r]++) sseq[nsseq++] = r; } else b->smap[s][sfill[r]++] = (int32_t)l;
                    }
                }
            }
        }
        free(b->lnid); free(b->node_xyz); free(b->etable); free(b->ntable);
        free(b->dn_id); free(b->dn_ptr); free(b->dn_anchor);
        b->lnid = lnid; b->node_xyz = xyz; b->etable = et; b->ntable = nt;
        b->dn_id = dn_id; b->dn_ptr = dn_ptr; b->dn_anchor = dn_anchor;
        b->owner = own; b->gid = gid;
        lnid = xyz = own = gid = dn_id = dn_ptr = dn_anchor = NULL; et = nt = NULL;
        if (b->edata) {                                      /* this rank's rows: its block of the element list */
            int64_t e_first = 0;
            while (e_first < E && HQH_ERANK(e_first) != me) e_first++;
            memmove(b->edata, b->edata + 4 * e_first, sizeof(float) * 4 * (size_t)ne);
        }
        b->E = ne; b->N = nh; b->ldnnum = ndn;
        rc = HQ_OK;
    }
#undef HQH_ERANK
done:
    for (int L = 0; L < HQH_MAXLEVELS; L++) free(cell[L]);
    free(harb); free(gowner); free(g2l); free(hang); free(vharb); free(met);
    free(lnid); free(xyz); free(own); free(gid); free(dn_id); free(dn_ptr); free(dn_anchor); free(et); free(nt);
    return rc;
}

/* Lysmer dashpot of one element corner from the element's six face bits
 * (compute_setflag + theIDBoundaryMatrix + compute_setboundary, psolve.c:5629-5804) */
static int face_dashpot(int face, int corner, int halfspace, float size, float Vp, float Vs, float rho, double out[3])
{
    int bits = 0;
    out[0] = out[1] = out[2] = 0.0;
    if (!face) return 0;
    for (int d = 0; d < 3; d++) {
        int near = (face >> d) & 1, far = (face >> (3 + d)) & 1;
        int cls = far ? 2 : (near ? 0 : 1);
        if (d == 2 && halfspace && cls == 0) cls = 1;
        int cfar = (corner >> d) & 1;
        if ((cls == 0 && !cfar) || (cls == 2 && cfar)) bits |= 1 << d;
    }
    double scale = rho * (size / 2) * (size / 2);
    int nf = (bits & 1) + ((bits >> 1) & 1) + ((bits >> 2) & 1);
    for (int d = 0; d < 3; d++) {
        if (nf == 3) out[d] = (Vp + 2 * Vs) * scale;
        else if (nf == 2) out[d] = (Vs + ((bits & (1 << d)) ? Vp : Vs)) * scale;
        else if (nf == 1) out[d] = ((bits & (1 << d)) ? Vp : Vs) * scale;
    }
    return 1;
}


/* ------------------------------------------------------------------------ */
/* per-rank construction of a layered octree box (no whole-box arrays)       */
/* ------------------------------------------------------------------------ */
/*
 * hqh_octbox_create_levels builds the WHOLE box and cuts a rank's part out (octbox_cut): 30 GB and a minute per rank
 * at 189 M elements.  octbox_local builds the same per-rank tables from the sorted leaf keys alone (the only whole-box
 * array, 8 B per element): everything a node's row needs is found by locating the leaves around it --
 *   leaf of a cell            level of its plane, origin = the cell rounded to that level's edge, global element id =
 *                             position of its key among the sorted leaf keys (octree pre-order = Z-order);
 *   vertices of a leaf        its eight corners; a node is a vertex of the <= 8 leaves around it whose corner it is;
 *   owner                     rank of the leaf that contains the (far-boundary adjusted) node (octor.c:5466-5475);
 *   harbored by               the ranks of the leaves it is a vertex of, its owner, and -- an anchor -- the owners of
 *                             the hanging nodes on it (indirect sharing, octor.c:5516-6040);
 *   n_t row                   the elements' contributions in global element order (psolve.c:3436-3471), then the
 *                             hanging nodes' mass parts in node order (compute_adjust on nTable, psolve.c:3502).
 * Equal table for table to octbox_cut's output (tests/test_host_partition.py) except gid (the global node index is
 * not computed: -1).
 */
typedef struct { int32_t o[3]; int32_t L; int64_t e; } oct_leaf_t;

#define OCT_BUCKET_BITS 20
typedef struct {
    const hqh_octbox* b;
    const uint64_t* ek;
    const int64_t* bucket;          /* [2^20 + 1] first leaf whose key's top 20 of 36 bits reach the bucket's */
    int64_t E;
    int P;
    int32_t far[3];
    const double (*lc)[4];
    const double *la, *lM;
    const float *lvp, *lh;
} oct_ctx_t;

static int oct_leaf_of_cell(const oct_ctx_t* C, int32_t px, int32_t py, int32_t pz, oct_leaf_t* out)
{
    if (px < 0 || py < 0 || pz < 0 || px >= C->far[0] || py >= C->far[1] || pz >= C->far[2]) return 0;
    const int L = octbox_level_at(C->b, pz), s = 1 << L;
    out->L = L;
    out->o[0] = px & ~(s - 1); out->o[1] = py & ~(s - 1); out->o[2] = pz & ~(s - 1);
    const uint64_t key = zvalue((uint32_t)out->o[0], (uint32_t)out->o[1], (uint32_t)out->o[2]);
    const int64_t bk = (int64_t)(key >> (36 - OCT_BUCKET_BITS));
    int64_t lo = C->bucket[bk], hi = C->bucket[bk + 1] - 1;
    if (hi < lo) return 0;
    while (lo < hi) { int64_t m = (lo + hi) / 2; if (C->ek[m] < key) lo = m + 1; else hi = m; }
    if (C->ek[lo] != key) return 0;
    out->e = lo;
    return 1;
}

#define OCT_ERANK(C, e) ((int)((((int64_t)(e) + 1) * (C)->P - 1) / (C)->E))      /* octor.c:4939-4944 */

/* the leaves node c is a vertex of, in global element order; corner[k] = which corner of leaf k it is */
static int oct_leaves_of_node(const oct_ctx_t* C, const int32_t c[3], oct_leaf_t leaf[8], int corner[8])
{
    int n = 0;
    for (int o = 0; o < 8; o++) {
        oct_leaf_t lf;
        if (!oct_leaf_of_cell(C, c[0] - (o & 1), c[1] - ((o >> 1) & 1), c[2] - ((o >> 2) & 1), &lf)) continue;
        const int s = 1 << lf.L;
        int cb = 0, ok = 1;
        for (int d = 0; d < 3; d++) {
            if (c[d] == lf.o[d] + s) cb |= 1 << d;
            else if (c[d] != lf.o[d]) ok = 0;
        }
        if (!ok) continue;
        int dup = 0;
        for (int k = 0; k < n; k++) dup |= (leaf[k].e == lf.e);
        if (dup) continue;
        int pos = n++;
        while (pos > 0 && leaf[pos - 1].e > lf.e) { leaf[pos] = leaf[pos - 1]; corner[pos] = corner[pos - 1]; pos--; }
        leaf[pos] = lf; corner[pos] = cb;
    }
    return n;
}

/* does node c hang?  -> its level (>= 1) and its anchors in octor's list order, or 0 */
static int oct_node_hangs(const oct_ctx_t* C, const int32_t c[3], int32_t anchor[4][3], int* deps)
{
    const hqh_octbox* b = C->b;
    const int Lh = octbox_level_at(b, c[2] < C->far[2] ? c[2] : C->far[2] - 1);
    if (Lh < 1 || c[2] != b->ztop[Lh]) return 0;
    const int32_t m = (1 << Lh) - 1, h = 1 << (Lh - 1);
    if (!((c[0] & m) || (c[1] & m))) return 0;
    int na = 0;
#define OCT_A(dx, dy) { anchor[na][0] = c[0] + (dx); anchor[na][1] = c[1] + (dy); anchor[na][2] = c[2]; na++; }
    if ((c[0] & m) && (c[1] & m)) { OCT_A(h, h) OCT_A(-h, h) OCT_A(h, -h) OCT_A(-h, -h) }     /* ZFACE */
    else if (c[0] & m) { OCT_A(h, 0) OCT_A(-h, 0) }                                          /* XEDGE */
    else { OCT_A(0, h) OCT_A(0, -h) }                                                         /* YEDGE */
#undef OCT_A
    *deps = na;
    return Lh;
}

static int oct_owner(const oct_ctx_t* C, const int32_t c[3])
{
    oct_leaf_t lf;
    const int32_t a[3] = { c[0] < C->far[0] ? c[0] : C->far[0] - 1, c[1] < C->far[1] ? c[1] : C->far[1] - 1,
                           c[2] < C->far[2] ? c[2] : C->far[2] - 1 };
    if (!oct_leaf_of_cell(C, a[0], a[1], a[2], &lf)) return -1;
    return OCT_ERANK(C, lf.e);
}

/* the element contributions to node c's n_t row (psolve.c:3436-3471), elements in global order; -> harbor bits of
 * the elements' ranks */
static uint64_t oct_node_row(const oct_ctx_t* C, const int32_t c[3], double np[7])
{
    const hqh_octbox* b = C->b;
    oct_leaf_t leaf[8];
    int corner[8];
    const int n = oct_leaves_of_node(C, c, leaf, corner);
    const double dt = b->p.deltaT;
    uint64_t bits = 0;
    for (int t = 0; t < 7; t++) np[t] = 0.0;
    for (int k = 0; k < n; k++) {
        const int L = leaf[k].L, s = 1 << L;
        const int32_t q = b->lay0[L] + ((leaf[k].o[2] - b->ztop[L]) >> L);
        const int face = (leaf[k].o[0] == 0) | ((leaf[k].o[1] == 0) << 1) | ((leaf[k].o[2] == 0) << 2) |
                         ((leaf[k].o[0] + s == C->far[0]) << 3) | ((leaf[k].o[1] + s == C->far[1]) << 4) |
                         ((leaf[k].o[2] + s == C->far[2]) << 5);
        double dash[3];
        const int bnd = face_dashpot(face, corner[k], b->p.halfspace, C->lh[q], C->lvp[q], b->vs[q], b->rho[q], dash);
        const double M = C->lM[q], a = C->la[q];
        nt_accumulate(np, b->p.solver_float == 4, dt, a, M, bnd, dash);
        bits |= 1ull << OCT_ERANK(C, leaf[k].e);
    }
    return bits;
}

static uint64_t oct_node_key(const oct_ctx_t* C, const int32_t c[3])
{
    uint32_t d[3];
    for (int q = 0; q < 3; q++) d[q] = (c[q] == C->far[q]) ? (uint32_t)(2 * c[q] - 1) : (uint32_t)(2 * c[q]);
    return zvalue(d[0], d[1], d[2]);
}

static void oct_key_node(const oct_ctx_t* C, uint64_t key, int32_t c[3])
{
    const uint32_t d[3] = { compact3(key), compact3(key >> 1), compact3(key >> 2) };
    for (int q = 0; q < 3; q++) c[q] = (d[q] & 1) ? C->far[q] : (int32_t)(d[q] >> 1);
}

static int64_t oct_find_key(const uint64_t* keys, int64_t n, uint64_t key)
{
    int64_t lo = 0, hi = n - 1;
    while (lo < hi) { int64_t m = (lo + hi) / 2; if (keys[m] < key) lo = m + 1; else hi = m; }
    return (n > 0 && keys[lo] == key) ? lo : -1;
}

static int octbox_local(hqh_octbox* b, const uint64_t* ek, int me, int P, const double (*lc)[4], const double* la,
                        const double* lM, const float* lvp, const float* lh)
{
    oct_ctx_t C;
    const int f32 = b->p.solver_float == 4;          /* (one rank's order: the float build on partitions never comes here) */
    C.b = b; C.ek = ek; C.E = b->E; C.P = P;
    C.far[0] = b->far_q[0]; C.far[1] = b->far_q[1]; C.far[2] = b->far_q[2];
    C.lc = lc; C.la = la; C.lM = lM; C.lvp = lvp; C.lh = lh;
    int64_t* bucket = (int64_t*)malloc(sizeof(int64_t) * (((size_t)1 << OCT_BUCKET_BITS) + 1));
    if (!bucket) return HQ_ERR_NOMEM;
    {
        const int64_t nbk = (int64_t)1 << OCT_BUCKET_BITS;
        int64_t e = 0;
        for (int64_t k = 0; k <= nbk; k++) {
            while (e < b->E && (int64_t)(ek[e] >> (36 - OCT_BUCKET_BITS)) < k) e++;
            bucket[k] = e;
        }
    }
    C.bucket = bucket;
    const int64_t E = b->E, elo = (int64_t)me * E / P, ehi = (int64_t)(me + 1) * E / P, ne = ehi - elo;
    int rc = HQ_ERR_NOMEM;
    uint64_t* keys = NULL;
    int32_t *lnid = NULL, *xyz = NULL, *own = NULL, *gid = NULL, *dn_id = NULL, *dn_ptr = NULL, *dn_anchor = NULL;
    double *et = NULL, *nt = NULL;
    uint64_t *harb = NULL, *vh = NULL;           /* vh: the ranks that have the node as an element vertex */
    uint8_t* hang = NULL;
    int64_t met[64];                             /* when this rank met the others (hqh_share_list) */
    /* 1. candidates: the vertices of my elements; the hanging nodes on top of my elements (their containing leaf is
     *    mine when I own them) and their anchors */
    int64_t cap = 8 * ne + 64, nk = 8 * ne;
    keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)cap);
    if (!keys) goto done;
#pragma omp parallel for schedule(static)
    for (int64_t e = elo; e < ehi; e++) {
        const int32_t i = (int32_t)compact3(ek[e]), j = (int32_t)compact3(ek[e] >> 1), k = (int32_t)compact3(ek[e] >> 2);
        const int s = 1 << octbox_level_at(b, k);
        for (int c = 0; c < 8; c++) {
            const int32_t p[3] = { i + s * (c & 1), j + s * ((c >> 1) & 1), k + s * ((c >> 2) & 1) };
            keys[8 * (e - elo) + c] = oct_node_key(&C, p);
        }
    }
    for (int64_t e = elo; e < ehi; e++) {                     /* (the top layers of the coarse slabs only) */
        const int32_t k = (int32_t)compact3(ek[e] >> 2);
        const int L = octbox_level_at(b, k), s = 1 << L;
        if (L < 1 || k != b->ztop[L]) continue;
        const int32_t i = (int32_t)compact3(ek[e]), j = (int32_t)compact3(ek[e] >> 1), h = s >> 1;
        for (int dy = 0; dy <= 2; dy++)
            for (int dx = 0; dx <= 2; dx++) {
                const int32_t p[3] = { i + dx * h, j + dy * h, k };
                int32_t an[4][3];
                int deps;
                if (!oct_node_hangs(&C, p, an, &deps) || oct_owner(&C, p) != me) continue;
                if (nk + 5 > cap) {
                    cap += cap / 8 + 64;
                    uint64_t* nw = (uint64_t*)realloc(keys, sizeof(uint64_t) * (size_t)cap);
                    if (!nw) goto done;
                    keys = nw;
                }
                keys[nk++] = oct_node_key(&C, p);
                for (int a = 0; a < deps; a++) keys[nk++] = oct_node_key(&C, an[a]);
            }
    }
    if (radix_sort_u64(keys, nk, 36) != 0) goto done;
    int64_t nh = 0;
    for (int64_t t = 0; t < nk; t++) if (t == 0 || keys[t] != keys[t - 1]) keys[nh++] = keys[t];
    {
        uint64_t* nw = (uint64_t*)realloc(keys, sizeof(uint64_t) * (size_t)(nh ? nh : 1));     /* 8 candidates per node before */
        if (nw) keys = nw;
    }
    /* 2. every harbored node's row */
    xyz = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(nh ? nh : 1));
    own = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nh ? nh : 1));
    gid = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nh ? nh : 1));
    nt = (double*)malloc(sizeof(double) * 7 * (size_t)(nh ? nh : 1));
    harb = (uint64_t*)calloc((size_t)(nh ? nh : 1), sizeof(uint64_t));
    vh = (uint64_t*)calloc((size_t)(nh ? nh : 1), sizeof(uint64_t));
    hang = (uint8_t*)calloc((size_t)(nh ? nh : 1), 1);
    lnid = (int32_t*)malloc(sizeof(int32_t) * 8 * (size_t)(ne ? ne : 1));
    et = (double*)malloc(sizeof(double) * 4 * (size_t)(ne ? ne : 1));
    if (!xyz || !own || !gid || !nt || !harb || !vh || !hang || !lnid || !et) goto done;
    int fault = 0;
    int64_t ndn = 0, nan = 0;
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : ndn, nan) reduction(| : fault)
    for (int64_t l = 0; l < nh; l++) {
        int32_t c[3];
        oct_key_node(&C, keys[l], c);
        memcpy(xyz + 3 * l, c, sizeof c);
        gid[l] = -1;
        double* np = nt + 7 * l;
        uint64_t bits = oct_node_row(&C, c, np);
        const int o = oct_owner(&C, c);
        if (o < 0 || !bits) { fault = 1; continue; }
        own[l] = o;
        vh[l] = bits;
        bits |= 1ull << o;
        int32_t an[4][3];
        int deps = 0;
        if (oct_node_hangs(&C, c, an, &deps)) {
            hang[l] = 1;
            if (o == me) { ndn++; nan += deps; }
        } else {
            /* an anchor?  the hanging nodes around it on its plane, in node order: their mass parts and their owners */
            const int Lp = octbox_level_at(b, c[2] < C.far[2] ? c[2] : C.far[2] - 1);
            if (Lp >= 1 && c[2] == b->ztop[Lp]) {
                const int32_t h = 1 << (Lp - 1);
                uint64_t hk[8];
                int32_t hc[8][3];
                int nhn = 0;
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dx && !dy) continue;
                        const int32_t p[3] = { c[0] + dx * h, c[1] + dy * h, c[2] };
                        if (p[0] < 0 || p[1] < 0 || p[0] > C.far[0] || p[1] > C.far[1]) continue;
                        int32_t pa[4][3];
                        int pd;
                        if (!oct_node_hangs(&C, p, pa, &pd)) continue;
                        int mine = 0;
                        for (int a = 0; a < pd; a++) mine |= (pa[a][0] == c[0] && pa[a][1] == c[1]);
                        if (!mine) continue;
                        const uint64_t key = oct_node_key(&C, p);
                        int pos = nhn++;
                        while (pos > 0 && hk[pos - 1] > key) { hk[pos] = hk[pos - 1]; memcpy(hc[pos], hc[pos - 1], sizeof hc[0]); pos--; }
                        hk[pos] = key; memcpy(hc[pos], p, sizeof hc[0]);
                    }
                for (int k = 0; k < nhn; k++) {
                    double hp[7];
                    int32_t pa[4][3];
                    int pd;
                    oct_node_hangs(&C, hc[k], pa, &pd);
                    const int ho = oct_owner(&C, hc[k]);
                    if (ho >= 0) bits |= 1ull << ho;                     /* indirect sharing */
                    oct_node_row(&C, hc[k], hp);
                    for (int q = 0; q < 7; q++) {
                        const double part = HQH_SF(f32, hp[q] / (uint32_t)pd);
                        np[q] = HQH_SF(f32, np[q] + part);
                    }
                }
            }
        }
        harb[l] = bits;
        if (!(bits & (1ull << me))) fault = 1;
    }
    if (fault) { rc = HQ_ERR_STATE; goto done; }
    /* 3. my elements */
#pragma omp parallel for schedule(static) reduction(| : fault)
    for (int64_t e = elo; e < ehi; e++) {
        const int32_t i = (int32_t)compact3(ek[e]), j = (int32_t)compact3(ek[e] >> 1), k = (int32_t)compact3(ek[e] >> 2);
        const int L = octbox_level_at(b, k), s = 1 << L;
        const int32_t q = b->lay0[L] + ((k - b->ztop[L]) >> L);
        for (int c4 = 0; c4 < 4; c4++) et[4 * (e - elo) + c4] = lc[q][c4];
        for (int c = 0; c < 8; c++) {
            const int32_t p[3] = { i + s * (c & 1), j + s * ((c >> 1) & 1), k + s * ((c >> 2) & 1) };
            const int64_t l = oct_find_key(keys, nh, oct_node_key(&C, p));
            if (l < 0) fault = 1;
            lnid[8 * (e - elo) + c] = (int32_t)l;
        }
    }
    if (fault) { rc = HQ_ERR_STATE; goto done; }
    /* when I met my neighbours (com_allocpctl's scan over my leaves, see hqh_share_list) */
    for (int q = 0; q < 64; q++) met[q] = -1;
    for (int64_t e = elo; e < ehi; e++) {
        int edge = 0;
        for (int c = 0; c < 8; c++) edge |= (vh[lnid[8 * (e - elo) + c]] & ~(1ull << me)) != 0;
        if (!edge) continue;
        const int32_t i = (int32_t)compact3(ek[e]), j = (int32_t)compact3(ek[e] >> 1), k = (int32_t)compact3(ek[e] >> 2);
        const int64_t sz = 1 << octbox_level_at(b, k);
        for (int kk = 0; kk < 4; kk++) {
            const int64_t z = hqh_probe_cell(k, sz, kk, C.far[2]);
            if (z < 0) continue;
            for (int jj = 0; jj < 4; jj++) {
                const int64_t y = hqh_probe_cell(j, sz, jj, C.far[1]);
                if (y < 0) continue;
                for (int ii = 0; ii < 4; ii++) {
                    const int64_t x = hqh_probe_cell(i, sz, ii, C.far[0]);
                    if (x < 0) continue;
                    oct_leaf_t lf;
                    if (!oct_leaf_of_cell(&C, (int32_t)x, (int32_t)y, (int32_t)z, &lf)) { fault = 1; continue; }
                    const int q = OCT_ERANK(&C, lf.e);
                    if (q != me && met[q] < 0) met[q] = 64 * e + (kk * 4 + jj) * 4 + ii;
                }
            }
        }
    }
    if (fault) { rc = HQ_ERR_STATE; goto done; }
    /* 4. dnodeTable of the hanging nodes I own, node order, anchors in octor's list order */
    dn_id = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ndn ? ndn : 1));
    dn_ptr = (int32_t*)malloc(sizeof(int32_t) * ((size_t)ndn + 1));
    dn_anchor = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nan ? nan : 1));
    if (!dn_id || !dn_ptr || !dn_anchor) goto done;
    {
        int32_t kd = 0, na = 0;
        dn_ptr[0] = 0;
        for (int64_t l = 0; l < nh; l++) {
            if (!hang[l] || own[l] != me) continue;
            int32_t an[4][3];
            int deps;
            oct_node_hangs(&C, xyz + 3 * l, an, &deps);
            dn_id[kd] = (int32_t)l;
            for (int a = 0; a < deps; a++) {
                const int64_t al = oct_find_key(keys, nh, oct_node_key(&C, an[a]));
                if (al < 0) { rc = HQ_ERR_STATE; goto done; }
                dn_anchor[na++] = (int32_t)al;
            }
            dn_ptr[++kd] = na;
        }
    }
    /* 5. schedule_build (psolve.c:4704-4863), as octbox_cut */
    {
        for (int s = 0; s < 2; s++) {
            int64_t ccount[64], scount[64], cfill[64], sfill[64];
            int cseq[64], sseq[64], ncseq = 0, nsseq = 0;     /* ranks in the order of first encounter */
            memset(ccount, 0, sizeof ccount); memset(scount, 0, sizeof scount);
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 1) {
                    int64_t ct = 0, st = 0;
                    for (int r = 0; r < P; r++) { ct += ccount[r]; st += scount[r]; b->nc[s] += ccount[r] > 0; b->ns[s] += scount[r] > 0; }
                    b->cmap[s] = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ct ? ct : 1));
                    b->smap[s] = (int32_t*)malloc(sizeof(int32_t) * (size_t)(st ? st : 1));
                    b->mc[s] = (hq_messenger*)calloc((size_t)(b->nc[s] ? b->nc[s] : 1), sizeof(hq_messenger));
                    b->ms[s] = (hq_messenger*)calloc((size_t)(b->ns[s] ? b->ns[s] : 1), sizeof(hq_messenger));
                    if (!b->cmap[s] || !b->smap[s] || !b->mc[s] || !b->ms[s]) goto done;
                    int64_t co = 0, so = 0;
                    int ic = 0, is = 0;
                    for (int q = ncseq - 1; q >= 0; q--) {   /* a new messenger goes to the HEAD of its list (build_schedule) */
                        const int r = cseq[q];
                        b->mc[s][ic].procid = r; b->mc[s][ic].nodecount = (int32_t)ccount[r]; b->mc[s][ic].mapping = b->cmap[s] + co; cfill[r] = co; co += ccount[r]; ic++;
                    }
                    for (int q = nsseq - 1; q >= 0; q--) {
                        const int r = sseq[q];
                        b->ms[s][is].procid = r; b->ms[s][is].nodecount = (int32_t)scount[r]; b->ms[s][is].mapping = b->smap[s] + so; sfill[r] = so; so += scount[r]; is++;
                    }
                }
                for (int64_t l = 0; l < nh; l++) {
                    if ((int)hang[l] != s) continue;
                    if (own[l] != me) {
                        if (pass == 0) { if (!ccount[own[l]]++) cseq[ncseq++] = own[l]; } else b->cmap[s][cfill[own[l]]++] = (int32_t)l;
                        continue;
                    }
                    int sl[64];                                  /* the node's share list, in its order */
                    const int nsl = hqh_share_list(harb[l], vh[l], me, P, met, sl);
                    for (int t = 0; t < nsl; t++) {
                        const int r = sl[t];
                        if (pass == 0) { if (!scount[r]++) sseq[nsseq++] = r; } else b->smap[s][sfill[r]++] = (int32_t)l;
                    }
                }
            }
        }
    }
    b->lnid = lnid; b->node_xyz = xyz; b->etable = et; b->ntable = nt;
    b->dn_id = dn_id; b->dn_ptr = dn_ptr; b->dn_anchor = dn_anchor;
    b->owner = own; b->gid = gid;
    lnid = xyz = own = gid = dn_id = dn_ptr = dn_anchor = NULL; et = nt = NULL;
    b->E = ne; b->N = nh; b->ldnnum = (int32_t)ndn;
    rc = HQ_OK;
done:
    free(keys); free(harb); free(vh); free(hang); free(bucket);
    free(lnid); free(xyz); free(own); free(gid); free(dn_id); free(dn_ptr); free(dn_anchor); free(et); free(nt);
    return rc;
}
#undef OCT_ERANK

/* element constants per element layer (mu_and_lambda psolve.c:3236-3272 + psolve.c:3387-3409, 3436-3437), as in
 * hqh_octbox_create_levels; the arrays are the caller's to free */
static int octbox_layer_constants(const hqh_octbox* b, double (**plc)[4], double** pla, double** plM, float** plvp, float** plh)
{
    const hqh_octlevels_params* p = &b->p;
    const int NL = p->nlevels;
    const int32_t nlay = b->lay0[NL];
    double aBase, bBase, dt = p->deltaT, dt2 = dt * dt;
    rayleigh_base(p->freq, p->damping, &aBase, &bBase);
    double (*lc)[4] = (double (*)[4])malloc(sizeof(double) * 4 * (size_t)nlay);
    double* la = (double*)malloc(sizeof(double) * (size_t)nlay);
    double* lM = (double*)malloc(sizeof(double) * (size_t)nlay);
    float* lvp = (float*)malloc(sizeof(float) * (size_t)nlay);
    float* lh = (float*)malloc(sizeof(float) * (size_t)nlay);
    int rc = HQ_OK;
    if (!lc || !la || !lM || !lvp || !lh) rc = HQ_ERR_NOMEM;
    for (int L = 0; L < NL && rc == HQ_OK; L++)
    for (int32_t q = b->lay0[L]; q < b->lay0[L + 1]; q++) {
        float Vp = b->vp[q], Vs = b->vs[q], rho = b->rho[q], h = (float)(p->h * (1 << L));
        lh[q] = h;
        double mu = rho * Vs * Vs, lambda;
        if (Vp > (Vs * p->threshold_vpvs)) lambda = rho * Vs * Vs * p->threshold_vpvs * p->threshold_vpvs - 2 * mu;
        else lambda = rho * Vp * Vp - 2 * mu;
        if (lambda < 0) {
            if (Vs < 500) Vp = 2.45 * Vs; else if (Vs < 1200) Vp = 2 * Vs; else Vp = 1.87 * Vs;
            lambda = rho * Vp * Vp;
        }
        if (lambda < 0) { rc = HQ_ERR_ARG; break; }
        lvp[q] = Vp;
        double zeta = 10 / Vs;
        if (zeta > p->threshold_damping) zeta = p->threshold_damping;
        double a = zeta * aBase, bb = zeta * bBase;
        lc[q][0] = dt2 * h * mu / 9; lc[q][1] = dt2 * h * lambda / 9;
        lc[q][2] = bb * dt * h * mu / 9; lc[q][3] = bb * dt * h * lambda / 9;
        la[q] = a;
        double mass = rho * h * h * h;
        lM[q] = mass / 8;
    }
    if (rc != HQ_OK) { free(lc); free(la); free(lM); free(lvp); free(lh); return rc; }
    *plc = lc; *pla = la; *plM = lM; *plvp = lvp; *plh = lh;
    return HQ_OK;
}

int hqh_octbox_create_levels(const hqh_octlevels_params* p, hqh_octbox** out)
{
    if (!p || !out) return HQ_ERR_ARG;
    *out = NULL;
    const int NL = p->nlevels;
    if (NL < 1 || NL > HQH_MAXLEVELS || !p->layers || !p->vp || !p->vs || !p->rho) return HQ_ERR_ARG;
    const int32_t nx = p->nx, ny = p->ny;
    if (nx < 1 || ny < 1 || (nx & ((1 << (NL - 1)) - 1)) || (ny & ((1 << (NL - 1)) - 1))) return HQ_ERR_ARG;
    if (p->h <= 0 || p->deltaT <= 0 || !hqh_sf_valid(p->solver_float)) return HQ_ERR_ARG;
    const int P = p->nranks > 1 ? p->nranks : 1;
    if (P > 64 || p->rank < 0 || p->rank >= P) return HQ_ERR_ARG;
    hqh_octbox* b = (hqh_octbox*)calloc(1, sizeof *b);
    if (!b) return HQ_ERR_NOMEM;
    b->p = *p;
    int64_t E = 0, N = 0;
    b->ztop[0] = 0; b->lay0[0] = 0;
    for (int L = 0; L < NL; L++) {
        b->layers[L] = p->layers[L];
        /* every slab at least one layer thick and aligned to the next level's cells (an octree) */
        if (p->layers[L] < 1) { free(b); return HQ_ERR_ARG; }
        b->lay0[L + 1] = b->lay0[L] + p->layers[L];
        b->ztop[L + 1] = b->ztop[L] + (p->layers[L] << L);
        if (L + 1 < NL && (b->ztop[L + 1] & ((2 << L) - 1))) { free(b); return HQ_ERR_ARG; }
        E += (int64_t)(nx >> L) * (ny >> L) * p->layers[L];
        /* node planes of the slab except its top one (which belongs to the finer slab above) */
        N += (int64_t)((nx >> L) + 1) * ((ny >> L) + 1) * p->layers[L];
    }
    const int32_t nzt = b->ztop[NL], nlay = b->lay0[NL];
    b->far_q[0] = nx; b->far_q[1] = ny; b->far_q[2] = nzt;
    N += (int64_t)(nx + 1) * (ny + 1);                           /* the free surface */
    b->vp = (float*)malloc(sizeof(float) * (size_t)nlay);
    b->vs = (float*)malloc(sizeof(float) * (size_t)nlay);
    b->rho = (float*)malloc(sizeof(float) * (size_t)nlay);
    if (!b->vp || !b->vs || !b->rho) { hqh_octbox_destroy(b); return HQ_ERR_NOMEM; }
    memcpy(b->vp, p->vp, sizeof(float) * (size_t)nlay);
    memcpy(b->vs, p->vs, sizeof(float) * (size_t)nlay);
    memcpy(b->rho, p->rho, sizeof(float) * (size_t)nlay);
    b->p.layers = b->layers; b->p.vp = b->vp; b->p.vs = b->vs; b->p.rho = b->rho;
    if (nx > 2047 || ny > 2047 || nzt > 2047) { hqh_octbox_destroy(b); return HQ_ERR_ARG; }
    if (E > 0x7fffffff / 8 || N > 0x7fffffff / 8) { hqh_octbox_destroy(b); return HQ_ERR_ARG; }
    b->E = E; b->N = N;
    {
        /* partitions of a large box: this rank's tables alone (octbox_local); HQH_OCTBOX_LOCAL = 1 / 0 forces / forbids.
         * Not with solver_float = 4: the N-rank float build adds the sharers' rows in the order of EVERY owner's messenger
         * list (nt_rank_rows), and a rank that sees its own neighbourhood only cannot know where another owner first met
         * whom -- those partitions are cut out of the whole box */
        const char* le = getenv("HQH_OCTBOX_LOCAL");
        if (P > 1 && p->solver_float != 4 && (le ? atoi(le) != 0 : E >= 4000000)) {
            uint64_t* lek = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)E);
            if (!lek) { hqh_octbox_destroy(b); return HQ_ERR_NOMEM; }
            int64_t te = 0;
            for (int L = 0; L < NL; L++) {
                const int32_t s = 1 << L;
                const int64_t per = (int64_t)(nx >> L) * (ny >> L);
#pragma omp parallel for schedule(static)
                for (int32_t kk = 0; kk < b->layers[L]; kk++) {
                    const int32_t k = b->ztop[L] + kk * s;
                    int64_t t2 = te + (int64_t)kk * per;
                    for (int32_t j = 0; j < ny; j += s)
                        for (int32_t i = 0; i < nx; i += s) lek[t2++] = zvalue((uint32_t)i, (uint32_t)j, (uint32_t)k);
                }
                te += per * b->layers[L];
            }
            double (*lc)[4] = NULL;
            double *la = NULL, *lM = NULL;
            float *lvp = NULL, *lh = NULL;
            int rc = radix_sort_u64(lek, E, 36) != 0 ? HQ_ERR_NOMEM : octbox_layer_constants(b, &lc, &la, &lM, &lvp, &lh);
            if (rc == HQ_OK) rc = octbox_local(b, lek, p->rank, P, (const double (*)[4])lc, la, lM, lvp, lh);
            free(lc); free(la); free(lM); free(lvp); free(lh); free(lek);
            if (rc != HQ_OK) { hqh_octbox_destroy(b); return rc; }
            *out = b;
            return HQ_OK;
        }
    }
    int64_t G = (int64_t)(nx + 1) * (ny + 1) * (nzt + 1);
    uint64_t* ek = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)E);
    uint64_t* nk = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N);
    int32_t* loc = (int32_t*)malloc(sizeof(int32_t) * (size_t)G);
    b->lnid = (int32_t*)malloc(sizeof(int32_t) * 8 * (size_t)E);
    b->node_xyz = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)N);
    b->etable = (double*)malloc(sizeof(double) * 4 * (size_t)E);
    b->ntable = (double*)calloc((size_t)N * 7, sizeof(double));
    if (!ek || !nk || !loc || !b->lnid || !b->node_xyz || !b->etable || !b->ntable) {
        free(ek); free(nk); free(loc); hqh_octbox_destroy(b); return HQ_ERR_NOMEM;
    }
    /* elements: octree pre-order = Z-order of the lower-left corner (finest-edge units) */
    int64_t t = 0;
    for (int L = 0; L < NL; L++) {
        const int32_t s = 1 << L;
        for (int32_t k = b->ztop[L]; k < b->ztop[L + 1]; k += s)
            for (int32_t j = 0; j < ny; j += s)
                for (int32_t i = 0; i < nx; i += s) ek[t++] = zvalue((uint32_t)i, (uint32_t)j, (uint32_t)k);
    }
    /* nodes: Z-order of the far-boundary-adjusted coordinates (octor.c:6100-6106, 6166); the
     * plane between two slabs carries the finer slab's grid */
    t = 0;
    for (int L = 0; L < NL; L++) {
        const int32_t s = 1 << L;
        for (int32_t k = b->ztop[L] + (L == 0 ? 0 : s); k <= b->ztop[L + 1]; k += s)
            for (int32_t j = 0; j <= ny; j += s)
                for (int32_t i = 0; i <= nx; i += s) {
                    uint32_t dx = (i == nx) ? (uint32_t)(2 * i - 1) : (uint32_t)(2 * i);
                    uint32_t dy = (j == ny) ? (uint32_t)(2 * j - 1) : (uint32_t)(2 * j);
                    uint32_t dz = (k == nzt) ? (uint32_t)(2 * k - 1) : (uint32_t)(2 * k);
                    nk[t++] = zvalue(dx, dy, dz);
                }
    }
    if (t != N || radix_sort_u64(ek, E, 36) != 0 || radix_sort_u64(nk, N, 36) != 0) {
        free(ek); free(nk); free(loc); hqh_octbox_destroy(b); return HQ_ERR_NOMEM;
    }
    /* a node of the plane on top of level L's slab (L >= 1) that is not on level L's grid hangs */
#define HQH_HANGS(c, Lout)                                                                     \
    (((Lout) = octbox_level_at(b, (c)[2] < nzt ? (c)[2] : nzt - 1)) >= 1 && (c)[2] == b->ztop[(Lout)] && \
     (((c)[0] & ((1 << (Lout)) - 1)) || ((c)[1] & ((1 << (Lout)) - 1))))
    memset(loc, 0xff, sizeof(int32_t) * (size_t)G);
    int32_t ldn = 0;
    for (int64_t n = 0; n < N; n++) {
        uint32_t d[3] = { compact3(nk[n]), compact3(nk[n] >> 1), compact3(nk[n] >> 2) };
        int32_t lim[3] = { nx, ny, nzt }, c[3];
        for (int q = 0; q < 3; q++) c[q] = (d[q] & 1) ? lim[q] : (int32_t)(d[q] >> 1);
        for (int q = 0; q < 3; q++) b->node_xyz[3 * n + q] = c[q];
        loc[((int64_t)c[2] * (ny + 1) + c[1]) * (nx + 1) + c[0]] = (int32_t)n;
        int Lh;
        if (HQH_HANGS(c, Lh)) ldn++;
    }
    free(nk);
    b->ldnnum = ldn;
    b->dn_id = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ldn ? ldn : 1));
    b->dn_ptr = (int32_t*)malloc(sizeof(int32_t) * ((size_t)ldn + 1));
    b->dn_anchor = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)(ldn ? ldn : 1));
    if (!b->dn_id || !b->dn_ptr || !b->dn_anchor) { free(ek); free(loc); hqh_octbox_destroy(b); return HQ_ERR_NOMEM; }
#define HQH_LOC(i, j, k) loc[((int64_t)(k) * (ny + 1) + (j)) * (nx + 1) + (i)]
    /* dnodeTable in node order; anchors in the order octor's prepending leaves them */
    {
        int32_t kdn = 0, na = 0;
        b->dn_ptr[0] = 0;
        for (int64_t n = 0; n < N; n++) {
            const int32_t* c = &b->node_xyz[3 * n];
            int Lh;
            if (!HQH_HANGS(c, Lh)) continue;
            const int32_t h = 1 << (Lh - 1), m = (1 << Lh) - 1, z = c[2];    /* finer edge; coarse-grid mask */
            b->dn_id[kdn] = (int32_t)n;
            if ((c[0] & m) && (c[1] & m)) {                      /* ZFACE: (+,+) (-,+) (+,-) (-,-) */
                b->dn_anchor[na++] = HQH_LOC(c[0] + h, c[1] + h, z);
                b->dn_anchor[na++] = HQH_LOC(c[0] - h, c[1] + h, z);
                b->dn_anchor[na++] = HQH_LOC(c[0] + h, c[1] - h, z);
                b->dn_anchor[na++] = HQH_LOC(c[0] - h, c[1] - h, z);
            } else if (c[0] & m) {                               /* XEDGE: +s then -s */
                b->dn_anchor[na++] = HQH_LOC(c[0] + h, c[1], z);
                b->dn_anchor[na++] = HQH_LOC(c[0] - h, c[1], z);
            } else {                                             /* YEDGE */
                b->dn_anchor[na++] = HQH_LOC(c[0], c[1] + h, z);
                b->dn_anchor[na++] = HQH_LOC(c[0], c[1] - h, z);
            }
            b->dn_ptr[++kdn] = na;
        }
    }
#undef HQH_HANGS
    /* element constants per element layer (mu_and_lambda + psolve.c:3387-3409, 3436-3437) */
    double aBase, bBase, dt = p->deltaT, dt2 = dt * dt;
    rayleigh_base(p->freq, p->damping, &aBase, &bBase);
    double (*lc)[4] = (double (*)[4])malloc(sizeof(double) * 4 * (size_t)nlay);
    double* la = (double*)malloc(sizeof(double) * (size_t)nlay);
    double* lM = (double*)malloc(sizeof(double) * (size_t)nlay);
    float* lvp = (float*)malloc(sizeof(float) * (size_t)nlay);
    float* lh = (float*)malloc(sizeof(float) * (size_t)nlay);
    if (!lc || !la || !lM || !lvp || !lh) {
        free(lc); free(la); free(lM); free(lvp); free(lh); free(ek); free(loc); hqh_octbox_destroy(b); return HQ_ERR_NOMEM;
    }
    for (int L = 0; L < NL; L++)
    for (int32_t q = b->lay0[L]; q < b->lay0[L + 1]; q++) {
        float Vp = b->vp[q], Vs = b->vs[q], rho = b->rho[q], h = (float)(p->h * (1 << L));
        lh[q] = h;
        double mu = rho * Vs * Vs, lambda;
        if (Vp > (Vs * p->threshold_vpvs)) lambda = rho * Vs * Vs * p->threshold_vpvs * p->threshold_vpvs - 2 * mu;
        else lambda = rho * Vp * Vp - 2 * mu;
        if (lambda < 0) {
            if (Vs < 500) Vp = 2.45 * Vs; else if (Vs < 1200) Vp = 2 * Vs; else Vp = 1.87 * Vs;
            lambda = rho * Vp * Vp;
        }
        if (lambda < 0) {
            free(lc); free(la); free(lM); free(lvp); free(lh); free(ek); free(loc); hqh_octbox_destroy(b); return HQ_ERR_ARG;
        }
        lvp[q] = Vp;
        double zeta = 10 / Vs;
        if (zeta > p->threshold_damping) zeta = p->threshold_damping;
        double a = zeta * aBase, bb = zeta * bBase;
        lc[q][0] = dt2 * h * mu / 9; lc[q][1] = dt2 * h * lambda / 9;
        lc[q][2] = bb * dt * h * mu / 9; lc[q][3] = bb * dt * h * lambda / 9;
        la[q] = a;
        double mass = rho * h * h * h;
        lM[q] = mass / 8;
    }
    /* connectivity, eTable, nTable (the reference's element loop, psolve.c:3360-3473) */
    nt_parts* parts = (p->solver_float == 4 && P > 1) ? nt_parts_new(b->N, P, E) : NULL;
    int nomem = (p->solver_float == 4 && P > 1 && !parts);
    for (int64_t e = 0; e < E && !nomem; e++) {
        int32_t i = (int32_t)compact3(ek[e]), j = (int32_t)compact3(ek[e] >> 1), k = (int32_t)compact3(ek[e] >> 2);
        int L = octbox_level_at(b, k), s = 1 << L;
        const int32_t q = b->lay0[L] + ((k - b->ztop[L]) >> L);  /* element layer */
        int face = (i == 0) | ((j == 0) << 1) | ((k == 0) << 2) | ((i + s == nx) << 3) | ((j + s == ny) << 4) |
                   ((k + s == nzt) << 5);
        for (int c4 = 0; c4 < 4; c4++) b->etable[4 * e + c4] = lc[q][c4];
        double M = lM[q], a = la[q];
        for (int c = 0; c < 8; c++) {
            int32_t n = HQH_LOC(i + s * (c & 1), j + s * ((c >> 1) & 1), k + s * ((c >> 2) & 1));
            b->lnid[8 * e + c] = n;
            double dash[3];
            int bnd = face_dashpot(face, c, p->halfspace, lh[q], lvp[q], b->vs[q], b->rho[q], dash);
            double* np = &b->ntable[7 * (int64_t)n];
            if (parts && nt_parts_row(parts, b->ntable, n, e)) { nomem = 1; break; }
            nt_accumulate(np, p->solver_float == 4, dt, a, M, bnd, dash);
        }
    }
    free(lc); free(la); free(lM); free(lvp); free(lh);
#undef HQH_LOC
    free(loc);
    if (nomem) { nt_parts_free(parts); free(ek); hqh_octbox_destroy(b); return HQ_ERR_NOMEM; }
    /* compute_adjust(nTable, 7, DISTRIBUTION), psolve.c:3502: hanging-node mass to the anchors (the float build on N ranks:
     * by the hanging nodes' owners, between the two exchanges -- nt_parts_combine in octbox_cut) */
    if (!parts) nt_distribute(b->ntable, p->solver_float == 4, b->ldnnum, b->dn_id, b->dn_ptr, b->dn_anchor);
    if (P > 1) {
        int rc = octbox_cut(b, ek, p->rank, P, parts);
        if (rc != HQ_OK) { nt_parts_free(parts); free(ek); hqh_octbox_destroy(b); return rc; }
    }
    nt_parts_free(parts);
    free(ek);
    *out = b;
    return HQ_OK;
}

/* ------------------------------------------------------------------------ */
/* mesh.e reader (read-only walk of the reference's etree / B-tree file)     */
/* ------------------------------------------------------------------------ */

static uint32_t rd_u32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static int64_t rd_i64(const unsigned char* p) { uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | p[i]; return (int64_t)v; }

int hqh_etree_read(const char* path, int64_t* n, int32_t* value_size, uint32_t** ticks, int32_t** level, void** values)
{
    if (!path || !n || !value_size || !ticks || !level || !values) return HQ_ERR_ARG;
    *n = 0; *ticks = NULL; *level = NULL; *values = NULL;
    FILE* f = fopen(path, "rb");
    if (!f) return HQ_ERR_ARG;
    int rc = HQ_ERR_ARG;
    unsigned char* page = NULL;
    uint32_t* tk = NULL; int32_t* lv = NULL; unsigned char* val = NULL;
    /* etree header (etree.c readheader): endian char, version, dimensions, rootlevel,
     * appmetasize, then leaf / index counts of 32 levels, 4 bytes each: 273 bytes */
    unsigned char eh[273], bh[33];
    if (fread(eh, 1, sizeof eh, f) != sizeof eh || eh[0] != 'L' || rd_u32(eh + 1) != 1 || rd_u32(eh + 5) != 3) goto done;
    /* B-tree meta data (btree.c metahdrsize): endian, pagesize, pagecount, rootpagenum, keysize,
     * valuesize, asciischemasize; page p lives at p * pagesize (page 0 = these headers) */
    if (fread(bh, 1, sizeof bh, f) != sizeof bh || bh[0] != 'L') goto done;
    const uint32_t pagesize = rd_u32(bh + 1), keysize = rd_u32(bh + 21), vsize = rd_u32(bh + 25);
    const int64_t pagecount = rd_i64(bh + 5), root = rd_i64(bh + 13);
    if (pagesize < 64 || pagesize > (1u << 24) || keysize != 13 || vsize == 0 || vsize > pagesize || root < 1 || root > pagecount) goto done;
    int64_t total = 0;
    for (int L = 0; L < 32; L++) total += rd_u32(eh + 17 + 8 * L);      /* leaf octants per level */
    page = (unsigned char*)malloc(pagesize);
    tk = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (size_t)(total ? total : 1));
    lv = (int32_t*)malloc(sizeof(int32_t) * (size_t)(total ? total : 1));
    val = (unsigned char*)malloc((size_t)vsize * (size_t)(total ? total : 1));
    if (!page || !tk || !lv || !val) { rc = HQ_ERR_NOMEM; goto done; }
    /* page header (btree.c setheader): right sibling i64 @0, (parent address) @8, count i32 @16,
     * (parent entry) @20, type 'l' / 'i' @24, entries from 25; index entry = key + child page i64 */
    int64_t pg = root;
    for (int depth = 0;; depth++) {
        if (depth > 64 || fseeko(f, (off_t)pg * pagesize, SEEK_SET) != 0 || fread(page, 1, pagesize, f) != pagesize) goto done;
        if (page[24] == 'l') break;
        if (page[24] != 'i' || (int32_t)rd_u32(page + 16) < 1) goto done;
        pg = rd_i64(page + 25 + keysize);                                 /* leftmost child */
        if (pg < 1 || pg > pagecount) goto done;
    }
    int64_t got = 0;
    for (int64_t guard = 0; pg != -1; guard++) {
        if (guard > pagecount || pg < 1 || pg > pagecount) goto done;
        if (fseeko(f, (off_t)pg * pagesize, SEEK_SET) != 0 || fread(page, 1, pagesize, f) != pagesize || page[24] != 'l') goto done;
        const int32_t cnt = (int32_t)rd_u32(page + 16);
        if (cnt < 0 || 25 + (int64_t)cnt * (keysize + vsize) > pagesize || got + cnt > total) goto done;
        for (int32_t e = 0; e < cnt; e++) {
            const unsigned char* k = page + 25 + (size_t)e * (keysize + vsize);
            if (!(k[0] & 0x80)) continue;                                 /* interior octant record */
            /* locational key (code.c): level | 0x80, then the 96-bit little-endian Morton code,
             * bit 3 i + d = bit i of coordinate d (x, y, z) */
            uint32_t c[3] = { 0, 0, 0 };
            for (int bit = 0; bit < 96; bit++)
                if (k[1 + (bit >> 3)] & (1u << (bit & 7))) c[bit % 3] |= 1u << (bit / 3);
            tk[3 * got] = c[0]; tk[3 * got + 1] = c[1]; tk[3 * got + 2] = c[2];
            lv[got] = k[0] & 0x7f;
            memcpy(val + (size_t)got * vsize, k + keysize, vsize);
            got++;
        }
        pg = rd_i64(page);
    }
    *n = got; *value_size = (int32_t)vsize; *ticks = tk; *level = lv; *values = val;
    tk = NULL; lv = NULL; val = NULL;
    rc = HQ_OK;
done:
    fclose(f);
    free(page); free(tk); free(lv); free(val);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* a CVM etree as the mesher's material model (cvm_query, quake/cvm/cvm.c:266-311) */
/* ------------------------------------------------------------------------ */

struct hqh_cvm {
    int64_t   n;                    /* leaf octants, in key (Z) order                                   */
    uint32_t* ticks;                /* [n][3] lower-left corners, etree ticks (root cube 2^31)           */
    int32_t*  level;                /* [n]                                                               */
    float*    val;                  /* [n][3] Vp, Vs, density (cvmpayload_t, cvm.h)                      */
    uint64_t* key;                  /* [n] Morton code of the corner: point location is a binary search */
    double    region[3];            /* region_length_east_m, _north_m, depth_deep_m - depth_shallow_m    */
    uint32_t  endpoint[3];          /* domain_endpoint_x / y / z, ticks                                  */
    double    ticksize;             /* metres per tick = region_length_east_m / domain_endpoint_x (cvm.c:289) */
    int32_t   minlevel, maxlevel;
};

static uint64_t hqh_morton63(uint32_t x, uint32_t y, uint32_t z)
{
    /* 21 bits per axis of the 31-bit tick coordinates' TOP bits would lose the low ones: interleave all 31 into the
     * order etree keys have (bit 3 i + d = bit i of coordinate d) using two words; compared lexicographically */
    uint64_t k = 0;
    for (int i = 20; i >= 0; i--) k = (k << 3) | (uint64_t)(((z >> (i + 10)) & 1u) << 2 | ((y >> (i + 10)) & 1u) << 1 | ((x >> (i + 10)) & 1u));
    return k;
}

void hqh_cvm_close(hqh_cvm* c)
{
    if (!c) return;
    free(c->ticks); free(c->level); free(c->val); free(c->key); free(c);
}

/*
 * Open a CVM etree: the leaves with their (Vp, Vs, density) payloads (hqh_etree_read's B-tree walk) and the database
 * control block cvm_getdbctl parses out of the etree's application meta data (cvm.c:62-215: five creation strings,
 * then origin latitude / longitude, region lengths east / north, depths shallow / deep, domain end point x y z), which
 * the etree library keeps as a text trailer behind the last page (etree.c storeappmeta).
 */
int hqh_cvm_open(const char* path, hqh_cvm** out)
{
    if (!path || !out) return HQ_ERR_ARG;
    *out = NULL;
    int64_t n = 0; int32_t vsize = 0; uint32_t* tk = NULL; int32_t* lv = NULL; void* vals = NULL;
    int rc = hqh_etree_read(path, &n, &vsize, &tk, &lv, &vals);
    if (rc != HQ_OK) return rc;
    hqh_cvm* c = (hqh_cvm*)calloc(1, sizeof *c);
    char* meta = NULL;
    rc = HQ_ERR_ARG;
    if (!c) { rc = HQ_ERR_NOMEM; goto fail; }
    if (n < 1 || vsize < 12) goto fail;                       /* "float Vp; float Vs; float density;" at least */
    {
        FILE* f = fopen(path, "rb");
        unsigned char eh[17];
        if (!f) goto fail;
        if (fread(eh, 1, sizeof eh, f) != sizeof eh) { fclose(f); goto fail; }
        const uint32_t msize = rd_u32(eh + 13);               /* appmetasize, the '\0' included */
        if (msize < 2 || msize > (1u << 20) || fseeko(f, -(off_t)msize, SEEK_END) != 0) { fclose(f); goto fail; }
        meta = (char*)malloc(msize + 1);
        if (!meta || fread(meta, 1, msize, f) != msize) { fclose(f); goto fail; }
        meta[msize] = 0;
        fclose(f);
    }
    {
        /* the LAST nine blank-separated tokens are the numbers (the creation strings in front may hold blanks) */
        double v[9];
        int got = 0;
        char* end = meta + strlen(meta);
        while (got < 9 && end > meta) {
            while (end > meta && (end[-1] == ' ' || end[-1] == '\n' || end[-1] == '\t')) end--;
            char* start = end;
            while (start > meta && start[-1] != ' ' && start[-1] != '\n' && start[-1] != '\t') start--;
            if (start == end) break;
            char save = *end; *end = 0;
            char* stop = NULL;
            v[8 - got] = strtod(start, &stop);
            const int ok = stop && *stop == 0;
            *end = save;
            if (!ok) break;
            got++;
            end = start;
        }
        if (got != 9) goto fail;
        c->region[0] = v[2]; c->region[1] = v[3]; c->region[2] = v[5] - v[4];
        for (int d = 0; d < 3; d++) { if (v[6 + d] < 1 || v[6 + d] > 2147483648.0) goto fail; c->endpoint[d] = (uint32_t)v[6 + d]; }
        if (!(c->region[0] > 0)) goto fail;
        c->ticksize = c->region[0] / (double)c->endpoint[0];
    }
    c->n = n; c->ticks = tk; c->level = lv; tk = NULL; lv = NULL;
    c->val = (float*)malloc(sizeof(float) * 3 * (size_t)n);
    c->key = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
    if (!c->val || !c->key) { rc = HQ_ERR_NOMEM; goto fail; }
    c->minlevel = 31; c->maxlevel = 0;
    for (int64_t i = 0; i < n; i++) {
        memcpy(c->val + 3 * i, (const unsigned char*)vals + (size_t)i * (size_t)vsize, 12);
        if (c->level[i] < 1 || c->level[i] > 21) goto fail;   /* (octants below 2^10 ticks: not this reader's keys) */
        if (c->level[i] < c->minlevel) c->minlevel = c->level[i];
        if (c->level[i] > c->maxlevel) c->maxlevel = c->level[i];
        c->key[i] = hqh_morton63(c->ticks[3 * i], c->ticks[3 * i + 1], c->ticks[3 * i + 2]);
        if (i > 0 && c->key[i] <= c->key[i - 1]) goto fail;   /* leaves arrive in key order */
    }
    free(vals); free(meta);
    *out = c;
    return HQ_OK;
fail:
    free(tk); free(lv); free(vals); free(meta);
    hqh_cvm_close(c);
    return rc;
}

int hqh_cvm_info(const hqh_cvm* c, int64_t* nleaves, int32_t levels[2], double region_m[3], double* ticksize)
{
    if (!c) return HQ_ERR_ARG;
    if (nleaves) *nleaves = c->n;
    if (levels) { levels[0] = c->minlevel; levels[1] = c->maxlevel; }
    if (region_m) for (int d = 0; d < 3; d++) region_m[d] = c->region[d];
    if (ticksize) *ticksize = c->ticksize;
    return HQ_OK;
}

/*
 * cvm_query (cvm.c:266-311): the payload of the leaf octant that holds the point -- ticks = (etree_tick_t)(metres /
 * tickSize) per axis, then etree_search's point location, here a binary search for the last leaf whose Morton code is
 * <= the point's, followed by the containment check.  -> 0, or -1 where no octant holds the point (as cvm_query).
 */
int hqh_cvm_query(const hqh_cvm* c, double east_m, double north_m, double depth_m, float payload[3])
{
    if (!c || !payload || east_m < 0 || north_m < 0 || depth_m < 0) return -1;
    const double q[3] = { east_m / c->ticksize, north_m / c->ticksize, depth_m / c->ticksize };
    uint32_t t[3];
    for (int d = 0; d < 3; d++) { if (!(q[d] < 2147483648.0)) return -1; t[d] = (uint32_t)q[d]; }
    const uint64_t k = hqh_morton63(t[0], t[1], t[2]);
    int64_t lo = 0, hi = c->n;                                 /* last i with key[i] <= k */
    while (hi - lo > 1) { const int64_t mid = (lo + hi) / 2; if (c->key[mid] <= k) lo = mid; else hi = mid; }
    if (c->key[lo] > k) return -1;
    const uint32_t edge = 1u << (31 - c->level[lo]);
    for (int d = 0; d < 3; d++)
        if (t[d] < c->ticks[3 * lo + d] || t[d] - c->ticks[3 * lo + d] >= edge) return -1;
    memcpy(payload, c->val + 3 * lo, 12);
    return 0;
}

/*
 * The database on a regular grid in the MESH's axes [k][y][x] for hqh_octree_generate: cells of the FINEST leaf level
 * present (every query that setrec makes, psolve.c:1307-1397, then finds the leaf it would find in the etree).  setrec
 * calls cvm_query(east = the mesh's y, north = the mesh's x) (psolve.c:1352), so the database's x ticks run along mesh y.
 * HQ_ERR_ARG where the leaves do not tile the domain or the grid would exceed 2^31 cells.  Release with hqh_free.
 */
int hqh_cvm_grid(const hqh_cvm* c, int32_t dims[3], double* cell_m, float** vp, float** vs, float** rho)
{
    if (!c || !dims || !cell_m || !vp || !vs || !rho) return HQ_ERR_ARG;
    *vp = *vs = *rho = NULL;
    const uint32_t ce = 1u << (31 - c->maxlevel);               /* cell edge, ticks */
    int64_t nd[3];                                              /* database axes: x (east), y (north), z */
    for (int d = 0; d < 3; d++) { if (c->endpoint[d] % ce) return HQ_ERR_ARG; nd[d] = c->endpoint[d] / ce; }
    const int64_t cells = nd[0] * nd[1] * nd[2];
    if (cells < 1 || cells > 0x7fffffffLL) return HQ_ERR_ARG;
    float* a[3];
    for (int q = 0; q < 3; q++) a[q] = (float*)malloc(sizeof(float) * (size_t)cells);
    unsigned char* seen = (unsigned char*)calloc((size_t)cells, 1);
    int rc = HQ_ERR_NOMEM;
    if (!a[0] || !a[1] || !a[2] || !seen) goto fail;
    rc = HQ_ERR_ARG;
    /* mesh axes: x = database north (y ticks), y = database east (x ticks) */
    const int64_t NX = nd[1], NY = nd[0];
    for (int64_t i = 0; i < c->n; i++) {
        const uint32_t edge = 1u << (31 - c->level[i]);
        const int64_t w = edge / ce, x0 = c->ticks[3 * i] / ce, y0 = c->ticks[3 * i + 1] / ce, z0 = c->ticks[3 * i + 2] / ce;
        if (c->ticks[3 * i] % ce || c->ticks[3 * i + 1] % ce || c->ticks[3 * i + 2] % ce) goto fail;
        for (int64_t z = z0; z < z0 + w && z < nd[2]; z++)
            for (int64_t xe = x0; xe < x0 + w && xe < nd[0]; xe++)
                for (int64_t yn = y0; yn < y0 + w && yn < nd[1]; yn++) {
                    const int64_t g = (z * NY + xe) * NX + yn;
                    if (seen[g]) goto fail;                     /* two leaves over one cell */
                    seen[g] = 1;
                    a[0][g] = c->val[3 * i]; a[1][g] = c->val[3 * i + 1]; a[2][g] = c->val[3 * i + 2];
                }
    }
    for (int64_t g = 0; g < cells; g++) if (!seen[g]) goto fail;     /* a hole in the database */
    free(seen);
    dims[0] = (int32_t)NX; dims[1] = (int32_t)NY; dims[2] = (int32_t)nd[2];
    *cell_m = c->ticksize * (double)ce;
    *vp = a[0]; *vs = a[1]; *rho = a[2];
    return HQ_OK;
fail:
    free(a[0]); free(a[1]); free(a[2]); free(seen);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* octree mesh from its leaves (octor_extractmesh + solver_init, one partition) */
/* ------------------------------------------------------------------------ */

int hqh_mesh_from_leaves(int64_t E, const uint32_t* et, const uint32_t* eedge, const float* edata,
                         const uint32_t far_ticks[3], const hqh_init_params* ip, hqh_octbox** out)
{
    if (E < 1 || !et || !eedge || !edata || !far_ticks || !ip || !out || ip->deltaT <= 0) return HQ_ERR_ARG;
    *out = NULL;
    if (E > 0x7fffffff / 8) return HQ_ERR_ARG;
    uint32_t emin = eedge[0];
    for (int64_t e = 1; e < E; e++) if (eedge[e] < emin) emin = eedge[e];
    if (emin == 0) return HQ_ERR_ARG;
    uint32_t nq[3];
    for (int d = 0; d < 3; d++) { if (far_ticks[d] % emin) return HQ_ERR_ARG; nq[d] = far_ticks[d] / emin; if (nq[d] >= (1u << 20)) return HQ_ERR_ARG; }
    hqh_octbox* b = (hqh_octbox*)calloc(1, sizeof *b);
    uint64_t* ck = (uint64_t*)malloc(sizeof(uint64_t) * 8 * (size_t)E);
    uint64_t* nkey = NULL;
    int32_t *touch = NULL, *small = NULL;
    int rc = HQ_ERR_NOMEM;
    nt_parts* parts = NULL;          /* solver_float = 4 on a partition: the rows in runs of ranks (nt_parts_combine) */
    if (!b || !ck) goto fail;
    const int P = ip->nranks > 1 ? ip->nranks : 1;
    if (P > 64 || ip->rank < 0 || ip->rank >= P || !hqh_sf_valid(ip->solver_float)) { rc = HQ_ERR_ARG; goto fail; }
    b->p.nlevels = 0; b->p.deltaT = ip->deltaT; b->p.rank = ip->rank; b->p.nranks = P; b->p.solver_float = ip->solver_float;
    for (int d = 0; d < 3; d++) b->far_q[d] = (int32_t)nq[d];
    b->E = E;
    b->lnid = (int32_t*)malloc(sizeof(int32_t) * 8 * (size_t)E);
    b->etable = (double*)malloc(sizeof(double) * 4 * (size_t)E);
    if (!b->lnid || !b->etable) goto fail;
    /* every element corner, keyed by the Z-value of its doubled, far-adjusted coordinates */
    {
        int badarg = 0;
#pragma omp parallel for schedule(static) reduction(| : badarg)
        for (int64_t e = 0; e < E; e++) {
            if (et[3 * e] % emin || et[3 * e + 1] % emin || et[3 * e + 2] % emin || eedge[e] % emin) { badarg |= 1; continue; }
            const uint32_t s = eedge[e] / emin;
            for (int c = 0; c < 8; c++) {
                uint32_t q[3], k2[3];
                for (int d = 0; d < 3; d++) {
                    q[d] = et[3 * e + d] / emin + (((c >> d) & 1) ? s : 0);
                    if (q[d] > nq[d]) { badarg |= 1; q[d] = nq[d]; }
                    k2[d] = (q[d] == nq[d]) ? 2 * q[d] - 1 : 2 * q[d];
                }
                ck[8 * e + c] = zvalue(k2[0], k2[1], k2[2]);
            }
        }
        if (badarg) { rc = HQ_ERR_ARG; goto fail; }
    }
    /* the distinct keys in order = the nodes (octor.c:6166); every corner then finds its node by bisection */
    int64_t N = 0;
    {
        uint64_t* sk = (uint64_t*)malloc(sizeof(uint64_t) * 8 * (size_t)E);
        if (!sk) goto fail;
        memcpy(sk, ck, sizeof(uint64_t) * 8 * (size_t)E);
        int kb = 0;
        for (int d = 0; d < 3; d++) { int w = 1; while (((uint64_t)2 * nq[d]) >> w) w++; if (w > kb) kb = w; }
        if (radix_sort_u64(sk, 8 * E, 3 * kb) != 0) { free(sk); goto fail; }
        for (int64_t i = 0; i < 8 * E; i++) if (i == 0 || sk[i] != sk[i - 1]) sk[N++] = sk[i];
        if (N > 0x7fffffff / 8) { free(sk); rc = HQ_ERR_ARG; goto fail; }
        nkey = (uint64_t*)realloc(sk, sizeof(uint64_t) * (size_t)N);
        if (!nkey) { free(sk); goto fail; }
    }
    b->N = N;
    b->node_xyz = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)N);
    b->ntable = (double*)calloc((size_t)N * 7, sizeof(double));
    touch = (int32_t*)calloc((size_t)N, sizeof(int32_t));
    small = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    if (!b->node_xyz || !b->ntable || !touch || !small) goto fail;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        uint32_t d2[3] = { compact3(nkey[n]), compact3(nkey[n] >> 1), compact3(nkey[n] >> 2) };
        for (int d = 0; d < 3; d++) b->node_xyz[3 * n + d] = (int32_t)((d2[d] & 1) ? nq[d] : (d2[d] >> 1));
        small[n] = 0x7fffffff;
    }
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; e++) {
        const int32_t s = (int32_t)(eedge[e] / emin);
        for (int c = 0; c < 8; c++) {
            const uint64_t key = ck[8 * e + c];
            int64_t lo = 0, hi = N - 1;
            while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (nkey[m] < key) lo = m + 1; else hi = m; }
            b->lnid[8 * e + c] = (int32_t)lo;
            __atomic_fetch_add(&touch[lo], 1, __ATOMIC_RELAXED);
            int32_t cur = __atomic_load_n(&small[lo], __ATOMIC_RELAXED);
            while (s < cur && !__atomic_compare_exchange_n(&small[lo], &cur, s, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
        }
    }
    free(ck); ck = NULL;
    /* node_setproperty: anchored unless it misses touches for where it sits; a hanging node sits
     * on the grid of its smallest toucher but off the next coarser one in 1 (edge) or 2 (face) axes */
    {
        int32_t ldn = 0, nanch = 0;
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) {
                b->ldnnum = ldn;
                b->dn_id = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ldn ? ldn : 1));
                b->dn_ptr = (int32_t*)malloc(sizeof(int32_t) * ((size_t)ldn + 1));
                b->dn_anchor = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nanch ? nanch : 1));
                if (!b->dn_id || !b->dn_ptr || !b->dn_anchor) goto fail;
                b->dn_ptr[0] = 0;
                ldn = 0; nanch = 0;
            }
            for (int64_t n = 0; n < N; n++) {
                const int32_t* c = &b->node_xyz[3 * n];
                int wh = 0, mods[3], nm = 0;
                const int tc = touch[n], s = small[n];
                for (int d = 0; d < 3; d++) wh += (c[d] == 0 || (uint32_t)c[d] == nq[d]);
                if (tc == 8 || (tc == 4 && wh == 1) || (tc == 2 && wh == 2) || (tc == 1 && wh == 3)) continue;
                for (int d = 0; d < 3; d++) { mods[d] = (c[d] % (2 * s)) != 0; nm += mods[d]; }
                if (!((tc == 6 && wh == 0 && nm == 1) || (tc == 4 && wh == 0 && (nm == 1 || nm == 2)) ||
                      (tc == 2 && (wh == 0 || wh == 1) && nm == 1))) { rc = HQ_ERR_ARG; goto fail; }   /* not a 2:1 octree */
                int32_t pts[4][3], np = 0;
                if (nm == 1) {                                    /* edge: -s then +s */
                    int d = mods[0] ? 0 : (mods[1] ? 1 : 2);
                    for (int sg = -1; sg <= 1; sg += 2) { memcpy(pts[np], c, sizeof pts[0]); pts[np][d] += sg * s; np++; }
                } else {                                          /* face: the two in-plane axes, low axis fastest */
                    int a = mods[0] ? 0 : 1, bb = mods[2] ? 2 : 1;
                    for (int dep = 0; dep < 4; dep++) {
                        memcpy(pts[np], c, sizeof pts[0]);
                        pts[np][a] += (dep & 1) ? s : -s;
                        pts[np][bb] += (dep & 2) ? s : -s;
                        np++;
                    }
                }
                if (pass == 1) {
                    b->dn_id[ldn] = (int32_t)n;
                    for (int i = np - 1; i >= 0; i--) {           /* the list is built by prepending */
                        uint32_t k2[3];
                        for (int d = 0; d < 3; d++) {
                            if (pts[i][d] < 0 || (uint32_t)pts[i][d] > nq[d]) { rc = HQ_ERR_ARG; goto fail; }
                            k2[d] = ((uint32_t)pts[i][d] == nq[d]) ? 2 * nq[d] - 1 : 2 * (uint32_t)pts[i][d];
                        }
                        const uint64_t key = zvalue(k2[0], k2[1], k2[2]);
                        int64_t lo = 0, hi = N - 1, hit = -1;
                        while (lo <= hi) { int64_t m = (lo + hi) / 2; if (nkey[m] < key) lo = m + 1; else if (nkey[m] > key) hi = m - 1; else { hit = m; break; } }
                        if (hit < 0) { rc = HQ_ERR_ARG; goto fail; }
                        b->dn_anchor[nanch + (np - 1 - i)] = (int32_t)hit;
                    }
                    b->dn_ptr[ldn + 1] = nanch + np;
                }
                ldn++; nanch += np;
            }
        }
    }
    free(nkey); nkey = NULL; free(touch); touch = NULL; free(small); small = NULL;
    /* solver_init's element loop (psolve.c:3360-3473) */
    {
        double aBase, bBase;
        const double dt = ip->deltaT, dt2 = dt * dt;
        rayleigh_base(ip->freq, ip->damping, &aBase, &bBase);
        b->edata = (float*)malloc(sizeof(float) * 4 * (size_t)E);
        if (!b->edata) goto fail;
        memcpy(b->edata, edata, sizeof(float) * 4 * (size_t)E);
        b->bbase = bBase; b->thr_damp = ip->threshold_damping; b->thr_vpvs = ip->threshold_vpvs;
        if (ip->solver_float == 4 && P > 1) {
            parts = nt_parts_new(b->N, P, E);
            if (!parts) goto fail;
        }
        for (int64_t e = 0; e < E; e++) {
            float h = edata[4 * e], Vp = edata[4 * e + 1], Vs = edata[4 * e + 2], rho = edata[4 * e + 3];
            double mu = rho * Vs * Vs, lambda;
            if (Vp > (Vs * ip->threshold_vpvs)) lambda = rho * Vs * Vs * ip->threshold_vpvs * ip->threshold_vpvs - 2 * mu;
            else lambda = rho * Vp * Vp - 2 * mu;
            if (lambda < 0) {
                if (Vs < 500) Vp = 2.45 * Vs; else if (Vs < 1200) Vp = 2 * Vs; else Vp = 1.87 * Vs;
                lambda = rho * Vp * Vp;
            }
            if (lambda < 0) { rc = HQ_ERR_ARG; goto fail; }
            b->edata[4 * e + 1] = Vp;                        /* mu_and_lambda rewrites edata->Vp (psolve.c:3253-3263) */
            double zeta = 10 / Vs;
            if (zeta > ip->threshold_damping) zeta = ip->threshold_damping;
            const double a = zeta * aBase, bb = zeta * bBase;
            b->etable[4 * e] = dt2 * h * mu / 9; b->etable[4 * e + 1] = dt2 * h * lambda / 9;
            b->etable[4 * e + 2] = bb * dt * h * mu / 9; b->etable[4 * e + 3] = bb * dt * h * lambda / 9;
            const double mass = rho * h * h * h, M = mass / 8;
            const uint32_t s = eedge[e] / emin, q[3] = { et[3 * e] / emin, et[3 * e + 1] / emin, et[3 * e + 2] / emin };
            const int face = (q[0] == 0) | ((q[1] == 0) << 1) | ((q[2] == 0) << 2) | ((q[0] + s == nq[0]) << 3) |
                             ((q[1] + s == nq[1]) << 4) | ((q[2] + s == nq[2]) << 5);
            for (int c = 0; c < 8; c++) {
                double dash[3];
                int bnd = face_dashpot(face, c, ip->halfspace, h, Vp, Vs, rho, dash);
                double* np = &b->ntable[7 * (int64_t)b->lnid[8 * e + c]];
                if (parts && nt_parts_row(parts, b->ntable, b->lnid[8 * e + c], e)) { rc = HQ_ERR_NOMEM; goto fail; }
                nt_accumulate(np, ip->solver_float == 4, dt, a, M, bnd, dash);
            }
        }
    }
    /* compute_adjust(nTable, 7, DISTRIBUTION), psolve.c:3502 (the float build on N ranks: nt_parts_combine in octbox_cut) */
    if (!parts) nt_distribute(b->ntable, ip->solver_float == 4, b->ldnnum, b->dn_id, b->dn_ptr, b->dn_anchor);
    if (P > 1) {
        /* the leaves' corners in Z-order = octree pre-order (checked), for point location */
        uint64_t* ek = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)E);
        if (!ek) goto fail;
        for (int64_t e = 0; e < E; e++) {
            ek[e] = zvalue(et[3 * e] / emin, et[3 * e + 1] / emin, et[3 * e + 2] / emin);
            if (e && ek[e] <= ek[e - 1]) { free(ek); rc = HQ_ERR_ARG; goto fail; }
        }
        rc = octbox_cut(b, ek, ip->rank, P, parts);
        free(ek);
        if (rc != HQ_OK) goto fail;
    }
    nt_parts_free(parts);
    *out = b;
    return HQ_OK;
fail:
    nt_parts_free(parts);
    free(ck); free(nkey); free(touch); free(small);
    hqh_octbox_destroy(b);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* layered model -> column of octree leaves (Vs rule + 2:1 balance)          */
/* ------------------------------------------------------------------------ */

typedef struct { double z0, edge; float vp, vs, rho; } hqh_leaf;

/* setrec (psolve.c:1307-1397) for a leaf of a depth-only model */
static void layered_setrec(const hqh_layered_model* m, double vscut, hqh_leaf* lf)
{
    static const double pts[3] = { 0.01, 1, 1.99 };
    const double half = lf->edge / 2;
    float bvs = FLT_MAX, bvp = 0, brho = 0;
    for (int i = 0; i < 3; i++) {
        const double z = lf->z0 + pts[i] * half;
        int L = 0;
        while (L + 1 < m->nlayers && z >= m->ztop[L + 1]) L++;
        if (m->vs[L] < bvs) { bvs = m->vs[L]; bvp = m->vp[L]; brho = m->rho[L]; }
        if (m->vs[L] <= vscut) break;
    }
    if (bvs <= vscut) {                                   /* adjust Vs and Vp, psolve.c:1389-1394 */
        const double ratio = bvp / bvs;
        bvs = (float)vscut;
        bvp = (float)(vscut * ratio);
    }
    lf->vp = bvp; lf->vs = bvs; lf->rho = brho;
}

int hqh_layered_column(const hqh_layered_model* m, double h0, int32_t ncoarse, double factor, double vscut,
                       int32_t cap, double* edge, float* vp, float* vs, float* rho, int32_t* nleaves)
{
    if (!m || m->nlayers < 1 || !m->ztop || !m->vp || !m->vs || !m->rho || h0 <= 0 || ncoarse < 1 || factor <= 0 ||
        cap < 1 || !edge || !vp || !vs || !rho || !nleaves)
        return HQ_ERR_ARG;
    int32_t n = 0, capw = cap;
    hqh_leaf* w = (hqh_leaf*)malloc(sizeof(hqh_leaf) * (size_t)capw);
    if (!w) return HQ_ERR_NOMEM;
#define HQH_SPLIT(i)                                                                         \
    do {                                                                                     \
        if (n + 1 > capw) { free(w); return HQ_ERR_ARG; }                                    \
        memmove(w + (i) + 1, w + (i), sizeof(hqh_leaf) * (size_t)(n - (i)));                 \
        n++;                                                                                 \
        w[(i)].edge /= 2;                                                                    \
        w[(i) + 1].edge = w[(i)].edge; w[(i) + 1].z0 = w[(i)].z0 + w[(i)].edge;              \
        layered_setrec(m, vscut, &w[(i)]); layered_setrec(m, vscut, &w[(i) + 1]);            \
    } while (0)
    if (ncoarse > capw) { free(w); return HQ_ERR_ARG; }
    for (int32_t i = 0; i < ncoarse; i++) { w[i].z0 = i * h0; w[i].edge = h0; layered_setrec(m, vscut, &w[i]); }
    n = ncoarse;
    /* octor_refinetree: vsrule (quake_util.c:215-225) */
    for (int32_t i = 0; i < n;) {
        if (!(w[i].edge <= w[i].vs / factor) && w[i].edge > h0 / 65536) HQH_SPLIT(i);
        else i++;
    }
    /* octor_balancetree: no leaf more than twice its neighbour */
    for (int again = 1; again;) {
        again = 0;
        for (int32_t i = 0; i + 1 < n; i++) {
            if (w[i].edge > 2 * w[i + 1].edge * (1 + 1e-12)) { HQH_SPLIT(i); again = 1; break; }
            if (w[i + 1].edge > 2 * w[i].edge * (1 + 1e-12)) { HQH_SPLIT(i + 1); again = 1; break; }
        }
    }
#undef HQH_SPLIT
    for (int32_t i = 0; i < n; i++) { edge[i] = w[i].edge; vp[i] = w[i].vp; vs[i] = w[i].vs; rho[i] = w[i].rho; }
    *nleaves = n;
    free(w);
    return HQ_OK;
}

#include "hq_mesher.h"

/* the two-level box: nz_fine layers of edge h over nz_coarse layers of edge 2h */
int hqh_octbox_create(const hqh_octbox_params* p, hqh_octbox** out)
{
    if (!p || !out) return HQ_ERR_ARG;
    *out = NULL;
    if (p->nx < 2 || p->ny < 2 || p->nz_fine < 2 || p->nz_coarse < 1 || (p->nx & 1) || (p->ny & 1) || (p->nz_fine & 1))
        return HQ_ERR_ARG;
    int32_t layers[2] = { p->nz_fine, p->nz_coarse };
    const int32_t nlay = p->nz_fine + p->nz_coarse;
    float* m = (float*)malloc(sizeof(float) * 3 * (size_t)nlay);
    if (!m) return HQ_ERR_NOMEM;
    for (int32_t i = 0; i < nlay; i++) {
        m[i] = i < p->nz_fine ? p->vp_top : p->vp_bot;
        m[nlay + i] = i < p->nz_fine ? p->vs_top : p->vs_bot;
        m[2 * nlay + i] = i < p->nz_fine ? p->rho_top : p->rho_bot;
    }
    hqh_octlevels_params q;
    memset(&q, 0, sizeof q);
    q.nx = p->nx; q.ny = p->ny; q.nlevels = 2; q.layers = layers; q.h = p->h;
    q.vp = m; q.vs = m + nlay; q.rho = m + 2 * nlay;
    q.deltaT = p->deltaT; q.freq = p->freq; q.damping = p->damping;
    q.threshold_damping = p->threshold_damping; q.threshold_vpvs = p->threshold_vpvs; q.halfspace = p->halfspace;
    q.rank = p->rank; q.nranks = p->nranks; q.solver_float = p->solver_float;
    int rc = hqh_octbox_create_levels(&q, out);
    free(m);
    return rc;
}

int hqh_octbox_solver_run(hq_ctx* ctx, const hqh_octbox* b, const hqh_run_params* rp, int32_t step0, int32_t nsteps)
{
    if (!b) return HQ_ERR_ARG;
    return hqh_solver_run_on(ctx, b->p.deltaT, (int32_t)b->N, rp, step0, nsteps);
}

int hqh_octbox_desc(const hqh_octbox* b, hq_desc* d)
{
    if (!b || !d) return HQ_ERR_ARG;
    memset(d, 0, sizeof *d);
    d->lenum = (int32_t)b->E; d->nharbored = (int32_t)b->N; d->ldnnum = b->ldnnum;
    d->lnid = b->lnid; d->node_xyz = b->node_xyz;
    d->dn_ldnid = b->dn_id; d->dn_ptr = b->dn_ptr; d->dn_lanid = b->dn_anchor;
    d->eTable = b->etable; d->nTable = b->ntable;
    d->an_sched.c_count = b->nc[0]; d->an_sched.first_c = b->mc[0];
    d->an_sched.s_count = b->ns[0]; d->an_sched.first_s = b->ms[0];
    d->dn_sched.c_count = b->nc[1]; d->dn_sched.first_c = b->mc[1];
    d->dn_sched.s_count = b->ns[1]; d->dn_sched.first_s = b->ms[1];
    d->deltaT = b->p.deltaT;
    d->rank = b->p.nranks > 1 ? b->p.rank : 0;
    d->nranks = b->p.nranks > 1 ? b->p.nranks : 1;
    d->variant = HQ_VARIANT_AUTO;
    if (b->edata) { d->edata = b->edata; d->mat_bbase = b->bbase; d->mat_threshold_damping = b->thr_damp; d->mat_threshold_vpvs = b->thr_vpvs; }
    return HQ_OK;
}

/* messenger lists for tests: sched 0 = an, 1 = dn; list 0 = c, 1 = s */
int hqh_octbox_schedule(const hqh_octbox* b, int32_t sched, int32_t list, int32_t* count, const hq_messenger** first)
{
    if (!b || !count || !first || sched < 0 || sched > 1 || list < 0 || list > 1) return HQ_ERR_ARG;
    *count = list ? b->ns[sched] : b->nc[sched];
    *first = list ? b->ms[sched] : b->mc[sched];
    return HQ_OK;
}

const void* hqh_octbox_view(const hqh_octbox* b, int32_t which, int64_t* count)
{
    if (!b || !count) return NULL;
    switch (which) {
    case 0: *count = b->E * 8; return b->lnid;
    case 1: *count = b->N * 3; return b->node_xyz;
    case 2: *count = b->ldnnum; return b->dn_id;
    case 3: *count = (int64_t)b->ldnnum + 1; return b->dn_ptr;
    case 4: *count = b->dn_ptr[b->ldnnum]; return b->dn_anchor;
    case 5: *count = b->E * 4; return b->etable;
    case 6: *count = b->N * 7; return b->ntable;
    case 7: *count = b->owner ? b->N : 0; return b->owner;
    case 8: *count = b->gid ? b->N : 0; return b->gid;
    }
    return NULL;
}
