/*
 * hq_patch.h -- "owner computes" fused step for gfx950.
 *
 * The reference accumulates element forces into a nodal force array
 * (stiffness.c:228-235, damping.c:88-98) and then sweeps the nodes
 * (solver_compute_displacement, psolve.c:4078-4111).  On the GPU that costs a
 * read-modify-write of `force` per element plus a read and a zeroing write per
 * node.  Here the Z-ordered node range (octor sorts nodes with
 * octor_zcompare, octor.c:6166) is cut into PATCHES of consecutive nodes that
 * fill one aligned octree cube.  One workgroup per patch
 *   1. stages u(t), u(t-dt) of the patch's nodes and of the one ring of
 *      neighbour ("halo") nodes its elements reach into LDS,
 *   2. evaluates EVERY element that touches an owned node (an element on a
 *      patch boundary is evaluated by each patch it touches), accumulating
 *      the forces of owned nodes in LDS with ds_add_f64,
 *   3. adds the source force and finishes the central-difference update for
 *      the owned nodes, writing u(t+dt) to a third buffer.
 * The force vector never reaches HBM and there is one launch per step.
 *
 * Host side (hq_patch_build): cube-aligned cuts from the node coordinates,
 * per-patch element lists, 16-bit patch-local connectivity, halo lists.
 */
#ifndef HQ_PATCH_H
#define HQ_PATCH_H

#ifdef _OPENMP
#include <omp.h>
#else
static inline int omp_get_num_threads(void) { return 1; }
static inline int omp_get_thread_num(void) { return 0; }
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <chrono>
#include <array>
#include <string>
#include <unordered_map>
#include <vector>
#include <cstring>

#include "hq_kernels.h"
#include "hq_opts.h"

#define HQ_PATCH_MAX_THREADS 512

/* Tuning knobs (environment overrides HQ_PATCH_THREADS / _PMAX / _PMERGE / _NLMAX are
 * read once per plan; defaults from the sweeps in profiles/). */
/* Lattice-SUBSET patches (domain faces, dashpots, far-face cubes, partition interfaces) through hq_k_patch_stencil too,
 * not only the full lattices?  Measured (docs/LABNOTES.md): 2-3 % faster than the element form on the whole 64 M box, 9 % on
 * its eight in-process partitions -- the default is yes.  HQ_PATCH_RAGGED=0: element form; 2: stencil form except for
 * patches with partition-interface nodes.  -> -1 (not set) or the value. */
static inline int hq_patch_ragged_env(void)
{
    return hq_opt_int("HQ_PATCH_RAGGED", -1);         /* 2: as 1, but patches with partition-interface nodes keep the element form */
}

struct hq_patch_cfg {
    int threads = 512;    /* workgroup size                                          */
    int pmax    = 768;    /* owned nodes per patch (9*9*9 = 729 fits)                */
    int pmerge  = 512;    /* small neighbouring cubes are merged up to this          */
    int psplit  = 0;      /* a cube with more owned nodes is halved; 0 = pmax (the 9-wide cubes on the far faces hold
                           * 576..729 nodes)                                                                         */
    int nlmax   = 1024;   /* owned + halo nodes staged in LDS (10*10*10 fits)        */
    int vmax    = 0;      /* extra force accumulators for hanging nodes whose anchors */
                          /* the patch owns (set by the planner when the mesh has any) */
};

static hq_patch_cfg hq_patch_cfg_from_env(void)
{
    hq_patch_cfg c;
    auto geti = [](const char* n, int def) { return hq_opt_int(n, def); };
    c.threads = geti("HQ_PATCH_THREADS", c.threads);
    c.pmax = geti("HQ_PATCH_PMAX", c.pmax);
    c.pmerge = geti("HQ_PATCH_PMERGE", c.pmerge);
    c.psplit = geti("HQ_PATCH_PSPLIT", 0);
    c.nlmax = geti("HQ_PATCH_NLMAX", c.nlmax);
    if (c.threads < 64) c.threads = 64;
    if (c.threads > HQ_PATCH_MAX_THREADS) c.threads = HQ_PATCH_MAX_THREADS;
    c.threads &= ~63;
    if (c.pmax < 8) c.pmax = 8;
    if (c.pmerge > c.pmax) c.pmerge = c.pmax;
    if (c.psplit > c.pmax || c.psplit < 8) c.psplit = c.pmax;
    if (c.pmerge < 1) c.pmerge = 1;
    if (c.nlmax < c.pmax + 8) c.nlmax = c.pmax + 8;
    if (c.nlmax > 0x7fff) c.nlmax = 0x7fff;          /* HQ_PIDX_ROW: 15-bit rows in the element row */
    c.vmax = geti("HQ_PATCH_VMAX", c.vmax);
    /* LDS: (6 nlmax + 3 (pmax + vmax)) doubles must fit 160 KiB */
    while ((6 * (size_t)c.nlmax + 3 * (size_t)(c.pmax + c.vmax)) * 8 > 160 * 1024) c.nlmax -= 8;
    return c;
}

struct hq_patch_desc {
    int32_t base;        /* first owned node (global id)          */
    int32_t nown;        /* owned nodes: [base, base + nown)       */
    int32_t nhalo;       /* halo nodes, local ids nown..nown+nhalo */
    int32_t npairs;      /* elements evaluated by this patch       */
    int64_t pair_off;    /* into pidx / pc1 / pc2 / pbeta          */
    int64_t halo_off;    /* into halo_ids                          */
    int32_t flags;       /* HQ_PATCH_ISO: every owned node has axis-independent n_t */
    int32_t nacc;        /* local nodes with a force accumulator: owned + the first   */
                         /* (nacc - nown) halo nodes = hanging nodes on owned anchors */
    int64_t pidx_off;    /* into pidx: this patch's rows, or the identical rows of an  */
                         /* earlier patch (regular regions repeat one local connectivity) */
};
#define HQ_PATCH_ISO 1
#define HQ_PATCH_UNIFORM 2   /* every element of the patch has the same (c1, c2, beta): read once */
#define HQ_PATCH_WFORM 4     /* uniform, and owned nodes <= nlmax / 2: hq_k_patch_pers keeps w = u1 + beta (u1 - u2) */
                             /* of all local nodes and u1, u2 of the owned ones in LDS instead of u1, u2 of all */
#define HQ_PATCH_NTSAME 8    /* ISO, and every owned node has the same n_t row: all lanes read the first one */
#define HQ_PATCH_LATTICE 16  /* rows and lanes of hq_lattice(): thread t's LDS row is lat_row[t] (hq_k_patch_pers / _seed only) */
#define HQ_PATCH_STENCIL 32  /* nodes and elements on the 10x10x10 lattice (all of it or a subset), uniform coefficients, no hanging
                              * node involved: stepped by hq_k_patch_stencil (the assembled 27-point stencil per owned node) */
#define HQ_PATCH_RAGGED 64   /* a STENCIL patch that is not the full lattice without dashpot: a domain face, a partition interface,
                              * a far-face patch of 9 layers (statistics only: the kernel reads the patch shape's table) */

struct hq_patch_host {
    std::vector<hq_patch_desc> desc;
    std::vector<uint16_t> pidx;      /* [npairs][8] patch-local node ids */
    std::vector<int32_t>  pelem;     /* [npairs] element id (host only)  */
    std::vector<int32_t>  halo;      /* halo node ids, patch p's list at p * hstride */
    int32_t               hstride = 0;
    int32_t               ndistinct = 0;   /* distinct local connectivities among the patches */
    std::vector<char>     lattice;   /* [P] 1 = lattice patch: rows and lanes of hq_lattice()      */
    std::vector<int32_t>  ds_ptr;    /* [P+1] hanging-node distribution entries per patch */
    std::vector<int32_t>  ds_ent;    /* [n][3] = {src local, dst local (owned anchor), deps} */
};

/* dnodeTable (octor.h:153-158) as CSR */
struct hq_dangling {
    int32_t n = 0;
    const int32_t* id = nullptr;
    const int32_t* ptr = nullptr;
    const int32_t* anchor = nullptr;
};

struct hq_patch_plan {
    hq_patch_cfg cfg;
    int32_t npatches = 0;
    int64_t npairs = 0;
    int64_t nhalo = 0;
    int32_t hstride = 0;             /* ints between consecutive patches' halo id lists */
    int32_t ndistinct = 0, nuniform = 0;
    hq_patch_desc* d_desc = nullptr;
    uint4*   d_pidx = nullptr;
    double*  d_pc1 = nullptr;
    double*  d_pc2 = nullptr;
    double*  d_pbeta = nullptr;
    int32_t* d_halo = nullptr;
    double*  d_nt3 = nullptr;        /* [N][3] {mass_simple, mass2_minusaM, mass_minusaM} for ISO patches */
    /* source entries grouped by patch (built by hq_patch_set_source) */
    int32_t* d_src_ptr = nullptr;    /* [npatches + 1]                    */
    int32_t* d_src_ent = nullptr;    /* [n][2] = {local node, loaded idx} */
    /* partition-interface nodes grouped by owning patch (hq_patch_set_interface) */
    int32_t* d_if_ptr = nullptr;     /* [npatches + 1]                    */
    int32_t* d_if_ent = nullptr;     /* [n][2] = {local node, slot}       */
    int32_t* d_order = nullptr;      /* patch ids: the nb interface patches first, then the rest */
    int32_t* d_tickets = nullptr;    /* per XCD, HQ_TICKET_STRIDE ints apart: {next slot of hq_k_patch_pers' work queue, workgroups done} */
    uint16_t* d_lat_row = nullptr;   /* [1024] LDS row of thread t's local node in a lattice patch             */
    int32_t  ne = 0, ns = 0, nr = 0; /* d_order = nb interface patches | ne other element-form patches | nr stencil patches with
                                      * interface nodes | ns other stencil patches */
    bool     ragged_default = false; /* set by the caller before hq_patch_build: see hq_patch_ragged_env */
    int32_t  ns_rg = 0;              /* of the ns: ragged patches of <= 512 nodes, between the full lattices and the big ones */
    int32_t  nr_big = 0, ns_big = 0; /* of the nr / ns: patches of more than 512 owned nodes, at the end of their part */
    int32_t  nragged = 0, nstencil = 0;  /* STENCIL patches (nr + ns entries: two for a patch of more than 512 nodes), RAGGED ones among them */
    uint32_t* d_rg_tab = nullptr;    /* tables of the stencil patches (hq_ragged_match), one per patch shape      */
#ifdef HQ_ST_TIMING
    bool timing_armed = false; unsigned long long* d_timing = nullptr;
#endif
    struct hq_st_desc* d_st_desc = nullptr;   /* [nr + ns] the stencil patches' records in launch order (hq_k_stencil_entries) */
    int2*    d_st_halo = nullptr;    /* [nr + ns][512] (halo node, its LDS word) in launch order                        */
    double*  d_pcoef = nullptr;      /* [P][4] c1, c2, beta of a stencil patch (beside the descriptor: no second hop)  */
    int64_t* d_rg_off = nullptr;     /* [P] offset of a patch's table in d_rg_tab                                  */
    double*  d_E1 = nullptr;         /* [576] element matrix blocks for (c1, c2) = (1, 0) ...                      */
    double*  d_E2 = nullptr;         /* ... and (0, 1)                                                              */
    int32_t* d_if_slot = nullptr;    /* [N] interface slot of a node or -1 (ragged patches on a partition)          */
    std::vector<int32_t> h_flags;    /* host copy of the patches' flags (hq_patch_set_interface edits them)     */
    int32_t  nlattice = 0;           /* lattice patches                                                        */
    int32_t  nrows = 0;              /* rows of hq_k_patch_pers' LDS image                                     */
    int32_t  grid_cus = 256;         /* persistent workgroups to launch: the device's CU count, a multiple of 8 */
    int32_t  max_nacc = 0;           /* accumulator rows the patches need (owned + hanging nodes on owned anchors; 729 for lattices) */
    bool     seeded = false;         /* hq_k_patch_seed: nt3 carries negative mass_simple for nodes whose seed is 0 */
    int32_t  pipe = 6;               /* HQ_PATCH_PIPE at plan time                                              */
    std::vector<char> patch_lat;     /* host copy of the lattice flags                                         */
    int32_t  nb = 0;
    int32_t* d_ds_ptr = nullptr;     /* hanging-node force distribution (compute_adjust) per patch */
    int32_t* d_ds_ent = nullptr;
    int32_t  max_nown = 0, max_nhalo = 0, max_npairs = 0;
    int32_t  step_nl = 0;            /* rows of hq_k_patch_step's LDS image: max over patches of owned + halo nodes */
    std::vector<int32_t> h_halo;     /* host copies kept only for meshes with hanging nodes */
    std::vector<int64_t> h_halo_off;
    std::vector<int32_t> h_nvirt;
    int64_t  n0 = 0;                 /* the nodes [0, n0) are brick nodes (hq_brick.h): no patch owns them */
    std::vector<int32_t> patch_base; /* host copy of desc[].base for lookups */
    std::vector<int32_t> patch_nown;
};

/* HQ_PATCH_PIPE: 6 (default) = hq_k_patch_seed where the plan fits it (else hq_k_patch_pers, else hq_k_patch_step),
 * 4 = hq_k_patch_pers where it fits, 0 = hq_k_patch_step always */
static int hq_patch_kernel_choice(void)              /* read when a plan is built and kept in it (hq_patch_plan.pipe) */
{
    return hq_opt_int("HQ_PATCH_PIPE", 6);
}

static thread_local std::string g_patch_err;
static const char* hq_patch_error(void) { return g_patch_err.c_str(); }

/* ------------------------------------------------------------------------ */
/* planner (host)                                                           */
/* ------------------------------------------------------------------------ */

static inline uint64_t hq_spread3(uint64_t v)
{
    v &= 0x1fffffULL;
    v = (v | (v << 32)) & 0x1f00000000ffffULL;
    v = (v | (v << 16)) & 0x1f0000ff0000ffULL;
    v = (v | (v << 8))  & 0x100f00f00f00f00fULL;
    v = (v | (v << 4))  & 0x10c30c30c30c30c3ULL;
    v = (v | (v << 2))  & 0x1249249249249249ULL;
    return v;
}

/*
 * LATTICE patches.  In a uniformly refined region a patch is 8x8x8 owned nodes, their one ring
 * (a 10x10x10 node lattice, (I, J, K) in [0,10)^3 with the owned nodes at 1..8) and 9x9x9 elements.
 * LDS operations are served in groups of 32 lanes; a b64 access to 24-byte rows is one pass when the
 * 32 rows are distinct modulo 32 (MI355X_MICROARCH.md, LDS; profiles/micro/lds_ops.hip: 2.3 against
 * 4.0 cycles per ds_read_b64, 8 against 16 per ds_add_f64 for id-ordered rows).  With
 *
 *     element (ei, ej, ek)  ->  lane  ei + 9 ej + 81 ek                 (729 dense lanes, 12 waves)
 *     node    (I, J, K)     ->  row   = I + 9 J + 81 K  (mod 32)
 *
 * corner c = (di, dj, dk) of lane L is the row  L + di + 9 dj + 81 dk  (mod 32): 32 consecutive
 * lanes hit 32 different classes for every corner, in the gathers and in the atomics alike.  Nodes
 * with I, J < 9 sit at exactly I + 9 J + 81 K (rows 0..809, so the accumulator of an owned node is its
 * row, < 729); the 190 nodes of the I = 9 and J = 9 faces take the first free row of their class
 * behind them (1051 rows in all).  The halo list of such a patch is kept in one canonical order
 * (neighbour cube by cube, Z-order inside), so thread t's row is the same for every lattice patch and
 * is read once per launch (lat_row), and all of them share ONE element-row block.
 */
#define HQ_LAT_ROWS 1056         /* image rows of a lattice patch (1051, rounded up to a multiple of 32) */
#define HQ_LAT_ACC  729          /* accumulator rows: owned nodes sit at rows < 729                      */
#define HQ_LAT_NOWN 512
#define HQ_LAT_NHALO 488
#define HQ_LAT_NELEM 729
#define HQ_PIDX_ACC 0x8000u      /* element-row entry: local row | HQ_PIDX_ACC where the patch accumulates that corner */
#define HQ_PIDX_ROW 0x7fffu

struct hq_lattice_tab {
    uint16_t row_of_ijk[1000];   /* I + 10 J + 100 K -> LDS row                               */
    int16_t  local_of_ijk[1000]; /* I + 10 J + 100 K -> canonical local node (owned: Z-order) */
    uint16_t row_of_local[1024]; /* canonical local node -> LDS row (pad: own index)          */
    uint16_t pidx[HQ_LAT_NELEM][8];
};

static const hq_lattice_tab& hq_lattice(void)
{
    static const hq_lattice_tab tab = [] {
        hq_lattice_tab t;
        std::vector<char> used(HQ_LAT_ROWS + 64, 0);
        for (int K = 0; K < 10; K++)
            for (int J = 0; J < 9; J++)
                for (int I = 0; I < 9; I++) {
                    int r = I + 9 * J + 81 * K;
                    t.row_of_ijk[I + 10 * J + 100 * K] = (uint16_t)r;
                    used[r] = 1;
                }
        for (int K = 0; K < 10; K++)
            for (int J = 0; J < 10; J++)
                for (int I = 0; I < 10; I++) {
                    if (I < 9 && J < 9) continue;
                    int r = (I + 9 * J + 81 * K) % 32;
                    while (used[r]) r += 32;
                    used[r] = 1;
                    t.row_of_ijk[I + 10 * J + 100 * K] = (uint16_t)r;
                }
        /* canonical local numbering: owned nodes in Z-order, then the halo neighbour cube by
         * neighbour cube (z, y, x of the cube slowest to fastest), Z-order inside each */
        auto deint = [](int m, int& x, int& y, int& z) {
            x = y = z = 0;
            for (int b = 0; b < 3; b++) {
                x |= ((m >> (3 * b)) & 1) << b; y |= ((m >> (3 * b + 1)) & 1) << b; z |= ((m >> (3 * b + 2)) & 1) << b;
            }
        };
        for (int i = 0; i < 1000; i++) t.local_of_ijk[i] = -1;
        for (int i = 0; i < 1024; i++) t.row_of_local[i] = (uint16_t)i;
        int nl = 0;
        for (int m = 0; m < 512; m++) {
            int x, y, z;
            deint(m, x, y, z);
            t.local_of_ijk[(x + 1) + 10 * (y + 1) + 100 * (z + 1)] = (int16_t)nl++;
        }
        for (int bz = -1; bz <= 1; bz++)
            for (int by = -1; by <= 1; by++)
                for (int bx = -1; bx <= 1; bx++) {
                    if (!bx && !by && !bz) continue;
                    for (int m = 0; m < 512; m++) {
                        int x, y, z;
                        deint(m, x, y, z);
                        /* position of the neighbour cube's node (x, y, z) on this patch's lattice */
                        const int I = 1 + 8 * bx + x, J = 1 + 8 * by + y, K = 1 + 8 * bz + z;
                        if (I < 0 || I > 9 || J < 0 || J > 9 || K < 0 || K > 9) continue;
                        t.local_of_ijk[I + 10 * J + 100 * K] = (int16_t)nl++;
                    }
                }
        for (int i = 0; i < 1000; i++) t.row_of_local[t.local_of_ijk[i]] = t.row_of_ijk[i];
        for (int ek = 0; ek < 9; ek++)
            for (int ej = 0; ej < 9; ej++)
                for (int ei = 0; ei < 9; ei++)
                    for (int c = 0; c < 8; c++) {
                        const int I = ei + (c & 1), J = ej + ((c >> 1) & 1), K = ek + ((c >> 2) & 1);
                        const bool own = I >= 1 && I <= 8 && J >= 1 && J <= 8 && K >= 1 && K <= 8;
                        t.pidx[ei + 9 * ej + 81 * ek][c] =
                            (uint16_t)(t.row_of_ijk[I + 10 * J + 100 * K] | (own ? HQ_PIDX_ACC : 0));
                    }
        return t;
    }();
    return tab;
}

/*
 * Is patch [base, base + 512) with elements `el` (729) and halo `h` (488 ids) a full lattice?  If so,
 * reorder `el` into lane order and `h` into the canonical halo order.  Checked from the node
 * coordinates alone (node_t.x/y/z): equal edge for all elements, owned nodes in Z-order of the
 * lattice, every element's corner c at (e + bits of c), every halo node on the shell.
 */
static bool hq_lattice_match(int32_t base, const int32_t* lnid, const int32_t* xyz, int32_t* el, std::vector<int32_t>& h)
{
    const hq_lattice_tab& T = hq_lattice();
    const int32_t* e0 = lnid + 8 * (int64_t)el[0];
    const int64_t s = (int64_t)xyz[3 * (int64_t)e0[1]] - xyz[3 * (int64_t)e0[0]];
    if (s <= 0) return false;
    const int64_t O[3] = { xyz[3 * (int64_t)base] - s, xyz[3 * (int64_t)base + 1] - s, xyz[3 * (int64_t)base + 2] - s };
    auto ijk = [&](int32_t n) -> int {
        int q[3];
        for (int d = 0; d < 3; d++) {
            const int64_t v = (int64_t)xyz[3 * (int64_t)n + d] - O[d];
            if (v < 0 || v % s || v / s > 9) return -1;
            q[d] = (int)(v / s);
        }
        return q[0] + 10 * q[1] + 100 * q[2];
    };
    for (int t = 0; t < HQ_LAT_NOWN; t++) {
        const int a = ijk(base + t);
        if (a < 0 || T.local_of_ijk[a] != t) return false;
    }
    int32_t canon[HQ_LAT_NHALO];
    for (int i = 0; i < HQ_LAT_NHALO; i++) canon[i] = -1;
    for (int32_t g : h) {
        const int a = ijk(g);
        if (a < 0) return false;
        const int l = T.local_of_ijk[a] - HQ_LAT_NOWN;
        if (l < 0 || canon[l] >= 0) return false;
        canon[l] = g;
    }
    int32_t lane[HQ_LAT_NELEM];
    for (int i = 0; i < HQ_LAT_NELEM; i++) lane[i] = -1;
    for (int q = 0; q < HQ_LAT_NELEM; q++) {
        const int32_t* id = lnid + 8 * (int64_t)el[q];
        const int a = ijk(id[0]);
        if (a < 0) return false;
        const int ei = a % 10, ej = (a / 10) % 10, ek = a / 100;
        if (ei > 8 || ej > 8 || ek > 8) return false;
        for (int c = 1; c < 8; c++)
            if (ijk(id[c]) != (ei + (c & 1)) + 10 * (ej + ((c >> 1) & 1)) + 100 * (ek + ((c >> 2) & 1))) return false;
        const int L = ei + 9 * ej + 81 * ek;
        if (lane[L] >= 0) return false;
        lane[L] = el[q];
    }
    for (int i = 0; i < HQ_LAT_NELEM; i++) el[i] = lane[i];
    for (int i = 0; i < HQ_LAT_NHALO; i++) h[i] = canon[i];
    return true;
}

/*
 * STENCIL patches.  In a lattice patch whose 729 elements share (c1, c2, beta) the force on an owned node is
 * a 27-point stencil with 3x3 blocks, S = c1 S1 + c2 S2, assembled from the element matrix: the SAME
 * operator -(c1 K1 + c2 K2) w that compute_addforce_effective + damping_addforce apply element by element
 * (stiffness.c:180-237, damping.c:29-103), summed per node instead of per element.  For the trilinear
 * hexahedron on a uniform lattice the blocks have the symmetry of the cube:
 *     S[d][a][a] depends only on (|d_a|; the two other |d|): 6 classes p[0..5] =
 *         (0;0,0) (1;0,0) (0;one 1) (1;one 1) (0;1,1) (1;1,1)
 *     S[d][a][b] = q[|d_c|] sgn(d_a) sgn(d_b)   (c the third axis; zero unless d_a and d_b are both != 0)
 * so a node costs 81 LDS reads and 153 fp64 FMAs instead of the 1.42 x (24 reads + 17 atomics, 230 fp64
 * instructions) of the element form, with no atomics and no accumulator.  The sixteen numbers are not typed
 * in: hq_stencil() assembles S from the kernels' own element arithmetic (hq_element_force run on the host on
 * unit vectors) and checks the symmetry; if the check failed no patch would be marked.
 * LDS rows of a stencil patch: node (I, J, K) of the 10x10x10 lattice at 10 I + 104 J + K (1036 rows): the
 * 32 consecutive owned nodes of a lane group are a 4x4x2 block of the Z-order, whose rows are then distinct
 * modulo 32 for every one of the 27 neighbour offsets (the smallest such pitches, found by search).
 */
#define HQ_ST_PX 10
#define HQ_ST_PY 104
#define HQ_ST_PZ 1
#define HQ_ST_ROWS 1040
struct hq_stencil_coef {
    double p1[6], p2[6];         /* diagonal-block classes of S1, S2 */
    double q1[2], q2[2];         /* off-diagonal magnitudes of S1, S2 for |d_c| = 0, 1 */
};
struct hq_stencil_tab {
    bool ok;
    bool face_ok;                    /* the half-space form of the stencil is what hq_k_brick's face planes assume (below) */
    hq_stencil_coef c;
    double E1[576], E2[576];         /* element matrix for (c1, c2) = (1, 0) / (0, 1): E[((o * 8 + m) * 3 + a) * 3 + b] = force on
                                      * corner o, component a, per unit displacement of corner m, component b */
};

static const hq_stencil_tab& hq_stencil(void)
{
    static const hq_stencil_tab tab = [] {
        hq_stencil_tab t;
        t.ok = true;
        for (int which = 0; which < 2; which++) {
            /* element matrix E[(n,a)][(m,b)] for (c1, c2) = (1, 0) / (0, 1): columns = forces of unit displacements */
            double E[24][24];
            for (int m = 0; m < 8; m++)
                for (int b = 0; b < 3; b++) {
                    double X[8] = { 0 }, Y[8] = { 0 }, Z[8] = { 0 };
                    (b == 0 ? X : b == 1 ? Y : Z)[m] = 1.0;
                    hq_element_force(X, Y, Z, which == 0 ? 1.0 : 0.0, which == 0 ? 0.0 : 1.0);
                    for (int n = 0; n < 8; n++) { E[3 * n][3 * m + b] = X[n]; E[3 * n + 1][3 * m + b] = Y[n]; E[3 * n + 2][3 * m + b] = Z[n]; }
                }
            {
                double* Et = which == 0 ? t.E1 : t.E2;
                for (int o = 0; o < 8; o++)
                    for (int m = 0; m < 8; m++)
                        for (int a = 0; a < 3; a++)
                            for (int b = 0; b < 3; b++) Et[((o * 8 + m) * 3 + a) * 3 + b] = E[3 * o + a][3 * m + b];
            }
            /* the node is corner o of the element whose low corner sits at offset -o (bit k of o set: offset -1 along k);
             * the neighbour at offset d is corner m of that element with m_k = d_k + o_k */
            double S[3][3][3][3][3] = {};
            for (int o = 0; o < 8; o++)
                for (int m = 0; m < 8; m++) {
                    int d[3];
                    for (int k = 0; k < 3; k++) d[k] = -((o >> k) & 1) + ((m >> k) & 1);
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) S[d[0] + 1][d[1] + 1][d[2] + 1][a][b] += E[3 * o + a][3 * m + b];
                }
            double* p = which == 0 ? t.c.p1 : t.c.p2;
            double* q = which == 0 ? t.c.q1 : t.c.q2;
            p[0] = S[1][1][1][0][0]; p[1] = S[2][1][1][0][0]; p[2] = S[1][2][1][0][0];
            p[3] = S[2][2][1][0][0]; p[4] = S[1][2][2][0][0]; p[5] = S[2][2][2][0][0];
            q[0] = S[2][2][1][0][1]; q[1] = S[2][2][2][0][1];
            double scale = 0;
            for (int i = 0; i < 6; i++) scale = std::max(scale, fabs(p[i]));
            auto cls = [](int a, int b, int c) { return (a ? 1 : 0) + 2 * ((b != 0) + (c != 0)); };
            for (int dx = -1; dx <= 1; dx++)
                for (int dy = -1; dy <= 1; dy++)
                    for (int dz = -1; dz <= 1; dz++) {
                        const int d[3] = { dx, dy, dz };
                        for (int a = 0; a < 3; a++)
                            for (int b = 0; b < 3; b++) {
                                const double v = S[dx + 1][dy + 1][dz + 1][a][b];
                                double want;
                                if (a == b) want = p[cls(d[a], d[(a + 1) % 3], d[(a + 2) % 3])];
                                else { const int c = 3 - a - b; want = (d[a] && d[b]) ? q[d[c] ? 1 : 0] * d[a] * d[b] : 0.0; }
                                if (fabs(v - want) > 1e-13 * scale) t.ok = false;
                            }
                    }
            /* A node on a domain face normal to z, its four elements on the +z side (the node is their low-z corner, o_z = 0):
             * its coupling to its own plane is  H(dx, dy) = S(dx, dy, 0) / 2 + T(dx, dy)  -- the part of S even under the
             * mirror z -> -z halves, the odd part T does not cancel any more -- and T couples z with x and y only,
             * antisymmetrically:  T[x][z] = -T[z][x] = r(|dy|) dx,  T[y][z] = -T[z][y] = r(|dx|) dy,  with r = +q for S1 and
             * r = -q for S2.  Hence, with q1 == q2 (checked), the face term is  rho (Uo_x, Uo_y, -Uo_z)  of the node's OWN
             * plane, rho = (c1 - c2) / (c1 + c2), Uo the odd-in-dz sums hq_k_brick forms for every plane anyway. */
            if (which == 0) t.face_ok = true;
            {
                double H[3][3][3][3] = {};
                for (int o = 0; o < 4; o++)                      /* o_z = 0 */
                    for (int m = 0; m < 4; m++) {                /* m_z = 0: dz = 0 */
                        const int dx = -(o & 1) + (m & 1), dy = -((o >> 1) & 1) + ((m >> 1) & 1);
                        for (int a = 0; a < 3; a++)
                            for (int b = 0; b < 3; b++) H[dx + 1][dy + 1][a][b] += E[3 * o + a][3 * m + b];
                    }
                const double sign = which == 0 ? 1.0 : -1.0;
                for (int dx = -1; dx <= 1; dx++)
                    for (int dy = -1; dy <= 1; dy++)
                        for (int a = 0; a < 3; a++)
                            for (int b = 0; b < 3; b++) {
                                double want = 0.5 * S[dx + 1][dy + 1][1][a][b];
                                if (a == 0 && b == 2) want += sign * q[dy ? 1 : 0] * dx;
                                if (a == 2 && b == 0) want -= sign * q[dy ? 1 : 0] * dx;
                                if (a == 1 && b == 2) want += sign * q[dx ? 1 : 0] * dy;
                                if (a == 2 && b == 1) want -= sign * q[dx ? 1 : 0] * dy;
                                if (fabs(H[dx + 1][dy + 1][a][b] - want) > 1e-13 * scale) t.face_ok = false;
                            }
            }
        }
        for (int i = 0; i < 2; i++)
            if (fabs(t.c.q1[i] - t.c.q2[i]) > 1e-14 * fabs(t.c.q1[i])) t.face_ok = false;
        if (!t.ok) t.face_ok = false;
        return t;
    }();
    return tab;
}

/*
 * STENCIL patches: the owned nodes (<= 729) and the ring (<= 512) of a patch lie on the 10x10x10 lattice of hq_stencil
 * -- all of it, or (RAGGED) not: the domain ends at a face, the partition ends, or the far-boundary layer makes the
 * patch 9 nodes wide.  Table of such a patch (uint32, shared between patches of one shape):
 *     nloc x: per local node (owned in id order, then the halo list) its lattice row 10 I + 104 J + K, for an owned
 *             node + 2^11 x the mask of its present elements (bit o: the element whose corner o the node is) + 2^19 x
 *             its index in the boundary list;
 *     nbnd x: the owned nodes with an incomplete mask (<= 256), the same word.
 * -> false if the patch is not a lattice subset (then it stays in the element form).
 */
#define HQ_RG_ROW(w) ((int)((w) & 0x7ffu))
#define HQ_RG_MASK(w) (((w) >> 11) & 0xffu)
#define HQ_RG_BIDX(w) ((int)(((w) >> 19) & 0xffu))
static bool hq_ragged_match(int32_t base, int32_t nown, const int32_t* lnid, const int32_t* xyz, const int32_t* el,
                            int32_t npairs, const std::vector<int32_t>& h, std::vector<uint32_t>& tab, int32_t* nbnd)
{
    if (nown < 1 || nown > 729 || npairs < 1 || npairs > 729 || h.empty() || h.size() > (nown > 512 ? 768u : 512u) || nown + (int32_t)h.size() > 1000) return false;
    const int32_t* e0 = lnid + 8 * (int64_t)el[0];
    const int64_t s = (int64_t)xyz[3 * (int64_t)e0[1]] - xyz[3 * (int64_t)e0[0]];
    if (s <= 0) return false;
    int64_t O[3];
    for (int d = 0; d < 3; d++) {
        int64_t mn = xyz[3 * (int64_t)base + d];
        for (int32_t t = 1; t < nown; t++) mn = std::min<int64_t>(mn, xyz[3 * ((int64_t)base + t) + d]);
        O[d] = mn - s;                                  /* the owned nodes start at lattice coordinate 1 */
    }
    auto ijk = [&](int32_t n, int q[3]) -> bool {
        for (int d = 0; d < 3; d++) {
            const int64_t v = (int64_t)xyz[3 * (int64_t)n + d] - O[d];
            if (v < 0 || v % s || v / s > 9) return false;
            q[d] = (int)(v / s);
        }
        return true;
    };
    const int32_t nloc = nown + (int32_t)h.size();
    std::vector<uint16_t> row((size_t)nloc);
    std::vector<int> own_ijk((size_t)nown);
    for (int32_t t = 0; t < nloc; t++) {
        int q[3];
        if (!ijk(t < nown ? base + t : h[(size_t)(t - nown)], q)) return false;
        if (t < nown) {
            if (q[0] < 1 || q[1] < 1 || q[2] < 1) return false;
            own_ijk[(size_t)t] = q[0] + 10 * q[1] + 100 * q[2];
        }
        row[(size_t)t] = (uint16_t)(HQ_ST_PX * q[0] + HQ_ST_PY * q[1] + HQ_ST_PZ * q[2]);
    }
    bool present[729] = { false };
    for (int32_t q = 0; q < npairs; q++) {
        const int32_t* id = lnid + 8 * (int64_t)el[q];
        int e[3];
        if (!ijk(id[0], e) || e[0] > 8 || e[1] > 8 || e[2] > 8) return false;
        for (int c = 1; c < 8; c++) {
            int v[3];
            if (!ijk(id[c], v) || v[0] != e[0] + (c & 1) || v[1] != e[1] + ((c >> 1) & 1) || v[2] != e[2] + ((c >> 2) & 1)) return false;
        }
        present[e[0] + 9 * e[1] + 81 * e[2]] = true;
    }
    std::vector<uint8_t> mask((size_t)nown);
    std::vector<uint16_t> bidx((size_t)nown, 0xffff), blist;
    for (int32_t t = 0; t < nown; t++) {
        const int I = own_ijk[(size_t)t] % 10, J = (own_ijk[(size_t)t] / 10) % 10, K = own_ijk[(size_t)t] / 100;
        uint8_t m = 0;
        for (int o = 0; o < 8; o++) {
            const int ei = I - (o & 1), ej = J - ((o >> 1) & 1), ek = K - ((o >> 2) & 1);
            if (ei >= 0 && ei < 9 && ej >= 0 && ej < 9 && ek >= 0 && ek < 9 && present[ei + 9 * ej + 81 * ek]) m |= (uint8_t)(1 << o);
        }
        if (m == 0) return false;                       /* a node no element touches */
        mask[(size_t)t] = m;
        if (m != 0xff) { bidx[(size_t)t] = (uint16_t)blist.size(); blist.push_back((uint16_t)t); }
    }
    if (blist.size() > 256) return false;               /* the boundary list is worked off by four waves at most */
    tab.clear();
    for (int32_t t = 0; t < nloc; t++)
        tab.push_back((uint32_t)row[(size_t)t] | (t < nown ? ((uint32_t)mask[(size_t)t] << 11) | ((uint32_t)(bidx[(size_t)t] & 0xff) << 19) : 0u));
    for (uint16_t t : blist) tab.push_back((uint32_t)row[t] | ((uint32_t)mask[t] << 11));
    while (tab.size() & 3) tab.push_back(0);             /* 16-byte granules */
    *nbnd = (int32_t)blist.size();
    return true;
}

/*
 * Cut [0,N) into runs of consecutive nodes.  With coordinates (node_t.x/y/z,
 * octor.h:133-147) the cuts follow aligned octree cubes holding at most PMAX
 * nodes; without them (or if the numbering is not Z-ordered) fixed runs.
 * n0: the nodes [0, n0) are brick nodes (hq_brick.h) and belong to no patch.
 */
static void hq_patch_cuts(const hq_patch_cfg& cfg, int64_t N, const int32_t* xyz, std::vector<int32_t>& cuts, int64_t n0 = 0)
{
    cuts.clear();
    auto fixed = [&]() {
        cuts.clear();
        for (int64_t i = n0; i < N; i += cfg.pmerge) cuts.push_back((int32_t)i);
        cuts.push_back((int32_t)N);
    };
    if (n0 >= N) { cuts.push_back((int32_t)N); return; }       /* every node is a brick node: no patch */
    if (!xyz) { fixed(); return; }

    uint32_t orall = 0;
    int32_t maxc[3] = { 0, 0, 0 };
    for (int64_t n = n0; n < N; n++)
        for (int d = 0; d < 3; d++) {
            int32_t v = xyz[3 * n + d];
            if (v < 0) { fixed(); return; }
            orall |= (uint32_t)v;
            maxc[d] = std::max(maxc[d], v);
        }
    int m = orall ? __builtin_ctz(orall) : 0;          /* common edge granularity 2^m ticks */
    std::vector<uint64_t> key((size_t)(N - n0));           /* of the nodes n0 .. N - 1 (the brick nodes below have no patch) */
    /* Nodes on the far boundary of the DOMAIN sort one tick inwards (octor.c:6100-6106).  On a partition
     * the largest coordinate of an axis is the domain's far face only for the partitions that touch it
     * (elsewhere those nodes belong to the next partition's cells and sort there), so: the set of axes
     * whose maximum is treated as far boundary is the first -- all three first -- under which the ids are
     * in Z-order. */
    bool sorted = false;
    for (int mask = 7; mask >= 0 && !sorted; mask--) {
        sorted = true;
        for (int64_t n = n0; n < N && sorted; n++) {
            uint64_t q[3];
            for (int d = 0; d < 3; d++) {
                int32_t v = xyz[3 * n + d];
                if (((mask >> d) & 1) && v == maxc[d] && v > 0) v -= 1;
                q[d] = (uint64_t)(v >> m);
                if (q[d] >> 21) { fixed(); return; }
            }
            key[n - n0] = hq_spread3(q[0]) | (hq_spread3(q[1]) << 1) | (hq_spread3(q[2]) << 2);
            if (n > n0 && key[n - n0] < key[n - n0 - 1]) sorted = false;
        }
    }
    if (!sorted) { fixed(); return; }                  /* not Z-ordered */

    /* k-d style descent over the bits of the Z-value (z, y, x of the coarsest level first):
     * a run of nodes sharing a key prefix is an axis-aligned box; split it at the next bit
     * until it holds at most pmax nodes.  8x8x8 -> 8x8x4 -> 8x4x4 -> 4x4x4 ... */
    std::vector<std::pair<int32_t, int32_t>> runs;
    struct item { int64_t lo, hi; int bit; };
    std::vector<item> stack;
    stack.push_back({ n0, N, 63 });
    while (!stack.empty()) {
        item it = stack.back();
        stack.pop_back();
        if (it.hi - it.lo <= cfg.psplit) { runs.push_back({ (int32_t)it.lo, (int32_t)it.hi }); continue; }
        if (it.bit <= 0) {
            for (int64_t i = it.lo; i < it.hi; i += cfg.pmerge)
                runs.push_back({ (int32_t)i, (int32_t)std::min<int64_t>(i + cfg.pmerge, it.hi) });
            continue;
        }
        int sh = it.bit - 1;
        /* first node whose bit `sh` is set (keys are sorted and share the bits above) */
        int64_t mid = std::partition_point(key.begin() + (it.lo - n0), key.begin() + (it.hi - n0),
                                           [sh](uint64_t k) { return ((k >> sh) & 1) == 0; }) - key.begin() + n0;
        if (mid < it.hi) stack.push_back({ mid, it.hi, sh });
        if (mid > it.lo) stack.push_back({ it.lo, mid, sh });
    }
    std::sort(runs.begin(), runs.end());
    /* merge small neighbours (coarse octree regions) */
    cuts.push_back((int32_t)n0);
    int32_t cur = 0;
    for (auto& r : runs) {
        int32_t n = r.second - r.first;
        if (cur > 0 && cur + n > cfg.pmerge) { cuts.push_back(r.first); cur = 0; }
        cur += n;
    }
    cuts.push_back((int32_t)N);
}

/*
 * Pair lists and local numbering for the node runs `cuts`.  A run whose halo
 * does not fit LDS is halved and the build repeated.
 */
/* the elements a patch can need when the nodes below n0 are brick nodes: those with a corner at or above n0, or a
 * hanging corner (its anchors' owners evaluate the element too) -- in ascending order.  Behind bricks that is the shell
 * of the mesh, a few per cent of it. */
static void hq_patch_candidates(int64_t E, const int32_t* lnid, const hq_dangling& dn, const std::vector<int32_t>& dn_of,
                                int64_t n0, std::vector<int32_t>& cand)
{
    cand.clear();
    int nth = 1;
#pragma omp parallel
    {
#pragma omp single
        nth = omp_get_num_threads();
    }
    /* slices are dealt by a worksharing loop, not by thread number: whatever team the runtime delivers, every slice
     * is scanned (round-3 advisor finding) */
    std::vector<std::vector<int32_t>> part((size_t)nth);
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < nth; t++) {
        const int64_t lo = E * t / nth, hi = E * (t + 1) / nth;
        std::vector<int32_t>& v = part[(size_t)t];
        for (int64_t e = lo; e < hi; e++) {
            const int32_t* id = lnid + 8 * e;
            bool take = false;
            for (int c = 0; c < 8 && !take; c++) take = id[c] >= n0 || (dn.n > 0 && dn_of[(size_t)id[c]] >= 0);
            if (take) v.push_back((int32_t)e);
        }
    }
    for (auto& v : part) cand.insert(cand.end(), v.begin(), v.end());
}

static int hq_patch_plan_host(const hq_patch_cfg& cfg, int64_t E, int64_t N, const int32_t* lnid,
                              const int32_t* xyz, const hq_dangling& dn, bool want_lattice, hq_patch_host* H, int64_t n0 = 0,
                              std::vector<int32_t>* cand_cache = nullptr)
{
    /* HQ_PATCH_VERBOSE >= 3: where the planner's own time goes */
    const bool lap_on = hq_opt_int("HQ_PATCH_VERBOSE", 0) > 2;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!lap_on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "    patch planner: %-30s %6.3f s\n", what, std::chrono::duration<double>(now - lap_t).count());
        lap_t = now;
    };
    std::vector<int32_t> cuts;
    hq_patch_cuts(cfg, N, xyz, cuts, n0);
    lap("cuts");
    /* hanging nodes: dn_of[n] = index into the dangling table or -1 */
    std::vector<int32_t> dn_of;
    if (dn.n > 0) {
        dn_of.assign((size_t)N, -1);
        for (int32_t k = 0; k < dn.n; k++) dn_of[dn.id[k]] = k;
        for (int32_t k = 0; k < dn.n; k++)
            for (int32_t a = dn.ptr[k]; a < dn.ptr[k + 1]; a++)
                if (dn_of[dn.anchor[a]] >= 0) { g_patch_err = "an anchor is itself a hanging node"; return -1; }
    }

    /* behind bricks only the shell's elements are looked at (every attempt below walks the element list twice) */
    std::vector<int32_t> cand_own;
    std::vector<int32_t>* cand = nullptr;
    if (n0 > 0) {
        cand = cand_cache ? cand_cache : &cand_own;
        if (cand->empty()) hq_patch_candidates(E, lnid, dn, dn_of, n0, *cand);
    }
    const int64_t NE = cand ? (int64_t)cand->size() : E;
    auto elem_at = [&](int64_t i) -> int64_t { return cand ? (int64_t)(*cand)[(size_t)i] : i; };
    lap("hanging-node table, candidates");

    for (int attempt = 0; attempt < 12; attempt++) {
        int32_t P = (int32_t)cuts.size() - 1;
        std::vector<int32_t> patch_of((size_t)N, -1);            /* -1: a brick node */
        for (int32_t p = 0; p < P; p++)
            for (int32_t n = cuts[p]; n < cuts[p + 1]; n++) patch_of[n] = p;

        /* count (patch, element) pairs */
        std::vector<int64_t> off((size_t)P + 1, 0);
        /* patches that must evaluate element e: the owners of its nodes and, for a hanging
         * node, the owners of its anchors (they need its complete force, psolve.c:5942-5987) */
        auto patches_of_elem = [&](int64_t e, int32_t out[40]) {
            int k = 0;
            auto add = [&](int32_t p) {
                if (p < 0) return;
                bool seen = false;
                for (int t = 0; t < k; t++) seen |= (out[t] == p);
                if (!seen) out[k++] = p;
            };
            for (int c = 0; c < 8; c++) {
                int32_t n = lnid[8 * e + c];
                add(patch_of[n]);
                if (dn.n > 0 && dn_of[n] >= 0)
                    for (int32_t a = dn.ptr[dn_of[n]]; a < dn.ptr[dn_of[n] + 1]; a++) add(patch_of[dn.anchor[a]]);
            }
            return k;
        };
        for (int64_t i = 0; i < NE; i++) {
            int32_t ps[40];
            int k = patches_of_elem(elem_at(i), ps);
            for (int t = 0; t < k; t++) off[ps[t] + 1]++;
        }
        for (int32_t p = 0; p < P; p++) off[p + 1] += off[p];
        int64_t npairs = off[P];
        lap("patch_of, pair counts");
        H->pelem.assign((size_t)npairs, 0);
        {
            std::vector<int64_t> fill(off.begin(), off.end() - 1);
            for (int64_t i = 0; i < NE; i++) {
                const int64_t e = elem_at(i);
                int32_t ps[40];
                int k = patches_of_elem(e, ps);
                for (int t = 0; t < k; t++) H->pelem[(size_t)fill[ps[t]]++] = (int32_t)e;
            }
        }

        lap("pair lists");
        /* halo lists */
        std::vector<int64_t> hoff((size_t)P + 1, 0);
        std::vector<std::vector<int32_t>> halos((size_t)P);
        std::vector<int32_t> nvirt((size_t)P, 0);
        std::vector<char> bad((size_t)P, 0);
        bool any_bad = false;
#pragma omp parallel for schedule(dynamic, 64)
        for (int32_t p = 0; p < P; p++) {
            int32_t base = cuts[p], nown = cuts[p + 1] - cuts[p];
            std::vector<int32_t>& h = halos[p];
            h.clear();
            for (int64_t q = off[p]; q < off[p + 1]; q++) {
                const int32_t* id = lnid + 8 * (int64_t)H->pelem[(size_t)q];
                for (int c = 0; c < 8; c++)
                    if (id[c] < base || id[c] >= base + nown) h.push_back(id[c]);
            }
            std::sort(h.begin(), h.end());
            h.erase(std::unique(h.begin(), h.end()), h.end());
            if (dn.n > 0) {
                /* halo order: first the hanging nodes that hang on an anchor this patch owns
                 * ("virtual" accumulators), then the rest; both ascending */
                auto hangs_here = [&](int32_t n) {
                    if (dn_of[n] < 0) return false;
                    for (int32_t a = dn.ptr[dn_of[n]]; a < dn.ptr[dn_of[n] + 1]; a++)
                        if (dn.anchor[a] >= base && dn.anchor[a] < base + nown) return true;
                    return false;
                };
                auto mid = std::stable_partition(h.begin(), h.end(), hangs_here);
                nvirt[p] = (int32_t)(mid - h.begin());
            }
            if (nown + (int64_t)h.size() > cfg.nlmax || nown > cfg.pmax || nvirt[p] > cfg.vmax ||
                off[p + 1] - off[p] > 0x7fffffff)
                bad[p] = 1;
        }
        for (int32_t p = 0; p < P; p++) any_bad |= (bad[p] != 0);
        lap("halo lists");
        if (any_bad) {
            std::vector<int32_t> nc;
            for (int32_t p = 0; p < P; p++) {
                nc.push_back(cuts[p]);
                if (bad[p]) {
                    int32_t n = cuts[p + 1] - cuts[p];
                    if (n < 2) { g_patch_err = "a single node reaches more neighbours than LDS can stage"; return -1; }
                    nc.push_back(cuts[p] + n / 2);
                }
            }
            nc.push_back((int32_t)N);
            cuts.swap(nc);
            continue;
        }

        /* halo id lists at a fixed stride: a workgroup finds its list from its patch number alone
         * and reads it while its descriptor is still in flight (one memory latency less in the
         * dependent chain descriptor -> ids -> node data), and can be prefetched by an earlier
         * workgroup.  Padded so the unconditional first-round id loads stay inside the table. */
        size_t hs = 32;
        for (int32_t p = 0; p < P; p++) hs = std::max(hs, (halos[p].size() + 31) / 32 * 32);
        H->hstride = (int32_t)hs;
        for (int32_t p = 0; p <= P; p++) hoff[p] = (int64_t)p * (int64_t)hs;
        H->halo.assign((size_t)hoff[P] + 4 * HQ_PATCH_MAX_THREADS / 3 + 64, 0);
        H->desc.assign((size_t)P, hq_patch_desc());
        H->pidx.assign((size_t)npairs * 8, 0);
        H->lattice.assign((size_t)P, 0);
        /* a patch with a hanging-node distribution entry keeps the id-ordered rows (its entries name them) */
        std::vector<char> has_ds((size_t)P, 0);
        for (int32_t k = 0; k < dn.n; k++)
            for (int32_t a = dn.ptr[k]; a < dn.ptr[k + 1]; a++) has_ds[patch_of[dn.anchor[a]]] = 1;
#pragma omp parallel for schedule(dynamic, 64)
        for (int32_t p = 0; p < P; p++) {
            hq_patch_desc& D = H->desc[p];
            D.base = cuts[p];
            D.nown = cuts[p + 1] - cuts[p];
            D.nhalo = (int32_t)halos[p].size();
            D.npairs = (int32_t)(off[p + 1] - off[p]);
            D.pair_off = off[p];
            D.halo_off = hoff[p];
            D.nacc = D.nown + nvirt[p];
            if (want_lattice && xyz && D.nown == HQ_LAT_NOWN && D.nhalo == HQ_LAT_NHALO && D.npairs == HQ_LAT_NELEM &&
                nvirt[p] == 0 && !has_ds[p] &&
                hq_lattice_match(D.base, lnid, xyz, &H->pelem[(size_t)off[p]], halos[p])) {
                H->lattice[p] = 1;
                memcpy(&H->pidx[(size_t)off[p] * 8], hq_lattice().pidx, sizeof(uint16_t) * 8 * HQ_LAT_NELEM);
                std::copy(halos[p].begin(), halos[p].end(), H->halo.begin() + hoff[p]);
                continue;
            }
            std::copy(halos[p].begin(), halos[p].end(), H->halo.begin() + hoff[p]);
            const std::vector<int32_t>& h = halos[p];
            const int32_t nv = nvirt[p];
            auto local_of = [&](int32_t g) -> int32_t {
                if (g >= D.base && g < D.base + D.nown) return g - D.base;
                auto v = std::lower_bound(h.begin(), h.begin() + nv, g);
                if (v != h.begin() + nv && *v == g) return D.nown + (int32_t)(v - h.begin());
                return D.nown + (int32_t)(std::lower_bound(h.begin() + nv, h.end(), g) - h.begin());
            };
            for (int64_t q = off[p]; q < off[p + 1]; q++) {
                const int32_t* id = lnid + 8 * (int64_t)H->pelem[(size_t)q];
                for (int c = 0; c < 8; c++) {
                    const int32_t l = local_of(id[c]);
                    H->pidx[(size_t)q * 8 + c] = (uint16_t)(l | (l < D.nacc ? (int32_t)HQ_PIDX_ACC : 0));
                }
            }
        }
        lap("element rows");
        /* regular regions repeat one local connectivity: a patch whose rows equal those of an
         * earlier patch reads that patch's rows (which then stay in L2) instead of its own */
        {
            std::unordered_map<uint64_t, std::vector<int32_t>> seen;
            int32_t ndistinct = 0;
            const bool dedup = !hq_opt_flag("HQ_PATCH_NO_DEDUP");
            for (int32_t p = 0; p < P; p++) {
                hq_patch_desc& D = H->desc[p];
                D.pidx_off = D.pair_off;
                if (!dedup || D.npairs == 0) continue;
                const uint16_t* blk = H->pidx.data() + 8 * (size_t)D.pair_off;
                const size_t nb = 16 * (size_t)D.npairs;
                uint64_t h = 1469598103934665603ull ^ (uint64_t)D.npairs;
                const uint64_t* w = reinterpret_cast<const uint64_t*>(blk);
                for (size_t i = 0; i < nb / 8; i++) { h ^= w[i]; h *= 1099511628211ull; }
                auto& cand = seen[h];
                bool found = false;
                for (int32_t r : cand) {
                    const hq_patch_desc& R = H->desc[r];
                    if (R.npairs == D.npairs && !memcmp(H->pidx.data() + 8 * (size_t)R.pair_off, blk, nb)) {
                        D.pidx_off = R.pair_off;
                        found = true;
                        break;
                    }
                }
                if (!found) { cand.push_back(p); ndistinct++; }
            }
            H->ndistinct = ndistinct;
        }
        if (dn.n > 0) {
            /* distribution entries, hanging nodes in table order (the reference's loop order) */
            std::vector<std::vector<int32_t>> ent((size_t)P);
            for (int32_t k = 0; k < dn.n; k++) {
                int32_t deps = dn.ptr[k + 1] - dn.ptr[k];
                for (int32_t a = dn.ptr[k]; a < dn.ptr[k + 1]; a++) {
                    int32_t p = patch_of[dn.anchor[a]];
                    const hq_patch_desc& D = H->desc[p];
                    const std::vector<int32_t>& h = halos[p];
                    int32_t g = dn.id[k], src;
                    if (g >= D.base && g < D.base + D.nown) src = g - D.base;
                    else src = D.nown + (int32_t)(std::lower_bound(h.begin(), h.begin() + nvirt[p], g) - h.begin());
                    ent[p].push_back(src);
                    ent[p].push_back(dn.anchor[a] - D.base);
                    ent[p].push_back(deps);
                }
            }
            H->ds_ptr.assign((size_t)P + 1, 0);
            for (int32_t p = 0; p < P; p++) {
                H->ds_ptr[p + 1] = H->ds_ptr[p] + (int32_t)(ent[p].size() / 3);
                H->ds_ent.insert(H->ds_ent.end(), ent[p].begin(), ent[p].end());
            }
        }
        return 0;
    }
    g_patch_err = "patch refinement did not converge";
    return -1;
}

/* ------------------------------------------------------------------------ */
/* kernel                                                                   */
/* ------------------------------------------------------------------------ */

/* element row (4 dwords): eight 16-bit entries, corner n in half (n & 1) of dword n >> 1 */
#define HQ_PIDX_UNPACK(l, raw)                                                                  \
    l[0] = (raw).x & HQ_PIDX_ROW; l[1] = ((raw).x >> 16) & HQ_PIDX_ROW;                          \
    l[2] = (raw).y & HQ_PIDX_ROW; l[3] = ((raw).y >> 16) & HQ_PIDX_ROW;                          \
    l[4] = (raw).z & HQ_PIDX_ROW; l[5] = ((raw).z >> 16) & HQ_PIDX_ROW;                          \
    l[6] = (raw).w & HQ_PIDX_ROW; l[7] = ((raw).w >> 16) & HQ_PIDX_ROW;
#define HQ_PIDX_WORD(raw, n) ((n) < 2 ? (raw).x : (n) < 4 ? (raw).y : (n) < 6 ? (raw).z : (raw).w)
#define HQ_PIDX_HAS_ACC(raw, n) ((HQ_PIDX_WORD(raw, n) & (((n) & 1) ? (HQ_PIDX_ACC << 16) : HQ_PIDX_ACC)) != 0)

struct hq_pair_data {
    uint4 raw;
    double beta, c1, c2;
};

#ifdef HQ_PATCH_PROFILING
/* DIAG == 6: thread 0 of every workgroup stamps the shader clock at the phase boundaries into
 * g_hq_stamps[patch][8]; hq_patch_report_stamps() prints the mean cycles per phase.  Never in
 * the shipped build. */
__device__ unsigned long long* g_hq_stamps = nullptr;
__device__ unsigned long long* g_hq_wg = nullptr;      /* [grid][2]: shader clock at workgroup start / exit */
#define HQ_WG_STAMP(k) do { if (threadIdx.x == 0 && g_hq_wg) g_hq_wg[2 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
static void hq_patch_report_stamps(void);
#define HQ_STAMP(k) do { if (DIAG == 6 && threadIdx.x == 0 && g_hq_stamps) g_hq_stamps[8 * (size_t)p + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#ifndef HQ_STAMP_TID
#define HQ_STAMP_TID 0     /* the thread whose view of the phases is recorded */
#endif
#define HQ_STAMPD(k) do { if (tid0 == HQ_STAMP_TID && p0 >= 0 && g_hq_stamps) g_hq_stamps[8 * (size_t)p0 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HQ_STAMP(k) do { } while (0)
#define HQ_STAMPD(k) do { } while (0)
#define HQ_WG_STAMP(k) do { } while (0)
#endif

typedef unsigned int hq_u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, typename T>
__device__ __forceinline__ T hq_ld(const T* p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

/* element row: local node ids at pidx[gi], coefficients at [gc] */
template <bool NT>
__device__ __forceinline__ hq_pair_data hq_pair_load(const uint4* __restrict__ pidx, const double* __restrict__ pc1,
                                                     const double* __restrict__ pc2,
                                                     const double* __restrict__ pbeta, int64_t gi, int64_t gc)
{
    hq_pair_data d;
    hq_u32x4 r = hq_ld<NT>(reinterpret_cast<const hq_u32x4*>(pidx) + gi);
    d.raw.x = r.x; d.raw.y = r.y; d.raw.z = r.z; d.raw.w = r.w;
    d.beta = hq_ld<NT>(&pbeta[gc]);
    d.c1 = hq_ld<NT>(&pc1[gc]);
    d.c2 = hq_ld<NT>(&pc2[gc]);
    return d;
}

/*
 * LDS: s_u1[3 nlmax] | s_u2[3 nlmax] | s_f[3 pmax]   (doubles, node-major AoS)
 *
 * Every global load whose address does not depend on LDS contents is issued as
 * early as possible (pair data of the first round before the staging barrier,
 * the next round's before the current round's arithmetic, the nodal constants
 * before the element loop) so the few waves a CU holds keep requests in flight.
 */
/* DIAG != 0 only in -DHQ_PATCH_PROFILING builds (results are wrong): 1 = skip the element
 * loop, 2 = skip the staging loads, 3 = skip the nodal update, 4 = skip the LDS atomics,
 * 5 = conflict-free LDS gathers.  Measured on the 64M box (ms/step): full 3.09, (1) 1.78,
 * (2) 2.14, (3) 2.85, (4) 3.00, (5) 2.83 -- the memory phases alone run at the HBM rate,
 * the element loop adds ~1.3 ms that two workgroups per CU do not overlap (round 2 work). */
template <bool NT, int DIAG>
__global__ void __launch_bounds__(HQ_PATCH_MAX_THREADS, 4)
hq_k_patch_step(int32_t npatches, int32_t per_xcd, const int32_t* __restrict__ order, int32_t nlmax,
                const hq_patch_desc* __restrict__ desc,
                const uint4* __restrict__ pidx, const double* __restrict__ pc1,
                const double* __restrict__ pc2, const double* __restrict__ pbeta,
                const int32_t* __restrict__ halo, const hq_real* __restrict__ u1g,
                const hq_real* __restrict__ u2g, hq_real* __restrict__ ung,
                const double* __restrict__ nt, const double* __restrict__ nt3,
                const int32_t* __restrict__ src_ptr,
                const int32_t* __restrict__ src_ent, const double* __restrict__ F, double dt2,
                const int32_t* __restrict__ if_ptr, const int32_t* __restrict__ if_ent,
                double* __restrict__ iforce, const int32_t* __restrict__ ds_ptr,
                const int32_t* __restrict__ ds_ent, int32_t hstride)
{
    extern __shared__ __align__(16) double s_mem[];
    double* __restrict__ s_u1 = s_mem;
    double* __restrict__ s_u2 = s_mem + 3 * nlmax;
    double* __restrict__ s_f = s_mem + 6 * nlmax;

    /* workgroups b and b+8 share an XCD (round-robin dispatch): give each XCD a
     * contiguous run of Z-ordered patches so halo reads hit its own L2 */
    const int slot = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (slot >= npatches) return;
    const int p = order ? order[slot] : slot;
    const int tid = threadIdx.x, T = blockDim.x;
    HQ_STAMP(0);
    /* first-round halo ids: their address needs only p, so they travel with the descriptor */
    const int32_t* __restrict__ hl = halo + (int64_t)p * hstride;
    int32_t hid[4];
#pragma unroll
    for (int k = 0; k < 4; k++) hid[k] = hq_ld<NT>(&hl[(k * T + tid) / 3]);
    const hq_patch_desc D = desc[p];
    const int own3 = D.nown * 3, halo3 = D.nhalo * 3;
    for (int i = own3 + tid; i < 3 * D.nacc; i += T) s_f[i] = 0.0;   /* hanging nodes on owned anchors */
    if (DIAG == 6 && D.nown > 0) HQ_STAMP(1);

    /* HQ_PATCH_WFORM (uniform beta, owned <= nlmax / 2): the LDS image is w = u1 + beta (u1 - u2) of
     * all local nodes | u1 of the owned | u2 of the owned, which halves the gathers of the element loop */
    const bool wf = (D.flags & HQ_PATCH_WFORM) != 0;
    const double wbeta = wf ? pbeta[D.pair_off] : 0.0;
    const int o2off = 3 * (nlmax / 2);
    hq_pair_data cur;
    const int cstep = (D.flags & HQ_PATCH_UNIFORM) ? 0 : 1;     /* uniform patch: every row reads coefficient 0 */
    if (tid < D.npairs) cur = hq_pair_load<NT>(pidx, pc1, pc2, pbeta, D.pidx_off + tid, D.pair_off + cstep * tid);

    {   /* stage: owned nodes are one contiguous run of doubles, halo nodes a gather */
        const hq_real* g1 = u1g + 3 * (int64_t)D.base;
        const hq_real* g2 = u2g + 3 * (int64_t)D.base;
        for (int i0 = 0; i0 < own3 || i0 < halo3; i0 += 4 * T) {
            double a1[4], a2[4], b1[4], b2[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int i = i0 + k * T + tid;
                if (DIAG == 2) { a1[k] = a2[k] = b1[k] = b2[k] = 1e-3 * i; continue; }
                if (i < own3) { a1[k] = g1[i]; a2[k] = g2[i]; }
                if (i < halo3) {
                    int h = i / 3, d = i - 3 * h;
                    int32_t id = hid[k];
                    if (i0 > 0) id = hq_ld<NT>(&hl[h]);
                    int64_t g = 3 * (int64_t)id + d;
                    b1[k] = u1g[g]; b2[k] = u2g[g];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int i = i0 + k * T + tid;
                if (wf) {
                    if (i < own3) { s_u1[i] = a1[k] + wbeta * (a1[k] - a2[k]); s_u2[i] = a1[k]; s_u2[o2off + i] = a2[k]; s_f[i] = 0.0; }
                    if (i < halo3) s_u1[own3 + i] = b1[k] + wbeta * (b1[k] - b2[k]);
                } else {
                    if (i < own3) { s_u1[i] = a1[k]; s_u2[i] = a2[k]; s_f[i] = 0.0; }
                    if (i < halo3) { s_u1[own3 + i] = b1[k]; s_u2[own3 + i] = b2[k]; }
                }
            }
        }
    }
    HQ_STAMP(2);
    __syncthreads();
    HQ_STAMP(3);

    /* nodal constants of "my" node for the update below: n_t (psolve.h:210-214), or its
     * 3-double form where no dashpot makes the axes differ */
    const bool iso = (D.flags & HQ_PATCH_ISO) != 0;
    double np[7];
    if (tid < D.nown) {
        if (iso) {
            const double* q = nt3 + 3 * ((int64_t)D.base + ((D.flags & HQ_PATCH_NTSAME) ? 0 : tid));
            np[0] = hq_ld<NT>(q);                /* (no copies of loaded values here: a copy waits */
            np[1] = hq_ld<NT>(q + 1);            /*  for the load; the axes pick at the update)    */
            np[4] = hq_ld<NT>(q + 2);
        } else {
            const double* q = nt + 7 * ((int64_t)D.base + tid);
#pragma unroll
            for (int k = 0; k < 7; k++) np[k] = hq_ld<NT>(q + k);
        }
    }

    for (int q = tid; q < (DIAG == 1 ? 0 : D.npairs); q += T) {
        hq_pair_data nxt;
        if (q + T < D.npairs)
            nxt = hq_pair_load<NT>(pidx, pc1, pc2, pbeta, D.pidx_off + q + T, D.pair_off + cstep * (q + T));
        const uint4 raw = cur.raw;
        const double beta = cur.beta;
        int l[8];
        HQ_PIDX_UNPACK(l, raw)
        double X[8], Y[8], Z[8];
        if (wf) {
#pragma unroll
            for (int n = 0; n < 8; n++) {
                const double* a = &s_u1[3 * (DIAG == 5 ? (tid & 7) : l[n])];
                X[n] = a[0]; Y[n] = a[1]; Z[n] = a[2];
            }
        } else {
#pragma unroll
            for (int n = 0; n < 8; n++) {
                const double* a = &s_u1[3 * (DIAG == 5 ? (tid & 7) : l[n])];
                const double* b = &s_u2[3 * (DIAG == 5 ? (tid & 7) : l[n])];
                double a0 = a[0], a1 = a[1], a2 = a[2];
                X[n] = a0 + beta * (a0 - b[0]);
                Y[n] = a1 + beta * (a1 - b[1]);
                Z[n] = a2 + beta * (a2 - b[2]);
            }
        }
        hq_element_force(X, Y, Z, cur.c1, cur.c2);
#pragma unroll
        for (int n = 0; n < 8; n++) {
            if (DIAG == 4) { if (X[n] + Y[n] + Z[n] == 1.2345e-300) s_f[n] = 1.0; continue; }
            if (HQ_PIDX_HAS_ACC(raw, n)) {
                atomicAdd(&s_f[3 * l[n] + 0], X[n]);
                atomicAdd(&s_f[3 * l[n] + 1], Y[n]);
                atomicAdd(&s_f[3 * l[n] + 2], Z[n]);
            }
        }
        cur = nxt;
    }
    HQ_STAMP(4);
    if (F) {                                         /* compute_addforce_s, psolve.c:5917-5927 */
        for (int k = src_ptr[p] + tid; k < src_ptr[p + 1]; k += T) {
            int ln = src_ent[2 * k], li = src_ent[2 * k + 1];
            for (int d = 0; d < 3; d++) atomicAdd(&s_f[3 * ln + d], F[3 * li + d] * dt2);
        }
    }
    if (ds_ptr && ds_ptr[p + 1] > ds_ptr[p]) {       /* compute_adjust DISTRIBUTION, psolve.c:5942-5987 */
        __syncthreads();
        for (int k = ds_ptr[p] + tid; k < ds_ptr[p + 1]; k += T) {
            const int src = ds_ent[3 * k], dst = ds_ent[3 * k + 1];
            const double deps = (double)(unsigned)ds_ent[3 * k + 2];
            for (int d = 0; d < 3; d++) atomicAdd(&s_f[3 * dst + d], s_f[3 * src + d] / deps);
        }
    }
    __syncthreads();
    HQ_STAMP(5);

    /* solver_compute_displacement, psolve.c:4078-4106: one thread per owned node */
    for (int n = tid; n < (DIAG == 3 ? (tid == 0 ? 1 : 0) : D.nown); n += T) {
        if (n != tid) {
            if (iso) {
                const double* q = nt3 + 3 * ((int64_t)D.base + ((D.flags & HQ_PATCH_NTSAME) ? 0 : n));
                np[0] = q[0]; np[1] = q[1]; np[4] = q[2];
            } else {
                const double* q = nt + 7 * ((int64_t)D.base + n);
#pragma unroll
                for (int k = 0; k < 7; k++) np[k] = q[k];
            }
        }
        hq_real* out = ung + 3 * ((int64_t)D.base + n);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double m2 = iso ? np[1] : np[1 + d], m1 = iso ? np[4] : np[4 + d];
            const double x1 = wf ? s_u2[3 * n + d] : s_u1[3 * n + d], x2 = wf ? s_u2[o2off + 3 * n + d] : s_u2[3 * n + d];
            double f = s_f[3 * n + d] + (m2 * x1 - m1 * x2);
            if (NT) __builtin_nontemporal_store((hq_real)(f / np[0]), out + d);
            else out[d] = f / np[0];
        }
    }
    if (if_ptr) {   /* partition interface: hand the partial force to the exchange (psolve.c:4301) */
        for (int k = if_ptr[p] + tid; k < if_ptr[p + 1]; k += T) {
            int ln = if_ent[2 * k];
            double* o = iforce + 3 * (int64_t)if_ent[2 * k + 1];
            o[0] = s_f[3 * ln]; o[1] = s_f[3 * ln + 1]; o[2] = s_f[3 * ln + 2];
        }
    }
    HQ_STAMP(6);
}

__device__ __forceinline__ hq_patch_desc hq_patch_desc_or_empty(const hq_patch_desc* __restrict__ desc, int p)
{
    hq_patch_desc D = desc[p < 0 ? 0 : p];
    if (p < 0) { D.nown = 0; D.nhalo = 0; D.npairs = 0; D.nacc = 0; D.flags = 0; }
    return D;
}

/*
 * hq_k_patch_pers: the patch step as ONE persistent 1024-thread workgroup per CU with two LDS
 * node buffers and a register prefetch one patch deep.  With 1024 threads a patch is one local
 * node (u1, u2: 12 registers), one element row (10) and one 3-double n_t row (6) per thread, so
 * the node data of patch k+1 can be requested at the top of iteration k, fly during the element
 * section of patch k, and be written to the other LDS buffer after it -- the CU's memory pipe is
 * busy while its VALU/LDS pipes are, which two independent workgroups per CU (hq_k_patch_step)
 * achieve only by chance.  The element row of patch k+1 is requested after the element section
 * (its registers are free then) and flies during the update.  Plain loads and __syncthreads
 * throughout: the compiler's own vmcnt waits are the right ones (issue order = order of use).
 *
 * LDS image of a patch (`nrows` rows of 3 doubles per array): u1 | u2 of all local nodes, or
 * (HQ_PATCH_WFORM) w = u1 + beta (u1 - u2) of all local nodes | u1 of the owned | u2 of the owned.
 * Thread t holds local node t; its LDS row is t, or lat_row[t] in a lattice patch (see hq_lattice:
 * rows and lanes on one lattice, no bank conflicts in gathers and atomics).  The accumulator of a
 * node is indexed by its row.
 */
#define HQ_PERS_THREADS 1024
#ifndef HQ_TICKET_STRIDE
#define HQ_TICKET_STRIDE 64      /* ints between the XCDs' work-queue counters: a 256-byte line each */
#endif
typedef __attribute__((address_space(3))) double hq_lds_double;
#define HQ_LDS_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
/* &base[3*i] for a row index i < 2^24.  gfx9 has no 32-bit mad, so the compiler takes
 * v_mad_u64_u32 with a don't-care upper addend register -- and when that register happens to be
 * the target of a load in flight, the address waits for the load (vmcnt(0) in front of a
 * ds_add_f64 group).  The 24-bit mad has no such operand. */
static __device__ __forceinline__ hq_lds_double* hq_lds_row3(hq_lds_double* base, int i) {
    unsigned b = (unsigned)(__SIZE_TYPE__)base, o;
    asm("v_mad_u32_u24 %0, %1, 24, %2" : "=v"(o) : "v"(i), "v"(b));
    return (hq_lds_double*)(__SIZE_TYPE__)o;
}

__global__ void __launch_bounds__(HQ_PERS_THREADS)
hq_k_patch_pers(int32_t count, int32_t per_xcd, const int32_t* __restrict__ order, int32_t nrows,
                int32_t nfacc, const hq_patch_desc* __restrict__ desc,
                const uint4* __restrict__ pidx, const double* __restrict__ pc1,
                const double* __restrict__ pc2, const double* __restrict__ pbeta,
                const int32_t* __restrict__ halo, const hq_real* __restrict__ u1g,
                const hq_real* __restrict__ u2g, hq_real* __restrict__ ung,
                const double* __restrict__ nt, const double* __restrict__ nt3,
                const int32_t* __restrict__ src_ptr, const int32_t* __restrict__ src_ent,
                const double* __restrict__ F, double dt2, const int32_t* __restrict__ if_ptr,
                const int32_t* __restrict__ if_ent, double* __restrict__ iforce,
                const int32_t* __restrict__ ds_ptr, const int32_t* __restrict__ ds_ent, int32_t hstride,
                int32_t* __restrict__ tickets, const uint16_t* __restrict__ lat_row)
{
    extern __shared__ __align__(16) double s_mem[];
    double* __restrict__ s_fg = s_mem + 12 * nrows;     /* after the two node buffers */
    int32_t* __restrict__ s_tick = reinterpret_cast<int32_t*>(s_fg + nfacc);   /* ring of 8: slots drawn 5 patches ahead */
    const int tid0 = threadIdx.x, T = HQ_PERS_THREADS;
    const int W = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7);
    const int end = min((xcd + 1) * per_xcd, count);
    /* The workgroups of an XCD draw the slots of its run of patches from a counter (tickets[xcd]),
     * four patches ahead of use (descriptors, gather ids and element rows are requested that far
     * ahead): patches differ in cost (dashpot faces, far faces, hanging nodes), and with a fixed
     * slot, slot + W, ... assignment the mean workgroup idled 3 % (64M box) to 9 % (8M box) of the
     * launch at the end.  The last workgroup out resets the counter for the next launch. */
    HQ_WG_STAMP(0);
#define HQ_SLOT_PATCH(s) ((s) < end ? (order ? order[(s)] : (s)) : -1)
#define HQ_DRAW() (xcd * per_xcd + atomicAdd(&tickets[HQ_TICKET_STRIDE * xcd], 1))
    /* halo id of this thread's local node of patch (P_, DD) (0 where that node is owned or absent) */
#define HQ_PERS_ID(P_, DD) \
    ((tid >= (DD).nown && tid < (DD).nown + (DD).nhalo) ? halo[(int64_t)(P_) * hstride + (tid - (DD).nown)] : 0)

    if (tid0 == 0) { for (int i = 0; i < 5; i++) s_tick[i] = HQ_DRAW(); }
    /* this thread's row in a lattice patch: the same for every such patch */
    int lrow0 = lat_row ? (int)lat_row[tid0] : tid0;
    __syncthreads();
    const int sl0 = __builtin_amdgcn_readfirstlane(s_tick[0]), sl1 = __builtin_amdgcn_readfirstlane(s_tick[1]),
              sl2 = __builtin_amdgcn_readfirstlane(s_tick[2]);
    int p0 = HQ_SLOT_PATCH(sl0), p1 = HQ_SLOT_PATCH(sl1), p2 = HQ_SLOT_PATCH(sl2);
#define HQ_PERS_EXIT()                                                                          \
    {                                                                                           \
        if (tid0 == 0 && atomicAdd(&tickets[HQ_TICKET_STRIDE * xcd + 1], 1) == W - 1) {   /* last workgroup of the XCD out */ \
            tickets[HQ_TICKET_STRIDE * xcd] = 0;                                                \
            tickets[HQ_TICKET_STRIDE * xcd + 1] = 0;                                            \
        }                                                                                       \
        HQ_WG_STAMP(1);                                                                         \
    }
    if (p0 < 0) {                                       /* the run was drawn empty before this workgroup got to it */
        HQ_PERS_EXIT()
        return;
    }
    hq_patch_desc D0 = hq_patch_desc_or_empty(desc, p0);
    hq_patch_desc D1 = hq_patch_desc_or_empty(desc, p1);
    hq_patch_desc D2 = hq_patch_desc_or_empty(desc, p2);
    hq_u32x4 c_raw = { 0, 0, 0, 0 };                    /* element row of the CURRENT patch (pidx; beta, c1, c2) */
    double c_beta = 0.0, c_c1 = 0.0, c_c2 = 0.0;
    int32_t idn;                                        /* gather id of the NEXT patch's local node */
#define HQ_PERS_ROW(DD)                                                                         \
    {                                                                                           \
        const int q_ = tid < (DD).npairs ? tid : 0;                                             \
        const int64_t gc_ = (DD).pair_off + (((DD).flags & HQ_PATCH_UNIFORM) ? 0 : q_);         \
        c_raw = *(reinterpret_cast<const hq_u32x4*>(pidx) + ((DD).pidx_off + q_));              \
        c_beta = pbeta[gc_]; c_c1 = pc1[gc_]; c_c2 = pc2[gc_];                                  \
    }
    {   /* prologue: patch 0 into buffer 0 */
        const int tid = tid0;
        for (int i = tid; i < nfacc; i += T) s_fg[i] = 0.0;
        HQ_PERS_ROW(D0)
        const int32_t id0 = HQ_PERS_ID(p0, D0);
        idn = HQ_PERS_ID((p1 < 0 ? 0 : p1), D1);
        if (tid < D0.nown + D0.nhalo) {
            const int64_t g = tid < D0.nown ? (int64_t)D0.base + tid : (int64_t)id0;
            const bool wf = (D0.flags & HQ_PATCH_WFORM) != 0;
            const int row = (D0.flags & HQ_PATCH_LATTICE) ? lrow0 : tid;
            const double b0 = pbeta[D0.pair_off];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const double x1 = u1g[3 * g + d], x2 = u2g[3 * g + d];
                if (wf) {
                    s_mem[3 * row + d] = x1 + b0 * (x1 - x2);
                    if (tid < D0.nown) { s_mem[3 * nrows + 3 * tid + d] = x1; s_mem[3 * nrows + 3 * (nrows / 2) + 3 * tid + d] = x2; }
                } else {
                    s_mem[3 * row + d] = x1;
                    s_mem[3 * nrows + 3 * row + d] = x2;
                }
            }
        }
        /* nothing loaded here may still be pending when the loop is entered: the compiler merges
         * this path with the loop's back-edge and would wait vmcnt(0) (i.e. for the previous
         * patch's stores) at the top of every iteration */
        asm volatile("" : "+v"(c_raw), "+v"(c_beta), "+v"(c_c1), "+v"(c_c2), "+v"(idn), "+v"(lrow0));
        __syncthreads();
    }

    for (int k = 0;; k++) {
        /* keep the per-patch address arithmetic inside the iteration: hipcc otherwise hoists
         * table + f(thread) for every table out of the loop and spills */
        HQ_STAMPD(0);
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        /* LDS pointers carry their address space: 32-bit address arithmetic (through generic pointers
         * hipcc builds LDS addresses with 64-bit multiply-adds whose unused high half can alias a
         * register with a load pending -- a wait in front of every LDS atomic) */
        hq_lds_double* __restrict__ s_u1 = (hq_lds_double*)s_mem + (k & 1) * 6 * nrows;
        hq_lds_double* __restrict__ s_u2 = s_u1 + 3 * nrows;
        hq_lds_double* __restrict__ n_u1 = (hq_lds_double*)s_mem + ((k + 1) & 1) * 6 * nrows;
        hq_lds_double* __restrict__ n_u2 = n_u1 + 3 * nrows;
        hq_lds_double* __restrict__ s_f = (hq_lds_double*)s_fg;
        const bool wf0 = (D0.flags & HQ_PATCH_WFORM) != 0, wf1 = (D1.flags & HQ_PATCH_WFORM) != 0;
        /* this thread's LDS row in patch k (accumulator, u1/u2 of its owned node) and in patch k+1 */
        const int row0 = (D0.flags & HQ_PATCH_LATTICE) ? lrow0 : tid;
        const int row1 = (D1.flags & HQ_PATCH_LATTICE) ? lrow0 : tid;

        /* 1. the request that flies during the element section: the node data of patch k+1 */
        /* (loads are unconditional, from a clamped address where the thread has nothing to load:
         * straight-line code lets the compiler count vmcnt exactly instead of waiting for all) */
        double a1[3], a2[3];
        const bool have_node = tid < D1.nown + D1.nhalo;
#define HQ_PERS_IMAGE_WRITE()                                                                   \
        if (have_node) {                                                                        \
            if (wf1) {                                                                          \
                /* c_beta: the row of patch k+1 is here, and every row of a uniform patch holds the patch's beta */ \
                _Pragma("unroll")                                                               \
                for (int d = 0; d < 3; d++) n_u1[3 * row1 + d] = a1[d] + c_beta * (a1[d] - a2[d]); \
                if (tid < D1.nown) {                                                            \
                    _Pragma("unroll")                                                           \
                    for (int d = 0; d < 3; d++) { n_u2[3 * tid + d] = a1[d]; n_u2[3 * (nrows / 2) + 3 * tid + d] = a2[d]; } \
                }                                                                               \
            } else {                                                                            \
                _Pragma("unroll")                                                               \
                for (int d = 0; d < 3; d++) { n_u1[3 * row1 + d] = a1[d]; n_u2[3 * row1 + d] = a2[d]; } \
            }                                                                                   \
        }
#define HQ_PERS_NODE_LOADS()                                                                    \
        {                                                                                       \
            const int64_t g = tid < D1.nown ? (int64_t)D1.base + tid : (have_node ? (int64_t)idn : 0); \
            _Pragma("unroll")                                                                   \
            for (int d = 0; d < 3; d++) { a1[d] = u1g[3 * g + d]; a2[d] = u2g[3 * g + d]; }     \
        }
        HQ_PERS_NODE_LOADS()
        HQ_STAMPD(7);
        const int slot3 = __builtin_amdgcn_readfirstlane(s_tick[(k + 3) & 7]);   /* drawn two iterations ago */
        const int p3 = HQ_SLOT_PATCH(slot3);
        const hq_patch_desc D3 = hq_patch_desc_or_empty(desc, p3);
        /* the slot of patch k+4, into the ring before the barrier (drawing it from the last, element-less
         * wave instead measured 1 % slower) */
        int32_t drawn = 0;
        if (tid == 0) drawn = HQ_DRAW();

        HQ_STAMPD(1);
        /* 2. element section of patch k on the current buffer: one element per thread (the
         *    planner keeps patches at <= 1024 elements) */
        const bool has_elem = tid < D0.npairs;
        int l[8];
        double X[8], Y[8], Z[8];
        hq_u32x4 rawk = c_raw;
        if (has_elem) {
            const hq_u32x4 raw = c_raw;
            const double beta = c_beta;
            HQ_PIDX_UNPACK(l, raw)
            if (wf0) {
#pragma unroll
                for (int n = 0; n < 8; n++) {
                    const hq_lds_double* a = &s_u1[3 * l[n]];
                    X[n] = a[0]; Y[n] = a[1]; Z[n] = a[2];
                }
            } else {
#pragma unroll
                for (int n = 0; n < 8; n++) {
                    const hq_lds_double* a = &s_u1[3 * l[n]];
                    const hq_lds_double* b = &s_u2[3 * l[n]];
                    double a0 = a[0], a1_ = a[1], a2_ = a[2];
                    X[n] = a0 + beta * (a0 - b[0]);
                    Y[n] = a1_ + beta * (a1_ - b[1]);
                    Z[n] = a2_ + beta * (a2_ - b[2]);
                }
            }
            hq_element_force(X, Y, Z, c_c1, c_c2);
        }
        HQ_STAMPD(2);
        /* 3. the element row is consumed: request what flies during the atomics, the barrier and
         *    the LDS write below: n_t of this patch's node (3-double form, or the 7-double row where
         *    a dashpot makes the axes differ), element row of patch k+1, halo id of patch k+2 */
        const bool iso = (D0.flags & HQ_PATCH_ISO) != 0;
        double np[7];
        {
            const int64_t nn = (int64_t)D0.base + ((tid < D0.nown && !(D0.flags & HQ_PATCH_NTSAME)) ? tid : 0);
            if (iso) {
                const double* q = nt3 + 3 * nn;           /* (no copies of loaded values here: they would */
                np[0] = q[0]; np[1] = q[1]; np[4] = q[2];  /*  wait for the load; the axes pick at the update) */
            } else {
                const double* q = nt + 7 * nn;
#pragma unroll
                for (int i = 0; i < 7; i++) np[i] = q[i];
            }
        }
        HQ_PERS_ROW(D1)
        int32_t idnn;
        {
            const int h = tid - D2.nown;
            idnn = halo[(int64_t)(p2 < 0 ? 0 : p2) * hstride + ((h >= 0 && h < D2.nhalo) ? h : 0)];
        }
        if (has_elem) {
            /* the local rows again from the packed element row (4 registers across the force arithmetic
             * instead of 8: the kernel sits at the 128-register limit of 16 waves per CU) */
            asm volatile("" : "+v"(rawk));
            HQ_PIDX_UNPACK(l, rawk)
#pragma unroll
            for (int n = 0; n < 8; n++) {
                if (HQ_PIDX_HAS_ACC(rawk, n)) {
                    hq_lds_double* a = hq_lds_row3(s_f, l[n]);
                    HQ_LDS_ADD(a + 0, X[n]);
                    HQ_LDS_ADD(a + 1, Y[n]);
                    HQ_LDS_ADD(a + 2, Z[n]);
                }
            }
        }

        if (F) {                                         /* compute_addforce_s, psolve.c:5917-5927 */
            for (int i = src_ptr[p0] + tid; i < src_ptr[p0 + 1]; i += T) {
                int ln = src_ent[2 * i], li = src_ent[2 * i + 1];
                for (int d = 0; d < 3; d++) HQ_LDS_ADD(&s_f[3 * ln + d], F[3 * li + d] * dt2);
            }
        }
        if (ds_ptr && ds_ptr[p0 + 1] > ds_ptr[p0]) {     /* compute_adjust DISTRIBUTION, psolve.c:5942-5987 */
            __syncthreads();
            for (int i = ds_ptr[p0] + tid; i < ds_ptr[p0 + 1]; i += T) {
                const int src = ds_ent[3 * i], dst = ds_ent[3 * i + 1];
                const double deps = (double)(unsigned)ds_ent[3 * i + 2];
                for (int d = 0; d < 3; d++) HQ_LDS_ADD(&s_f[3 * dst + d], s_f[3 * src + d] / deps);
            }
        }
        __syncthreads();
        HQ_STAMPD(3);

        /* 4. patch k+1 into the other buffer (last read an iteration ago) */
        HQ_PERS_IMAGE_WRITE()
        /* the element row and gather id requested above are the youngest loads: the compiler's wait
         * for them sits here, before the update's stores are in the queue */
        asm volatile("" : "+v"(c_raw), "+v"(c_beta), "+v"(c_c1), "+v"(c_c2), "+v"(idnn), "+v"(drawn));
        if (tid == 0) {
            s_tick[(k + 5) & 7] = drawn;                 /* its old content was read at iteration k-6 */
        }
        HQ_STAMPD(4);
        /* 5. interface partial forces (psolve.c:4301), then update + re-zero the accumulators */
        if (if_ptr && if_ptr[p0 + 1] > if_ptr[p0]) {
            for (int i = if_ptr[p0] + tid; i < if_ptr[p0 + 1]; i += T) {
                int ln = if_ent[2 * i];
                double* o = iforce + 3 * (int64_t)if_ent[2 * i + 1];
                o[0] = s_f[3 * ln]; o[1] = s_f[3 * ln + 1]; o[2] = s_f[3 * ln + 2];
            }
            __syncthreads();
        }
        if (tid < D0.nown) {                             /* solver_compute_displacement, psolve.c:4078-4106 */
            const int n = tid;
            hq_real* out = ung + 3 * ((int64_t)D0.base + n);
            /* u1, u2 of the owned node: compact by owned index behind w (w-form), else its image row */
            const hq_lds_double* __restrict__ o_u1 = wf0 ? s_u2 + 3 * n : s_u1 + 3 * row0;
            const hq_lds_double* __restrict__ o_u2 = wf0 ? s_u2 + 3 * (nrows / 2) + 3 * n : s_u2 + 3 * row0;
            hq_lds_double* __restrict__ acc = s_f + 3 * row0;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const double m2 = iso ? np[1] : np[1 + d], m1 = iso ? np[4] : np[4 + d];
                double f = acc[d] + (m2 * o_u1[d] - m1 * o_u2[d]);
                acc[d] = 0.0;
                out[d] = f / np[0];
            }
        }
        for (int i = 3 * D0.nown + tid; i < 3 * D0.nacc; i += T) s_f[i] = 0.0;   /* hanging nodes' accumulators (id-ordered patches) */
        HQ_STAMPD(5);
        __syncthreads();
        HQ_STAMPD(6);
        if (p1 < 0) break;
        p0 = p1; p1 = p2; p2 = p3;
        D0 = D1; D1 = D2; D2 = D3;
        idn = idnn;
    }
    HQ_PERS_EXIT()
#undef HQ_PERS_EXIT
#undef HQ_SLOT_PATCH
#undef HQ_DRAW
#undef HQ_PERS_ID
#undef HQ_PERS_ROW
}


/*
 * hq_k_patch_seed: hq_k_patch_pers with ONE barrier per patch.
 *
 *   un = (f + m2 u1 - m1 u2) / m0      (solver_compute_displacement, psolve.c:4078-4106)
 *
 * The thread that loads node t of patch k+1 (u1, u2 in registers) also has the node's n_t row, so it
 * SEEDS the node's force accumulator with m2 u1 - m1 u2 when it writes the LDS image; after the
 * element forces are added (atomics) the accumulator holds the whole numerator and the update is one
 * LDS read and a division.  u1, u2 of the owned nodes are never stored in LDS, the accumulators are
 * never re-zeroed, and with three accumulator arrays and two images nothing written in iteration k is
 * read before the barrier of iteration k, nothing read after it is overwritten before the barrier of
 * iteration k+1:
 *
 *   iteration k:  node loads k+1 | gather, arithmetic, atomics of patch k  (image[k&1], acc[k%3])
 *                 | image[(k+1)&1], seed acc[(k+1)%3] | BARRIER | update of patch k from acc[k%3]
 *
 * so waves drift apart by up to an iteration: one wave's update and node requests run beside another
 * wave's element arithmetic.  LDS image: w = u1 + beta (u1 - u2) (uniform patches) or u1 | u2, `nrows`
 * rows per array.  Nodes whose update belongs to someone else -- partition-interface nodes
 * (hq_k_interface_update finishes them from the pure partial force) and hanging nodes (compute_adjust
 * overwrites them, and their pure force is what is distributed) -- carry a NEGATIVE mass_simple in the
 * kernel's private n_t table (nt3): seed 0, divide by |m0|.
 */
__global__ void __launch_bounds__(HQ_PERS_THREADS)
hq_k_patch_seed(int32_t count, int32_t per_xcd, const int32_t* __restrict__ order, int32_t nrows,
                int32_t nfacc, const hq_patch_desc* __restrict__ desc,
                const uint4* __restrict__ pidx, const double* __restrict__ pc1,
                const double* __restrict__ pc2, const double* __restrict__ pbeta,
                const int32_t* __restrict__ halo, const hq_real* __restrict__ u1g,
                const hq_real* __restrict__ u2g, hq_real* __restrict__ ung,
                const double* __restrict__ nt, const double* __restrict__ nt3,
                const int32_t* __restrict__ src_ptr, const int32_t* __restrict__ src_ent,
                const double* __restrict__ F, double dt2, const int32_t* __restrict__ if_ptr,
                const int32_t* __restrict__ if_ent, double* __restrict__ iforce,
                const int32_t* __restrict__ ds_ptr, const int32_t* __restrict__ ds_ent, int32_t hstride,
                int32_t* __restrict__ tickets, const uint16_t* __restrict__ lat_row)
{
    extern __shared__ __align__(16) double s_mem[];
    /* LDS: image[2][2][3 nrows] | acc[3][nfacc] | ticket ring */
    double* __restrict__ s_fg = s_mem + 12 * nrows;
    int32_t* __restrict__ s_tick = reinterpret_cast<int32_t*>(s_fg + 3 * nfacc);   /* ring of 8: slots drawn 5 patches ahead */
    const int tid0 = threadIdx.x, T = HQ_PERS_THREADS;
    const int W = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7);
    const int end = min((xcd + 1) * per_xcd, count);
    HQ_WG_STAMP(0);
#define HQ_SLOT_PATCH(s) ((s) < end ? (order ? order[(s)] : (s)) : -1)
#define HQ_DRAW() (xcd * per_xcd + atomicAdd(&tickets[HQ_TICKET_STRIDE * xcd], 1))
#define HQ_PERS_ID(P_, DD) \
    ((tid >= (DD).nown && tid < (DD).nown + (DD).nhalo) ? halo[(int64_t)(P_) * hstride + (tid - (DD).nown)] : 0)
    if (tid0 == 0) { for (int i = 0; i < 5; i++) s_tick[i] = HQ_DRAW(); }
    int lrow0 = lat_row ? (int)lat_row[tid0] : tid0;   /* this thread's row in a lattice patch: the same for every such patch */
    __syncthreads();
    const int sl0 = __builtin_amdgcn_readfirstlane(s_tick[0]), sl1 = __builtin_amdgcn_readfirstlane(s_tick[1]),
              sl2 = __builtin_amdgcn_readfirstlane(s_tick[2]);
    int p0 = HQ_SLOT_PATCH(sl0), p1 = HQ_SLOT_PATCH(sl1), p2 = HQ_SLOT_PATCH(sl2);
#define HQ_PERS_EXIT()                                                                          \
    {                                                                                           \
        if (tid0 == 0 && atomicAdd(&tickets[HQ_TICKET_STRIDE * xcd + 1], 1) == W - 1) {   /* last workgroup of the XCD out */ \
            tickets[HQ_TICKET_STRIDE * xcd] = 0;                                                \
            tickets[HQ_TICKET_STRIDE * xcd + 1] = 0;                                            \
        }                                                                                       \
        HQ_WG_STAMP(1);                                                                         \
    }
    if (p0 < 0) {                                       /* the run was drawn empty before this workgroup got to it */
        HQ_PERS_EXIT()
        return;
    }
    hq_patch_desc D0 = hq_patch_desc_or_empty(desc, p0);
    hq_patch_desc D1 = hq_patch_desc_or_empty(desc, p1);
    hq_patch_desc D2 = hq_patch_desc_or_empty(desc, p2);
    hq_u32x4 c_raw = { 0, 0, 0, 0 };                    /* element row of the CURRENT patch (pidx; beta, c1, c2) */
    double c_beta = 0.0, c_c1 = 0.0, c_c2 = 0.0;
    double m0 = 1.0;                                    /* |mass_simple| of this thread's owned node of the CURRENT patch */
    int32_t idn;                                        /* gather id of the NEXT patch's local node */
#define HQ_PERS_ROW(DD)                                                                         \
    {                                                                                           \
        const int q_ = tid < (DD).npairs ? tid : 0;                                             \
        const int64_t gc_ = (DD).pair_off + (((DD).flags & HQ_PATCH_UNIFORM) ? 0 : q_);         \
        c_raw = *(reinterpret_cast<const hq_u32x4*>(pidx) + ((DD).pidx_off + q_));              \
        c_beta = pbeta[gc_]; c_c1 = pc1[gc_]; c_c2 = pc2[gc_];                                  \
    }
    /* n_t of this thread's node of patch DD into np[0..6]: mass_simple (negative: seed 0) from the private
     * 3-double table, the axis terms from it (no dashpot on the patch) or from the 7-double rows */
#define HQ_SEED_NT(DD)                                                                          \
    {                                                                                           \
        const int64_t nn_ = (int64_t)(DD).base + ((tid < (DD).nown && !((DD).flags & HQ_PATCH_NTSAME)) ? tid : 0); \
        const double* q3_ = nt3 + 3 * nn_;                                                      \
        np[0] = q3_[0];                                                                         \
        if ((DD).flags & HQ_PATCH_ISO) { np[1] = q3_[1]; np[4] = q3_[2]; }                      \
        else {                                                                                  \
            const double* q7_ = nt + 7 * nn_;                                                   \
            _Pragma("unroll")                                                                   \
            for (int i_ = 1; i_ < 7; i_++) np[i_] = q7_[i_];                                    \
        }                                                                                       \
    }
    /* image rows of patch DD from (x1, x2) and the seed of its accumulator: buffers ib_ (image), ab_ (accumulators) */
#define HQ_SEED_WRITE(DD, x1, x2, beta_, ib_, ab_)                                              \
    {                                                                                           \
        const int row_ = ((DD).flags & HQ_PATCH_LATTICE) ? lrow0 : tid;                         \
        hq_lds_double* iu1_ = (hq_lds_double*)s_mem + (ib_) * 6 * nrows;                        \
        hq_lds_double* iu2_ = iu1_ + 3 * nrows;                                                 \
        hq_lds_double* ac_ = (hq_lds_double*)s_fg + (ab_) * nfacc;                              \
        if (tid < (DD).nown + (DD).nhalo) {                                                     \
            if ((DD).flags & HQ_PATCH_WFORM) {                                                  \
                _Pragma("unroll")                                                               \
                for (int d = 0; d < 3; d++) iu1_[3 * row_ + d] = x1[d] + (beta_) * (x1[d] - x2[d]); \
            } else {                                                                            \
                _Pragma("unroll")                                                               \
                for (int d = 0; d < 3; d++) { iu1_[3 * row_ + d] = x1[d]; iu2_[3 * row_ + d] = x2[d]; } \
            }                                                                                   \
        }                                                                                       \
        if (tid < (DD).nown) {                                                                  \
            const bool iso_ = ((DD).flags & HQ_PATCH_ISO) != 0;                                 \
            _Pragma("unroll")                                                                   \
            for (int d = 0; d < 3; d++) {                                                       \
                const double m2_ = iso_ ? np[1] : np[1 + d], m1_ = iso_ ? np[4] : np[4 + d];    \
                ac_[3 * row_ + d] = np[0] < 0.0 ? 0.0 : (m2_ * x1[d] - m1_ * x2[d]);            \
            }                                                                                   \
        } else if (tid < (DD).nacc) {               /* hanging nodes on owned anchors (id-ordered patches) */ \
            _Pragma("unroll")                                                                   \
            for (int d = 0; d < 3; d++) ac_[3 * tid + d] = 0.0;                                 \
        }                                                                                       \
    }
    {   /* prologue: patch 0 into image 0, its seeds into accumulator array 0 */
        const int tid = tid0;
        for (int i = tid; i < 3 * nfacc; i += T) s_fg[i] = 0.0;
        __syncthreads();
        HQ_PERS_ROW(D0)
        const int32_t id0 = HQ_PERS_ID(p0, D0);
        idn = HQ_PERS_ID((p1 < 0 ? 0 : p1), D1);
        double np[7] = { 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
        HQ_SEED_NT(D0)
        const int64_t g = tid < D0.nown ? (int64_t)D0.base + tid : (tid < D0.nown + D0.nhalo ? (int64_t)id0 : 0);
        double x1[3], x2[3];
#pragma unroll
        for (int d = 0; d < 3; d++) { x1[d] = u1g[3 * g + d]; x2[d] = u2g[3 * g + d]; }
        const double b0 = pbeta[D0.pair_off];
        HQ_SEED_WRITE(D0, x1, x2, b0, 0, 0)
        m0 = fabs(np[0]);
        /* nothing loaded here may still be pending when the loop is entered (see hq_k_patch_pers) */
        asm volatile("" : "+v"(c_raw), "+v"(c_beta), "+v"(c_c1), "+v"(c_c2), "+v"(idn), "+v"(lrow0), "+v"(m0));
        __syncthreads();
    }

    int ab = 0;                                         /* accumulator array of the current patch: k % 3 */
    for (int k = 0;; k++) {
        HQ_STAMPD(0);
        int tid = tid0;
        asm volatile("" : "+v"(tid));                    /* (the per-patch address arithmetic stays inside the iteration) */
        hq_lds_double* __restrict__ s_u1 = (hq_lds_double*)s_mem + (k & 1) * 6 * nrows;
        hq_lds_double* __restrict__ s_u2 = s_u1 + 3 * nrows;
        hq_lds_double* __restrict__ s_f = (hq_lds_double*)s_fg + ab * nfacc;
        const int abn = ab == 2 ? 0 : ab + 1;
        const bool wf0 = (D0.flags & HQ_PATCH_WFORM) != 0;
        const int row0 = (D0.flags & HQ_PATCH_LATTICE) ? lrow0 : tid;

        /* 1. the request that flies during the element section: the node data of patch k+1
         * (unconditional, from a clamped address: straight-line code keeps the compiler's vmcnt exact) */
        double a1[3], a2[3];
        {
            const int64_t g = tid < D1.nown ? (int64_t)D1.base + tid : (tid < D1.nown + D1.nhalo ? (int64_t)idn : 0);
#pragma unroll
            for (int d = 0; d < 3; d++) { a1[d] = u1g[3 * g + d]; a2[d] = u2g[3 * g + d]; }
        }
        HQ_STAMPD(7);
        const int slot3 = __builtin_amdgcn_readfirstlane(s_tick[(k + 3) & 7]);   /* drawn two iterations ago */
        const int p3 = HQ_SLOT_PATCH(slot3);
        const hq_patch_desc D3 = hq_patch_desc_or_empty(desc, p3);
        int32_t drawn = 0;
        if (tid == 0) drawn = HQ_DRAW();

        HQ_STAMPD(1);
        /* 2. element section of patch k: one element per thread */
        const bool has_elem = tid < D0.npairs;
        int l[8];
        double X[8], Y[8], Z[8];
        hq_u32x4 rawk = c_raw;
        if (has_elem) {
            const hq_u32x4 raw = c_raw;
            const double beta = c_beta;
            HQ_PIDX_UNPACK(l, raw)
            if (wf0) {
#pragma unroll
                for (int n = 0; n < 8; n++) {
                    const hq_lds_double* a = &s_u1[3 * l[n]];
                    X[n] = a[0]; Y[n] = a[1]; Z[n] = a[2];
                }
            } else {
#pragma unroll
                for (int n = 0; n < 8; n++) {
                    const hq_lds_double* a = &s_u1[3 * l[n]];
                    const hq_lds_double* b = &s_u2[3 * l[n]];
                    double a0 = a[0], a1_ = a[1], a2_ = a[2];
                    X[n] = a0 + beta * (a0 - b[0]);
                    Y[n] = a1_ + beta * (a1_ - b[1]);
                    Z[n] = a2_ + beta * (a2_ - b[2]);
                }
            }
            hq_element_force(X, Y, Z, c_c1, c_c2);
        }
        HQ_STAMPD(2);
        /* 3. the element row is consumed: request n_t of this thread's node of patch k+1, the element row of
         *    patch k+1 and the gather id of patch k+2; they fly during the atomics.  (Requested at the top of
         *    the iteration instead -- 16 more registers -- the kernel is 12 % SLOWER: every vector-memory
         *    instruction in the burst at the top delays the element section behind it.) */
        double np[7];
        HQ_SEED_NT(D1)
        hq_u32x4 n_raw;
        double n_beta, n_c1, n_c2;
        {
            const int q_ = tid < D1.npairs ? tid : 0;
            const int64_t gc_ = D1.pair_off + ((D1.flags & HQ_PATCH_UNIFORM) ? 0 : q_);
            n_beta = pbeta[gc_];
            n_raw = *(reinterpret_cast<const hq_u32x4*>(pidx) + (D1.pidx_off + q_));
            n_c1 = pc1[gc_]; n_c2 = pc2[gc_];
        }
        int32_t idnn;
        {
            const int h = tid - D2.nown;
            idnn = halo[(int64_t)(p2 < 0 ? 0 : p2) * hstride + ((h >= 0 && h < D2.nhalo) ? h : 0)];
        }
        if (has_elem) {
            asm volatile("" : "+v"(rawk));               /* the rows again from the packed element row (4 registers across the arithmetic) */
            HQ_PIDX_UNPACK(l, rawk)
#pragma unroll
            for (int n = 0; n < 8; n++) {
                if (HQ_PIDX_HAS_ACC(rawk, n)) {
                    hq_lds_double* a = hq_lds_row3(s_f, l[n]);
                    HQ_LDS_ADD(a + 0, X[n]);
                    HQ_LDS_ADD(a + 1, Y[n]);
                    HQ_LDS_ADD(a + 2, Z[n]);
                }
            }
        }
        if (F) {                                         /* compute_addforce_s, psolve.c:5917-5927 */
            for (int i = src_ptr[p0] + tid; i < src_ptr[p0 + 1]; i += T) {
                int ln = src_ent[2 * i], li = src_ent[2 * i + 1];
                for (int d = 0; d < 3; d++) HQ_LDS_ADD(&s_f[3 * ln + d], F[3 * li + d] * dt2);
            }
        }
        /* 4. patch k+1: image into the other buffer (last read before the previous barrier), seeds into the next
         *    accumulator array (last read after the barrier before the previous one) */
        HQ_SEED_WRITE(D1, a1, a2, n_beta, (k + 1) & 1, abn)    /* n_beta: a uniform patch's rows all hold its beta */
        const double m0n = fabs(np[0]);
        asm volatile("" : "+v"(n_raw), "+v"(n_c1), "+v"(n_c2), "+v"(idnn), "+v"(drawn));
        HQ_STAMPD(3);
        __syncthreads();
        HQ_STAMPD(4);
        if (tid == 0) s_tick[(k + 5) & 7] = drawn;       /* its old content was read at iteration k-6 */
        if (ds_ptr && ds_ptr[p0 + 1] > ds_ptr[p0]) {     /* compute_adjust DISTRIBUTION, psolve.c:5942-5987 */
            for (int i = ds_ptr[p0] + tid; i < ds_ptr[p0 + 1]; i += T) {
                const int src = ds_ent[3 * i], dst = ds_ent[3 * i + 1];
                const double deps = (double)(unsigned)ds_ent[3 * i + 2];
                for (int d = 0; d < 3; d++) HQ_LDS_ADD(&s_f[3 * dst + d], s_f[3 * src + d] / deps);
            }
            __syncthreads();
        }
        /* 5. interface partial forces (psolve.c:4301: pure element force, their seed is 0), then the update */
        if (if_ptr && if_ptr[p0 + 1] > if_ptr[p0]) {
            for (int i = if_ptr[p0] + tid; i < if_ptr[p0 + 1]; i += T) {
                int ln = if_ent[2 * i];
                double* o = iforce + 3 * (int64_t)if_ent[2 * i + 1];
                o[0] = s_f[3 * ln]; o[1] = s_f[3 * ln + 1]; o[2] = s_f[3 * ln + 2];
            }
        }
        if (tid < D0.nown) {                             /* solver_compute_displacement, psolve.c:4078-4106 */
            hq_real* out = ung + 3 * ((int64_t)D0.base + tid);
            const hq_lds_double* __restrict__ acc = s_f + 3 * row0;
#pragma unroll
            for (int d = 0; d < 3; d++) out[d] = acc[d] / m0;
        }
        HQ_STAMPD(5);
        HQ_STAMPD(6);
        if (p1 < 0) break;
        p0 = p1; p1 = p2; p2 = p3;
        D0 = D1; D1 = D2; D2 = D3;
        idn = idnn;
        m0 = m0n;
        ab = abn;
        c_raw = n_raw; c_beta = n_beta; c_c1 = n_c1; c_c2 = n_c2;
    }
    HQ_PERS_EXIT()
#undef HQ_PERS_EXIT
#undef HQ_SLOT_PATCH
#undef HQ_DRAW
#undef HQ_PERS_ID
#undef HQ_PERS_ROW
#undef HQ_SEED_NT
#undef HQ_SEED_WRITE
}


/*
 * What a stencil workgroup needs to start, in LAUNCH order (entry k of the stencil part of d_order): the facts of the
 * patch in one 64-byte record and, per thread, the halo node it loads with the LDS word of that node -- so the
 * workgroup's first loads depend on nothing but its slot (patch number -> descriptor -> halo list -> rows is two
 * memory latencies longer).  Built on the device from the plan (hq_k_stencil_entries) whenever the order changes.
 */
struct hq_st_desc {
    int32_t base, nown, nhalo, flags, nbnd, p;
    int64_t loc_off;                 /* into rg_tab: row, mask, boundary index of the local nodes; then the boundary list */
    double  c1, c2, beta, pad;
};
#define HQ_ST_HSTRIDE 512            /* halo entries per patch: (node, word); past the list: the patch's first node */

__global__ void __launch_bounds__(HQ_ST_HSTRIDE)
hq_k_stencil_entries(int32_t count, const int32_t* __restrict__ order, const hq_patch_desc* __restrict__ desc,
                     const double* __restrict__ pcoef, const int64_t* __restrict__ rg_off,
                     const uint32_t* __restrict__ rg_tab, const int32_t* __restrict__ halo, int32_t hstride,
                     hq_st_desc* __restrict__ st_desc, int2* __restrict__ st_halo)
{
    const int k = blockIdx.x, t = threadIdx.x;
    if (k >= count) return;
    const int p = order[k];
    const hq_patch_desc D = desc[p];
    const int64_t rgo = rg_off[p];
    const uint32_t* loc = rg_tab + (rgo & 0xffffffffffll);
    int2 e;
    if (t < D.nhalo) { e.x = halo[(int64_t)p * hstride + t]; e.y = (int)loc[D.nown + t]; }
    else { e.x = D.base; e.y = (int)loc[0]; }
    st_halo[(int64_t)k * HQ_ST_HSTRIDE + t] = e;
    if (t == 0) {
        hq_st_desc q;
        q.base = D.base; q.nown = D.nown; q.nhalo = D.nhalo; q.flags = D.flags; q.nbnd = (int32_t)(rgo >> 40); q.p = p;
        q.loc_off = rgo & 0xffffffffffll;
        q.c1 = pcoef[4 * (int64_t)p]; q.c2 = pcoef[4 * (int64_t)p + 1]; q.beta = pcoef[4 * (int64_t)p + 2]; q.pad = 0.0;
        st_desc[k] = q;
    }
}

/*
 * hq_k_patch_stencil: one step of a STENCIL patch (see hq_stencil and hq_ragged_match): the patch's nodes and elements
 * lie on the 10x10x10 lattice -- all of it (the interior of a uniform region) or a subset (a domain face, a partition
 * interface, a 9-wide far-face cube: RAGGED).  One workgroup per patch, not persistent: 512 threads (768 for the
 * far-face cubes of 513 .. 729 owned nodes), few registers and 25 KB of LDS (42 KB in the launches over ragged
 * patches), so three workgroups share a CU and the hardware overlaps one patch's loads with another's arithmetic; the
 * patches of a launch are in Z-order, so neighbours' rings meet in the XCD's L2.  (A persistent, software-pipelined
 * form -- next patch's rows requested before this patch's stencil, two images -- was measured and is slower, as are
 * two patches per workgroup and the ragged patches mixed into the full lattices' launch: docs/LABNOTES.md.)
 * Thread t owns owned node t and loads it and halo node t.  Everything a thread needs is requested up front in the
 * order of the dependency chain: the patch's record and the thread's halo entry ride on the slot alone
 * (hq_k_stencil_entries), the owned rows and the table words need the record, the halo rows the halo entry;
 * w = u1 + beta (u1 - u2) of both nodes goes to the LDS image, and after ONE barrier
 *   B: (ragged patches) the owned nodes with an incomplete element mask, <= 256, compacted list of the table: the sum
 *      over their PRESENT elements' blocks, f = sum_o sum_m E[o][m] w(m), E = c1 E1 + c2 E2 built per patch in LDS,
 *      the octants dealt to the waves (partial sums in LDS, added in octant order: no atomics, one summation order),
 *      and a second barrier;
 *   A: every owned node with all eight elements around it: the assembled 27-point stencil  f = S w;
 *   then the owner's update  u(t+dt) = (f + m2 u1 - m1 u2) / m0  (solver_compute_displacement, psolve.c:4078-4106)
 *   with the node's own n_t row (7 doubles where a dashpot acts); in the launch over the interface patches a node on
 *   the partition interface hands its pure force to the exchange (psolve.c:4301) -- its seed is 0 (negative m0), the
 *   interface kernel finishes it.
 * No atomics, no accumulators.  Same operator as the element kernels, other summation order.
 */
#define HQ_ST_THREADS 512
#define HQ_RG_MAXB 256
#define HQ_ST_LDS_RAGGED (8 * (576 + 3 * 2 * HQ_RG_MAXB))
#ifdef HQ_ST_TIMING      /* experiment builds only: per-patch clock stamps of hq_k_patch_stencil */
__device__ unsigned long long* g_hq_st_time = nullptr;
#define HQ_ST_STAMP(k) do { if (threadIdx.x == 0 && g_hq_st_time) g_hq_st_time[4 * (size_t)slot + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HQ_ST_STAMP(k) do { } while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define HQ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define HQ_SCHED_FENCE() do { } while (0)
#endif

template <int NT>                /* 512, or 768 for the far-face patches of 513 .. 729 owned nodes */
__global__ void __launch_bounds__(NT)
hq_k_patch_stencil(int32_t count, int32_t per_xcd, const hq_st_desc* __restrict__ st_desc,
                   const int2* __restrict__ st_halo, const hq_real* __restrict__ u1g,
                   const hq_real* __restrict__ u2g, hq_real* __restrict__ ung, const double* __restrict__ nt,
                   const double* __restrict__ nt3, const int32_t* __restrict__ src_ptr,
                   const int32_t* __restrict__ src_ent, const double* __restrict__ F, double dt2,
                   const uint32_t* __restrict__ rg_tab,
                   const double* __restrict__ E1, const double* __restrict__ E2,
                   const int32_t* __restrict__ if_slot, double* __restrict__ iforce,
                   const uint16_t* __restrict__ lat_row, hq_stencil_coef sc)
{
    __shared__ __align__(16) double s_w[3 * HQ_ST_ROWS];
    extern __shared__ __align__(16) double s_dyn[];      /* HQ_ST_LDS_RAGGED bytes where the launch has ragged patches */
    double* s_E = s_dyn;
    double* s_fb = s_dyn + 576;
    /* workgroups b and b + 8 share an XCD: each XCD walks a contiguous run of Z-ordered patches */
    const int slot = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (slot >= count) return;
    HQ_ST_STAMP(0);
    const int t = threadIdx.x;
    const hq_st_desc D = st_desc[slot];
    const int p = D.p, nown = D.nown, nhalo = D.nhalo, nbnd = D.nbnd;
    const uint32_t* __restrict__ loc = rg_tab + D.loc_off;                /* row, mask, boundary index of the local nodes */
    const uint32_t* __restrict__ blist = loc + (nown + nhalo);            /* the boundary nodes' words */
    const double c1 = D.c1, c2 = D.c2, beta = D.beta;
    hq_lds_double* __restrict__ img = (hq_lds_double*)s_w;
    /* halo node t and its LDS word: from the slot alone (past the list: the patch's first node again) */
    const int2 hw = st_halo[(int64_t)slot * HQ_ST_HSTRIDE + (NT == HQ_ST_THREADS ? t : (t & (HQ_ST_HSTRIDE - 1)))];
    /* lanes past the owned nodes read the last one again (same row, same value) */
    const bool owner = t < nown;
    const int lA = owner ? t : nown - 1;
    const int64_t gA = (int64_t)D.base + lA;
    double x1[3], x2[3], y1[3], y2[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { x1[d] = u1g[3 * gA + d]; x2[d] = u2g[3 * gA + d]; }
    const int64_t gB = (int64_t)hw.x;
    const uint32_t wA = loc[lA], wB = (uint32_t)hw.y;
    /* B's lane (j, group): boundary node j, the group's octants; groups = 512 / (boundary nodes rounded up to 64, 128
     * or 256) = 8, 4 or 2 of 1, 2 or 4 octants: a wave works on ONE octant at a time (uniform E) */
    const int sh = nbnd <= 64 ? 6 : (nbnd <= 128 ? 7 : 8);
    const int j = t & ((1 << sh) - 1), grp = t >> sh;
    const uint32_t wJ = blist[j < nbnd ? j : 0];
    /* the element matrix blocks of B (L2-resident, the same for every patch): requested with the first loads */
    double e1a = 0.0, e2a = 0.0, e1b = 0.0, e2b = 0.0;
    if (nbnd > 0) { e1a = E1[t & 511]; e2a = E2[t & 511]; e1b = E1[512 + (t & 63)]; e2b = E2[512 + (t & 63)]; }
    HQ_SCHED_FENCE();
#pragma unroll
    for (int d = 0; d < 3; d++) { y1[d] = u1g[3 * gB + d]; y2[d] = u2g[3 * gB + d]; }
    /* the owned node's n_t: mass_simple (negative: the node's seed is 0, its update belongs to the interface kernel or
     * to compute_adjust) from the private 3-double table; the axis terms from it or, where a dashpot acts on the patch,
     * from the 7-double rows (then NTSAME is not set) */
    const int64_t nn = (int64_t)D.base + ((D.flags & HQ_PATCH_NTSAME) ? 0 : lA);
    const double* q3 = nt3 + 3 * nn;
    double m0 = q3[0], rs[3];
    const double m2s = q3[1], m1s = q3[2];
    HQ_SCHED_FENCE();
    /* the node's own contribution to its update, m2 u1 - m1 u2, now: u1, u2 and two of the three masses need not live
     * through the stencil */
#pragma unroll
    for (int d = 0; d < 3; d++) rs[d] = m2s * x1[d] - m1s * x2[d];
    if (!(D.flags & HQ_PATCH_ISO)) {
        const double* q7 = nt + 7 * nn;
#pragma unroll
        for (int d = 0; d < 3; d++) rs[d] = q7[1 + d] * x1[d] - q7[4 + d] * x2[d];
    }
    if (m0 < 0.0) { rs[0] = rs[1] = rs[2] = 0.0; m0 = -m0; }
    const int myrow = HQ_RG_ROW(wA), rowB = HQ_RG_ROW(wB);
    const unsigned mymask = HQ_RG_MASK(wA);
    const unsigned bm = j < nbnd ? HQ_RG_MASK(wJ) : 0u;
    const int brow = HQ_RG_ROW(wJ);
#pragma unroll
    for (int d = 0; d < 3; d++) img[3 * myrow + d] = x1[d] + beta * (x1[d] - x2[d]);
#pragma unroll
    for (int d = 0; d < 3; d++) img[3 * rowB + d] = y1[d] + beta * (y1[d] - y2[d]);
    if (nbnd > 0 && t < HQ_ST_THREADS) {
        s_E[t] = c1 * e1a + c2 * e2a;
        if (t < 64) s_E[512 + t] = c1 * e1b + c2 * e2b;
    }
    /* S = c1 S1 + c2 S2: eight wave-uniform numbers */
    double P[6], Q[2];
#pragma unroll
    for (int i = 0; i < 6; i++) P[i] = c1 * sc.p1[i] + c2 * sc.p2[i];
#pragma unroll
    for (int i = 0; i < 2; i++) Q[i] = c1 * sc.q1[i] + c2 * sc.q2[i];
    __syncthreads();
    HQ_ST_STAMP(1);

    if (nbnd > 0) {                                      /* B: the first 512 threads */
      if (NT == HQ_ST_THREADS || t < HQ_ST_THREADS) {
        const int per = 1 << (sh - 6);
        double g[3] = { 0.0, 0.0, 0.0 };
        const hq_lds_double* __restrict__ Es = (const hq_lds_double*)s_E;
#pragma unroll 1
        for (int k = 0; k < per; k++) {
            const int o = grp * per + k;
            if (__builtin_amdgcn_ballot_w64((bm >> o) & 1) == 0) continue;     /* no lane has this element */
            if ((bm >> o) & 1) {
                /* the node is corner o of the element: its corner mm sits at offset (mm - o) per axis */
                const hq_lds_double* c0 = img + 3 * (brow - (HQ_ST_PX * (o & 1) + HQ_ST_PY * ((o >> 1) & 1) + HQ_ST_PZ * ((o >> 2) & 1)));
                const hq_lds_double* e = Es + o * 72;
#pragma unroll 2
                for (int mm = 0; mm < 8; mm++) {
                    const hq_lds_double* q = c0 + 3 * (HQ_ST_PX * (mm & 1) + HQ_ST_PY * ((mm >> 1) & 1) + HQ_ST_PZ * ((mm >> 2) & 1));
                    const double ux = q[0], uy = q[1], uz = q[2];
#pragma unroll
                    for (int a = 0; a < 3; a++) g[a] = fma(e[9 * mm + 3 * a], ux, fma(e[9 * mm + 3 * a + 1], uy, fma(e[9 * mm + 3 * a + 2], uz, g[a])));
                }
            }
        }
        /* partial sums [group][j]; 2 * 256 slots hold 8 x 64, 4 x 128 or 2 x 256 */
        hq_lds_double* fb = (hq_lds_double*)s_fb + 3 * ((grp << sh) + j);
        fb[0] = g[0]; fb[1] = g[1]; fb[2] = g[2];
      }
        __syncthreads();
    }
    HQ_ST_STAMP(2);
    if (!owner) return;

    double f[3] = { 0.0, 0.0, 0.0 };
    if (mymask == 0xffu) {                               /* A: all eight elements around the node */
        const hq_lds_double* __restrict__ ctr = img + 3 * myrow;
#pragma unroll
        for (int dz = -1; dz <= 1; dz++)
#pragma unroll
            for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                for (int dx = -1; dx <= 1; dx++) {
                    const hq_lds_double* q = ctr + 3 * (HQ_ST_PX * dx + HQ_ST_PY * dy + HQ_ST_PZ * dz);
                    const double ux = q[0], uy = q[1], uz = q[2];
                    const int ax = dx != 0, ay = dy != 0, az = dz != 0;
                    /* diagonal blocks: class = (own axis off the node?) + 2 x (how many of the other two) */
                    f[0] = fma(P[ax + 2 * (ay + az)], ux, f[0]);
                    f[1] = fma(P[ay + 2 * (ax + az)], uy, f[1]);
                    f[2] = fma(P[az + 2 * (ax + ay)], uz, f[2]);
                    /* off-diagonal blocks: q[|third axis|] sgn sgn */
                    if (dx && dy) { const double c = dx * dy > 0 ? Q[az] : -Q[az]; f[0] = fma(c, uy, f[0]); f[1] = fma(c, ux, f[1]); }
                    if (dx && dz) { const double c = dx * dz > 0 ? Q[ay] : -Q[ay]; f[0] = fma(c, uz, f[0]); f[2] = fma(c, ux, f[2]); }
                    if (dy && dz) { const double c = dy * dz > 0 ? Q[ax] : -Q[ax]; f[1] = fma(c, uz, f[1]); f[2] = fma(c, uy, f[2]); }
                }
    } else {                                             /* B's partial sums, in octant order */
        const hq_lds_double* fb = (const hq_lds_double*)s_fb + 3 * HQ_RG_BIDX(wA);
        for (int gq = 0; gq < (HQ_ST_THREADS >> sh); gq++) {
            f[0] += fb[0]; f[1] += fb[1]; f[2] += fb[2];
            fb += 3 << sh;
        }
    }
    if (F && src_ptr[p + 1] > src_ptr[p]) {              /* compute_addforce_s, psolve.c:5917-5927 (entries name the local node,
                                                          * on a lattice patch its row in the element-form image) */
        const int key = (D.flags & HQ_PATCH_LATTICE) ? (int)lat_row[t] : t;
        for (int i = src_ptr[p]; i < src_ptr[p + 1]; i++)
            if (src_ent[2 * i] == key) {
                const int li = src_ent[2 * i + 1];
                for (int d = 0; d < 3; d++) f[d] += F[3 * li + d] * dt2;
            }
    }
    if (if_slot) {                                       /* partition interface: the pure force goes to the exchange */
        const int32_t sl = if_slot[(int64_t)D.base + t];
        if (sl >= 0) { double* o = iforce + 3 * (int64_t)sl; o[0] = f[0]; o[1] = f[1]; o[2] = f[2]; }
    }
    hq_real* out = ung + 3 * ((int64_t)D.base + t);
#pragma unroll
    for (int d = 0; d < 3; d++) out[d] = (f[d] + rs[d]) / m0;
    HQ_ST_STAMP(3);
}


/* ------------------------------------------------------------------------ */
/* device plan                                                              */
/* ------------------------------------------------------------------------ */

#ifdef HQ_ST_TIMING
static void hq_st_timing_report(hq_patch_plan* P)
{
    const int32_t n = P->ns;
    if (n <= 0 || !P->d_order) return;
    unsigned long long* d = nullptr;
    hipMalloc((void**)&d, 32 * (size_t)n);
    hipMemset(d, 0, 32 * (size_t)n);
    hipMemcpyToSymbol(HIP_SYMBOL(g_hq_st_time), &d, sizeof d);
    hipDeviceSynchronize();
    P->timing_armed = true;
    P->d_timing = d;
}
static void hq_st_timing_print(hq_patch_plan* P)
{
    if (!P->d_timing) return;
    const int32_t n = P->ns;
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(4 * (size_t)n);
    std::vector<int32_t> ord((size_t)n);
    hipMemcpy(h.data(), P->d_timing, 32 * (size_t)n, hipMemcpyDeviceToHost);
    hipMemcpy(ord.data(), P->d_order + P->nb + P->ne + P->nr, 4 * (size_t)n, hipMemcpyDeviceToHost);
    double sum[4][4] = {}; long cnt[4] = {};
    unsigned long long t0 = ~0ull, t1 = 0, xend[8] = {};
    const int per_xcd = (n + 7) / 8;
    for (int32_t i = 0; i < n; i++) {
        const unsigned long long* q = &h[4 * (size_t)i];
        if (!q[0]) continue;
        const int32_t p = ord[(size_t)i];
        const bool sec = false;
        const int cls = !(P->h_flags[p] & HQ_PATCH_RAGGED) ? 0 : (P->patch_nown[p] <= HQ_ST_THREADS ? 1 : (sec ? 3 : 2));
        cnt[cls]++;
        for (int k = 1; k < 4; k++) sum[cls][k] += (double)(q[k] - q[0]);
        t0 = std::min(t0, q[0]); t1 = std::max(t1, q[3]);
        xend[i / per_xcd] = std::max(xend[i / per_xcd], q[3]);
    }
    const char* nm[4] = { "full", "ragged<=512", "far first", "far second" };
    fprintf(stderr, "[hq_st_timing] last launch: %.1f us (100 MHz clock)\n", (double)(t1 - t0) / 100.0);
    for (int c = 0; c < 4; c++)
        if (cnt[c]) fprintf(stderr, "[hq_st_timing] %-12s n=%6ld  barrier %.2f us  afterB %.2f us  end %.2f us\n", nm[c], cnt[c],
                            sum[c][1] / cnt[c] / 100.0, sum[c][2] / cnt[c] / 100.0, sum[c][3] / cnt[c] / 100.0);
    for (int x = 0; x < 8; x++) fprintf(stderr, "[hq_st_timing] xcd %d ends at %.1f us\n", x, (double)(xend[x] - t0) / 100.0);
}
#endif

static void hq_patch_free(hq_patch_plan* P)
{
#ifdef HQ_ST_TIMING
    hq_st_timing_print(P);
#endif
#ifdef HQ_PATCH_PROFILING
    if (P->npatches) hq_patch_report_stamps();
#endif
    void* ptrs[] = { P->d_desc, P->d_pidx, P->d_pc1, P->d_pc2, P->d_pbeta, P->d_halo, P->d_src_ptr, P->d_src_ent,
                     P->d_if_ptr, P->d_if_ent, P->d_order, P->d_nt3, P->d_ds_ptr, P->d_ds_ent, P->d_tickets, P->d_lat_row,
                     P->d_rg_tab, P->d_rg_off, P->d_E1, P->d_E2, P->d_if_slot, P->d_pcoef, P->d_st_desc, P->d_st_halo };
    for (void* p : ptrs) if (p) hipFree(p);
    *P = hq_patch_plan();
}

/*
 * Launch order of the patches: [nb element-form patches that own interface nodes | ne other element-form patches |
 * nr stencil patches that own interface nodes | ns other stencil patches], each part in Z-order.
 * if_ptr (or null): CSR of the interface entries.
 */
static int hq_patch_build_order(hq_patch_plan* P, const int32_t* if_ptr, int64_t* bytes)
{
    const int32_t np = P->npatches;
    std::vector<int32_t> order;
    order.reserve((size_t)np);
    auto is_if = [&](int32_t p) { return if_ptr && if_ptr[p + 1] > if_ptr[p]; };
    auto st = [&](int32_t p) { return (P->h_flags[p] & HQ_PATCH_STENCIL) != 0; };
    P->nr_big = P->ns_big = 0;
    for (int32_t p = 0; p < np; p++) if (is_if(p) && !st(p)) order.push_back(p);
    P->nb = (int32_t)order.size();
    for (int32_t p = 0; p < np; p++) if (!is_if(p) && !st(p)) order.push_back(p);
    P->ne = (int32_t)order.size() - P->nb;
    /* stencil patches of more than 512 owned nodes (far faces, HQ_PATCH_RAGGED=1) behind the others of their part:
     * they are stepped by 768-thread workgroups */
    auto big = [&](int32_t p) { return P->patch_nown[(size_t)p] > HQ_ST_THREADS; };
    int32_t mark = (int32_t)order.size();
    for (int32_t p = 0; p < np; p++) if (is_if(p) && st(p) && !big(p)) order.push_back(p);
    for (int32_t p = 0; p < np; p++) if (is_if(p) && st(p) && big(p)) { order.push_back(p); P->nr_big++; }
    P->nr = (int32_t)order.size() - mark;
    mark = (int32_t)order.size();
    /* part 1: the full lattices (no boundary list: 25 KB of LDS) | the ragged patches of <= 512 nodes | the big ones */
    auto rg = [&](int32_t p) { return (P->h_flags[p] & HQ_PATCH_RAGGED) != 0; };
    P->ns_rg = 0;
    for (int32_t p = 0; p < np; p++) if (!is_if(p) && st(p) && !big(p) && !rg(p)) order.push_back(p);
    for (int32_t p = 0; p < np; p++) if (!is_if(p) && st(p) && !big(p) && rg(p)) { order.push_back(p); P->ns_rg++; }
    for (int32_t p = 0; p < np; p++) if (!is_if(p) && st(p) && big(p)) { order.push_back(p); P->ns_big++; }
    P->ns = (int32_t)order.size() - mark;
    P->nragged = P->nstencil = 0;
    for (int32_t p = 0; p < np; p++) {
        P->nragged += (P->h_flags[p] & HQ_PATCH_RAGGED) != 0;
        P->nstencil += (P->h_flags[p] & HQ_PATCH_STENCIL) != 0;
    }
    if (P->nb == 0 && P->ns == 0 && P->nr == 0) return 0;          /* identity order: no table */
    if (!P->d_order) {
        if (hipMalloc((void**)&P->d_order, 4 * (size_t)(np ? np : 1)) != hipSuccess) { g_patch_err = "hipMalloc failed"; return -2; }
        *bytes += (int64_t)(4 * (size_t)np);
    }
    hipMemcpy(P->d_order, order.data(), 4 * order.size(), hipMemcpyHostToDevice);
    if (P->nr + P->ns > 0) {                             /* the stencil part's records and halo entries in launch order */
        const size_t n = (size_t)(P->nr + P->ns);
        if (!P->d_st_desc) {
            if (hipMalloc((void**)&P->d_st_desc, sizeof(hq_st_desc) * n) != hipSuccess ||
                hipMalloc((void**)&P->d_st_halo, sizeof(int2) * HQ_ST_HSTRIDE * n) != hipSuccess) { g_patch_err = "hipMalloc failed"; return -2; }
            *bytes += (int64_t)((sizeof(hq_st_desc) + sizeof(int2) * HQ_ST_HSTRIDE) * n);
        }
        hq_k_stencil_entries<<<(unsigned)n, HQ_ST_HSTRIDE>>>((int32_t)n, P->d_order + P->nb + P->ne, P->d_desc, P->d_pcoef,
                                                             P->d_rg_off, P->d_rg_tab, P->d_halo, P->hstride, P->d_st_desc,
                                                             P->d_st_halo);
        if (hipDeviceSynchronize() != hipSuccess) { g_patch_err = "building the stencil entries failed"; return -3; }
    }
#ifdef HQ_ST_TIMING
    if (!P->timing_armed) hq_st_timing_report(P);
#endif
    return 0;
}

static bool hq_patch_uses_pers(const hq_patch_plan* P);

static int hq_patch_build(hq_patch_plan* P, int64_t E, int64_t N, const int32_t* lnid, const int32_t* xyz,
                          const double* c1, const double* c2, const double* beta, const double* ntab,
                          const hq_dangling& dn, const char* seed0, int64_t* bytes, int64_t n0 = 0)
{
    hq_patch_host H;
    const bool plap_on = hq_opt_int("HQ_PATCH_VERBOSE", 0) > 1;
    auto plap_t = std::chrono::steady_clock::now();
    auto plap = [&](const char* what) {
        if (!plap_on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  patch plan: %-36s %6.2f s\n", what, std::chrono::duration<double>(now - plap_t).count());
        plap_t = now;
    };
    P->n0 = n0;
    P->cfg = hq_patch_cfg_from_env();
    P->pipe = hq_patch_kernel_choice();
    if (dn.n > 0 && P->cfg.vmax == 0) {
        P->cfg.vmax = 384;
        while ((6 * (size_t)P->cfg.nlmax + 3 * (size_t)(P->cfg.pmax + P->cfg.vmax)) * 8 > 160 * 1024) P->cfg.nlmax -= 8;
    }
    /* lattice patches exist for hq_k_patch_pers only: plan with them when the configuration can run it,
     * and again without if this mesh's patches then turn out not to fit it (> 1024 elements in one) */
    auto pers_fits = [&](int32_t nrows, int32_t max_npairs) {
        return (P->pipe == 4 || P->pipe == 6) && P->cfg.nlmax <= HQ_PERS_THREADS &&
               max_npairs <= HQ_PERS_THREADS &&
               (12 * (size_t)nrows + 3 * (size_t)(P->cfg.pmax + P->cfg.vmax) + 36) * sizeof(double) <= 160 * 1024;
    };
    bool want_lattice = !hq_opt_flag("HQ_PATCH_NO_LATTICE") && xyz && P->cfg.pmax >= HQ_LAT_ACC &&
                        pers_fits(std::max(P->cfg.nlmax, HQ_LAT_ROWS), 0);
    std::vector<int32_t> cand;                       /* the shell's elements: found once, used by every plan below */
    for (;;) {
        if (hq_patch_plan_host(P->cfg, E, N, lnid, xyz, dn, want_lattice, &H, n0, &cand) != 0) return -1 /* HQ_ERR_ARG */;
        /* behind bricks the patches are the shell of the mesh -- a few thousand, 15 per CU on the 64M box: the
         * persistent kernels' prologue and work queue cost more than they save there, and one workgroup per patch
         * (hq_k_patch_step, two per CU) fills the device at once: 64M box 1.104 -> 1.060 ms per step, 8M box
         * 0.159 -> 0.148 on the same box.  HQ_PATCH_PIPE overrides. */
        if (n0 > 0 && !hq_opt_has("HQ_PATCH_PIPE") && P->pipe != 0 && H.desc.size() <= 16384) {
            P->pipe = 0;
            if (want_lattice) { want_lattice = false; H = hq_patch_host(); continue; }
        }
        int32_t mp = 0, nl = 0;
        for (size_t p = 0; p < H.desc.size(); p++) { mp = std::max(mp, H.desc[p].npairs); nl += H.lattice[p]; }
        P->nlattice = nl;
        P->nrows = std::max(P->cfg.nlmax, nl ? HQ_LAT_ROWS : 0);
        if (nl == 0 || pers_fits(P->nrows, mp)) break;
        want_lattice = false;
        H = hq_patch_host();
    }
    for (size_t p = 0; p < H.desc.size(); p++) H.desc[p].flags = H.lattice[p] ? HQ_PATCH_LATTICE : 0;
    P->patch_lat = H.lattice;
    {
        int32_t mp = 0;
        P->max_nacc = P->nlattice ? HQ_LAT_ACC : 0;
        for (auto& D : H.desc) { P->max_nacc = std::max(P->max_nacc, D.nacc); mp = std::max(mp, D.npairs); }
        /* hq_k_patch_seed: two images and THREE accumulator arrays must fit the 160 KiB of LDS */
        /* (and the persistent form itself must be the one that runs: hq_patch_uses_pers -- the fallback kernel
         * hq_k_patch_step divides by mass_simple as it stands and must never see a negated one) */
        P->max_npairs = std::max(P->max_npairs, mp);
        P->seeded = P->pipe == 6 && seed0 && hq_patch_uses_pers(P) &&
                    P->cfg.nlmax <= HQ_PERS_THREADS && mp <= HQ_PERS_THREADS &&
                    (12 * (size_t)P->nrows + 9 * (size_t)P->max_nacc + 16) * sizeof(double) <= 160 * 1024;
    }
    plap("patches cut");
    /* ISO patches: mass2_minusaM / mass_minusaM (psolve.c:3454-3468) equal on the three axes
     * for every owned node, i.e. no dashpot touches the patch */
    std::vector<double> nt3((size_t)N * 3);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        /* hq_k_patch_seed: a negative mass_simple marks a node whose accumulator is seeded with 0 (its update belongs
         * to the interface kernel or to compute_adjust, and its pure force is handed on) */
        nt3[3 * n] = (P->seeded && seed0[n]) ? -ntab[7 * n] : ntab[7 * n];
        nt3[3 * n + 1] = ntab[7 * n + 1]; nt3[3 * n + 2] = ntab[7 * n + 4];
    }
    const bool use_iso = !hq_opt_flag("HQ_PATCH_NO_ISO");
    int32_t nntsame = 0;
    for (auto& D : H.desc) {
        bool iso = use_iso;
        for (int32_t n = D.base; n < D.base + D.nown && iso; n++) {
            const double* q = ntab + 7 * (int64_t)n;
            iso = (q[1] == q[2]) && (q[1] == q[3]) && (q[4] == q[5]) && (q[4] == q[6]);
        }
        D.flags = (D.flags & HQ_PATCH_LATTICE) | (iso ? HQ_PATCH_ISO : 0);
        /* interior of a homogeneous region: one n_t row serves the whole patch (bitwise equal rows) */
        bool same = iso && D.nown > 0 && !hq_opt_flag("HQ_PATCH_NO_NTSAME");
        for (int32_t n = D.base + 1; n < D.base + D.nown && same; n++)
            same = memcmp(&nt3[3 * (size_t)n], &nt3[3 * (size_t)D.base], 24) == 0;
        if (same) { D.flags |= HQ_PATCH_NTSAME; nntsame++; }
    }
    int32_t nuniform = 0;
    const bool wform = !(hq_opt_off("HQ_PATCH_WFORM"));
    if (!hq_opt_flag("HQ_PATCH_NO_UNIFORM")) {
        for (auto& D : H.desc) {
            bool uni = D.npairs > 0;
            const int32_t e0 = D.npairs > 0 ? H.pelem[(size_t)D.pair_off] : 0;
            for (int64_t q = D.pair_off; q < D.pair_off + D.npairs && uni; q++) {
                const int32_t e = H.pelem[(size_t)q];
                uni = c1[e] == c1[e0] && c2[e] == c2[e0] && beta[e] == beta[e0];
            }
            if (uni) {
                D.flags |= HQ_PATCH_UNIFORM;
                nuniform++;
                if (wform && 2 * D.nown <= P->cfg.nlmax) D.flags |= HQ_PATCH_WFORM;
            }
        }
    }
    {   /* what hq_k_patch_step's LDS image really needs: a thin shell's patches stage far fewer than cfg.nlmax nodes, and
         * every KB less is a workgroup more per CU */
        int32_t nl = 8;
        for (auto& D : H.desc) nl = std::max(nl, std::max(D.nown + D.nhalo, (D.flags & HQ_PATCH_WFORM) ? 2 * D.nown : 0));
        P->step_nl = std::min(P->cfg.nlmax, (nl + 7) & ~7);
    }
    plap("uniform / n_t classes");
    /* stencil patches: uniform coefficients, nodes and elements a subset of the lattice, no hanging node's force to
     * distribute (full_only: only the full lattice without dashpot) */
    std::vector<uint32_t> rg_tab;
    std::vector<int64_t> rg_off(H.desc.size(), 0);       /* table offset + 2^40 x boundary nodes */
    if (hq_stencil().ok && xyz && !hq_opt_flag("HQ_PATCH_NO_STENCIL")) {
        const bool full_only = !(hq_patch_ragged_env() >= 0 ? hq_patch_ragged_env() != 0 : P->ragged_default);
        std::vector<std::vector<uint32_t>> tabs(H.desc.size());
        std::vector<int32_t> nbnds(H.desc.size(), 0);
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t p = 0; p < (int64_t)H.desc.size(); p++) {
            const hq_patch_desc& D = H.desc[(size_t)p];
            if (!(D.flags & HQ_PATCH_UNIFORM) || D.nacc != D.nown) continue;
            if (full_only && !((D.flags & HQ_PATCH_LATTICE) && (D.flags & HQ_PATCH_ISO))) continue;
            if (!H.ds_ptr.empty() && H.ds_ptr[(size_t)p + 1] > H.ds_ptr[(size_t)p]) continue;
            std::vector<int32_t> h(H.halo.begin() + D.halo_off, H.halo.begin() + D.halo_off + D.nhalo);
            if (!hq_ragged_match(D.base, D.nown, lnid, xyz, &H.pelem[(size_t)D.pair_off], D.npairs, h, tabs[(size_t)p], &nbnds[(size_t)p]))
                tabs[(size_t)p].clear();
        }
        std::unordered_map<uint64_t, std::vector<int64_t>> seen;        /* one table per patch shape */
        for (size_t p = 0; p < H.desc.size(); p++) {
            const std::vector<uint32_t>& tb = tabs[p];
            if (tb.empty()) continue;
            uint64_t hsh = 1469598103934665603ull ^ tb.size();
            for (uint32_t v : tb) { hsh ^= v; hsh *= 1099511628211ull; }
            int64_t off = -1;
            for (int64_t c : seen[hsh])
                if (c + (int64_t)tb.size() <= (int64_t)rg_tab.size() && !memcmp(&rg_tab[(size_t)c], tb.data(), 4 * tb.size())) { off = c; break; }
            if (off < 0) { off = (int64_t)rg_tab.size(); rg_tab.insert(rg_tab.end(), tb.begin(), tb.end()); seen[hsh].push_back(off); }
            rg_off[p] = off | ((int64_t)nbnds[p] << 40);
            H.desc[p].flags |= HQ_PATCH_STENCIL;
            if (nbnds[p] != 0 || !(H.desc[p].flags & HQ_PATCH_ISO) || H.desc[p].nown != HQ_LAT_NOWN || H.desc[p].nhalo != HQ_LAT_NHALO)
                H.desc[p].flags |= HQ_PATCH_RAGGED;
        }
        for (int i = 0; i < 4; i++) rg_tab.push_back(0); /* a lane past the boundary list reads its first entry's place */
    }
    P->h_flags.resize(H.desc.size());
    for (size_t p = 0; p < H.desc.size(); p++) P->h_flags[p] = H.desc[p].flags;
    P->ndistinct = H.ndistinct;
    P->nuniform = nuniform;
    {
        int nst = 0;
        for (auto& D : H.desc) nst += (D.flags & HQ_PATCH_STENCIL) != 0;
        if (hq_opt_flag("HQ_PATCH_VERBOSE"))
        fprintf(stderr, "hq patch plan: %zu patches (%d lattice, %d stencil), %d distinct local connectivities, %d with uniform coefficients, %d with one n_t row\n",
                H.desc.size(), P->nlattice, nst, H.ndistinct, nuniform, nntsame);
    }
    P->npatches = (int32_t)H.desc.size();
    plap("stencil tables");
    P->npairs = (int64_t)H.pelem.size();
    P->nhalo = (int64_t)H.halo.size();
    P->hstride = H.hstride;
    std::vector<double> v((size_t)P->npairs);
    size_t np = (size_t)P->npairs ? (size_t)P->npairs : 1, nh = H.halo.size() ? H.halo.size() : 1;
#define HQ_PA(ptr, bytes_)                                                                   \
    if (hipMalloc((void**)&(ptr), (bytes_)) != hipSuccess) { g_patch_err = "hipMalloc failed"; hq_patch_free(P); return -2; } \
    *bytes += (int64_t)(bytes_);
    HQ_PA(P->d_desc, sizeof(hq_patch_desc) * H.desc.size())
    HQ_PA(P->d_pidx, 16 * np)
    HQ_PA(P->d_pc1, 8 * np)
    HQ_PA(P->d_pc2, 8 * np)
    HQ_PA(P->d_pbeta, 8 * np)
    HQ_PA(P->d_halo, 4 * nh)
    HQ_PA(P->d_nt3, 8 * nt3.size())
    HQ_PA(P->d_tickets, 4 * 8 * HQ_TICKET_STRIDE)
    hipMemset(P->d_tickets, 0, 4 * 8 * HQ_TICKET_STRIDE);
    if (P->nlattice) {
        HQ_PA(P->d_lat_row, sizeof(uint16_t) * 1024)
        hipMemcpy(P->d_lat_row, hq_lattice().row_of_local, sizeof(uint16_t) * 1024, hipMemcpyHostToDevice);
    }
    if (!rg_tab.empty()) {
        HQ_PA(P->d_rg_tab, sizeof(uint32_t) * rg_tab.size())
        HQ_PA(P->d_rg_off, sizeof(int64_t) * rg_off.size())
        {
            std::vector<double> pco(4 * H.desc.size(), 0.0);
            for (size_t q = 0; q < H.desc.size(); q++)
                if (H.desc[q].flags & HQ_PATCH_STENCIL) {
                    const int32_t e0 = H.pelem[(size_t)H.desc[q].pair_off];
                    pco[4 * q] = c1[e0]; pco[4 * q + 1] = c2[e0]; pco[4 * q + 2] = beta[e0];
                }
            HQ_PA(P->d_pcoef, sizeof(double) * pco.size())
            hipMemcpy(P->d_pcoef, pco.data(), sizeof(double) * pco.size(), hipMemcpyHostToDevice);
        }
        HQ_PA(P->d_E1, sizeof(double) * 576)
        HQ_PA(P->d_E2, sizeof(double) * 576)
        hipMemcpy(P->d_rg_tab, rg_tab.data(), sizeof(uint32_t) * rg_tab.size(), hipMemcpyHostToDevice);
        hipMemcpy(P->d_rg_off, rg_off.data(), sizeof(int64_t) * rg_off.size(), hipMemcpyHostToDevice);
        hipMemcpy(P->d_E1, hq_stencil().E1, sizeof(double) * 576, hipMemcpyHostToDevice);
        hipMemcpy(P->d_E2, hq_stencil().E2, sizeof(double) * 576, hipMemcpyHostToDevice);
    }
    if (dn.n > 0) {
        HQ_PA(P->d_ds_ptr, 4 * H.ds_ptr.size())
        HQ_PA(P->d_ds_ent, 4 * (H.ds_ent.size() ? H.ds_ent.size() : 1))
        hipMemcpy(P->d_ds_ptr, H.ds_ptr.data(), 4 * H.ds_ptr.size(), hipMemcpyHostToDevice);
        hipMemcpy(P->d_ds_ent, H.ds_ent.data(), 4 * H.ds_ent.size(), hipMemcpyHostToDevice);
        P->h_halo = H.halo;
        P->h_halo_off.resize(H.desc.size());
        P->h_nvirt.resize(H.desc.size());
        for (size_t p = 0; p < H.desc.size(); p++) {
            P->h_halo_off[p] = H.desc[p].halo_off;
            P->h_nvirt[p] = H.desc[p].nacc - H.desc[p].nown;
        }
    }
#undef HQ_PA
    hipMemcpy(P->d_nt3, nt3.data(), 8 * nt3.size(), hipMemcpyHostToDevice);
    hipMemcpy(P->d_desc, H.desc.data(), sizeof(hq_patch_desc) * H.desc.size(), hipMemcpyHostToDevice);
    hipMemcpy(P->d_pidx, H.pidx.data(), 16 * (size_t)P->npairs, hipMemcpyHostToDevice);
    hipMemcpy(P->d_halo, H.halo.data(), 4 * H.halo.size(), hipMemcpyHostToDevice);
    const double* src[3] = { c1, c2, beta };
    double* dst[3] = { P->d_pc1, P->d_pc2, P->d_pbeta };
    for (int k = 0; k < 3; k++) {
        for (int64_t q = 0; q < P->npairs; q++) v[(size_t)q] = src[k][H.pelem[(size_t)q]];
        hipMemcpy(dst[k], v.data(), 8 * (size_t)P->npairs, hipMemcpyHostToDevice);
    }
    for (auto& D : H.desc) {
        P->max_nown = std::max(P->max_nown, D.nown);
        P->max_nhalo = std::max(P->max_nhalo, D.nhalo);
        P->max_npairs = std::max(P->max_npairs, D.npairs);
    }
    P->patch_base.resize(H.desc.size());
    P->patch_nown.resize(H.desc.size());
    for (size_t p = 0; p < H.desc.size(); p++) { P->patch_base[p] = H.desc[p].base; P->patch_nown[p] = H.desc[p].nown; }
    if (hq_patch_build_order(P, nullptr, bytes) != 0) return -2;
    plap("uploads");
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

/* group the loaded nodes (Global.theNodesLoadedList, psolve.c:5917-5918) by owning patch */
static int hq_patch_set_source(hq_patch_plan* P, int32_t nloaded, const int32_t* loaded, int64_t* bytes)
{
    if (P->d_src_ptr) { hipFree(P->d_src_ptr); P->d_src_ptr = nullptr; }
    if (P->d_src_ent) { hipFree(P->d_src_ent); P->d_src_ent = nullptr; }
    if (nloaded <= 0) return 0;
    /* (patch, local accumulator, loaded index): the owning patch, plus every patch that keeps a
     * "virtual" accumulator of a loaded hanging node (its force is distributed to their anchors) */
    std::vector<std::array<int32_t, 3>> rec;
    for (int32_t i = 0; i < nloaded; i++) {
        if (loaded[i] < P->n0) continue;                          /* a brick node: hq_brick_set_source */
        int32_t p = (int32_t)(std::upper_bound(P->patch_base.begin(), P->patch_base.end(), loaded[i]) -
                              P->patch_base.begin()) - 1;
        const int32_t lo = loaded[i] - P->patch_base[p];          /* accumulators are indexed by LDS row */
        rec.push_back({ p, P->patch_lat[p] ? (int32_t)hq_lattice().row_of_local[lo] : lo, i });
        for (size_t q = 0; q < P->h_nvirt.size(); q++) {
            const int32_t* v = P->h_halo.data() + P->h_halo_off[q];
            const int32_t* hit = std::lower_bound(v, v + P->h_nvirt[q], loaded[i]);
            if (hit != v + P->h_nvirt[q] && *hit == loaded[i])
                rec.push_back({ (int32_t)q, P->patch_nown[q] + (int32_t)(hit - v), i });
        }
    }
    std::sort(rec.begin(), rec.end());
    std::vector<int32_t> ptr((size_t)P->npatches + 1, 0), ent(rec.size() * 2);
    for (auto& r : rec) ptr[r[0] + 1]++;
    for (int32_t p = 0; p < P->npatches; p++) ptr[p + 1] += ptr[p];
    for (size_t k = 0; k < rec.size(); k++) { ent[2 * k] = rec[k][1]; ent[2 * k + 1] = rec[k][2]; }
    if (hipMalloc((void**)&P->d_src_ptr, 4 * ptr.size()) != hipSuccess) return -2;
    if (hipMalloc((void**)&P->d_src_ent, 4 * ent.size()) != hipSuccess) return -2;
    *bytes += (int64_t)(4 * ptr.size() + 4 * ent.size());
    hipMemcpy(P->d_src_ptr, ptr.data(), 4 * ptr.size(), hipMemcpyHostToDevice);
    hipMemcpy(P->d_src_ent, ent.data(), 4 * ent.size(), hipMemcpyHostToDevice);
    return 0;
}

/* slot[n] >= 0 for nodes on the partition interface */
static int hq_patch_set_interface(hq_patch_plan* P, const int32_t* slot, int64_t nnodes, int64_t* bytes)
{
    std::vector<int32_t> ptr((size_t)P->npatches + 1, 0), ent;
    for (int32_t p = 0; p < P->npatches; p++) {
        for (int32_t n = 0; n < P->patch_nown[p]; n++) {
            int32_t sl = slot[P->patch_base[p] + n];
            if (sl >= 0) { ent.push_back(P->patch_lat[p] ? (int32_t)hq_lattice().row_of_local[n] : n); ent.push_back(sl); }
        }
        ptr[p + 1] = (int32_t)(ent.size() / 2);
    }
    if (ent.empty()) return 0;
    /* launch order: patches owning interface nodes first, so their partial forces can travel while the rest of
     * the partition is still being computed; they hand partial forces on, so they take the element form */
    bool any_st_if = false;
    for (int32_t p = 0; p < P->npatches; p++)
        if (ptr[p + 1] > ptr[p] && (P->h_flags[p] & HQ_PATCH_STENCIL)) {
            if (hq_patch_ragged_env() == 2) { P->h_flags[p] &= ~(HQ_PATCH_STENCIL | HQ_PATCH_RAGGED); continue; }
            P->h_flags[p] |= HQ_PATCH_RAGGED;                /* hands partial forces on: launched ahead of the exchange */
            any_st_if = true;
        }
    if (P->d_if_slot) { hipFree(P->d_if_slot); P->d_if_slot = nullptr; }
    if (any_st_if) {
        if (hipMalloc((void**)&P->d_if_slot, 4 * (size_t)nnodes) != hipSuccess) { g_patch_err = "hipMalloc failed"; return -2; }
        *bytes += (int64_t)(4 * (size_t)nnodes);
        hipMemcpy(P->d_if_slot, slot, 4 * (size_t)nnodes, hipMemcpyHostToDevice);
    }
    if (hq_patch_build_order(P, ptr.data(), bytes) != 0) return -2;
    if (hipMalloc((void**)&P->d_if_ptr, 4 * ptr.size()) != hipSuccess) { g_patch_err = "hipMalloc failed"; return -2; }
    if (hipMalloc((void**)&P->d_if_ent, 4 * ent.size()) != hipSuccess) { g_patch_err = "hipMalloc failed"; return -2; }
    *bytes += (int64_t)(4 * ptr.size() + 4 * ent.size());
    hipMemcpy(P->d_if_ptr, ptr.data(), 4 * ptr.size(), hipMemcpyHostToDevice);
    hipMemcpy(P->d_if_ent, ent.data(), 4 * ent.size(), hipMemcpyHostToDevice);
    return 0;
}

#ifdef HQ_PATCH_PROFILING
static unsigned long long* g_hq_stamp_buf = nullptr;
static unsigned long long* g_hq_wg_buf = nullptr;
static int32_t g_hq_stamp_n = 0;
static void hq_patch_report_stamps(void)
{
    if (!g_hq_stamp_buf) return;
    std::vector<unsigned long long> h((size_t)g_hq_stamp_n * 8);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), g_hq_stamp_buf, 64 * (size_t)g_hq_stamp_n, hipMemcpyDeviceToHost);
    static const int pipe_ = hq_patch_kernel_choice() == 0 ? 0 : (hq_patch_kernel_choice() == 6 ? 6 : 4);
    const char* name0[6] = { "descriptor", "issue+wait staging, LDS write", "barrier 1", "element loop", "src/ds + barrier 2", "update + stores issued" };
    const char* name4[6] = { "row wait + node loads issued", "element section", "row/n_t loads, src/ds, barrier", "wait node data, LDS write", "update + stores issued", "barrier" };
    const char* name6[6] = { "node requests issued, descriptor", "gather + arithmetic", "row/n_t requests, atomics, image + seeds", "barrier", "update + stores issued", "-" };
    const char** name = pipe_ == 4 ? name4 : (pipe_ == 6 ? name6 : name0);
    double sum[6] = { 0, 0, 0, 0, 0, 0 };
    long cnt = 0;
    for (int32_t p = 0; p < g_hq_stamp_n; p++) {
        const unsigned long long* s = &h[8 * (size_t)p];
        if (!s[6] || !s[0]) continue;
        for (int k = 0; k < 6; k++) sum[k] += (double)(s[k + 1] - s[k]);
        cnt++;
    }
    if ((pipe_ == 4 || pipe_ == 6) && g_hq_wg_buf) {
        unsigned long long w[512];
        hipMemcpy(w, g_hq_wg_buf, sizeof w, hipMemcpyDeviceToHost);
        /* s_memrealtime: the 100 MHz reference clock, common to the chip */
        unsigned long long t0 = ~0ull, t1 = 0; double me = 0; int c = 0;
        for (int i = 0; i < 256; i++) if (w[2 * i] && w[2 * i + 1]) { if (w[2 * i] < t0) t0 = w[2 * i]; if (w[2 * i + 1] > t1) t1 = w[2 * i + 1]; c++; }
        for (int i = 0; i < 256; i++) if (w[2 * i] && w[2 * i + 1]) me += (double)(w[2 * i + 1] - t0);
        if (c) fprintf(stderr, "  workgroups of the last launch: %d; first start -> last exit %.1f us, mean exit at %.1f us "
                       "(the mean workgroup idles %.1f %% at the end)\n", c, (double)(t1 - t0) / 100.0, me / c / 100.0,
                       100.0 * ((double)(t1 - t0) - me / c) / (double)(t1 - t0));
    }
    if (pipe_ == 4 || pipe_ == 6) {
        double pre = 0; long c7 = 0;
        for (int32_t p = 0; p < g_hq_stamp_n; p++) {
            const unsigned long long* s = &h[8 * (size_t)p];
            if (!s[6] || !s[0] || !s[7]) continue;
            pre += (double)(s[7] - s[0]); c7++;
        }
        fprintf(stderr, "  (of the first phase, up to and including the issue of the node loads: %.0f)\n", pre / (c7 ? c7 : 1));
    }
    fprintf(stderr, "hq patch stamps (mean shader cycles per workgroup over %ld patches, last step):\n", cnt);
    double tot = 0;
    for (int k = 0; k < 6; k++) tot += sum[k] / (cnt ? cnt : 1);
    for (int k = 0; k < 6; k++) fprintf(stderr, "  %-34s %9.0f  (%4.1f %%)\n", name[k], sum[k] / (cnt ? cnt : 1), 100.0 * sum[k] / (cnt ? cnt : 1) / tot);
    fprintf(stderr, "  %-34s %9.0f\n", "total", tot);
}
#endif


/* hq_k_patch_pers wants one local node and one element per thread and two node buffers in LDS
 * (the planner keeps owned + halo nodes of every patch <= cfg.nlmax) */
static bool hq_patch_uses_pers(const hq_patch_plan* P)
{
    if (P->pipe != 4 && P->pipe != 6) return false;
    if (P->cfg.nlmax > HQ_PERS_THREADS || P->max_npairs > HQ_PERS_THREADS) return false;
    return (12 * (size_t)P->nrows + 3 * (size_t)(P->cfg.pmax + P->cfg.vmax) + 36) * sizeof(double) <= 160 * 1024;
}

/* launch patches order[first .. first+count) (order == identity when there is no interface) */
static void hq_patch_launch(const hq_patch_plan* P, int32_t first, int32_t count, const hq_real* u1,
                            const hq_real* u2, hq_real* un, const double* nt, const double* F, double dt2,
                            double* iforce, hipStream_t stream, int reserve_cus = 0)
{
    if (count <= 0) return;
#ifdef HQ_EXPERIMENT            /* profiles/tools only: what a step costs WITHOUT its shell (results are wrong) */
    { static const bool skip = getenv("HQ_X_NO_SHELL") != nullptr; if (skip) return; }
#endif
    int per_xcd = (count + 7) / 8;
    /* the LDS image as large as the plan's patches need it (a thin shell stages far fewer than cfg.nlmax nodes) */
    const int32_t step_nl = P->step_nl > 0 ? P->step_nl : P->cfg.nlmax;
    const int32_t step_na = P->step_nl > 0 && P->max_nacc > 0 ? P->max_nacc : P->cfg.pmax + P->cfg.vmax;
    size_t lds = (6 * (size_t)step_nl + 3 * (size_t)step_na) * sizeof(double);
    static const bool nt_hint = hq_opt_on("HQ_PATCH_NT");
#ifdef HQ_PATCH_PROFILING
    if (getenv("HQ_PATCH_DIAG") && atoi(getenv("HQ_PATCH_DIAG")) == 6) {
        static unsigned long long* d_st = nullptr;
        if (!d_st) {
            hipMalloc((void**)&d_st, 64 * (size_t)P->npatches);
            hipMemset(d_st, 0, 64 * (size_t)P->npatches);
            hipMemcpyToSymbol(HIP_SYMBOL(g_hq_stamps), &d_st, sizeof d_st);
            g_hq_stamp_buf = d_st; g_hq_stamp_n = P->npatches;
            hipMalloc((void**)&g_hq_wg_buf, 16 * 256);
            hipMemset(g_hq_wg_buf, 0, 16 * 256);
            hipMemcpyToSymbol(HIP_SYMBOL(g_hq_wg), &g_hq_wg_buf, sizeof g_hq_wg_buf);
        }
    }
#endif
    /* (the planner keeps owned + halo nodes of every patch <= cfg.nlmax) */
    if (hq_patch_uses_pers(P)) {
        const int32_t nfacc = 3 * (P->cfg.pmax + P->cfg.vmax);
        size_t lds4 = (12 * (size_t)P->nrows + (size_t)nfacc + 4) * sizeof(double);   /* + ticket ring */
        {
            /* a persistent workgroup holds its CU until the queue is empty and stream priorities do not
             * preempt resident waves: on a partition the interior launch leaves `reserve_cus` CUs (a multiple
             * of 8, one or more per XCD) to the exchange chain's kernels (pack, RCCL, interface update) */
            int grid = std::max(8, P->grid_cus - (reserve_cus & ~7));
            while (grid > 8 && (grid >> 3) > per_xcd) grid -= 8;
            if (P->seeded) {
                const int32_t nfa = 3 * P->max_nacc;
                const size_t ldss = (12 * (size_t)P->nrows + 3 * (size_t)nfa + 4) * sizeof(double);
                hq_k_patch_seed<<<grid, HQ_PERS_THREADS, ldss, stream>>>(
                    count, per_xcd, P->d_order ? P->d_order + first : nullptr, P->nrows, nfa, P->d_desc, P->d_pidx,
                    P->d_pc1, P->d_pc2, P->d_pbeta, P->d_halo, u1, u2, un, nt, P->d_nt3, P->d_src_ptr, P->d_src_ent,
                    (P->d_src_ptr ? F : nullptr), dt2, P->d_if_ptr, P->d_if_ent, iforce, P->d_ds_ptr, P->d_ds_ent,
                    P->hstride, P->d_tickets, P->d_lat_row);
                return;
            }
            hq_k_patch_pers<<<grid, HQ_PERS_THREADS, lds4, stream>>>(
                count, per_xcd, P->d_order ? P->d_order + first : nullptr, P->nrows, nfacc, P->d_desc, P->d_pidx,
                P->d_pc1, P->d_pc2, P->d_pbeta, P->d_halo, u1, u2, un, nt, P->d_nt3, P->d_src_ptr, P->d_src_ent,
                (P->d_src_ptr ? F : nullptr), dt2, P->d_if_ptr, P->d_if_ent, iforce, P->d_ds_ptr, P->d_ds_ent,
                P->hstride, P->d_tickets, P->d_lat_row);
            return;
        }
    }
    auto kern = nt_hint ? hq_k_patch_step<true, 0> : hq_k_patch_step<false, 0>;
#ifdef HQ_PATCH_PROFILING   /* ablation builds for profiles/: results are WRONG by construction */
    static const int diag = hq_opt_int("HQ_PATCH_DIAG", 0);
    if (diag == 1) kern = hq_k_patch_step<false, 1>;
    if (diag == 2) kern = hq_k_patch_step<false, 2>;
    if (diag == 3) kern = hq_k_patch_step<false, 3>;
    if (diag == 4) kern = hq_k_patch_step<false, 4>;
    if (diag == 5) kern = hq_k_patch_step<false, 5>;
    if (diag == 6) kern = hq_k_patch_step<false, 6>;
#endif
    kern<<<per_xcd * 8, P->cfg.threads, lds, stream>>>(
        count, per_xcd, P->d_order ? P->d_order + first : nullptr, step_nl, P->d_desc, P->d_pidx, P->d_pc1,
        P->d_pc2, P->d_pbeta, P->d_halo, u1, u2, un, nt, P->d_nt3, P->d_src_ptr, P->d_src_ent,
        (P->d_src_ptr ? F : nullptr), dt2, P->d_if_ptr, P->d_if_ent, iforce, P->d_ds_ptr, P->d_ds_ent,
        P->hstride);
}

/* the stencil patches: part 0 = those that own interface nodes, d_order[nb + ne .. nb + ne + nr) (they also hand their
 * pure force to the exchange), part 1 = the others, d_order[nb + ne + nr .. nb + ne + nr + ns) */
static void hq_patch_launch_stencil(const hq_patch_plan* P, int part, const hq_real* u1, const hq_real* u2, hq_real* un,
                                    const double* nt, const double* F, double dt2, double* iforce, hipStream_t stream)
{
    const int32_t total = part == 0 ? P->nr : P->ns, nbig = part == 0 ? P->nr_big : P->ns_big;
    const int32_t nrg = part == 0 ? total - nbig : P->ns_rg;             /* part 0: every patch hands forces on (ragged) */
    const int64_t first = part == 0 ? 0 : P->nr;         /* entry of the part's first patch */
    /* three launches: full lattices (no boundary list, 25 KB of LDS), ragged <= 512 nodes, ragged 513..729 nodes */
    const int32_t cnt[3] = { total - nbig - nrg, nrg, nbig };
    int64_t e0 = first;
    for (int k = 0; k < 3; k++) {
        const int32_t count = cnt[k];
        if (count <= 0) continue;
        const int per_xcd = (count + 7) / 8;
        const size_t lds = k == 0 ? 0 : HQ_ST_LDS_RAGGED;
#define HQ_ST_ARGS count, per_xcd, P->d_st_desc + e0, P->d_st_halo + e0 * HQ_ST_HSTRIDE, u1, u2, un, nt, P->d_nt3, P->d_src_ptr,  \
        P->d_src_ent, (P->d_src_ptr ? F : nullptr), dt2, P->d_rg_tab, P->d_E1, P->d_E2,                                             \
        part == 0 ? P->d_if_slot : nullptr, iforce, P->d_lat_row, hq_stencil().c
        if (k < 2) hq_k_patch_stencil<HQ_ST_THREADS><<<per_xcd * 8, HQ_ST_THREADS, lds, stream>>>(HQ_ST_ARGS);
        else hq_k_patch_stencil<768><<<per_xcd * 8, 768, lds, stream>>>(HQ_ST_ARGS);
#undef HQ_ST_ARGS
        e0 += count;
    }
}

#endif /* HQ_PATCH_H */
