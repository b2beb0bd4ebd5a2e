/*
 * hq_brick.h -- BRICKS: the bulk of a uniformly refined, homogeneous region stepped by a z-marching kernel on a
 * tile-major node layout (round 3).
 *
 * Same operator as hq_k_patch_stencil -- the assembled 27-point stencil f = S w, S = c1 S1 + c2 S2 = -(c1 K1 + c2 K2)
 * summed over the eight elements around a node (compute_addforce_effective stiffness.c:180-237 + damping_addforce
 * damping.c:29-103 through w = u1 + beta (u1 - u2)), followed by solver_compute_displacement (psolve.c:4072-4114) --
 * but organised around what the 8x8x8 patches of the Z-ordered node array cannot avoid: a patch loads 488 halo rows
 * for 512 owned ones, gathered 24 bytes at a time (7.19 GB per step on the 64M box for 4.87 GB compulsory).
 *
 *   SIMPLE node   all eight elements around it exist, have one size and one (c1, c2, beta); its n_t row has no
 *                 dashpot term; it is neither a hanging node, an anchor, nor named in a communication schedule.
 *   TILE COLUMN   TX x TY (64 x 8) lattice positions, a run of planes along z, ALL of whose nodes are simple.
 *   LAYOUT        the engine renumbers the nodes (hq_create: a permutation between the caller's ids and the device's):
 *                 the nodes of a tile column are consecutive, [plane][y][x] -- a plane is one contiguous 12 KB run --
 *                 and every other node follows in its original (Z-) order, where the patch planner takes over.
 *   UNIT          one tile column x <= CZ planes = one workgroup of 512 threads, thread (x, y) -> its node of every
 *                 plane.  The workgroup marches along z: plane p+1 is requested while plane p is worked on; the
 *                 arriving plane's w goes to one of TWO LDS slots (31 KB in all, one barrier per plane) together with
 *                 its ring of 2 (nx + 2) + 2 ny neighbours outside the tile (ids from a table: they may be nodes of
 *                 other columns or non-brick nodes); a thread reduces the 9 rows around its node ONCE to the 22
 *                 in-plane sums the cube symmetry of S leaves distinct, and those feed the three output planes p-1,
 *                 p, p+1 (accumulators in registers): 27 LDS reads and ~85 fp64 operations per node instead of 81 and
 *                 153+.  HBM: the compulsory 72 B per node + the ring (148 rows of 48 B per 512 nodes, most of them
 *                 L2 hits of the neighbouring columns marching beside this one) + 2 planes per unit + the id tables:
 *                 measured 84 B per node on the 64M box (hq_k_patch_stencil: 106).
 * Measured before integration (profiles/micro/march_stencil.hip, 512 x 512 x 256 nodes): 0.98 ms per step against
 * 0.94 ms for the same loads and stores without the stencil, i.e. the compulsory 72 B per node at 4.9 TB/s.
 * Summation order differs from the element kernels and from hq_k_patch_stencil (same operator): GPU parity bar 1e-9.
 */
#ifndef HQ_BRICK_H
#define HQ_BRICK_H

#include <chrono>
#include <map>

#include "hq_patch.h"

#define HQ_BK_TX 64
#ifndef HQ_BK_TY            /* tile rows = waves of a workgroup (experiment builds: -DHQ_BK_TY=16) */
#define HQ_BK_TY 8
#endif
#define HQ_BK_THREADS (HQ_BK_TX * HQ_BK_TY)
#define HQ_BK_PY (HQ_BK_TX + 2)
#define HQ_BK_PLANE ((HQ_BK_TX + 2) * (HQ_BK_TY + 2))
#ifndef HQ_BK_ATTR            /* experiment builds: -DHQ_BK_ATTR='__attribute__((amdgpu_num_vgpr(120)))' */
#define HQ_BK_ATTR
#endif
#define HQ_BK_NTSAME 1           /* every node of the unit has the same n_t row: it is in the unit's record */
#define HQ_BK_HET 2              /* the elements around the unit's nodes have coefficients of their own: hq_k_brick_het */
#define HQ_BK_TOPFACE 8          /* the plane above the unit's first one (cap plane za - 1) is a DOMAIN FACE normal to z whose nodes
                                  * the unit steps too: their four elements are the ones between the two planes, so everything
                                  * their update needs is in the workgroup already -- the plane's own sums (halved: the stencil's
                                  * even part; plus rho (Uo_x, Uo_y, -Uo_z): its odd part, hq_stencil()), the next plane's g + Uo,
                                  * and their n_t row (7 doubles: dashpots differ per axis), one for the whole face of the unit */
#define HQ_BK_BOTFACE 16         /* the same for the plane below the unit's last one (za + np)                          */
#define HQ_BK_RAGGED 32          /* (round 5) the unit owns a SUBSET of its tile's lattice positions: beside a level interface or a
                                  * material boundary that cuts through the footprint the planes are not full, and what used to be left
                                  * to the patches (a laterally refined basin: 11 % of the nodes at a fifth of the bricks' rate) marches
                                  * too -- every plane's 512 positions come out of an id table [np + 2][ny][nx] behind the ring table
                                  * (v >= 0: a node the unit owns; v <= -2: node -v - 2 of somebody else, loaded for the stencils of the
                                  * owned ones and never written; -1: no node there), the owned nodes are numbered plane by plane
                                  * without gaps from U.base on, and all of them share (c1, c2, beta) and one n_t row (HQ_BK_NTSAME) */
#define HQ_BK_PACKED 4           /* HET, and every element's (c1, c2, beta) comes out of three floats bit for bit
                                  * (hq_material_coef) and every node's n_t row out of two doubles: 12 + 16 bytes per element /
                                  * node and step instead of 24 + 24; the unit's record carries dt^2 h and h in c1, c2 */
/* tile of a HET unit: 62 x 7 owned nodes -- the 64 x 8 threads of the workgroup each evaluate ONE element of the layer,
 * the elements around the owned nodes (element (i, j) has its low corner at node (i - 1, j - 1)); the 78 threads that own
 * no node (row 7, columns 62 and 63) load the <= 142 ring nodes, two each, in the registers the owners use for their node */
#ifndef HQ_BH_WAVES          /* tile rows of threads = waves of a HET workgroup: 8 (two workgroups of 512 per CU) */
#define HQ_BH_WAVES 8
#endif
#define HQ_BH_THREADS (64 * HQ_BH_WAVES)
#define HQ_BH_NRT (64 + 2 * (HQ_BH_WAVES - 1))
#define HQ_BH_TX 62
#define HQ_BH_TY (HQ_BH_WAVES - 1)
#define HQ_BH_PY 65
#define HQ_BH_CS HQ_BH_THREADS            /* stride between c1, c2, beta of an element in a layer's coefficient block */
#define HQ_BH_ROWS (65 * (HQ_BH_WAVES + 1))

struct hq_brick_unit {
    int64_t base;                /* device id of node (0, 0) of the unit's first plane; the unit's nodes are
                                  * base + (plane * ny + y) * nx + x                                            */
    int64_t tab;                 /* the unit's id table in d_tab: ring ids [np + 2][nr] (planes za - 1 .. za + np,
                                  * nr = 2 (nx + 2) + 2 ny: row y = -1, row y = ny, column x = -1, column x = nx),
                                  * then the nodes of plane za - 1 [ny][nx] and of plane za + np [ny][nx]        */
    int32_t nx, ny, np, flags;
    double  c1, c2, beta;        /* of the elements around the unit's nodes                                     */
    double  m0, m2, m1;          /* HQ_BK_NTSAME: mass_simple, mass2_minusaM, mass_minusaM of every node        */
    double  ft[7], fb[7];        /* HQ_BK_TOPFACE / BOTFACE: the n_t row (psolve.h:210-214) of every node of that face plane     */
    int64_t coef;                /* HQ_BK_HET: the unit's element coefficients in d_coef, [np + 1 layers]
                                  * [c1 | c2 | beta][8][64]: layer l lies between the planes za - 1 + l and za + l, element
                                  * (i, j) has its low corner at node (i - 1, j - 1) of the tile; 0 where there is none.
                                  * HQ_BK_PACKED: in d_coef32 (floats), [np + 1 layers][rho | Vs | Vp][8][64]            */
};

struct hq_brick_cfg {
    int cz = 32;                 /* planes per unit (HQ_BRICK_CZ)                                                */
    int minz = 4;                /* shortest run of planes worth a tile column (HQ_BRICK_MINZ)                   */
    int minnodes = 512;          /* fewest nodes worth a tile column (HQ_BRICK_MINNODES)                         */
    int ragged = 1;              /* HQ_BRICK_RAGGED: the second planner round takes partly filled tiles (HQ_BK_RAGGED); 0: round 5's
                                  * first arrangement, 32-wide full tiles (HQ_BRICK_HALF_TILES)                                   */
    int minfill = 128;           /* HQ_BRICK_RAGGED_MINFILL: fewest owned nodes of a plane of a ragged tile column (of 512)      */
    int ragged_het = 1;          /* HQ_BRICK_RAGGED_HET: (round 6) partly filled tiles of the per-element kernel too               */
};

static hq_brick_cfg hq_brick_cfg_from_env(void)
{
    hq_brick_cfg c;
    auto geti = [](const char* n, int def) { return hq_opt_int(n, def); };
    c.cz = std::max(2, geti("HQ_BRICK_CZ", c.cz));
    c.minz = std::max(1, geti("HQ_BRICK_MINZ", c.minz));
    c.minnodes = std::max(1, geti("HQ_BRICK_MINNODES", c.minnodes));
    c.ragged = geti("HQ_BRICK_RAGGED", c.ragged) != 0;
    c.minfill = std::min(HQ_BK_THREADS, std::max(1, geti("HQ_BRICK_RAGGED_MINFILL", c.minfill)));
    c.ragged_het = geti("HQ_BRICK_RAGGED_HET", c.ragged_het) != 0;
    return c;
}

struct hq_brick_host {
    int64_t nb = 0;                          /* brick nodes: device ids [0, nb)                                  */
    std::vector<int32_t> perm;               /* caller's node id -> device id (all N nodes); empty: identity      */
    std::vector<hq_brick_unit> units;        /* launch order: the HQ_BK_NTSAME units first                        */
    int32_t nsame = 0;                       /* units with HQ_BK_NTSAME                                           */
    int32_t nrag = 0;                        /* of those, HQ_BK_RAGGED: the last of the NTSAME units                */
    std::vector<int32_t> tab;                /* id tables (device ids)                                            */
    std::vector<double> coef;                /* element coefficients of the HQ_BK_HET units                       */
    std::vector<float> coef32;               /* ... of the HQ_BK_PACKED ones: rho (sign: see hq_material_coef), Vs, Vp */
    std::vector<double> nt2;                 /* [N][2] {mass_simple, mass_simple - mass_minusaM} of the nodes of packed units (0 elsewhere) */
    int32_t nhet = 0;                        /* HQ_BK_HET units: the last of the launch order                     */
    int32_t npacked = 0;                     /* of those, HQ_BK_PACKED: the last of the HET units                 */
    int32_t nrhet = 0, nrpacked = 0;         /* HQ_BK_HET | HQ_BK_RAGGED units behind them, the packed ones last  */
    int32_t ncolumns = 0, nlevels = 0;
};

struct hq_brick_plan {
    int64_t nb = 0;
    int32_t nunits = 0, nsame = 0, nrag = 0, nhet = 0, npacked = 0, nrhet = 0, nrpacked = 0;
    hq_brick_unit* d_units = nullptr;
    int32_t* d_tab = nullptr;
    double* d_coef = nullptr;
    float* d_coef32 = nullptr;
    double* d_nt2 = nullptr;
    hq_mat_const mat = { 0, 0, 0, 0, 0, 0 };  /* dt, bBase and the thresholds of the packed units (A and h ride in their records) */
    int32_t* d_src_ptr = nullptr;            /* [nunits + 1] source entries per unit (hq_brick_set_source)        */
    int32_t* d_src_ent = nullptr;            /* [n][2] = {node of the unit (plane * ny + y) * nx + x, loaded idx} */
    std::vector<int64_t> h_base;             /* units' first ids, ascending, and their launch slots: owner lookup */
    std::vector<int32_t> h_slot;
    std::vector<int64_t> h_size;
    std::vector<int64_t> h_first;            /* the unit's first PLANE node (U.base): a node's index in the unit is id - h_first,
                                              * negative in its top face plane, >= nx ny np in its bottom face plane */
};

/*
 * Plan the bricks of a mesh (host only).  excl[n] != 0: node n must stay with the patches (hanging nodes, anchors,
 * nodes a schedule names).  Leaves B->nb = 0 (and no permutation) where nothing qualifies.
 * -> 0, or -1 with g_patch_err set (only on inconsistent input: a level whose geometry cannot be understood is
 * skipped, not refused).
 */
/* what solver_init built the eTable from, where the caller hands it over (hq_desc.edata, mat_*): lets HET units pack */
struct hq_mat_src {
    const float* edata = nullptr;            /* [E][4] edgesize, Vp, Vs, rho (edata_t, psolve.h:95-97)            */
    double dt = 0, bbase = 0, thr_damp = 0, thr_vpvs = 0;
};

static int hq_brick_plan_host(int64_t E, int64_t N, const int32_t* lnid, const int32_t* xyz, const double* c1,
                              const double* c2, const double* beta, const double* ntab, const char* excl,
                              hq_brick_host* B, const hq_mat_src* MS = nullptr)
{
    *B = hq_brick_host();
    if (!xyz || E <= 0 || N <= 0) return 0;
    const hq_brick_cfg cfg = hq_brick_cfg_from_env();
    const bool verbose = (hq_opt_int("HQ_PATCH_VERBOSE", 0) > 1);
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  brick plan: %-36s %6.2f s\n", what, std::chrono::duration<double>(now - t_lap).count());
        t_lap = now;
    };
    const int TX = HQ_BK_TX, TY = HQ_BK_TY;
    const bool want_het = !(hq_opt_on("HQ_BRICK_NO_HET"));
    /* the uniform units use the assembled stencil's coefficients: without a verified table (hq_stencil().ok, the gate
     * hq_k_patch_stencil has too) their nodes go to the element-by-element HET units or stay with the patches */
    const bool stencil_ok = hq_stencil().ok;

    /* levels: elements by edge length (ticks) */
    std::vector<int32_t> hs((size_t)E);
    std::map<int32_t, int64_t> cnt;
    {
        bool bad_order = false;
#pragma omp parallel
        {
            std::map<int32_t, int64_t> mine;
#pragma omp for schedule(static) reduction(|| : bad_order)
            for (int64_t e = 0; e < E; e++) {
                const int32_t* id = lnid + 8 * e;
                const int64_t h = (int64_t)xyz[3 * (int64_t)id[1]] - xyz[3 * (int64_t)id[0]];
                if (h <= 0 || h > 0x7fffffff) { bad_order = true; hs[(size_t)e] = 0; continue; }
                hs[(size_t)e] = (int32_t)h;
                mine[(int32_t)h]++;
            }
#pragma omp critical
            for (auto& kv : mine) cnt[kv.first] += kv.second;
        }
        if (bad_order) return 0;                                 /* not the corner order of octor.c:6444-6470: no bricks */
    }
    lap("edge lengths");

    /* elements around every node (any level): a node with four of them can be the interior of a domain face */
    const bool want_faces = stencil_ok && hq_stencil().face_ok && !hq_opt_flag("HQ_BRICK_NO_FACES") && !hq_opt_flag("HQ_BRICK_NO_NTSAME");
    std::vector<uint8_t> touch;
    if (want_faces) {
        touch.assign((size_t)N, 0);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < E * 8; i++) {
            uint8_t& t8 = touch[(size_t)lnid[i]];
#pragma omp atomic
            t8++;
        }
    }

    struct level_t {
        int32_t h;
        int64_t O[3];
        int64_t D[3];                        /* cells per axis; nodes per axis = D + 1 */
        std::vector<int32_t> Ng;             /* node at a lattice position, or -1 */
        std::vector<int32_t> Eg;             /* element in a lattice cell, or -1 */
    };
    struct column_t { int lvl; int32_t x0, y0, nx, ny, z0, nz, ti, tj; int64_t base; int het; int top, bot; int rag; };
    /* a ragged column (column_t.rag = 1 + its index here): which positions it owns, how many before each plane, and the
     * coefficients and n_t row all of them share */
    struct ragged_t { std::vector<uint8_t> own; std::vector<int32_t> pfx; double c[3], m[3]; };
    std::vector<level_t> levels;
    std::vector<column_t> cols;
    std::vector<ragged_t> rags;
    std::vector<int32_t> ntx_of_level;

    for (auto& kv : cnt) {
        const int32_t h = kv.first;
        if (kv.second < (int64_t)cfg.minnodes) continue;
        level_t L;
        L.h = h;
        int64_t mn[3] = { INT64_MAX, INT64_MAX, INT64_MAX }, mx[3] = { INT64_MIN, INT64_MIN, INT64_MIN };
        {
            int64_t mn0 = INT64_MAX, mn1 = INT64_MAX, mn2 = INT64_MAX, mx0 = INT64_MIN, mx1 = INT64_MIN, mx2 = INT64_MIN;
#pragma omp parallel for schedule(static) reduction(min : mn0, mn1, mn2) reduction(max : mx0, mx1, mx2)
            for (int64_t e = 0; e < E; e++) {
                if (hs[(size_t)e] != h) continue;
                const int32_t* p = xyz + 3 * (int64_t)lnid[8 * e];
                mn0 = std::min<int64_t>(mn0, p[0]); mn1 = std::min<int64_t>(mn1, p[1]); mn2 = std::min<int64_t>(mn2, p[2]);
                mx0 = std::max<int64_t>(mx0, p[0]); mx1 = std::max<int64_t>(mx1, p[1]); mx2 = std::max<int64_t>(mx2, p[2]);
            }
            mn[0] = mn0; mn[1] = mn1; mn[2] = mn2; mx[0] = mx0; mx[1] = mx1; mx[2] = mx2;
        }
        bool ok = true;
        for (int d = 0; d < 3; d++) {
            L.O[d] = mn[d];
            if ((mx[d] - mn[d]) % h) ok = false;
            L.D[d] = (mx[d] - mn[d]) / h + 1;
        }
        if (!ok) continue;
        const int64_t NX = L.D[0] + 1, NY = L.D[1] + 1, NZ = L.D[2] + 1;
        const double vol = (double)NX * (double)NY * (double)NZ;
        if (vol > 8.0 * (double)kv.second + 65536.0 || vol > 2.0e9) continue;       /* a sparse level: the patches keep it */
        std::vector<int32_t>& Eg = L.Eg;
        Eg.assign((size_t)(L.D[0] * L.D[1] * L.D[2]), -1);
        L.Ng.assign((size_t)(NX * NY * NZ), -1);
        /* fill in parallel (two elements of one cell, or two ids at one position, overwrite each other), then verify:
         * every element must find itself in its cell and its eight ids at their positions */
        for (int pass = 0; pass < 2 && ok; pass++) {
            bool bad = false;
#pragma omp parallel for schedule(static) reduction(|| : bad)
            for (int64_t e = 0; e < E; e++) {
                if (hs[(size_t)e] != h) continue;
                const int32_t* id = lnid + 8 * e;
                int64_t q[3];
                bool good = true;
                for (int d = 0; d < 3; d++) {
                    const int64_t v = (int64_t)xyz[3 * (int64_t)id[0] + d] - L.O[d];
                    if (v % h) good = false;
                    q[d] = v / h;
                }
                if (!good) { bad = true; continue; }
                int32_t& cell = Eg[(size_t)((q[2] * L.D[1] + q[1]) * L.D[0] + q[0])];
                if (pass == 0) cell = (int32_t)e; else if (cell != (int32_t)e) bad = true;
                for (int c = 0; c < 8; c++) {
                    const int64_t X = q[0] + (c & 1), Y = q[1] + ((c >> 1) & 1), Z = q[2] + ((c >> 2) & 1);
                    const int32_t* p = xyz + 3 * (int64_t)id[c];
                    if (p[0] != L.O[0] + X * h || p[1] != L.O[1] + Y * h || p[2] != L.O[2] + Z * h) { bad = true; break; }
                    int32_t& g = L.Ng[(size_t)((Z * NY + Y) * NX + X)];
                    if (pass == 0) g = id[c]; else if (g != id[c]) bad = true;
                }
            }
            if (bad) ok = false;
        }
        if (!ok) continue;
        lap("level grids");
        /* simple nodes */
        std::vector<char> S((size_t)(NX * NY * NZ), 0);
        int64_t sx0 = INT64_MAX, sx1 = -1, sy0 = INT64_MAX, sy1 = -1;
        int64_t nsimple = 0;
#pragma omp parallel for schedule(static) reduction(min : sx0, sy0) reduction(max : sx1, sy1) reduction(+ : nsimple)
        for (int64_t Z = 1; Z < L.D[2]; Z++)
            for (int64_t Y = 1; Y < L.D[1]; Y++)
                for (int64_t X = 1; X < L.D[0]; X++) {
                    const int32_t n = L.Ng[(size_t)((Z * NY + Y) * NX + X)];
                    if (n < 0 || (excl && excl[n])) continue;
                    const double* q = ntab + 7 * (int64_t)n;
                    if (!((q[1] == q[2]) && (q[1] == q[3]) && (q[4] == q[5]) && (q[4] == q[6]))) continue;
                    int32_t e0 = -1;
                    bool s = true, uni = true;
                    for (int o = 0; o < 8 && s; o++) {
                        const int64_t cx = X - (o & 1), cy = Y - ((o >> 1) & 1), cz = Z - ((o >> 2) & 1);
                        const int32_t e = Eg[(size_t)((cz * L.D[1] + cy) * L.D[0] + cx)];
                        if (e < 0) { s = false; break; }
                        if (e0 < 0) e0 = e;
                        else uni = uni && c1[e] == c1[e0] && c2[e] == c2[e0] && beta[e] == beta[e0];
                    }
                    if (!s) continue;
                    if (!stencil_ok) uni = false;
                    if (!uni && !want_het) continue;
                    S[(size_t)((Z * NY + Y) * NX + X)] = uni ? 2 : 1;   /* 1: all eight elements, coefficients of their own */
                    nsimple++;
                    sx0 = std::min(sx0, X); sx1 = std::max(sx1, X); sy0 = std::min(sy0, Y); sy1 = std::max(sy1, Y);
                }
        if (nsimple < cfg.minnodes) continue;
        /* face nodes: on the first / last plane of the level's grid, touched by exactly the four elements of this level on
         * the inner side (so no element of ANY level lies beyond: a domain face, not a level interface), those with one
         * (c1, c2, beta), inside the grid in x and y, not excluded.  3: top (elements towards +z), 4: bottom */
        if (want_faces && L.D[2] >= 2) {
#pragma omp parallel for schedule(static) collapse(2)
            for (int side = 0; side < 2; side++)
                for (int64_t Y = 1; Y < L.D[1]; Y++)
                    for (int64_t X = 1; X < L.D[0]; X++) {
                        const int64_t Z = side ? L.D[2] : 0, cz = side ? L.D[2] - 1 : 0;
                        const int32_t n = L.Ng[(size_t)((Z * NY + Y) * NX + X)];
                        if (n < 0 || (excl && excl[n]) || touch[(size_t)n] != 4) continue;
                        int32_t e0 = -1;
                        bool s4 = true;
                        for (int o = 0; o < 4 && s4; o++) {
                            const int32_t e = Eg[(size_t)((cz * L.D[1] + (Y - ((o >> 1) & 1))) * L.D[0] + (X - (o & 1)))];
                            if (e < 0) { s4 = false; break; }
                            if (e0 < 0) e0 = e;
                            else s4 = c1[e] == c1[e0] && c2[e] == c2[e0] && beta[e] == beta[e0];
                        }
                        if (s4 && c1[e0] + c2[e0] != 0.0) S[(size_t)((Z * NY + Y) * NX + X)] = side ? 4 : 3;   /* (rho = (c1 - c2) / (c1 + c2)) */
                    }
        }
        lap("simple nodes");
        /* tile columns: footprints on a TX x TY grid from the first simple node; runs of planes all of whose nodes
         * in the footprint are simple.  Two passes: 64 x 8 tiles of nodes whose eight elements share their coefficients
         * (hq_k_brick), then 63 x 7 tiles of what is left (hq_k_brick_het: per-element coefficients) */
        const int lvl = (int)levels.size();
        std::vector<std::vector<column_t>> found;
        int32_t ntx_lvl = 1;
        /* rounds: the 64-wide tiles, then (round 5) 32-wide ones over what is left of the uniform simple nodes -- beside a
         * level interface or a material boundary that runs along y or z a strip of up to 63 nodes per row is left over,
         * and half a workgroup's lanes in the marching kernel still beat the patches by a wide margin -- then the het tiles */
        /* ... or (the default since the ragged units exist) 64-wide tiles again that need not be full: HQ_BK_RAGGED */
        const int rag = cfg.ragged && stencil_ok && !hq_opt_flag("HQ_BRICK_NO_NTSAME") ? 1 : 0;
        const int half = rag || hq_opt_int("HQ_BRICK_HALF_TILES", 1) != 0 ? 1 : 0;
        /* ... and (round 6) behind the full het tiles a round of RAGGED het tiles: 62 x 7 footprints of which every plane of a
         * run holds >= minfill simple nodes of ANY material (hq_k_brick_het<., RAGGED>) -- on a mesh whose material differs
         * from element to element the one-material ragged columns above find nothing */
        const int rag_het = cfg.ragged && want_het && cfg.ragged_het ? 1 : 0;
        for (int round = 0; round < 2 + half + rag_het; round++) {
            const bool rh_round = rag_het && round == 2 + half;
            const int pass = (round == 1 + half || rh_round) ? 1 : 0;
            const bool rag_round = rag && round == 1;
            const int PTX = pass == 1 ? HQ_BH_TX : (round == 0 || rag_round ? TX : TX / 2), PTY = pass == 0 ? TY : HQ_BH_TY;
            /* pass 0: uniform simple nodes (2); pass 1: what is left of them and the per-element ones (1).  3 / 4 are face
             * nodes: never part of a run, but a run of pass 0 that starts / ends beside a full face plane takes it along */
            auto in_run = [pass](char v) { return pass == 0 ? v == 2 : (v == 1 || v == 2); };
            if (round >= 1) {
                sx0 = INT64_MAX; sx1 = -1; sy0 = INT64_MAX; sy1 = -1;
                for (int64_t Z = 0; Z < NZ; Z++)
                    for (int64_t Y = 0; Y < NY; Y++)
                        for (int64_t X = 0; X < NX; X++)
                            if (in_run(S[(size_t)((Z * NY + Y) * NX + X)])) { sx0 = std::min(sx0, X); sx1 = std::max(sx1, X); sy0 = std::min(sy0, Y); sy1 = std::max(sy1, Y); }
                if (sx1 < 0) { if (pass == 1) break; else continue; }
            }
            const int32_t ntx = (int32_t)((sx1 - sx0) / PTX + 1), nty = (int32_t)((sy1 - sy0) / PTY + 1);
            if (round == 0) ntx_lvl = ntx;
            const size_t f0 = found.size();
            found.resize(f0 + (size_t)nty);
            std::vector<std::vector<ragged_t>> found_r(rag_round || rh_round ? (size_t)nty : 0);
            const int32_t minfill = rh_round ? std::max(1, cfg.minfill * (HQ_BH_TX * HQ_BH_TY) / HQ_BK_THREADS) : cfg.minfill;
#pragma omp parallel for schedule(dynamic, 1)
            for (int32_t tj = 0; tj < nty; tj++) {
                const int64_t y0 = sy0 + (int64_t)tj * PTY;
                const int32_t ny = (int32_t)std::min<int64_t>(PTY, sy1 - y0 + 1);
                std::vector<int32_t> cntz;
                for (int32_t ti = 0; ti < ntx; ti++) {
                    const int64_t x0 = sx0 + (int64_t)ti * PTX;
                    const int32_t nx = (int32_t)std::min<int64_t>(PTX, sx1 - x0 + 1);
                    if (rag_round || rh_round) {
                        /* ragged tile columns: runs of planes each of which holds >= minfill uniform simple nodes of ONE
                         * material and n_t row (the first candidate's; up to four materials per footprint, one after the other);
                         * rh_round: >= minfill simple nodes, whatever their elements' coefficients and their n_t rows */
                        cntz.assign((size_t)NZ, 0);
                        for (int iter = 0; iter < (rh_round ? 1 : 4); iter++) {
                            int32_t n0 = -1;
                            int64_t Z0 = 0, X0 = 0, Y0 = 0;
                            auto cand = [&](char v) { return rh_round ? (v == 1 || v == 2) : v == 2; };
                            for (int64_t Z = 1; Z < L.D[2] && n0 < 0; Z++)
                                for (int64_t y = y0; y < y0 + ny && n0 < 0; y++) {
                                    const char* row = &S[(size_t)((Z * NY + y) * NX + x0)];
                                    for (int32_t x = 0; x < nx; x++)
                                        if (cand(row[x])) { n0 = L.Ng[(size_t)((Z * NY + y) * NX + x0 + x)]; Z0 = Z; Y0 = y; X0 = x0 + x; break; }
                                }
                            if (n0 < 0) break;
                            const int32_t e0 = L.Eg[(size_t)(((Z0 - 1) * L.D[1] + (Y0 - 1)) * L.D[0] + (X0 - 1))];
                            const double rc[3] = { c1[e0], c2[e0], beta[e0] };
                            const double* q0 = ntab + 7 * (int64_t)n0;
                            auto match = [&](int64_t X, int64_t Y, int64_t Z) -> bool {
                                if (rh_round) return true;
                                const double* q = ntab + 7 * (int64_t)L.Ng[(size_t)((Z * NY + Y) * NX + X)];
                                if (q[0] != q0[0] || q[1] != q0[1] || q[4] != q0[4]) return false;
                                const int32_t e = L.Eg[(size_t)(((Z - 1) * L.D[1] + (Y - 1)) * L.D[0] + (X - 1))];
                                return c1[e] == rc[0] && c2[e] == rc[1] && beta[e] == rc[2];
                            };
                            for (int64_t Z = Z0; Z < L.D[2]; Z++) {
                                int32_t cnt = 0;
                                for (int64_t y = y0; y < y0 + ny; y++) {
                                    const char* row = &S[(size_t)((Z * NY + y) * NX + x0)];
                                    for (int32_t x = 0; x < nx; x++) cnt += cand(row[x]) && match(x0 + x, y, Z);
                                }
                                cntz[(size_t)Z] = cnt;
                            }
                            int64_t ra = -1;
                            for (int64_t Z = Z0; Z <= L.D[2]; Z++) {
                                const bool in = Z < L.D[2] && cntz[(size_t)Z] >= minfill;
                                if (in) { if (ra < 0) ra = Z; continue; }
                                if (ra < 0) continue;
                                const int64_t nz = Z - ra;
                                int64_t total = 0;
                                for (int64_t z = ra; z < Z; z++) total += cntz[(size_t)z];
                                if (nz >= cfg.minz && total >= cfg.minnodes) {
                                    ragged_t R;
                                    R.own.assign((size_t)(nz * nx * ny), 0);
                                    R.pfx.assign((size_t)nz + 1, 0);
                                    for (int d = 0; d < 3; d++) R.c[d] = rc[d];
                                    R.m[0] = q0[0]; R.m[1] = q0[1]; R.m[2] = q0[4];
                                    for (int64_t z = ra; z < Z; z++) {
                                        int32_t k = 0;
                                        for (int64_t y = y0; y < y0 + ny; y++) {
                                            char* row = &S[(size_t)((z * NY + y) * NX + x0)];
                                            for (int32_t x = 0; x < nx; x++)
                                                if (cand(row[x]) && match(x0 + x, y, z)) { R.own[(size_t)(((z - ra) * ny + (y - y0)) * nx + x)] = 1; row[x] = 0; k++; }
                                        }
                                        R.pfx[(size_t)(z - ra) + 1] = R.pfx[(size_t)(z - ra)] + k;
                                    }
                                    found_r[(size_t)tj].push_back(std::move(R));
                                    found[f0 + (size_t)tj].push_back({ lvl, (int32_t)x0, (int32_t)y0, nx, ny, (int32_t)ra, (int32_t)nz, ti, tj, 0, rh_round ? 1 : 0, 0, 0,
                                                                       (int)found_r[(size_t)tj].size() });
                                }
                                ra = -1;
                            }
                            /* what the runs left of this material waits (5) until the round is over */
                            for (int64_t Z = Z0; Z < L.D[2]; Z++)
                                for (int64_t y = y0; y < y0 + ny; y++) {
                                    char* row = &S[(size_t)((Z * NY + y) * NX + x0)];
                                    for (int32_t x = 0; x < nx; x++) if (!rh_round && row[x] == 2 && match(x0 + x, y, Z)) row[x] = 5;
                                }
                        }
                        continue;
                    }
                    int64_t run0 = -1;
                    for (int64_t Z = 0; Z <= NZ; Z++) {
                        bool full = Z < NZ;
                        for (int64_t y = y0; y < y0 + ny && full; y++) {
                            const char* row = &S[(size_t)((Z * NY + y) * NX + x0)];
                            for (int32_t x = 0; x < nx; x++) if (!in_run(row[x])) { full = false; break; }
                        }
                        if (full) { if (run0 < 0) run0 = Z; continue; }
                        if (run0 >= 0) {
                            const int64_t nz = Z - run0;
                            if (nz >= cfg.minz && nz * nx * ny >= cfg.minnodes) {
                                /* a face plane beside the run: every node of the footprint a face node with ONE n_t row */
                                auto face = [&](int64_t Zf, char cls) -> int {
                                    if (pass != 0 || Zf < 0 || Zf > L.D[2]) return 0;
                                    const double* q0 = nullptr;
                                    for (int64_t y = y0; y < y0 + ny; y++)
                                        for (int32_t x = 0; x < nx; x++) {
                                            if (S[(size_t)((Zf * NY + y) * NX + x0 + x)] != cls) return 0;
                                            const double* q = ntab + 7 * (int64_t)L.Ng[(size_t)((Zf * NY + y) * NX + x0 + x)];
                                            if (!q0) q0 = q;
                                            else if (memcmp(q, q0, 7 * sizeof(double)) != 0) return 0;
                                        }
                                    return 1;
                                };
                                /* ... and only beside a column whose own nodes share one n_t row (its units are then all
                                 * HQ_BK_NTSAME: the kernel form that carries the face code) */
                                auto one_row = [&]() -> bool {
                                    const double* q0 = ntab + 7 * (int64_t)L.Ng[(size_t)((run0 * NY + y0) * NX + x0)];
                                    for (int64_t z = run0; z < Z; z++)
                                        for (int64_t y = y0; y < y0 + ny; y++)
                                            for (int32_t x = 0; x < nx; x++) {
                                                const double* q = ntab + 7 * (int64_t)L.Ng[(size_t)((z * NY + y) * NX + x0 + x)];
                                                if (q[0] != q0[0] || q[1] != q0[1] || q[4] != q0[4]) return false;
                                            }
                                    return true;
                                };
                                int top = run0 == 1 ? face(0, 3) : 0, bot = Z == L.D[2] ? face(L.D[2], 4) : 0;
                                if ((top || bot) && !one_row()) top = bot = 0;
                                found[f0 + (size_t)tj].push_back({ lvl, (int32_t)x0, (int32_t)y0, nx, ny, (int32_t)run0, (int32_t)nz, ti, tj, 0, pass, top, bot });
                            }
                            run0 = -1;
                        }
                    }
                }
            }
            /* the nodes the pass took are gone for the next one */
            for (size_t f = f0; f < found.size(); f++)
                for (auto& c : found[f]) {
                    if (c.rag) continue;                 /* (a ragged column has cleared what it owns) */
                    for (int32_t z = -c.top; z < c.nz + c.bot; z++)
                        for (int32_t y = 0; y < c.ny; y++)
                            memset(&S[(size_t)(((int64_t)(c.z0 + z) * NY + (c.y0 + y)) * NX + c.x0)], 0, (size_t)c.nx);
                }
            if (rag_round || rh_round) {
                for (int32_t tj = 0; tj < nty; tj++) {
                    const size_t r0 = rags.size();
                    for (auto& c : found[f0 + (size_t)tj]) c.rag += (int)r0;
                    for (auto& R : found_r[(size_t)tj]) rags.push_back(std::move(R));
                }
                const int64_t nS = NX * NY * NZ;
#pragma omp parallel for schedule(static)
                for (int64_t i = 0; i < nS; i++) if (S[(size_t)i] == 5) S[(size_t)i] = 2;
            }
            if (!want_het && round + 1 >= 1 + half) break;
            if (rh_round) break;
        }
        lap("tile columns");
        const int32_t ntx = ntx_lvl;
        size_t before = cols.size();
        for (auto& v : found) cols.insert(cols.end(), v.begin(), v.end());
        if (cols.size() == before) continue;
        ntx_of_level.push_back(ntx);
        levels.push_back(std::move(L));
    }
    if (cols.empty()) return 0;

    /* device numbering: tile columns first (plane-major inside a column), everything else behind in its old order */
    int64_t nb = 0;
    /* a column's nodes: [its top face plane][its planes][its bottom face plane] */
    for (auto& c : cols) { c.base = nb; nb += c.rag ? (int64_t)rags[(size_t)c.rag - 1].pfx[(size_t)c.nz] : (int64_t)c.nx * c.ny * (c.nz + c.top + c.bot); }
    if (nb > 0x7fffffff) return 0;
    B->perm.assign((size_t)N, -1);
    {
        /* columns are disjoint (checked: the nodes numbered must come out as nb, and every id below is taken once) */
        int64_t twice = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : twice)
        for (int64_t ci = 0; ci < (int64_t)cols.size(); ci++) {
            const column_t& c = cols[(size_t)ci];
            const level_t& L = levels[(size_t)c.lvl];
            const int64_t NX = L.D[0] + 1, NY = L.D[1] + 1;
            const ragged_t* R = c.rag ? &rags[(size_t)c.rag - 1] : nullptr;
            int64_t next = c.base;                   /* a ragged column numbers what it owns, plane by plane */
            for (int32_t z = -c.top; z < c.nz + c.bot; z++)
                for (int32_t y = 0; y < c.ny; y++)
                    for (int32_t x = 0; x < c.nx; x++) {
                        if (R && !R->own[(size_t)(((int64_t)z * c.ny + y) * c.nx + x)]) continue;
                        const int32_t n = L.Ng[(size_t)(((int64_t)(c.z0 + z) * NY + (c.y0 + y)) * NX + (c.x0 + x))];
                        int32_t& slot = B->perm[(size_t)n];
                        int32_t was;
                        const int32_t id = (int32_t)(R ? next++ : c.base + ((int64_t)(z + c.top) * c.ny + y) * c.nx + x);
#pragma omp atomic capture
                        { was = slot; slot = id; }
                        twice += was != -1;
                    }
        }
        if (twice) { g_patch_err = "brick plan: a node lies in two tile columns"; return -1; }
    }
    {
        /* everything else behind, in its old order: a count per chunk of ids, then the fill */
        int nth = 1;
#pragma omp parallel
        {
#pragma omp single
            nth = omp_get_num_threads();
        }
        std::vector<int64_t> first((size_t)nth + 1, 0);
        /* slices dealt by worksharing loops: independent of the team the runtime delivers */
#pragma omp parallel for schedule(static, 1)
        for (int t = 0; t < nth; t++) {
            const int64_t lo = N * t / nth, hi = N * (t + 1) / nth;
            int64_t cnt = 0;
            for (int64_t n = lo; n < hi; n++) cnt += B->perm[(size_t)n] < 0;
            first[(size_t)t + 1] = cnt;
        }
        for (int i = 0; i < nth; i++) first[(size_t)i + 1] += first[(size_t)i];
#pragma omp parallel for schedule(static, 1)
        for (int t = 0; t < nth; t++) {
            const int64_t lo = N * t / nth, hi = N * (t + 1) / nth;
            int64_t k = nb + first[(size_t)t];
            for (int64_t n = lo; n < hi; n++) if (B->perm[(size_t)n] < 0) B->perm[(size_t)n] = (int32_t)k++;
        }
        if (nb + first[(size_t)nth] != N) { g_patch_err = "brick plan: numbering is not a permutation"; return -1; }
    }
    lap("numbering");
    B->nb = nb;
    B->ncolumns = (int32_t)cols.size();
    B->nlevels = (int32_t)levels.size();

    /* units: a column in chunks of <= cz planes.  Launch order: slabs of tile rows (about 64 tiles per slab and
     * chunk level: what an XCD keeps resident), inside a slab chunk by chunk -- neighbouring columns march side by
     * side on one XCD and find each other's ring rows in its L2 */
    struct unit_t { int col; int32_t za, np; int64_t key; };
    std::vector<unit_t> us;
    /* planes per unit: 32 where that still gives two units per CU; a small mesh (or a rank's share of one) gets shorter
     * chunks -- each costs two more planes of loads, but 64 workgroups do not fill 256 CUs (1 M-element box: 0.068 ms
     * per step with 32 planes per unit, 0.039 with 8; 8 M box: 0.164 / 0.172).  HQ_BRICK_CZ fixes it. */
    int cz = cfg.cz;
    if (!hq_opt_has("HQ_BRICK_CZ")) {
        for (cz = 32; cz > 8; cz /= 2) {
            int64_t n = 0;
            for (auto& c : cols) n += (c.nz + cz - 1) / cz;
            if (n >= 512) break;
        }
    }
    for (size_t ci = 0; ci < cols.size(); ci++) {
        const column_t& c = cols[ci];
        const int32_t nch = (c.nz + cz - 1) / cz;
        const int32_t G = std::max(1, 64 / std::max(1, ntx_of_level[(size_t)c.lvl]));
        for (int32_t k = 0; k < nch; k++) {
            const int32_t za = (int32_t)((int64_t)c.nz * k / nch), zb = (int32_t)((int64_t)c.nz * (k + 1) / nch);
            /* key: level | slab | plane of the chunk's start | tile row | tile column */
            const int64_t key = ((int64_t)c.lvl << 56) | ((int64_t)(c.tj / G) << 42) | ((int64_t)((c.z0 + za) / cz) << 28) |
                                ((int64_t)(c.tj % G) << 20) | (int64_t)c.ti;
            us.push_back({ (int)ci, c.z0 + za, zb - za, key });
        }
    }
    std::stable_sort(us.begin(), us.end(), [](const unit_t& a, const unit_t& b) { return a.key < b.key; });
    B->units.resize(us.size());
    std::vector<int64_t> toff(us.size() + 1, 0);
    for (size_t u = 0; u < us.size(); u++) {
        const column_t& c = cols[(size_t)us[u].col];
        const int64_t nr = 2 * (c.nx + 2) + 2 * c.ny;
        /* (a ragged unit: a table row of nx ny ids for EVERY plane, the two cap planes included) */
        toff[u + 1] = toff[u] + ((int64_t)(us[u].np + 2) * nr + (c.rag ? us[u].np + 2 : 2) * (int64_t)c.nx * c.ny + 3) / 4 * 4;
    }
    B->tab.assign((size_t)toff[us.size()] + 64, 0);
    std::vector<int64_t> coff(us.size() + 1, 0);         /* coefficient blocks of the HET units */
    for (size_t u = 0; u < us.size(); u++)
        coff[u + 1] = coff[u] + (cols[(size_t)us[u].col].het ? (int64_t)(us[u].np + 1) * HQ_BH_THREADS * 3 : 0);
    B->coef.assign((size_t)coff[us.size()] + 8, 0.0);
    const bool try_pack = MS && MS->edata && MS->dt > 0 && !hq_opt_flag("HQ_BRICK_NO_PACK");
    const bool no_ntsame = hq_opt_flag("HQ_BRICK_NO_NTSAME");     /* read here: the workers of the loop below see no options (hq_opts.h) */
    if (try_pack) { B->coef32.assign((size_t)coff[us.size()] + 8, 0.0f); B->nt2.assign(2 * (size_t)N, 0.0); }        /* [N]: a unit also LOADS the rows of its two cap planes, which may be anybody's nodes */
    int fault = 0;                                       /* written by many threads: atomic writes only */
    auto set_fault = [&]() {
#pragma omp atomic write
        fault = 1;
    };
    std::vector<char> same(us.size(), 0);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t u = 0; u < (int64_t)us.size(); u++) {
        const column_t& c = cols[(size_t)us[(size_t)u].col];
        const level_t& L = levels[(size_t)c.lvl];
        const int64_t NX = L.D[0] + 1, NY = L.D[1] + 1;
        const int32_t za = us[(size_t)u].za, np = us[(size_t)u].np, nx = c.nx, ny = c.ny, nr = 2 * (nx + 2) + 2 * ny;
        auto dev = [&](int64_t X, int64_t Y, int64_t Z) -> int32_t {
            if (X < 0 || Y < 0 || Z < 0 || X > L.D[0] || Y > L.D[1] || Z > L.D[2]) { set_fault(); return 0; }
            const int32_t n = L.Ng[(size_t)((Z * NY + Y) * NX + X)];
            if (n < 0) { set_fault(); return 0; }
            return B->perm[(size_t)n];
        };
        int32_t* t = B->tab.data() + toff[(size_t)u];
        hq_brick_unit& U = B->units[(size_t)u];
        if (c.rag) {
            /* HQ_BK_RAGGED: a position of the ring or of a plane may hold no node of this level at all (-1: beyond the
             * level's region; no owned node has it for a neighbour -- a simple node's 26 neighbours exist) */
            const ragged_t& R = rags[(size_t)c.rag - 1];
            auto any = [&](int64_t X, int64_t Y, int64_t Z) -> int32_t {
                if (X < 0 || Y < 0 || Z < 0 || X > L.D[0] || Y > L.D[1] || Z > L.D[2]) return -1;
                const int32_t n = L.Ng[(size_t)((Z * NY + Y) * NX + X)];
                return n < 0 ? -1 : B->perm[(size_t)n];
            };
            /* does the UNIT own the node at (i, j) of plane k (k = 1 .. np)? */
            auto owns = [&](int32_t i, int32_t j, int32_t k) -> bool {
                return i >= 0 && i < nx && j >= 0 && j < ny && k >= 1 && k <= np && R.own[(size_t)((((int64_t)za - 1 + k - c.z0) * ny + j) * nx + i)];
            };
            /* ... and is a position the neighbour of a node the unit owns?  Only those are loaded: the others (the far
             * side of a level interface, another material's nodes) are -1 like the positions without a node */
            auto wanted = [&](int32_t i, int32_t j, int32_t k) -> bool {
                for (int32_t dk = -1; dk <= 1; dk++)
                    for (int32_t dj = -1; dj <= 1; dj++)
                        for (int32_t di = -1; di <= 1; di++)
                            if (owns(i + di, j + dj, k + dk)) return true;
                return false;
            };
            for (int32_t k = 0; k < np + 2; k++) {
                const int64_t Z = (int64_t)za - 1 + k;
                int32_t* r = t + (int64_t)k * nr;
                for (int32_t i = 0; i < nx + 2; i++) {
                    r[i] = wanted(i - 1, -1, k) ? any(c.x0 - 1 + i, c.y0 - 1, Z) : -1;
                    r[nx + 2 + i] = wanted(i - 1, ny, k) ? any(c.x0 - 1 + i, c.y0 + ny, Z) : -1;
                }
                for (int32_t j = 0; j < ny; j++) {
                    r[2 * (nx + 2) + j] = wanted(-1, j, k) ? any(c.x0 - 1, c.y0 + j, Z) : -1;
                    r[2 * (nx + 2) + ny + j] = wanted(nx, j, k) ? any(c.x0 + nx, c.y0 + j, Z) : -1;
                }
                int32_t* pl = t + (int64_t)(np + 2) * nr + (int64_t)k * nx * ny;
                for (int32_t j = 0; j < ny; j++)
                    for (int32_t i = 0; i < nx; i++) {
                        const bool own = owns(i, j, k);
                        const int32_t q = own || wanted(i, j, k) ? any(c.x0 + i, c.y0 + j, Z) : -1;
                        if (own && q < 0) set_fault();
                        pl[j * nx + i] = own ? q : (q < 0 ? -1 : -q - 2);
                    }
            }
            U.base = c.base + R.pfx[(size_t)(za - c.z0)];
            U.tab = toff[(size_t)u];
            U.nx = nx; U.ny = ny; U.np = np; U.flags = HQ_BK_NTSAME | HQ_BK_RAGGED;
            memset(U.ft, 0, sizeof U.ft); memset(U.fb, 0, sizeof U.fb);
            U.coef = 0;
            if (!c.het) {
                U.c1 = R.c[0]; U.c2 = R.c[1]; U.beta = R.c[2];
                U.m0 = R.m[0]; U.m2 = R.m[1]; U.m1 = R.m[2];
                same[(size_t)u] = 4;
                continue;
            }
            /* a ragged unit of the per-element kernel: its coefficient block is filled with the full het units' below */
            U.flags = HQ_BK_RAGGED;
            U.c1 = U.c2 = U.beta = 0.0; U.m0 = 1.0; U.m2 = U.m1 = 0.0;
        }
        if (!c.rag) {
        for (int32_t k = 0; k < np + 2; k++) {
            const int64_t Z = (int64_t)za - 1 + k;
            int32_t* r = t + (int64_t)k * nr;
            for (int32_t i = 0; i < nx + 2; i++) { r[i] = dev(c.x0 - 1 + i, c.y0 - 1, Z); r[nx + 2 + i] = dev(c.x0 - 1 + i, c.y0 + ny, Z); }
            for (int32_t j = 0; j < ny; j++) { r[2 * (nx + 2) + j] = dev(c.x0 - 1, c.y0 + j, Z); r[2 * (nx + 2) + ny + j] = dev(c.x0 + nx, c.y0 + j, Z); }
        }
        int32_t* cap = t + (int64_t)(np + 2) * nr;
        for (int32_t j = 0; j < ny; j++)
            for (int32_t i = 0; i < nx; i++) {
                cap[j * nx + i] = dev(c.x0 + i, c.y0 + j, (int64_t)za - 1);
                cap[nx * ny + j * nx + i] = dev(c.x0 + i, c.y0 + j, (int64_t)za + np);
            }
        U.base = c.base + (int64_t)(za - c.z0 + c.top) * nx * ny;
        U.tab = toff[(size_t)u];
        U.nx = nx; U.ny = ny; U.np = np; U.flags = 0;
        memset(U.ft, 0, sizeof U.ft); memset(U.fb, 0, sizeof U.fb);
        if (c.top && za == c.z0) {
            U.flags |= HQ_BK_TOPFACE;
            memcpy(U.ft, ntab + 7 * (int64_t)L.Ng[(size_t)((((int64_t)za - 1) * NY + c.y0) * NX + c.x0)], sizeof U.ft);
        }
        if (c.bot && za + np == c.z0 + c.nz) {
            U.flags |= HQ_BK_BOTFACE;
            memcpy(U.fb, ntab + 7 * (int64_t)L.Ng[(size_t)((((int64_t)za + np) * NY + c.y0) * NX + c.x0)], sizeof U.fb);
        }
        /* coefficients: those of any element around the first node (all eight are equal: the node is simple) */
        const int32_t n0 = L.Ng[(size_t)(((int64_t)za * NY + c.y0) * NX + c.x0)];
        {
            /* the element whose corner 7 the node is: one cell down on every axis (all eight are equal: the node is simple) */
            const int32_t e0 = L.Eg[(size_t)((((int64_t)za - 1) * L.D[1] + (c.y0 - 1)) * L.D[0] + (c.x0 - 1))];
            if (e0 < 0) { set_fault(); continue; }
            U.c1 = c1[e0]; U.c2 = c2[e0]; U.beta = beta[e0];
        }
        bool sm = true;
        const double* q0 = ntab + 7 * (int64_t)n0;
        for (int32_t z = 0; z < np && sm; z++)
            for (int32_t y = 0; y < ny && sm; y++)
                for (int32_t x = 0; x < nx; x++) {
                    const int32_t n = L.Ng[(size_t)((((int64_t)za + z) * NY + (c.y0 + y)) * NX + (c.x0 + x))];
                    const double* q = ntab + 7 * (int64_t)n;
                    if (q[0] != q0[0] || q[1] != q0[1] || q[4] != q0[4]) { sm = false; break; }
                }
        if (sm && !c.het && !no_ntsame) { U.flags |= HQ_BK_NTSAME; same[(size_t)u] = 1; }
        U.m0 = q0[0]; U.m2 = q0[1]; U.m1 = q0[4];
        U.coef = 0;
        }       /* (!c.rag) */
        if (c.het) {
            U.flags |= HQ_BK_HET;
            same[(size_t)u] = c.rag ? 5 : 2;
            U.coef = coff[(size_t)u];
            double* cf = B->coef.data() + coff[(size_t)u];
            for (int32_t l = 0; l <= np; l++)
                for (int32_t j = 0; j < HQ_BH_WAVES; j++)
                    for (int32_t i = 0; i < 64; i++) {
                        const int64_t cx = (int64_t)c.x0 - 1 + i, cy = (int64_t)c.y0 - 1 + j, cz = (int64_t)za - 1 + l;
                        double* o = cf + (int64_t)l * (3 * HQ_BH_THREADS) + (j * 64 + i);      /* [layer][c1 | c2 | beta][thread] */
                        int32_t e = -1;
                        if (cx >= 0 && cy >= 0 && cz >= 0 && cx < L.D[0] && cy < L.D[1] && cz < L.D[2])
                            e = L.Eg[(size_t)((cz * L.D[1] + cy) * L.D[0] + cx)];
                        if (e >= 0) { o[0] = c1[e]; o[HQ_BH_CS] = c2[e]; o[2 * HQ_BH_CS] = beta[e]; }
                        else if (!c.rag && i <= nx && j <= ny) set_fault();       /* an element around an owned node is missing */
                        /* (a ragged unit: cells without an element of this level keep their zeros -- no owned node touches
                         *  them, a simple node's eight elements exist) */
                    }
            /* the packed form: every element of the block out of (rho, Vs, Vp) bit for bit, every owned node's n_t row out
             * of {m0, m0 - m1} (m1 exactly; m2 = 2 m0 - (m0 - m1) to 1e-15: it is summed in another order, psolve.c:3436-3471) */
            if (try_pack) {
                bool ok = true;
                int32_t e0 = L.Eg[(size_t)(((int64_t)za * L.D[1] + c.y0) * L.D[0] + c.x0)];
                if (c.rag) {                 /* any element of the level will do for the edge length: the first one of the block */
                    e0 = -1;
                    for (int32_t l = 0; l <= np && e0 < 0; l++)
                        for (int32_t j = 0; j < HQ_BH_WAVES && e0 < 0; j++)
                            for (int32_t i = 0; i < 64 && e0 < 0; i++) {
                                const int64_t cx = (int64_t)c.x0 - 1 + i, cy = (int64_t)c.y0 - 1 + j, cz = (int64_t)za - 1 + l;
                                if (cx >= 0 && cy >= 0 && cz >= 0 && cx < L.D[0] && cy < L.D[1] && cz < L.D[2]) e0 = L.Eg[(size_t)((cz * L.D[1] + cy) * L.D[0] + cx)];
                            }
                }
                const float hf = e0 >= 0 ? MS->edata[4 * (int64_t)e0] : 0.0f;
                hq_mat_const K = { (MS->dt * MS->dt) * (double)hf, (double)hf, MS->dt, MS->bbase, MS->thr_damp, MS->thr_vpvs };
                float* cq = B->coef32.data() + coff[(size_t)u];
                for (int32_t l = 0; l <= np && ok; l++)
                    for (int32_t j = 0; j < HQ_BH_WAVES && ok; j++)
                        for (int32_t i = 0; i < 64; i++) {
                            const int64_t cx = (int64_t)c.x0 - 1 + i, cy = (int64_t)c.y0 - 1 + j, cz = (int64_t)za - 1 + l;
                            int32_t e = -1;
                            if (cx >= 0 && cy >= 0 && cz >= 0 && cx < L.D[0] && cy < L.D[1] && cz < L.D[2])
                                e = L.Eg[(size_t)((cz * L.D[1] + cy) * L.D[0] + cx)];
                            float* o = cq + (int64_t)l * (3 * HQ_BH_THREADS) + (j * 64 + i);
                            o[0] = o[HQ_BH_CS] = o[2 * HQ_BH_CS] = 0.0f;
                            if (e < 0) continue;
                            const float* ed = MS->edata + 4 * (int64_t)e;
                            if (ed[0] != hf || !(ed[3] > 0.0f)) { ok = false; break; }
                            bool hit = false;
                            for (int fixed = 0; fixed < 2 && !hit; fixed++) {
                                double k1, k2, kb;
                                hq_material_coef(fixed ? -ed[3] : ed[3], ed[2], ed[1], K, &k1, &k2, &kb);
                                if (k1 == c1[e] && k2 == c2[e] && kb == beta[e]) { o[0] = fixed ? -ed[3] : ed[3]; o[HQ_BH_CS] = ed[2]; o[2 * HQ_BH_CS] = ed[1]; hit = true; }
                            }
                            if (!hit) { ok = false; break; }
                        }
                const int32_t* own_ids = c.rag ? t + (int64_t)(np + 2) * nr + (int64_t)nx * ny : nullptr;      /* plane 1 .. np of a ragged unit */
                for (int32_t z = 0; z < np && ok; z++)
                    for (int32_t y = 0; y < ny && ok; y++)
                        for (int32_t x = 0; x < nx; x++) {
                            if (own_ids && own_ids[((int64_t)z * ny + y) * nx + x] < 0) continue;            /* not this unit's node */
                            const int32_t n = L.Ng[(size_t)((((int64_t)za + z) * NY + (c.y0 + y)) * NX + (c.x0 + x))];
                            const double* q = ntab + 7 * (int64_t)n;
                            const double sdiff = q[0] - q[4];
                            if (q[0] - sdiff != q[4] || fabs((2.0 * q[0] - sdiff) - q[1]) > 1e-15 * fabs(q[1])) { ok = false; break; }
                            double* o2 = B->nt2.data() + 2 * (own_ids ? (int64_t)own_ids[((int64_t)z * ny + y) * nx + x] : U.base + ((int64_t)z * ny + y) * nx + x);
                            o2[0] = q[0]; o2[1] = sdiff;
                        }
                if (ok) { U.flags |= HQ_BK_PACKED; same[(size_t)u] = c.rag ? 6 : 3; U.c1 = K.A; U.c2 = K.h; }
            }
        }
    }
    lap("unit tables");
    if (fault) { g_patch_err = "brick plan: a neighbour of a simple node is missing"; return -1; }
    /* launch order: the units whose nodes share one n_t row (the row rides in the record), then those with per-node
     * rows, then the HET units -- a launch each */
    {
        std::vector<hq_brick_unit> a, r, b, h, hp, rh, rhp;
        for (size_t u = 0; u < us.size(); u++)
            (same[u] == 1 ? a : (same[u] == 4 ? r : (same[u] == 2 ? h : (same[u] == 3 ? hp : (same[u] == 5 ? rh : (same[u] == 6 ? rhp : b)))))).push_back(B->units[u]);
        B->nsame = (int32_t)(a.size() + r.size());
        B->nrag = (int32_t)r.size();
        B->nhet = (int32_t)(h.size() + hp.size());
        B->npacked = (int32_t)hp.size();
        B->nrhet = (int32_t)(rh.size() + rhp.size());
        B->nrpacked = (int32_t)rhp.size();
        a.insert(a.end(), r.begin(), r.end());
        a.insert(a.end(), b.begin(), b.end());
        a.insert(a.end(), h.begin(), h.end());
        a.insert(a.end(), hp.begin(), hp.end());
        a.insert(a.end(), rh.begin(), rh.end());
        a.insert(a.end(), rhp.begin(), rhp.end());
        B->units.swap(a);
        if (B->npacked + B->nrpacked == 0) { std::vector<float>().swap(B->coef32); std::vector<double>().swap(B->nt2); }
        if (verbose) fprintf(stderr, "  brick plan: %d units: %d with one n_t row of which %d ragged, %d per-element coefficients of which %d packed, %d ragged per-element of which %d packed\n",
                             (int)B->units.size(), B->nsame, B->nrag, B->nhet, B->npacked, B->nrhet, B->nrpacked);
        if (verbose && B->nrag) {
            int64_t pos = 0, own = 0, other = 0;
            for (const hq_brick_unit& U : B->units) {
                if (!(U.flags & HQ_BK_RAGGED)) continue;
                const int32_t* pl = B->tab.data() + U.tab + (int64_t)(U.np + 2) * (2 * (U.nx + 2) + 2 * U.ny);
                for (int64_t i = 0; i < (int64_t)U.nx * U.ny * (U.np + 2); i++) { pos++; own += pl[i] >= 0; other += pl[i] <= -2; }
            }
            fprintf(stderr, "  brick plan: ragged units: %lld positions, %lld owned, %lld loaded for their neighbours' sake\n", (long long)pos, (long long)own, (long long)other);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* kernel                                                                   */
/* ------------------------------------------------------------------------ */

/*
 * hq_k_brick: one unit per workgroup (see the head of this file).  PERNODE: the nodes' n_t rows differ (read from
 * the 3-double table with the node); otherwise the row is in the unit's record.
 * Plane k = 0 .. np + 1 of the march is plane za - 1 + k of the column; plane k completes the output of plane k - 1.
 *   fA  accumulator of output plane k - 1: contributions of the planes k - 2 and k - 1 and the node's own term
 *   fB  accumulator of output plane k: contribution of plane k - 1 and the node's own term m2 u1 - m1 u2
 */
/* a value every lane holds alike, moved to scalar registers */
static __device__ __forceinline__ double hq_uniform(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <bool PERNODE, bool BYCOMP, bool RAGGED = false>
/* RAGGED (HQ_BK_RAGGED): every plane's nodes through the unit's id table, output only where the unit owns the node.
 * PERNODE (a caller's nTable whose rows differ inside a homogeneous region: no mesh solver_init builds has one) holds ten
 * more values per lane -- the plane's n_t row and the reciprocal masses of the two unfinished planes; with the stencil
 * numbers in scalar registers it fits 128 VGPRs too (round 3: 33 spilled) */
__global__ void HQ_BK_ATTR __launch_bounds__(HQ_BK_THREADS, 4)   /* 4 waves per SIMD = two workgroups per CU: <= 128 VGPRs */
hq_k_brick(int32_t count, int32_t per_xcd, const hq_brick_unit* __restrict__ units, const int32_t* __restrict__ tab,
           const hq_real* __restrict__ u1g, const hq_real* __restrict__ u2g, hq_real* __restrict__ ung,
           const double* __restrict__ nt3, const int32_t* __restrict__ src_ptr, const int32_t* __restrict__ src_ent,
           const double* __restrict__ F, double dt2, hq_stencil_coef sc)
{
    __shared__ __align__(16) double s_w[2 * 3 * HQ_BK_PLANE];
    const int slot = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (slot >= count) return;
    const hq_brick_unit U = units[slot];
    const int nx = U.nx, ny = U.ny, np = U.np, nxy = nx * ny, nr = 2 * (nx + 2) + 2 * ny;
    const int t = threadIdx.x, lx = t & (HQ_BK_TX - 1), ly = t >> 6;
    const bool active = lx < nx && ly < ny;
    const int sidx = active ? ly * nx + lx : 0;
    const int myrow = (ly + 1) * HQ_BK_PY + lx + 1;
    const bool ring = t < nr;
    int rrow;
    {
        int rx, ry;
        if (t < nx + 2) { rx = t - 1; ry = -1; }
        else if (t < 2 * (nx + 2)) { rx = t - (nx + 2) - 1; ry = ny; }
        else if (t < 2 * (nx + 2) + ny) { rx = -1; ry = t - 2 * (nx + 2); }
        else { rx = nx; ry = t - 2 * (nx + 2) - ny; }
        rrow = (ry + 1) * HQ_BK_PY + rx + 1;
    }
    const int32_t* __restrict__ utab = tab + U.tab;                  /* uniform: scalar registers */
    const int rofs = ring ? t : 0;
    const int32_t* __restrict__ cap = utab + (int64_t)(np + 2) * nr;    /* RAGGED: the plane table [np + 2][ny][nx] */
    /* RAGGED: the table values of the planes k - 1 (the one being completed), k, k + 1 (being loaded) and k + 2 (fetched a
     * plane ahead, as the ring ids are: the node loads must not wait for their ids) */
    int32_t vOut = -1, vMid = RAGGED ? cap[sidx] : 0, vCur = RAGGED ? cap[nxy + sidx] : 0, vNext = 0;
    const int32_t id_lo = RAGGED ? (vMid >= 0 ? vMid : -vMid - 2) : cap[sidx];
    const double beta = U.beta;
    /* the eight stencil numbers are the same for every lane, but fp64 products are vector instructions: without the
     * readfirstlane their results would sit in 16 VGPRs for the whole march.  In SGPRs (a VALU instruction takes one
     * scalar operand, and no fma below has two of them) the kernel allocates <= 112 VGPRs instead of 128: 4 waves of it
     * per SIMD then leave 64 registers per lane to the exchange chain's small kernels, which can become resident on a
     * CU BESIDE two brick workgroups instead of behind them */
    double P[6], Q[2];
#pragma unroll
    for (int i = 0; i < 6; i++) P[i] = hq_uniform(U.c1 * sc.p1[i] + U.c2 * sc.p2[i]);
#pragma unroll
    for (int i = 0; i < 2; i++) Q[i] = hq_uniform(U.c1 * sc.q1[i] + U.c2 * sc.q2[i]);
    const bool has_src = F && src_ptr[slot + 1] > src_ptr[slot];

    /* (the registers the loads land in have the STATE's type: a float state is widened where it is used, at the PUT --
     *  a conversion beside the load would wait for it there and then, and the loads of a plane are meant to stay in flight
     *  while the plane before is worked on: float build 0.95 -> see docs/LABNOTES.md, round 5) */
    hq_real x1[3] = { 0, 0, 0 }, x2[3] = { 0, 0, 0 }, y1[3] = { 0, 0, 0 }, y2[3] = { 0, 0, 0 };
    double mn[3] = { U.m0, U.m2, U.m1 };     /* n_t of the plane being loaded */
    double m0A = hq_uniform(1.0 / U.m0), m0B = m0A;      /* 1 / mass_simple of the output planes k - 1, k */
    double fA[3] = { 0.0, 0.0, 0.0 }, fB[3] = { 0.0, 0.0, 0.0 };
    int32_t rid = utab[rofs];                /* ring id of the plane to load next */

#define HQ_BK_LOAD(node_)                                                                             \
    {                                                                                                 \
        const int64_t a_ = (node_);                                                                   \
        if (active && (!RAGGED || a_ >= 0)) {                                                         \
            _Pragma("unroll") for (int d = 0; d < 3; d++) { x1[d] = u1g[3 * a_ + d]; x2[d] = u2g[3 * a_ + d]; } \
            if (PERNODE) { _Pragma("unroll") for (int d = 0; d < 3; d++) mn[d] = nt3[3 * a_ + d]; }    \
        }                                                                                             \
        if (ring && (!RAGGED || rid >= 0)) {                                                          \
            const int64_t b_ = (int64_t)rid;                                                          \
            _Pragma("unroll") for (int d = 0; d < 3; d++) { y1[d] = u1g[3 * b_ + d]; y2[d] = u2g[3 * b_ + d]; } \
        }                                                                                             \
    }
    /* the loaded plane -> LDS slot s_; its nodes' own term m2 u1 - m1 u2 joins the accumulator acc_ */
    /* (M2_, M1_: mass2_minusaM and mass_minusaM of the plane's nodes, expressions in the axis d -- the unit's row, or a
     *  face plane's 7-double row whose dashpot terms differ per axis) */
#define HQ_BK_PUT(s_, acc_, M2_, M1_)                                                                 \
    {                                                                                                 \
        hq_lds_double* img_ = (hq_lds_double*)s_w + 3 * HQ_BK_PLANE * (s_);                            \
        if (active) {                                                                                 \
            _Pragma("unroll") for (int d = 0; d < 3; d++) {                                           \
                const double a1_ = x1[d], a2_ = x2[d];                                                \
                img_[3 * myrow + d] = a1_ + beta * (a1_ - a2_);                                       \
                acc_[d] += (M2_) * a1_ - (M1_) * a2_;                                                 \
            }                                                                                         \
        }                                                                                             \
        if (ring) { _Pragma("unroll") for (int d = 0; d < 3; d++) { const double b1_ = y1[d], b2_ = y2[d]; img_[3 * rrow + d] = b1_ + beta * (b1_ - b2_); } } \
    }

    /* face planes (HQ_BK_TOPFACE / BOTFACE): uniform per unit */
    /* (only columns whose nodes all share one n_t row take their faces along -- the planner sees to it --, so the PERNODE
     *  form, which no mesh of solver_init reaches, does not carry the face code: it would spill) */
    const bool topf = !PERNODE && !RAGGED && (U.flags & HQ_BK_TOPFACE) != 0, botf = !PERNODE && !RAGGED && (U.flags & HQ_BK_BOTFACE) != 0;
    const double rho = hq_uniform((U.c1 - U.c2) / (U.c1 + U.c2));

    HQ_BK_LOAD((int64_t)id_lo)
    rid = utab[nr + rofs];
    if (topf) HQ_BK_PUT(0, fB, U.ft[1 + d], U.ft[4 + d])        /* the face plane's own term m2 u1 - m1 u2, per axis */
    else { double dummy[3] = { 0.0, 0.0, 0.0 }; HQ_BK_PUT(0, dummy, mn[1], mn[2]) }
    for (int k = 0; k <= np + 1; k++) {
        if (k <= np) {                       /* request plane k + 1 */
            if (RAGGED) {
                HQ_BK_LOAD((int64_t)(vCur >= 0 ? vCur : -vCur - 2))
                if (k < np) vNext = cap[(k + 2) * nxy + sidx];
            } else HQ_BK_LOAD(k == np ? (int64_t)cap[nxy + sidx] : U.base + (int64_t)k * nxy + sidx)
            if (k < np) rid = utab[(k + 2) * nr + rofs];
        }
        __syncthreads();
        const hq_lds_double* __restrict__ q = (const hq_lds_double*)s_w + 3 * (HQ_BK_PLANE * (k & 1) + myrow);
        double m[3], g[3], Uo[3];
        /* BYCOMP: the plane sums component by component behind scheduling barriers -- 9 LDS values live at a time instead
         * of 27: 100 VGPRs instead of 118.  On one GPU that is 1.6 % slower (64M box 1.112 against 1.095 ms, same box);
         * on a PARTITION, where the exchange chain's kernels must find room on CUs two brick workgroups occupy, four waves
         * of 104 registers leave 96 per lane instead of 32 -- three chain workgroups per CU instead of one: the interface
         * update beside the bricks takes 38 instead of 65 us, and a rank's step keeps its 180 us with 20 us of transport
         * latency per exchange where the 118-register kernel goes to 205 (profiles/r04/rank_alone_latency.txt).  The
         * launcher picks it for contexts with a transport. */
        if (BYCOMP) {
            double A_x, A_y;
            {   /* z */
                const double C = q[2], XM = q[2 - 3], XP = q[2 + 3], YM = q[2 - 3 * HQ_BK_PY], YP = q[2 + 3 * HQ_BK_PY];
                const double MM = q[2 - 3 * HQ_BK_PY - 3], PM = q[2 - 3 * HQ_BK_PY + 3], MP = q[2 + 3 * HQ_BK_PY - 3], PP = q[2 + 3 * HQ_BK_PY + 3];
                const double sxy = (XM + XP) + (YM + YP), dg = (MM + PP) + (PM + MP);
                m[2] = fma(P[0], C, fma(P[2], sxy, P[4] * dg));
                g[2] = fma(P[1], C, fma(P[3], sxy, P[5] * dg));
                Uo[0] = fma(Q[0], XP - XM, Q[1] * ((PP - MP) + (PM - MM)));
                Uo[1] = fma(Q[0], YP - YM, Q[1] * ((PP - PM) + (MP - MM)));
            }
            __builtin_amdgcn_sched_barrier(0);
            {   /* x */
                const double C = q[0], XM = q[0 - 3], XP = q[0 + 3], YM = q[0 - 3 * HQ_BK_PY], YP = q[0 + 3 * HQ_BK_PY];
                const double MM = q[0 - 3 * HQ_BK_PY - 3], PM = q[0 - 3 * HQ_BK_PY + 3], MP = q[0 + 3 * HQ_BK_PY - 3], PP = q[0 + 3 * HQ_BK_PY + 3];
                const double sx = XM + XP, sy = YM + YP, dg = (MM + PP) + (PM + MP);
                A_x = (PP + MM) - (PM + MP);
                m[0] = fma(P[0], C, fma(P[1], sx, fma(P[2], sy, P[3] * dg)));
                g[0] = fma(P[2], C, fma(P[3], sx, fma(P[4], sy, P[5] * dg)));
                Uo[2] = fma(Q[0], XP - XM, Q[1] * ((PP - MP) + (PM - MM)));
            }
            __builtin_amdgcn_sched_barrier(0);
            {   /* y */
                const double C = q[1], XM = q[1 - 3], XP = q[1 + 3], YM = q[1 - 3 * HQ_BK_PY], YP = q[1 + 3 * HQ_BK_PY];
                const double MM = q[1 - 3 * HQ_BK_PY - 3], PM = q[1 - 3 * HQ_BK_PY + 3], MP = q[1 + 3 * HQ_BK_PY - 3], PP = q[1 + 3 * HQ_BK_PY + 3];
                const double sx = XM + XP, sy = YM + YP, dg = (MM + PP) + (PM + MP);
                A_y = (PP + MM) - (PM + MP);
                m[1] = fma(P[0], C, fma(P[1], sy, fma(P[2], sx, fma(P[3], dg, Q[0] * A_x))));
                g[1] = fma(P[2], C, fma(P[3], sy, fma(P[4], sx, fma(P[5], dg, Q[1] * A_x))));
                Uo[2] += fma(Q[0], YP - YM, Q[1] * ((PP - PM) + (MP - MM)));
            }
            m[0] = fma(Q[0], A_y, m[0]);
            g[0] = fma(Q[1], A_y, g[0]);
        } else {
            double C[3], XM[3], XP[3], YM[3], YP[3], MM[3], PM[3], MP[3], PP[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                C[d] = q[d]; XM[d] = q[d - 3]; XP[d] = q[d + 3]; YM[d] = q[d - 3 * HQ_BK_PY]; YP[d] = q[d + 3 * HQ_BK_PY];
                MM[d] = q[d - 3 * HQ_BK_PY - 3]; PM[d] = q[d - 3 * HQ_BK_PY + 3];
                MP[d] = q[d + 3 * HQ_BK_PY - 3]; PP[d] = q[d + 3 * HQ_BK_PY + 3];
            }
            /* in-plane sums (first sign: x, second: y) */
            double sx[3], sy[3], dg[3];
#pragma unroll
            for (int d = 0; d < 3; d++) { sx[d] = XM[d] + XP[d]; sy[d] = YM[d] + YP[d]; dg[d] = (MM[d] + PP[d]) + (PM[d] + MP[d]); }
            const double A_x = (PP[0] + MM[0]) - (PM[0] + MP[0]), A_y = (PP[1] + MM[1]) - (PM[1] + MP[1]);
            const double Bx0_z = XP[2] - XM[2], Bx0_x = XP[0] - XM[0], By0_z = YP[2] - YM[2], By0_y = YP[1] - YM[1];
            const double Bx1_z = (PP[2] - MP[2]) + (PM[2] - MM[2]), Bx1_x = (PP[0] - MP[0]) + (PM[0] - MM[0]);
            const double By1_z = (PP[2] - PM[2]) + (MP[2] - MM[2]), By1_y = (PP[1] - PM[1]) + (MP[1] - MM[1]);
            const double sxy_z = sx[2] + sy[2];
            /* the plane seen from its own nodes (dz = 0) */
            m[0] = fma(P[0], C[0], fma(P[1], sx[0], fma(P[2], sy[0], fma(P[3], dg[0], Q[0] * A_y))));
            m[1] = fma(P[0], C[1], fma(P[1], sy[1], fma(P[2], sx[1], fma(P[3], dg[1], Q[0] * A_x))));
            m[2] = fma(P[0], C[2], fma(P[2], sxy_z, P[4] * dg[2]));
            /* seen from the planes below and above (|dz| = 1): the part even in dz ... */
            g[0] = fma(P[2], C[0], fma(P[3], sx[0], fma(P[4], sy[0], fma(P[5], dg[0], Q[1] * A_y))));
            g[1] = fma(P[2], C[1], fma(P[3], sy[1], fma(P[4], sx[1], fma(P[5], dg[1], Q[1] * A_x))));
            g[2] = fma(P[1], C[2], fma(P[3], sxy_z, P[5] * dg[2]));
            /* ... and the part odd in dz (sign: dz as the output node sees it) */
            Uo[0] = fma(Q[0], Bx0_z, Q[1] * Bx1_z);
            Uo[1] = fma(Q[0], By0_z, Q[1] * By1_z);
            Uo[2] = fma(Q[0], Bx0_x, fma(Q[1], Bx1_x, fma(Q[0], By0_y, Q[1] * By1_y)));
        }
        if (k == 0 && topf) {                /* the face plane seen from its own nodes: half the even part, rho x the odd part */
            m[0] = fma(rho, Uo[0], 0.5 * m[0]); m[1] = fma(rho, Uo[1], 0.5 * m[1]); m[2] = fma(-rho, Uo[2], 0.5 * m[2]);
        }
        if (k == np + 1 && botf && active) { /* the bottom face plane: elements on its -z side only (the mirror image) */
            double f[3];
            f[0] = fB[0] + fma(-rho, Uo[0], 0.5 * m[0]); f[1] = fB[1] + fma(-rho, Uo[1], 0.5 * m[1]); f[2] = fB[2] + fma(rho, Uo[2], 0.5 * m[2]);
            const int local = np * nxy + sidx;
            if (has_src) {
                for (int i = src_ptr[slot]; i < src_ptr[slot + 1]; i++)
                    if (src_ent[2 * i] == local) {
                        const int li = src_ent[2 * i + 1];
                        for (int d = 0; d < 3; d++) f[d] += F[3 * li + d] * dt2;
                    }
            }
            hq_real* out = ung + 3 * (int64_t)cap[nxy + sidx];
            const double rm = 1.0 / U.fb[0];
#pragma unroll
            for (int d = 0; d < 3; d++) out[d] = f[d] * rm;
        }
        const bool outface = k == 1 && topf; /* plane 1 completes the TOP face plane (cap plane za - 1) */
        if ((k >= 2 || outface) && active && (!RAGGED || vOut >= 0)) { /* plane k is at dz = +1 of output plane k - 1 = node plane k - 2 of the unit */
            double f[3];
#pragma unroll
            for (int d = 0; d < 3; d++) f[d] = fA[d] + (g[d] + Uo[d]);
            const int local = RAGGED ? (int)(vOut - U.base) : (k - 2) * nxy + sidx;          /* (the top face plane: -nxy + sidx) */
            if (has_src) {                   /* compute_addforce_s, psolve.c:5917-5927 */
                for (int i = src_ptr[slot]; i < src_ptr[slot + 1]; i++)
                    if (src_ent[2 * i] == local) {
                        const int li = src_ent[2 * i + 1];
                        for (int d = 0; d < 3; d++) f[d] += F[3 * li + d] * dt2;
                    }
            }
            hq_real* out = ung + 3 * (RAGGED ? (int64_t)vOut : (outface ? (int64_t)cap[sidx] : U.base + (int64_t)local));
            const double rm = outface ? 1.0 / U.ft[0] : m0A;
#pragma unroll
            for (int d = 0; d < 3; d++) out[d] = f[d] * rm;
        }
#pragma unroll
        for (int d = 0; d < 3; d++) { fA[d] = fB[d] + m[d]; fB[d] = g[d] - Uo[d]; }
        if (RAGGED) { vOut = vMid; vMid = vCur; vCur = vNext; }
        if (PERNODE) { m0A = m0B; m0B = 1.0 / mn[0]; }
        if (k < np || (k == np && !botf)) HQ_BK_PUT((k + 1) & 1, fB, mn[1], mn[2])
        else if (k == np) HQ_BK_PUT((k + 1) & 1, fB, U.fb[1 + d], U.fb[4 + d])
    }
#undef HQ_BK_LOAD
#undef HQ_BK_PUT
}


/*
 * hq_k_brick_het: the same march for units whose elements have coefficients of their own (HQ_BK_HET) -- what
 * solver_init builds on any real CVM mesh (psolve.c:3360-3409): no assembled stencil applies, the force is summed
 * element by element, f_e = -(c1_e K1 + c2_e K2)(u1 + beta_e (u1 - u2)) (hq_element_force: the butterfly form of
 * compute_addforce_effective stiffness.c:180-237 + damping_addforce damping.c:29-103).
 * The tile owns 63 x 7 nodes; its 64 x 8 threads each evaluate ONE element of the layer between the planes l and
 * l + 1 -- element (i, j) has its low corner at node (i - 1, j - 1) -- from u1 and v = u1 - u2 of the two planes in
 * LDS (two slots of two fields, 56 KB), and the eight corner forces are summed to the nodes WITHOUT atomics:
 *   x: thread (i, j) owns node (i, j) = the high-x corners of its own element; the low-x corners of element (i + 1, j)
 *      come from lane i + 1 by DPP (wave_shl:1; a tile row is one wave);
 *   y: the low-y corners of row j + 1 come through a 24 KB exchange buffer;
 *   z: the accumulators of the two unfinished node planes live in registers, as in hq_k_brick.
 * Two barriers per layer; 2 workgroups per CU (80.7 KB of LDS, <= 128 VGPRs).  Per node and step: 72 B of state +
 * 24 B of n_t + 24 B x 512 / 441 of coefficients + the ring.
 */
#define HQ_BH_LDS (8 * (2 * 2 * 3 * HQ_BH_ROWS + 6 * HQ_BH_THREADS))
#ifndef HQ_BH_ABL            /* experiment builds only (-DHQ_BH_ABL=n, results wrong by construction): 1 no ring loads,
                              * 2 no coefficient loads, 3 no element arithmetic, 4 no n_t loads */
#define HQ_BH_ABL 0
#endif

static __device__ __forceinline__ double hq_dpp_from_next_lane(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);      /* wave_shl:1: lane i <- lane i + 1 (lane 63 <- 0: */
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);      /* bound_ctrl, no old value to set up; never an owner) */
    return __hiloint2double(hi, lo);
}

/* PACKED (HQ_BK_PACKED units): `coef` holds three FLOATS per element (rho, Vs, Vp: hq_material_coef expands them to the
 * caller's very doubles, ~50 VALU operations per element and layer), `nt3` two doubles per node {m0, m0 - m1}: 28 bytes
 * per node and step instead of 52 */
/* RAGGED (round 6; HQ_BK_HET | HQ_BK_RAGGED units): the unit owns a SUBSET of its tile's positions -- beside a level
 * interface that cuts the footprint sideways no 63 x 7 tile is full of simple nodes, and a mesh with material of its own in
 * every element (any real CVM) has no one-material columns for hq_k_brick<.., RAGGED> either.  Every plane's positions come
 * out of the id table [np + 2][ny][nx] behind the ring table, exactly as there (v >= 0: a node the unit owns, numbered plane
 * by plane from U.base; v <= -2: node -v - 2 of somebody else, loaded because an owned node has it for a neighbour, never
 * written; -1: nothing an owned node needs -- the lane loads node 0 instead, whose values only reach elements without an
 * owned corner), read one plane ahead like the ring ids.  Elements that do not exist at this level carry zero coefficients
 * (rho = 0 in the packed form).  Same march, same LDS image, same sums; only the stores are conditional. */
static __device__ __forceinline__ int64_t hq_rag_node(int32_t v) { return v >= 0 ? (int64_t)v : (v <= -2 ? (int64_t)(-v - 2) : (int64_t)0); }

template <bool PACKED, bool RAGGED = false>
__global__ void __launch_bounds__(HQ_BH_THREADS, HQ_BH_WAVES == 12 ? 3 : 4)   /* 8 waves: 4 per SIMD = two workgroups per CU, <= 128 VGPRs */
hq_k_brick_het(int32_t count, int32_t per_xcd, const hq_brick_unit* __restrict__ units, const int32_t* __restrict__ tab,
               const void* __restrict__ coef_any, const hq_real* __restrict__ u1g, const hq_real* __restrict__ u2g,
               hq_real* __restrict__ ung, const double* __restrict__ nt3, const int32_t* __restrict__ src_ptr,
               const int32_t* __restrict__ src_ent, const double* __restrict__ F, double dt2, hq_mat_const mat)
{
    extern __shared__ __align__(16) double s_het[];
    hq_lds_double* const img = (hq_lds_double*)s_het;                  /* [slot][u1 | v][3 x rows] */
    hq_lds_double* const xch = img + 2 * 2 * 3 * HQ_BH_ROWS;           /* [6][threads] */
    const int slot = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (slot >= count) return;
    const hq_brick_unit U = units[slot];
    const int nx = U.nx, ny = U.ny, np = U.np, nxy = nx * ny, nr = 2 * (nx + 2) + 2 * ny;
    const int t = threadIdx.x, lx = t & 63, ly = t >> 6;
    const bool owner = lx < nx && ly < ny;
    const int sidx = owner ? ly * nx + lx : 0;
    const int myrow = (ly + 1) * HQ_BH_PY + lx + 1;                    /* node (lx, ly) */
    const int row0 = ly * HQ_BH_PY + lx;                               /* low corner of element (lx, ly): node (lx - 1, ly - 1) */
    /* ring threads: the 78 that never own a node (nx <= 62, ny <= 7); thread q loads the ring slots q and q + 78 */
    const int rq = ly == HQ_BH_WAVES - 1 ? lx : (lx >= 62 ? 64 + 2 * ly + (lx - 62) : -1);
    const bool ringA = rq >= 0 && rq < nr, ringB = rq >= 0 && rq + HQ_BH_NRT < nr;
    auto ring_row = [&](int r) {
        int rx, ry;
        if (r < nx + 2) { rx = r - 1; ry = -1; }
        else if (r < 2 * (nx + 2)) { rx = r - (nx + 2) - 1; ry = ny; }
        else if (r < 2 * (nx + 2) + ny) { rx = -1; ry = r - 2 * (nx + 2); }
        else { rx = nx; ry = r - 2 * (nx + 2) - ny; }
        return (ry + 1) * HQ_BH_PY + rx + 1;
    };
    const int rrowA = ringA ? ring_row(rq) : 0, rrowB = ringB ? ring_row(rq + HQ_BH_NRT) : 0;
    const int32_t* __restrict__ rtabA = tab + U.tab + (ringA ? rq : 0);
    const int32_t* __restrict__ rtabB = tab + U.tab + (ringB ? rq + HQ_BH_NRT : 0);
    const int32_t* __restrict__ cap = tab + U.tab + (int64_t)(np + 2) * nr;      /* RAGGED: [np + 2][nxy], every plane's ids */
    const int64_t id_lo = RAGGED ? 0 : cap[sidx], id_hi = RAGGED ? 0 : cap[nxy + sidx];
    /* RAGGED: vload = the table entry of the plane whose loads are in flight, vnext = of the plane behind it (read one plane
     * ahead), idB / idA = of the planes whose accumulators are accB / accA (>= 0: this unit owns the node and stores it) */
    int32_t vload = RAGGED ? cap[sidx] : 0, vnext = RAGGED ? cap[nxy + sidx] : 0, idA = -1, idB = -1;
    const double* __restrict__ cf = (const double*)coef_any + (PACKED ? 0 : U.coef + t);   /* [layer][c1 | c2 | beta][thread] */
    const float* __restrict__ cf32 = (const float*)coef_any + (PACKED ? U.coef + t : 0);      /* [layer][rho | Vs | Vp][thread] */
    hq_mat_const K = mat;
    if (PACKED) { K.A = hq_uniform(U.c1); K.h = hq_uniform(U.c2); }
    const int third = (PACKED && owner) ? 1 : 2;                       /* PACKED: an owner's third n_t load repeats the second */
    const bool has_src = F && src_ptr[slot + 1] > src_ptr[slot];

    /* an owner's registers: x1, x2 = u1, u2 of its node of the plane in flight, mn = its n_t row, accA / accB = the
     * accumulators of its two unfinished planes.  A ring thread's: x1, x2 = u1, u2 of ring node A, mn / accB = u1 / u2
     * of ring node B (the same registers: a thread is either the one or the other) */
    /* (x1, x2 in the state's type, widened where they are used: see hq_k_brick.  A FLOAT state cannot share mn / accB
     *  -- doubles -- with ring node B: its lanes land that node in rb1, rb2) */
    hq_real x1[3] = { 0, 0, 0 }, x2[3] = { 0, 0, 0 }, rb1[3] = { 0, 0, 0 }, rb2[3] = { 0, 0, 0 };
    constexpr bool SAME = sizeof(hq_real) == sizeof(double);
    double mn[3] = { 1.0, 0.0, 0.0 };
    double accA[3] = { 0.0, 0.0, 0.0 }, accB[3] = { 0.0, 0.0, 0.0 }, m0A = 1.0, m0B = 1.0;    /* m0: 1 / mass_simple */
    int32_t ridA = rtabA[0], ridB = rtabB[0];
    /* RAGGED: a ring position no owned node needs is -1 in the table -- node 0 is loaded in its place (its values reach only
     * elements without an owned corner).  Clamped where the ids are TAKEN (here, and behind the PUT below), never beside
     * their loads */
    if (RAGGED) { ridA = ridA < 0 ? 0 : ridA; ridB = ridB < 0 ? 0 : ridB; }

    /* One load per register for all lanes of a wave, the address chosen per lane -- an owner's node or a ring thread's
     * ring node A into x1, x2; the owner's n_t row or u1 of ring node B into mn -- and no branch around them: two loads
     * into one register (an owner's and a ring lane's) would have the second wait for the first, and loads behind a
     * branch cannot be counted by the compiler, whose waits then drain everything.  Only u2 of ring node B (into accB,
     * which an owner needs for itself) sits behind a branch, as the last of a request. */
#define HQ_BH_LOAD(node_)                                                                             \
    {                                                                                                 \
        const int64_t a_ = owner ? (int64_t)(node_) : (int64_t)ridA;                                  \
        const double* __restrict__ pb_ = (owner || sizeof(hq_real) != sizeof(double)) ? nt3 + (PACKED ? 2 : 3) * a_ : (const double*)(const void*)(u1g + 3 * (int64_t)ridB); \
        _Pragma("unroll") for (int d = 0; d < 3; d++) { x1[d] = u1g[3 * a_ + d]; x2[d] = u2g[3 * a_ + d]; } \
        if (SAME || owner) {                                                                          \
            if (HQ_BH_ABL != 4) { mn[0] = pb_[0]; mn[1] = pb_[1]; mn[2] = pb_[third]; }                \
        } else {                     /* (a float state: the n_t row and ring node B are loads of two types) */ \
            const hq_real* __restrict__ pu_ = u1g + 3 * (int64_t)ridB;                                \
            rb1[0] = pu_[0]; rb1[1] = pu_[1]; rb1[2] = pu_[2];                                         \
        }                                                                                             \
    }
#define HQ_BH_LOAD_B()                                                                                \
    {                                                                                                 \
        if (ringB && HQ_BH_ABL != 1) {                                                                \
            const int64_t b_ = (int64_t)ridB;                                                         \
            _Pragma("unroll") for (int d = 0; d < 3; d++) { if (SAME) accB[d] = u2g[3 * b_ + d]; else rb2[d] = u2g[3 * b_ + d]; } \
        }                                                                                             \
    }
    /* the loaded plane -> slot s_; acc_ / m0_: seed m2 u1 - m1 u2 and 1 / mass_simple of its owned node (one division per
     * node instead of three: the quotient differs from the reference's by <= 1 ulp, as the summation order does) */
#define HQ_BH_PUT(s_, acc_, m0_)                                                                      \
    {                                                                                                 \
        hq_lds_double* iu_ = img + (size_t)(s_) * (2 * 3 * HQ_BH_ROWS);                                \
        hq_lds_double* iv_ = iu_ + 3 * HQ_BH_ROWS;                                                    \
        if (owner) {                                                                                  \
            const double m2_ = PACKED ? 2.0 * mn[0] - mn[1] : mn[1], m1_ = PACKED ? mn[0] - mn[1] : mn[2]; \
            _Pragma("unroll") for (int d = 0; d < 3; d++) {                                           \
                const double a1_ = x1[d], a2_ = x2[d];                                                \
                iu_[3 * myrow + d] = a1_; iv_[3 * myrow + d] = a1_ - a2_;                             \
                acc_[d] = m2_ * a1_ - m1_ * a2_;                                                      \
            }                                                                                         \
            m0_ = 1.0 / mn[0];                                                                        \
            if (RAGGED) idB = vload;                                                                  \
        }                                                                                             \
        if (ringA) { _Pragma("unroll") for (int d = 0; d < 3; d++) { const double a1_ = x1[d], a2_ = x2[d]; iu_[3 * rrowA + d] = a1_; iv_[3 * rrowA + d] = a1_ - a2_; } } \
        if (ringB) { _Pragma("unroll") for (int d = 0; d < 3; d++) { const double b1_ = SAME ? mn[d] : (double)rb1[d], b2_ = SAME ? accB[d] : (double)rb2[d]; iu_[3 * rrowB + d] = b1_; iv_[3 * rrowB + d] = b1_ - b2_; } } \
    }

    /* request plane p_ (1 .. np + 1) of the march, the ring ids of the plane after it and the coefficients of layer p_ - 1 */
    double nc1 = 0.0, nc2 = 0.0, nbeta = 0.0, c1 = 0.0, c2 = 0.0, beta = 0.0;
    float nrho = 0.0f, nvs = 0.0f, nvp = 0.0f;
#define HQ_BH_REQUEST(p_)                                                                             \
    {                                                                                                 \
        const int pp_ = (p_);                                                                         \
        if (HQ_BH_ABL != 2 && !PACKED) { const double* q_ = cf + (int64_t)(pp_ - 1) * (3 * HQ_BH_THREADS); nc1 = q_[0]; nc2 = q_[HQ_BH_CS]; nbeta = q_[2 * HQ_BH_CS]; } \
        if (HQ_BH_ABL != 2 && PACKED) { const float* q_ = cf32 + (int64_t)(pp_ - 1) * (3 * HQ_BH_THREADS); nrho = q_[0]; nvs = q_[HQ_BH_CS]; nvp = q_[2 * HQ_BH_CS]; } \
        if (RAGGED) vload = vnext;                                                                    \
        HQ_BH_LOAD(RAGGED ? hq_rag_node(vload) : (pp_ == np + 1 ? id_hi : U.base + (int64_t)(pp_ - 1) * nxy + sidx)) \
        HQ_BH_LOAD_B()                                                                                \
        { const int64_t r_ = (int64_t)(pp_ < np + 1 ? pp_ + 1 : np + 1) * nr; ridA = rtabA[r_]; ridB = rtabB[r_]; } \
        if (RAGGED) vnext = cap[(int64_t)(pp_ < np + 1 ? pp_ + 1 : np + 1) * nxy + sidx];              \
    }
    if (HQ_BH_ABL == 2) { nc1 = dt2; nc2 = dt2; nbeta = dt2; }
    HQ_BH_LOAD(RAGGED ? hq_rag_node(vload) : id_lo)
    HQ_BH_LOAD_B()
    ridA = rtabA[nr]; ridB = rtabB[nr];
    if (RAGGED) { ridA = ridA < 0 ? 0 : ridA; ridB = ridB < 0 ? 0 : ridB; }
    /* The loop starts two steps early: steps -2 and -1 only put the planes 0 and 1 into LDS and request the planes 1
     * and 2.  ONE request site: a plane and the coefficients of the next layer are requested as soon as the registers
     * are free -- right behind the PUT of the plane before, ahead of the barrier -- and stay in flight through the
     * whole element step that follows (a second site ahead of the loop, with registers of its own, had the compiler wait
     * at the top of every step for loads that only the first step could still have pending). */
    for (int l = -2; l <= np; l++) {
        if (l >= 0) {
            /* the element between the planes l and l + 1 */
            double X[8], Y[8], Z[8];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const hq_lds_double* pu = img + (size_t)((l + (m >> 2)) & 1) * (2 * 3 * HQ_BH_ROWS) + 3 * (row0 + (m & 1) + ((m >> 1) & 1) * HQ_BH_PY);
                const hq_lds_double* pv = pu + 3 * HQ_BH_ROWS;
                X[m] = fma(beta, pv[0], pu[0]); Y[m] = fma(beta, pv[1], pu[1]); Z[m] = fma(beta, pv[2], pu[2]);
            }
            if (HQ_BH_ABL != 3) hq_element_force<true>(X, Y, Z, c1, c2);      /* X[0..3] = a, X[4..7] = b: f(near z) = a - b, f(far z) = a + b */
            /* x: the node's high-x corners are this element's, its low-x corners the next lane's element's */
            double G[2][2][3];                       /* [y bit][a | b] */
#pragma unroll
            for (int yb = 0; yb < 2; yb++)
#pragma unroll
                for (int zb = 0; zb < 2; zb++) {
                    const int m1 = 1 + 2 * yb + 4 * zb, m0 = 2 * yb + 4 * zb;
                    G[yb][zb][0] = X[m1] + hq_dpp_from_next_lane(X[m0]);
                    G[yb][zb][1] = Y[m1] + hq_dpp_from_next_lane(Y[m0]);
                    G[yb][zb][2] = Z[m1] + hq_dpp_from_next_lane(Z[m0]);
                }
            /* y: the low-y corners belong to the node one row down */
#pragma unroll
            for (int zb = 0; zb < 2; zb++)
#pragma unroll
                for (int d = 0; d < 3; d++) xch[(3 * zb + d) * HQ_BH_THREADS + t] = G[0][zb][d];
            __syncthreads();
            if (owner) {
                double H0[3], H1[3];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    H0[d] = G[1][0][d] + xch[d * HQ_BH_THREADS + t + 64];
                    H1[d] = G[1][1][d] + xch[(3 + d) * HQ_BH_THREADS + t + 64];
                }
                if (l >= 1 && (!RAGGED || idA >= 0)) {   /* plane l of the march = plane l - 1 of the unit is complete */
                    double f[3];
#pragma unroll
                    for (int d = 0; d < 3; d++) f[d] = accA[d] + (H0[d] - H1[d]);
                    const int local = RAGGED ? (int)((int64_t)idA - U.base) : (l - 1) * nxy + sidx;
                    if (has_src) {                   /* compute_addforce_s, psolve.c:5917-5927 */
                        for (int i = src_ptr[slot]; i < src_ptr[slot + 1]; i++)
                            if (src_ent[2 * i] == local) {
                                const int li = src_ent[2 * i + 1];
                                for (int d = 0; d < 3; d++) f[d] += F[3 * li + d] * dt2;
                            }
                    }
                    hq_real* out = ung + 3 * (U.base + (int64_t)local);
#pragma unroll
                    for (int d = 0; d < 3; d++) out[d] = f[d] * m0A;
                }
#pragma unroll
                for (int d = 0; d < 3; d++) accA[d] = accB[d] + (H0[d] + H1[d]);
                m0A = m0B;
                if (RAGGED) idA = idB;
            }
        }
        if (l < np) HQ_BH_PUT(l & 1, accB, m0B)
        if (PACKED) hq_material_coef(nrho, nvs, nvp, K, &c1, &c2, &beta);
        else { c1 = nc1; c2 = nc2; beta = nbeta; }
        /* the coefficients and the ring ids are taken HERE, behind the PUT that has waited for the loads of their batch: a
         * wait for them further down would have to drain the loads requested next (the compiler cannot count loads
         * behind branches) */
        asm volatile("" : "+v"(c1), "+v"(c2), "+v"(beta), "+v"(ridA), "+v"(ridB));
        if (RAGGED) { asm volatile("" : "+v"(vnext)); ridA = ridA < 0 ? 0 : ridA; ridB = ridB < 0 ? 0 : ridB; }
        if (l + 1 < np) HQ_BH_REQUEST(l + 3)
        __syncthreads();
    }
#undef HQ_BH_REQUEST
#undef HQ_BH_LOAD
#undef HQ_BH_LOAD_B
#undef HQ_BH_PUT
}

/* ------------------------------------------------------------------------ */
/* device plan                                                              */
/* ------------------------------------------------------------------------ */

static void hq_brick_free(hq_brick_plan* P)
{
    void* ptrs[] = { P->d_units, P->d_tab, P->d_coef, P->d_coef32, P->d_nt2, P->d_src_ptr, P->d_src_ent };
    for (void* p : ptrs) if (p) hipFree(p);
    *P = hq_brick_plan();
}

static int hq_brick_upload(hq_brick_plan* P, const hq_brick_host& B, int64_t* bytes)
{
    P->nb = B.nb;
    P->nunits = (int32_t)B.units.size();
    P->nsame = B.nsame;
    P->nrag = B.nrag;
    P->nhet = B.nhet;
    P->npacked = B.npacked;
    P->nrhet = B.nrhet;
    P->nrpacked = B.nrpacked;
    if (P->nunits == 0) return 0;
    if (P->nhet + P->nrhet > 0) {
        /* (the double block of a packed unit is never read: only the blocks of the unpacked ones travel) */
        if (P->nhet > P->npacked || P->nrhet > P->nrpacked) {
            if (hipMalloc((void**)&P->d_coef, 8 * B.coef.size()) != hipSuccess) { g_patch_err = "hipMalloc failed"; return -2; }
            *bytes += (int64_t)(8 * B.coef.size());
            if (hipMemcpy(P->d_coef, B.coef.data(), 8 * B.coef.size(), hipMemcpyHostToDevice) != hipSuccess) { g_patch_err = "brick coefficient upload failed"; return -3; }
        }
        if (P->npacked + P->nrpacked > 0) {
            if (hipMalloc((void**)&P->d_coef32, 4 * B.coef32.size()) != hipSuccess || hipMalloc((void**)&P->d_nt2, 8 * B.nt2.size()) != hipSuccess) { g_patch_err = "hipMalloc failed"; return -2; }
            *bytes += (int64_t)(4 * B.coef32.size() + 8 * B.nt2.size());
            if (hipMemcpy(P->d_coef32, B.coef32.data(), 4 * B.coef32.size(), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(P->d_nt2, B.nt2.data(), 8 * B.nt2.size(), hipMemcpyHostToDevice) != hipSuccess) { g_patch_err = "brick coefficient upload failed"; return -3; }
        }
        static bool attr_set = false;            /* 80.7 KB of dynamic LDS per workgroup */
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)hq_k_brick_het<false>, hipFuncAttributeMaxDynamicSharedMemorySize, HQ_BH_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)hq_k_brick_het<true>, hipFuncAttributeMaxDynamicSharedMemorySize, HQ_BH_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)hq_k_brick_het<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, HQ_BH_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)hq_k_brick_het<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, HQ_BH_LDS) != hipSuccess) { g_patch_err = "hq_k_brick_het: LDS attribute"; return -3; }
            attr_set = true;
        }
    }
    if (hipMalloc((void**)&P->d_units, sizeof(hq_brick_unit) * B.units.size()) != hipSuccess ||
        hipMalloc((void**)&P->d_tab, 4 * B.tab.size()) != hipSuccess) { g_patch_err = "hipMalloc failed"; return -2; }
    *bytes += (int64_t)(sizeof(hq_brick_unit) * B.units.size() + 4 * B.tab.size());
    if (hipMemcpy(P->d_units, B.units.data(), sizeof(hq_brick_unit) * B.units.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(P->d_tab, B.tab.data(), 4 * B.tab.size(), hipMemcpyHostToDevice) != hipSuccess) { g_patch_err = "brick table upload failed"; return -3; }
    /* owner lookup for hq_brick_set_source */
    std::vector<std::pair<int64_t, int32_t>> by_base;
    for (int32_t u = 0; u < P->nunits; u++) by_base.push_back({ B.units[(size_t)u].base, u });
    std::sort(by_base.begin(), by_base.end());
    for (auto& pr : by_base) {
        const hq_brick_unit& U = B.units[(size_t)pr.second];
        const int64_t nxy = (int64_t)U.nx * U.ny;
        const int top = (U.flags & HQ_BK_TOPFACE) != 0, bot = (U.flags & HQ_BK_BOTFACE) != 0;
        int64_t size = nxy * (U.np + top + bot);
        if (U.flags & HQ_BK_RAGGED) {            /* what it owns: ids [base, base + count), a node's index in the unit = id - base */
            const int32_t* pl = B.tab.data() + U.tab + (int64_t)(U.np + 2) * (2 * (U.nx + 2) + 2 * U.ny);
            size = 0;
            for (int64_t i = nxy; i < nxy * (U.np + 1); i++) size += pl[i] >= 0;
        }
        P->h_base.push_back(pr.first - top * nxy); P->h_slot.push_back(pr.second); P->h_size.push_back(size);
        P->h_first.push_back(pr.first);
    }
    return 0;
}

/* group the loaded nodes (device ids) that are brick nodes by their unit; the others are the patches' */
static int hq_brick_set_source(hq_brick_plan* P, int32_t nloaded, const int32_t* loaded, int64_t* bytes)
{
    if (P->d_src_ptr) { hipFree(P->d_src_ptr); P->d_src_ptr = nullptr; }
    if (P->d_src_ent) { hipFree(P->d_src_ent); P->d_src_ent = nullptr; }
    if (nloaded <= 0 || P->nunits == 0) return 0;
    std::vector<std::array<int32_t, 3>> rec;
    for (int32_t i = 0; i < nloaded; i++) {
        if (loaded[i] >= P->nb) continue;
        const size_t k = (size_t)(std::upper_bound(P->h_base.begin(), P->h_base.end(), (int64_t)loaded[i]) - P->h_base.begin()) - 1;
        rec.push_back({ P->h_slot[k], (int32_t)(loaded[i] - P->h_first[k]), i });
    }
    if (rec.empty()) return 0;
    std::sort(rec.begin(), rec.end());
    std::vector<int32_t> ptr((size_t)P->nunits + 1, 0), ent(rec.size() * 2);
    for (auto& r : rec) ptr[(size_t)r[0] + 1]++;
    for (int32_t u = 0; u < P->nunits; u++) ptr[(size_t)u + 1] += ptr[(size_t)u];
    for (size_t k = 0; k < rec.size(); k++) { ent[2 * k] = rec[k][1]; ent[2 * k + 1] = rec[k][2]; }
    if (hipMalloc((void**)&P->d_src_ptr, 4 * ptr.size()) != hipSuccess) { P->d_src_ptr = nullptr; return -2; }
    if (hipMalloc((void**)&P->d_src_ent, 4 * ent.size()) != hipSuccess) {
        hipFree(P->d_src_ptr); P->d_src_ptr = nullptr; P->d_src_ent = nullptr;
        return -2;
    }
    if (hipMemcpy(P->d_src_ptr, ptr.data(), 4 * ptr.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(P->d_src_ent, ent.data(), 4 * ent.size(), hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(P->d_src_ptr); hipFree(P->d_src_ent); P->d_src_ptr = nullptr; P->d_src_ent = nullptr;
        return -3;
    }
    *bytes += (int64_t)(4 * ptr.size() + 4 * ent.size());
    return 0;
}

/* one step of all units: the HQ_BK_NTSAME units, then (a launch each) the units with per-node n_t rows and the HET units */
static void hq_brick_launch(const hq_brick_plan* P, const hq_real* u1, const hq_real* u2, hq_real* un, const double* nt3,
                            const double* F, double dt2, hipStream_t stream, bool light = false)
{
    const int32_t cnt[7] = { P->nsame - P->nrag, P->nunits - P->nsame - P->nhet - P->nrhet, P->nhet - P->npacked, P->npacked, P->nrag,
                             P->nrhet - P->nrpacked, P->nrpacked };
    int32_t first = 0;
    for (int k4 = 0; k4 < 7; k4++) {
        const int k = k4 == 0 ? 0 : (k4 == 1 ? 4 : (k4 < 5 ? k4 - 1 : k4));       /* launch order = unit order: one row, ragged, per-node rows, HET, packed, ragged HET, ragged packed */
        const int32_t count = cnt[k];
        if (count <= 0) continue;
        const int per_xcd = (count + 7) / 8;
        const int32_t* sp = P->d_src_ptr ? P->d_src_ptr + first : nullptr;
#define HQ_BK_ARGS count, per_xcd, P->d_units + first, P->d_tab, u1, u2, un, nt3, sp, P->d_src_ent, (sp ? F : nullptr), dt2, hq_stencil().c
#ifdef HQ_EXPERIMENT            /* profiles/tools only: unused dynamic LDS so that ONE brick workgroup fits a CU -- does half the residency still stream? */
        static const unsigned xpad = getenv("HQ_X_BRICK_LDS_PAD") ? (unsigned)atoi(getenv("HQ_X_BRICK_LDS_PAD")) : 0u;
        if (xpad && k == 0) {
            static bool once = false;
            if (!once) {
                once = true;
                hipFuncSetAttribute((const void*)hq_k_brick<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xpad);
                hipFuncSetAttribute((const void*)hq_k_brick<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xpad);
            }
            if (light) hq_k_brick<false, true><<<per_xcd * 8, HQ_BK_THREADS, xpad, stream>>>(HQ_BK_ARGS);
            else hq_k_brick<false, false><<<per_xcd * 8, HQ_BK_THREADS, xpad, stream>>>(HQ_BK_ARGS);
        } else
#endif
        if (k == 0 && light) hq_k_brick<false, true><<<per_xcd * 8, HQ_BK_THREADS, 0, stream>>>(HQ_BK_ARGS);
        else if (k == 0) hq_k_brick<false, false><<<per_xcd * 8, HQ_BK_THREADS, 0, stream>>>(HQ_BK_ARGS);
        else if (k == 1 && light) hq_k_brick<true, true><<<per_xcd * 8, HQ_BK_THREADS, 0, stream>>>(HQ_BK_ARGS);
        else if (k == 1) hq_k_brick<true, false><<<per_xcd * 8, HQ_BK_THREADS, 0, stream>>>(HQ_BK_ARGS);
        else if (k == 4 && light) hq_k_brick<false, true, true><<<per_xcd * 8, HQ_BK_THREADS, 0, stream>>>(HQ_BK_ARGS);
        else if (k == 4) hq_k_brick<false, false, true><<<per_xcd * 8, HQ_BK_THREADS, 0, stream>>>(HQ_BK_ARGS);
        else if (k == 2) hq_k_brick_het<false><<<per_xcd * 8, HQ_BH_THREADS, HQ_BH_LDS, stream>>>(count, per_xcd, P->d_units + first, P->d_tab, P->d_coef, u1, u2, un, nt3, sp,
                                                                                               P->d_src_ent, (sp ? F : nullptr), dt2, P->mat);
        else if (k == 3) hq_k_brick_het<true><<<per_xcd * 8, HQ_BH_THREADS, HQ_BH_LDS, stream>>>(count, per_xcd, P->d_units + first, P->d_tab, P->d_coef32, u1, u2, un, P->d_nt2, sp,
                                                                                   P->d_src_ent, (sp ? F : nullptr), dt2, P->mat);
        else if (k == 5) hq_k_brick_het<false, true><<<per_xcd * 8, HQ_BH_THREADS, HQ_BH_LDS, stream>>>(count, per_xcd, P->d_units + first, P->d_tab, P->d_coef, u1, u2, un, nt3, sp,
                                                                                                     P->d_src_ent, (sp ? F : nullptr), dt2, P->mat);
        else hq_k_brick_het<true, true><<<per_xcd * 8, HQ_BH_THREADS, HQ_BH_LDS, stream>>>(count, per_xcd, P->d_units + first, P->d_tab, P->d_coef32, u1, u2, un, P->d_nt2, sp,
                                                                                          P->d_src_ent, (sp ? F : nullptr), dt2, P->mat);
#undef HQ_BK_ARGS
        first += count;
    }
}

#endif /* HQ_BRICK_H */
