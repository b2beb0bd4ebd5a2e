/*
 * hq_mesher.h -- octree leaves from a gridded material model, as the reference's mesher makes them.
 * Part of libhq_host.so (included by hq_host.c).  C99 + OpenMP, host only.
 *
 * What it restates (mesh_generate psolve.c:1925-2070):
 *   octor_newtree      octor.c:4057-4160   root = cube of 2^30 ticks, far end point from the domain's
 *                                          aspect ratio, ticksize = x / farendp[0]
 *   octor_refinetree   octor.c:4337, oct_expand :1678-1745, oct_sprout :1549-1625
 *                                          a leaf splits if it straddles the domain or toexpand says so;
 *                                          children outside the domain are not created
 *   toexpand / vsrule  psolve.c:2185-2210, quake_util.c:215-225   edgesize > Vs / factor
 *   setrec             psolve.c:1307-1397  record of minimum Vs among 27 samples (0.01, 1, 1.99 half
 *                                          edges per axis, x outermost), early stop at Vs <= vscut, Vs
 *                                          raised to vscut at the same Vp/Vs; the query is
 *                                          cvm_query(east = y, north = x, depth = z) (cvm.c:266-311)
 *   octor_balancetree  octor.c:4398-4775, tree_pushdown :2376-2425
 *                                          2:1 across faces AND edges (dir L..UF, not corners), neighbours
 *                                          outside the domain ignored; new leaves get setrec, not toexpand
 *
 * The reference walks pointers and ripples level by level; here the tree is a stack of dense bitmaps, one per
 * level, bit = "this octant is split".  Refinement fills them top-down; balancing is the closure of the rule
 * "an octant of level l exists => the octants of level l-1 around it (18 directions) exist", which for the
 * parent P of that octant means six octants of level l-2 (the parent's own one excluded, the corner one
 * unreachable): the least fixed point of that rule is the tree prioritized ripple propagation leaves, whatever
 * the order.  Leaves come out in pre-order (= Z-order of their corners) with their setrec record.
 */
#ifndef HQ_MESHER_H
#define HQ_MESHER_H

#define HQM_PIXELLEVEL 30
#define HQM_MAXLEVELS  16          /* finest leaf level the bitmaps go to (the root is level 0) */

typedef struct {
    int        nlev;                          /* bitmaps exist for levels 0 .. nlev-1 */
    int64_t    n[HQM_MAXLEVELS + 1][3];       /* octants per axis that intersect the domain, per level */
    uint64_t*  split[HQM_MAXLEVELS + 1];
    uint32_t   far[3];
    double     ticksize;
    const hqh_grid_model* m;
    double     factor, vscut;
} hqm_tree;

static inline int hqm_get(const hqm_tree* t, int l, int64_t x, int64_t y, int64_t z)
{
    const int64_t i = (z * t->n[l][1] + y) * t->n[l][0] + x;
    return (int)((t->split[l][i >> 6] >> (i & 63)) & 1);
}

/* returns the previous value */
static inline int hqm_set(hqm_tree* t, int l, int64_t x, int64_t y, int64_t z)
{
    const int64_t i = (z * t->n[l][1] + y) * t->n[l][0] + x;
    const uint64_t bit = (uint64_t)1 << (i & 63);
    if (t->split[l][i >> 6] & bit) return 1;
    return (int)((__atomic_fetch_or(&t->split[l][i >> 6], bit, __ATOMIC_RELAXED) & bit) != 0);
}

/* setrec: the record of a leaf of level l at (x, y, z) (level-l units) -> edgesize, Vp, Vs, rho; 0 if no sample hit the model */
static int hqm_setrec(const hqm_tree* t, int l, int64_t x, int64_t y, int64_t z, float rec[4])
{
    static const double pts[3] = { 0.01, 1, 1.99 };
    const hqh_grid_model* m = t->m;
    const uint32_t half = (uint32_t)1 << (HQM_PIXELLEVEL - l - 1);
    const uint32_t lx = (uint32_t)x << (HQM_PIXELLEVEL - l), ly = (uint32_t)y << (HQM_PIXELLEVEL - l), lz = (uint32_t)z << (HQM_PIXELLEVEL - l);
    float bvs = FLT_MAX, bvp = NAN, brho = NAN;
    int hit = 0;
    rec[0] = (float)(t->ticksize * half * 2);
    for (int ix = 0; ix < 3; ix++) {
        const double xm = (lx + pts[ix] * half) * t->ticksize;
        for (int iy = 0; iy < 3; iy++) {
            const double ym = (ly + pts[iy] * half) * t->ticksize;
            for (int iz = 0; iz < 3; iz++) {
                const double zm = (lz + pts[iz] * half) * t->ticksize;
                const int64_t ci = (int64_t)(xm / m->cell), cj = (int64_t)(ym / m->cell), ck = (int64_t)(zm / m->cell);
                if (ci >= m->nx || cj >= m->ny || ck >= m->nz) continue;          /* the query fails: psolve.c:1354-1356 */
                const int64_t q = (ck * m->ny + cj) * m->nx + ci;
                hit = 1;
                if (m->vs[q] < bvs) { bvs = m->vs[q]; bvp = m->vp[q]; brho = m->rho[q]; }
                if (m->vs[q] <= t->vscut) goto done;
            }
        }
    }
done:
    if (!hit) return 0;
    if (bvs <= t->vscut) {                                   /* psolve.c:1389-1394 */
        const double ratio = bvp / bvs;
        bvs = (float)t->vscut;
        bvp = (float)(t->vscut * ratio);
    }
    rec[1] = bvp; rec[2] = bvs; rec[3] = brho;
    return 1;
}

/* does the octant of level l at (x, y, z) reach beyond the domain (oct_expand's isOverlapped)? */
static inline int hqm_overlaps(const hqm_tree* t, int l, int64_t x, int64_t y, int64_t z)
{
    const int sh = HQM_PIXELLEVEL - l;
    return (((uint64_t)(x + 1) << sh) > t->far[0]) || (((uint64_t)(y + 1) << sh) > t->far[1]) || (((uint64_t)(z + 1) << sh) > t->far[2]);
}

static int hqm_alloc_level(hqm_tree* t, int l)
{
    const int sh = HQM_PIXELLEVEL - l;
    for (int d = 0; d < 3; d++) t->n[l][d] = (int64_t)(((uint64_t)t->far[d] + ((uint64_t)1 << sh) - 1) >> sh);
    const int64_t cells = t->n[l][0] * t->n[l][1] * t->n[l][2];
    if (cells > ((int64_t)1 << 36)) return HQ_ERR_ARG;
    t->split[l] = (uint64_t*)calloc((size_t)((cells + 63) >> 6) + 1, sizeof(uint64_t));
    return t->split[l] ? HQ_OK : HQ_ERR_NOMEM;
}

static void hqm_free(hqm_tree* t)
{
    for (int l = 0; l <= HQM_MAXLEVELS; l++) { free(t->split[l]); t->split[l] = NULL; }
}

/* for every set bit of level l (in parallel): fn(t, l, x, y, z, arg) */
typedef void (*hqm_visit)(hqm_tree* t, int l, int64_t x, int64_t y, int64_t z, void* arg);
static void hqm_foreach_split(hqm_tree* t, int l, hqm_visit fn, void* arg)
{
    const int64_t nx = t->n[l][0], ny = t->n[l][1], cells = nx * ny * t->n[l][2], words = (cells + 63) >> 6;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t w = 0; w < words; w++) {
        uint64_t v = __atomic_load_n(&t->split[l][w], __ATOMIC_RELAXED);
        while (v) {
            const int b = __builtin_ctzll(v);
            v &= v - 1;
            const int64_t i = (w << 6) + b;
            fn(t, l, i % nx, (i / nx) % ny, i / (nx * ny), arg);
        }
    }
}

/* refinement: the children of a split octant of level l are set up and asked whether they split */
static void hqm_refine_visit(hqm_tree* t, int l, int64_t x, int64_t y, int64_t z, void* arg)
{
    int* bad = (int*)arg;
    for (int which = 0; which < 8; which++) {
        const int64_t cx = 2 * x + (which & 1), cy = 2 * y + ((which >> 1) & 1), cz = 2 * z + ((which >> 2) & 1);
        if (cx >= t->n[l + 1][0] || cy >= t->n[l + 1][1] || cz >= t->n[l + 1][2]) continue;      /* outside: not created */
        int expand = hqm_overlaps(t, l + 1, cx, cy, cz);
        if (!expand) {
            float rec[4];
            if (!hqm_setrec(t, l + 1, cx, cy, cz, rec)) { __atomic_store_n(bad, 1, __ATOMIC_RELAXED); continue; }
            expand = !((double)rec[0] <= (double)rec[2] / t->factor);                                /* vsrule */
        }
        if (expand) hqm_set(t, l + 1, cx, cy, cz);
    }
}

/* balancing: the octant P = (x, y, z) of level l is split, so its children (level l+1) exist and the octants of level l
 * around them must: per axis the only level-(l-1) octant that is not P's own parent lies towards -1 if the coordinate
 * is even, +1 if odd; all combinations but "none" and "all three" (a corner direction) */
static void hqm_balance_visit(hqm_tree* t, int l, int64_t x, int64_t y, int64_t z, void* arg)
{
    (void)arg;
    if (l < 1) return;
    const int64_t c[3] = { x, y, z };
    int64_t own[3], oth[3];
    int ok[3];
    for (int d = 0; d < 3; d++) {
        own[d] = c[d] >> 1;
        /* the child position that reaches over: 2c - 1 (level l+1) or 2c + 2; inside the domain? (octor.c:4545-4553) */
        const int64_t p = (c[d] & 1) ? 2 * c[d] + 2 : 2 * c[d] - 1;
        ok[d] = p >= 0 && (((uint64_t)p << (HQM_PIXELLEVEL - (l + 1))) < t->far[d]);
        oth[d] = (c[d] & 1) ? own[d] + 1 : own[d] - 1;
    }
    for (int s = 1; s < 7; s++) {
        if (((s & 1) && !ok[0]) || ((s & 2) && !ok[1]) || ((s & 4) && !ok[2])) continue;
        int64_t a[3] = { (s & 1) ? oth[0] : own[0], (s & 2) ? oth[1] : own[1], (s & 4) ? oth[2] : own[2] };
        /* that octant of level l-1 must be split (so that its children, level l, exist) -- and so must its ancestors */
        for (int la = l - 1; la >= 0; la--) {
            if (hqm_set(t, la, a[0], a[1], a[2])) break;
            a[0] >>= 1; a[1] >>= 1; a[2] >>= 1;
        }
    }
}

typedef struct { int8_t level; int32_t x, y, z; int64_t count, first; } hqm_task;

static int64_t hqm_count(const hqm_tree* t, int l, int64_t x, int64_t y, int64_t z)
{
    if (l >= t->nlev || !hqm_get(t, l, x, y, z)) return 1;
    int64_t n = 0;
    for (int which = 0; which < 8; which++) {
        const int64_t cx = 2 * x + (which & 1), cy = 2 * y + ((which >> 1) & 1), cz = 2 * z + ((which >> 2) & 1);
        if (cx >= t->n[l + 1][0] || cy >= t->n[l + 1][1] || cz >= t->n[l + 1][2]) continue;
        n += hqm_count(t, l + 1, cx, cy, cz);
    }
    return n;
}

static int64_t hqm_emit(const hqm_tree* t, int l, int64_t x, int64_t y, int64_t z, int64_t at, uint32_t* ticks, uint32_t* edge, float* edata, int* bad)
{
    if (l >= t->nlev || !hqm_get(t, l, x, y, z)) {
        const int sh = HQM_PIXELLEVEL - l;
        ticks[3 * at] = (uint32_t)x << sh; ticks[3 * at + 1] = (uint32_t)y << sh; ticks[3 * at + 2] = (uint32_t)z << sh;
        edge[at] = (uint32_t)1 << sh;
        if (!hqm_setrec(t, l, x, y, z, &edata[4 * at])) __atomic_store_n(bad, 1, __ATOMIC_RELAXED);
        return at + 1;
    }
    for (int which = 0; which < 8; which++) {
        const int64_t cx = 2 * x + (which & 1), cy = 2 * y + ((which >> 1) & 1), cz = 2 * z + ((which >> 2) & 1);
        if (cx >= t->n[l + 1][0] || cy >= t->n[l + 1][1] || cz >= t->n[l + 1][2]) continue;
        at = hqm_emit(t, l + 1, cx, cy, cz, at, ticks, edge, edata, bad);
    }
    return at;
}

/* the octants of level <= cut in pre-order: those of level cut (split or not) and the leaves above it */
static int hqm_tasks(const hqm_tree* t, int l, int64_t x, int64_t y, int64_t z, int cut, hqm_task** tasks, int64_t* n, int64_t* cap)
{
    if (l == cut || l >= t->nlev || !hqm_get(t, l, x, y, z)) {
        if (*n == *cap) {
            *cap = *cap ? 2 * *cap : 4096;
            hqm_task* q = (hqm_task*)realloc(*tasks, sizeof(hqm_task) * (size_t)*cap);
            if (!q) return HQ_ERR_NOMEM;
            *tasks = q;
        }
        hqm_task k = { (int8_t)l, (int32_t)x, (int32_t)y, (int32_t)z, 0, 0 };
        (*tasks)[(*n)++] = k;
        return HQ_OK;
    }
    for (int which = 0; which < 8; which++) {
        const int64_t cx = 2 * x + (which & 1), cy = 2 * y + ((which >> 1) & 1), cz = 2 * z + ((which >> 2) & 1);
        if (cx >= t->n[l + 1][0] || cy >= t->n[l + 1][1] || cz >= t->n[l + 1][2]) continue;
        int rc = hqm_tasks(t, l + 1, cx, cy, cz, cut, tasks, n, cap);
        if (rc != HQ_OK) return rc;
    }
    return HQ_OK;
}

void hqh_free(void* p) { free(p); }

int hqh_octree_generate(const hqh_grid_model* m, const hqh_mesher_params* p, int64_t* E_out, uint32_t** ticks_out,
                        uint32_t** edge_out, float** edata_out, uint32_t far_ticks[3], double* ticksize_out)
{
    if (!m || !p || !E_out || !ticks_out || !edge_out || !edata_out || !far_ticks) return HQ_ERR_ARG;
    if (m->nx < 1 || m->ny < 1 || m->nz < 1 || m->cell <= 0 || !m->vp || !m->vs || !m->rho || p->factor <= 0) return HQ_ERR_ARG;
    *E_out = 0; *ticks_out = NULL; *edge_out = NULL; *edata_out = NULL;
    hqm_tree t;
    memset(&t, 0, sizeof t);
    t.m = m; t.factor = p->factor; t.vscut = p->vscut;
    {
        /* octor_newtree, octor.c:4122-4146 */
        int32_t u[3] = { (int32_t)p->domain[0], (int32_t)p->domain[1], (int32_t)p->domain[2] };
        if (u[0] < 1 || u[1] < 1 || u[2] < 1) return HQ_ERR_ARG;
        int32_t g = u[0], b = u[1];
        while (b) { int32_t r = g % b; g = b; b = r; }
        b = u[2];
        while (b) { int32_t r = g % b; g = b; b = r; }
        int32_t mx = 0;
        for (int d = 0; d < 3; d++) { u[d] /= g; if (u[d] > mx) mx = u[d]; }
        int pw = 0;
        while ((mx >> (pw + 1)) != 0) pw++;                           /* LOG2_32b: index of the highest set bit */
        for (int d = 0; d < 3; d++) {
            const uint64_t f = (uint64_t)u[d] << (HQM_PIXELLEVEL - pw);
            if (f > ((uint64_t)1 << HQM_PIXELLEVEL)) return HQ_ERR_ARG;  /* the root cube does not hold such a domain */
            t.far[d] = (uint32_t)f;
        }
        t.ticksize = p->domain[0] / t.far[0];
    }
    const int maxlev = p->max_level > 0 && p->max_level < HQM_MAXLEVELS ? p->max_level : HQM_MAXLEVELS;
    int rc = hqm_alloc_level(&t, 0);
    if (rc != HQ_OK) return rc;
    t.split[0][0] = 1;                                                /* toexpand(data == NULL) = 1: the root always splits */
    int bad = 0;
    /* octor_refinetree */
    for (int l = 0;; l++) {
        const int64_t cells = t.n[l][0] * t.n[l][1] * t.n[l][2];
        int any = 0;
        for (int64_t w = 0; w < (cells + 63) >> 6 && !any; w++) any = t.split[l][w] != 0;
        if (!any) { t.nlev = l; break; }
        if (l + 1 > maxlev) { hqm_free(&t); return HQ_ERR_ARG; }     /* the Vs rule asks for leaves below max_level */
        if ((rc = hqm_alloc_level(&t, l + 1)) != HQ_OK) { hqm_free(&t); return rc; }
        hqm_foreach_split(&t, l, hqm_refine_visit, &bad);
        if (bad) { hqm_free(&t); return HQ_ERR_ARG; }                /* a leaf inside the domain that the model does not cover */
    }
    /* octor_balancetree: from the finest level up; bits set at level l-1 are visited when that level's turn comes */
    for (int l = t.nlev - 1; l >= 1; l--) hqm_foreach_split(&t, l, hqm_balance_visit, NULL);
    /* leaves in pre-order */
    hqm_task* tasks = NULL;
    int64_t nt = 0, cap = 0;
    int cut = t.nlev < 5 ? t.nlev : 5;
    rc = hqm_tasks(&t, 0, 0, 0, 0, cut, &tasks, &nt, &cap);
    if (rc != HQ_OK) { free(tasks); hqm_free(&t); return rc; }
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t i = 0; i < nt; i++) tasks[i].count = hqm_count(&t, tasks[i].level, tasks[i].x, tasks[i].y, tasks[i].z);
    int64_t E = 0;
    for (int64_t i = 0; i < nt; i++) { tasks[i].first = E; E += tasks[i].count; }
    uint32_t* ticks = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (size_t)(E ? E : 1));
    uint32_t* edge = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(E ? E : 1));
    float* edata = (float*)malloc(sizeof(float) * 4 * (size_t)(E ? E : 1));
    if (!ticks || !edge || !edata) { free(ticks); free(edge); free(edata); free(tasks); hqm_free(&t); return HQ_ERR_NOMEM; }
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t i = 0; i < nt; i++)
        hqm_emit(&t, tasks[i].level, tasks[i].x, tasks[i].y, tasks[i].z, tasks[i].first, ticks, edge, edata, &bad);
    free(tasks);
    for (int d = 0; d < 3; d++) far_ticks[d] = t.far[d];
    if (ticksize_out) *ticksize_out = t.ticksize;
    hqm_free(&t);
    if (bad) { free(ticks); free(edge); free(edata); return HQ_ERR_ARG; }
    *E_out = E; *ticks_out = ticks; *edge_out = edge; *edata_out = edata;
    return HQ_OK;
}

#endif /* HQ_MESHER_H */
