/* Round-1 experiment, measured and not kept as a product kernel (+1.5 % on the 64M box, opt-in
 * HQ_PATCH_PIPE=5 until round 2 removed it from the shipping translation unit).  Not compiled. */
/*
 * hq_k_patch_roles (HQ_PATCH_PIPE=5): hq_k_patch_pers with two wave roles.  In the one-role
 * kernel every wave requests its node of patch k+1 at the top of iteration k, and the stamps
 * (profiles/r01/stamps_c3_patch_v9.txt) show all 16 waves held ~4k cycles in that burst -- the
 * CU's vector-memory queue takes the requests at the rate the memory system serves them -- with
 * the element section (2.6k) only starting behind it.  Here
 *   waves 0-11  (one element per thread) run the element section of patch k, request the n_t row of
 *               their node and the element row of patch k+1, and later do the nodal update;
 *   waves 12-15 (four local nodes per thread) request the node data of patch k+1 and write its
 *               LDS image into the other buffer -- they are the ones that sit in the queue;
 * both meet at the barrier the one-role kernel has after its atomics.  Same tables, same LDS
 * layout, same work queue.
 */
#define HQ_ROLE_ETHREADS 768
#define HQ_ROLE_LTHREADS 256
#define HQ_ROLE_NR 4             /* local nodes per loader thread: nlmax <= 1024 */

__global__ void __launch_bounds__(HQ_PERS_THREADS)
hq_k_patch_roles(int32_t count, int32_t per_xcd, const int32_t* __restrict__ order, int32_t nlmax,
                 int32_t nfacc, const hq_patch_desc* __restrict__ desc,
                 const uint4* __restrict__ pidx, const double* __restrict__ pc1,
                 const double* __restrict__ pc2, const double* __restrict__ pbeta,
                 const int32_t* __restrict__ halo, const double* __restrict__ u1g,
                 const double* __restrict__ u2g, double* __restrict__ ung,
                 const double* __restrict__ nt, const double* __restrict__ nt3,
                 const int32_t* __restrict__ src_ptr, const int32_t* __restrict__ src_ent,
                 const double* __restrict__ F, double dt2, const int32_t* __restrict__ if_ptr,
                 const int32_t* __restrict__ if_ent, double* __restrict__ iforce,
                 const int32_t* __restrict__ ds_ptr, const int32_t* __restrict__ ds_ent, int32_t hstride,
                 int32_t* __restrict__ tickets)
{
    extern __shared__ __align__(16) double s_mem[];
    double* __restrict__ s_fg = s_mem + 12 * nlmax;
    int32_t* __restrict__ s_tick = reinterpret_cast<int32_t*>(s_fg + nfacc);   /* ring of 8: slots drawn 5 patches ahead */
    const int tid0 = threadIdx.x, T = HQ_PERS_THREADS;
    const int W = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7);
    const int end = min((xcd + 1) * per_xcd, count);
#define HQ_SLOT_PATCH(s) ((s) < end ? (order ? order[(s)] : (s)) : -1)
#define HQ_DRAW() (xcd * per_xcd + atomicAdd(&tickets[xcd], 1))
    /* halo id of loader thread st's r-th local node of patch (P_, DD): clamped, unconditional */
#define HQ_ROLE_ID1(R_, P_, DD)                                                                 \
    halo[(int64_t)((P_) < 0 ? 0 : (P_)) * hstride +                                             \
         ((st + (R_) * HQ_ROLE_LTHREADS >= (DD).nown && st + (R_) * HQ_ROLE_LTHREADS < (DD).nown + (DD).nhalo) \
              ? st + (R_) * HQ_ROLE_LTHREADS - (DD).nown : 0)]
#define HQ_ROLE_IDS(P_, DD)                                                                     \
    {                                                                                           \
        c_raw.x = (uint32_t)HQ_ROLE_ID1(0, P_, DD); c_raw.y = (uint32_t)HQ_ROLE_ID1(1, P_, DD); \
        c_raw.z = (uint32_t)HQ_ROLE_ID1(2, P_, DD); c_raw.w = (uint32_t)HQ_ROLE_ID1(3, P_, DD); \
    }
#define HQ_ROLE_ROW(DD, Q_)                                                                     \
    {                                                                                           \
        const int q_ = (Q_) < (DD).npairs ? (Q_) : 0;                                           \
        const int64_t gc_ = (DD).pair_off + (((DD).flags & HQ_PATCH_UNIFORM) ? 0 : q_);         \
        c_raw = *(reinterpret_cast<const hq_u32x4*>(pidx) + ((DD).pidx_off + q_));              \
        c_beta = pbeta[gc_]; c_c1 = pc1[gc_]; c_c2 = pc2[gc_];                                  \
    }
#define HQ_ROLE_EXIT()                                                                          \
    {                                                                                           \
        if (tid0 == 0 && atomicAdd(&tickets[8 + xcd], 1) == W - 1) {                            \
            tickets[xcd] = 0;                                                                   \
            tickets[8 + xcd] = 0;                                                               \
        }                                                                                       \
    }
    if (tid0 == 0) { for (int i = 0; i < 5; i++) s_tick[i] = HQ_DRAW(); }
    __syncthreads();
    const int sl0 = __builtin_amdgcn_readfirstlane(s_tick[0]), sl1 = __builtin_amdgcn_readfirstlane(s_tick[1]),
              sl2 = __builtin_amdgcn_readfirstlane(s_tick[2]);
    int p0 = HQ_SLOT_PATCH(sl0), p1 = HQ_SLOT_PATCH(sl1), p2 = HQ_SLOT_PATCH(sl2);
    if (p0 < 0) {
        HQ_ROLE_EXIT()
        return;
    }
    hq_patch_desc D0 = hq_patch_desc_or_empty(desc, p0);
    hq_patch_desc D1 = hq_patch_desc_or_empty(desc, p1);
    hq_patch_desc D2 = hq_patch_desc_or_empty(desc, p2);
    /* carried across iterations, 4 dwords both roles share: element threads the packed local node
     * ids of their element of the CURRENT patch (+ beta, c1, c2); loader threads the gather ids of
     * their four local nodes of the NEXT patch */
    hq_u32x4 c_raw = { 0, 0, 0, 0 };
    double c_beta = 0.0, c_c1 = 0.0, c_c2 = 0.0;
    {   /* prologue: patch 0 into buffer 0 (one node per thread), first rows / ids */
        const int tid = tid0, st = tid0 - HQ_ROLE_ETHREADS;
        for (int i = tid; i < nfacc; i += T) s_fg[i] = 0.0;
        const int32_t id0 = (tid >= D0.nown && tid < D0.nown + D0.nhalo) ? halo[(int64_t)p0 * hstride + (tid - D0.nown)] : 0;
        if (tid < HQ_ROLE_ETHREADS) {
            HQ_ROLE_ROW(D0, tid)
        } else {
            HQ_ROLE_IDS(p1, D1)
        }
        if (tid < D0.nown + D0.nhalo) {
            const int64_t g = tid < D0.nown ? (int64_t)D0.base + tid : (int64_t)id0;
            const bool wf = (D0.flags & HQ_PATCH_WFORM) != 0;
            const double b0 = pbeta[D0.pair_off];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const double x1 = u1g[3 * g + d], x2 = u2g[3 * g + d];
                if (wf) {
                    s_mem[3 * tid + d] = x1 + b0 * (x1 - x2);
                    if (tid < D0.nown) { s_mem[3 * nlmax + 3 * tid + d] = x1; s_mem[3 * nlmax + 3 * (nlmax / 2) + 3 * tid + d] = x2; }
                } else {
                    s_mem[3 * tid + d] = x1;
                    s_mem[3 * nlmax + 3 * tid + d] = x2;
                }
            }
        }
        asm volatile("" : "+v"(c_raw), "+v"(c_beta), "+v"(c_c1), "+v"(c_c2));
        __syncthreads();
    }

    for (int k = 0;; k++) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));                    /* (address arithmetic stays inside the iteration) */
        const int st = tid - HQ_ROLE_ETHREADS;
        hq_lds_double* __restrict__ s_u1 = (hq_lds_double*)s_mem + (k & 1) * 6 * nlmax;
        hq_lds_double* __restrict__ s_u2 = s_u1 + 3 * nlmax;
        hq_lds_double* __restrict__ n_u1 = (hq_lds_double*)s_mem + ((k + 1) & 1) * 6 * nlmax;
        hq_lds_double* __restrict__ n_u2 = n_u1 + 3 * nlmax;
        hq_lds_double* __restrict__ s_f = (hq_lds_double*)s_fg;
        const bool wf0 = (D0.flags & HQ_PATCH_WFORM) != 0, wf1 = (D1.flags & HQ_PATCH_WFORM) != 0;
        const bool iso = (D0.flags & HQ_PATCH_ISO) != 0;
        const int slot3 = __builtin_amdgcn_readfirstlane(s_tick[(k + 3) & 7]);
        const int p3 = HQ_SLOT_PATCH(slot3);
        const hq_patch_desc D3 = hq_patch_desc_or_empty(desc, p3);
        int32_t drawn = 0;
        if (tid == 0) drawn = HQ_DRAW();
        double np[7];
#pragma unroll
        for (int i = 0; i < 7; i++) np[i] = 0.0;

        if (tid < HQ_ROLE_ETHREADS) {
            /* ---- element waves: patch k on the current buffer ---- */
            asm volatile("" : "+v"(c_raw), "+v"(c_beta), "+v"(c_c1), "+v"(c_c2));   /* the row requested last iteration */
            for (int q = tid; q < D0.npairs; q += HQ_ROLE_ETHREADS) {
                if (q != tid) HQ_ROLE_ROW(D0, q)          /* patches with more than 768 elements: late row */
                hq_u32x4 rawk = c_raw;
                const double beta = c_beta;
                int l[8];
                double X[8], Y[8], Z[8];
                l[0] = rawk.x & 0xffff; l[1] = rawk.x >> 16;
                l[2] = rawk.y & 0xffff; l[3] = rawk.y >> 16;
                l[4] = rawk.z & 0xffff; l[5] = rawk.z >> 16;
                l[6] = rawk.w & 0xffff; l[7] = rawk.w >> 16;
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
                asm volatile("" : "+v"(rawk));            /* ids again from the packed row: 4 registers across the arithmetic */
                l[0] = rawk.x & 0xffff; l[1] = rawk.x >> 16;
                l[2] = rawk.y & 0xffff; l[3] = rawk.y >> 16;
                l[4] = rawk.z & 0xffff; l[5] = rawk.z >> 16;
                l[6] = rawk.w & 0xffff; l[7] = rawk.w >> 16;
#pragma unroll
                for (int n = 0; n < 8; n++) {
                    if (l[n] < D0.nacc) {
                        hq_lds_double* a = hq_lds_row3(s_f, l[n]);
                        HQ_LDS_ADD(a + 0, X[n]);
                        HQ_LDS_ADD(a + 1, Y[n]);
                        HQ_LDS_ADD(a + 2, Z[n]);
                    }
                }
            }
            /* n_t of this thread's node (psolve.h:210-214; 3-double form where no dashpot acts),
             * then the element row of patch k+1: both fly over the barrier */
            {
                const int64_t nn = (int64_t)D0.base + ((tid < D0.nown && !(D0.flags & HQ_PATCH_NTSAME)) ? tid : 0);
                if (iso) {
                    const double* q = nt3 + 3 * nn;
                    np[0] = q[0]; np[1] = q[1]; np[4] = q[2];
                } else {
                    const double* q = nt + 7 * nn;
#pragma unroll
                    for (int i = 0; i < 7; i++) np[i] = q[i];
                }
            }
            HQ_ROLE_ROW(D1, tid)
            if (F) {                                     /* compute_addforce_s, psolve.c:5917-5927 */
                for (int i = src_ptr[p0] + tid; i < src_ptr[p0 + 1]; i += HQ_ROLE_ETHREADS) {
                    int ln = src_ent[2 * i], li = src_ent[2 * i + 1];
                    for (int d = 0; d < 3; d++) HQ_LDS_ADD(&s_f[3 * ln + d], F[3 * li + d] * dt2);
                }
            }
        } else {
            /* ---- loader waves: patch k+1 into the other buffer ---- */
            asm volatile("" : "+v"(c_raw));              /* the gather ids requested last iteration */
            const int32_t idn[HQ_ROLE_NR] = { (int32_t)c_raw.x, (int32_t)c_raw.y, (int32_t)c_raw.z, (int32_t)c_raw.w };
            const int nl1 = D1.nown + D1.nhalo;
            const double beta1 = pbeta[D1.pair_off];     /* the uniform beta of patch k+1 (used if wf1) */
            double a1[HQ_ROLE_NR][3], a2[HQ_ROLE_NR][3];
#pragma unroll
            for (int r = 0; r < HQ_ROLE_NR; r++) {
                const int j = st + r * HQ_ROLE_LTHREADS;
                const int64_t g = j < D1.nown ? (int64_t)D1.base + j : (j < nl1 ? (int64_t)idn[r] : 0);
#pragma unroll
                for (int d = 0; d < 3; d++) { a1[r][d] = u1g[3 * g + d]; a2[r][d] = u2g[3 * g + d]; }
            }
            HQ_ROLE_IDS(p2, D2)                          /* gather ids of patch k+2: used next iteration */
#pragma unroll
            for (int r = 0; r < HQ_ROLE_NR; r++) {
                const int j = st + r * HQ_ROLE_LTHREADS;
                if (j < nl1) {
                    if (wf1) {
#pragma unroll
                        for (int d = 0; d < 3; d++) n_u1[3 * j + d] = a1[r][d] + beta1 * (a1[r][d] - a2[r][d]);
                        if (j < D1.nown) {
#pragma unroll
                            for (int d = 0; d < 3; d++) { n_u2[3 * j + d] = a1[r][d]; n_u2[3 * (nlmax / 2) + 3 * j + d] = a2[r][d]; }
                        }
                    } else {
#pragma unroll
                        for (int d = 0; d < 3; d++) { n_u1[3 * j + d] = a1[r][d]; n_u2[3 * j + d] = a2[r][d]; }
                    }
                }
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
        if (tid == 0) s_tick[(k + 5) & 7] = drawn;       /* its old content was read at iteration k-6 */
        if (if_ptr && if_ptr[p0 + 1] > if_ptr[p0]) {     /* partial forces to the exchange, psolve.c:4301 */
            for (int i = if_ptr[p0] + tid; i < if_ptr[p0 + 1]; i += T) {
                int ln = if_ent[2 * i];
                double* o = iforce + 3 * (int64_t)if_ent[2 * i + 1];
                o[0] = s_f[3 * ln]; o[1] = s_f[3 * ln + 1]; o[2] = s_f[3 * ln + 2];
            }
            __syncthreads();
        }
        if (tid < D0.nown) {                             /* solver_compute_displacement, psolve.c:4078-4106 */
            const int n = tid;                           /* (owned nodes <= 768: element threads) */
            double* out = ung + 3 * ((int64_t)D0.base + n);
            const hq_lds_double* __restrict__ o_u1 = wf0 ? s_u2 : s_u1;
            const hq_lds_double* __restrict__ o_u2 = wf0 ? s_u2 + 3 * (nlmax / 2) : s_u2;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const double m2 = iso ? np[1] : np[1 + d], m1 = iso ? np[4] : np[4 + d];
                double f = s_f[3 * n + d] + (m2 * o_u1[3 * n + d] - m1 * o_u2[3 * n + d]);
                s_f[3 * n + d] = 0.0;
                out[d] = f / np[0];
            }
        }
        for (int i = 3 * D0.nown + tid; i < 3 * D0.nacc; i += T) s_f[i] = 0.0;
        __syncthreads();
        if (p1 < 0) break;
        p0 = p1; p1 = p2; p2 = p3;
        D0 = D1; D1 = D2; D2 = D3;
    }
    HQ_ROLE_EXIT()
#undef HQ_ROLE_EXIT
#undef HQ_SLOT_PATCH
#undef HQ_DRAW
#undef HQ_ROLE_ID1
#undef HQ_ROLE_IDS
#undef HQ_ROLE_ROW
}

