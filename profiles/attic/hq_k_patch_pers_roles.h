/* Shelved experiment (round 1): hq_k_patch_pers with wave roles (12 element waves, 4 service
 * waves doing update + all node traffic).  Parity-green; 2.74 ms/step on the 64M box vs 2.65 for
 * the one-role form: four waves issue 26 loads each at ~200 cycles per load (the CU's texture
 * addresser serves them in turn), so the service path (10.6k cycles per patch) is longer than the
 * element path (7.9k).  Not compiled. */
/*
 * hq_k_patch_pers: the patch step as ONE persistent 1024-thread workgroup per CU whose waves have
 * two roles, over two LDS node buffers and two accumulator arrays:
 *
 *   element waves (12, one element per thread): gather patch k from node buffer k&1, K w, LDS
 *       atomics into accumulators k&1; then request the element row of patch k+1;
 *   service waves (4, local nodes st, st+256, ...): request the node data of patch k+1, do the
 *       nodal update of patch k-1 (accumulators and node buffer (k-1)&1), write the node data of
 *       patch k+1 into node buffer (k+1)&1 -- the slots the same thread has just read for the
 *       update; the gather ids of patch k+2 are requested beside the node data.
 *
 * One barrier per patch.  The stamps of the one-role form (profiles/r01/stamps_c3_patch_v5.txt)
 * showed element section, barrier, LDS write, update, barrier in sequence, the VALU 40 % busy;
 * here the update, the stores and all node traffic of the neighbouring patches run beside the
 * element arithmetic on the same SIMDs (3 element waves + 1 service wave each).
 * Patches with a source, hanging nodes or interface nodes (few) get a workgroup-wide step between
 * barrier and update.  Plain loads and __syncthreads: the compiler's vmcnt waits are the right
 * ones (loads are unconditional from clamped addresses, so it can count them).
 */
#define HQ_PERS_THREADS 1024
#define HQ_PERS_ETHREADS 768     /* element waves: threads [0, 768) */
#define HQ_PERS_STHREADS 256     /* service waves: threads [768, 1024) */
#define HQ_PERS_NR 4             /* local nodes per service thread: nlmax <= 1024 */

__global__ void __launch_bounds__(HQ_PERS_THREADS)
hq_k_patch_pers(int32_t count, int32_t per_xcd, const int32_t* __restrict__ order, int32_t nlmax,
                int32_t nfacc, const hq_patch_desc* __restrict__ desc,
                const uint4* __restrict__ pidx, const double* __restrict__ pc1,
                const double* __restrict__ pc2, const double* __restrict__ pbeta,
                const int32_t* __restrict__ halo, const double* __restrict__ u1g,
                const double* __restrict__ u2g, double* __restrict__ ung,
                const double* __restrict__ nt, const double* __restrict__ nt3,
                const int32_t* __restrict__ src_ptr, const int32_t* __restrict__ src_ent,
                const double* __restrict__ F, double dt2, const int32_t* __restrict__ if_ptr,
                const int32_t* __restrict__ if_ent, double* __restrict__ iforce,
                const int32_t* __restrict__ ds_ptr, const int32_t* __restrict__ ds_ent, int32_t hstride)
{
    extern __shared__ __align__(16) double s_mem[];
    double* __restrict__ s_fb = s_mem + 12 * nlmax;     /* two accumulator arrays after the two node buffers */
    const int tid0 = threadIdx.x, T = HQ_PERS_THREADS;
    const int W = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7);
    const int end = min((xcd + 1) * per_xcd, count);
    int slot = xcd * per_xcd + (int)(blockIdx.x >> 3);
    if (slot >= end) return;
#define HQ_SLOT_PATCH(s) ((s) < end ? (order ? order[(s)] : (s)) : -1)
    /* gather ids of service thread st's local nodes of patch (P_, DD): clamped, unconditional */
#define HQ_PERS_ID1(R_, P_, DD) \
    halo[(int64_t)((P_) < 0 ? 0 : (P_)) * hstride + \
         ((st + (R_) * HQ_PERS_STHREADS >= (DD).nown && st + (R_) * HQ_PERS_STHREADS < (DD).nown + (DD).nhalo) \
              ? st + (R_) * HQ_PERS_STHREADS - (DD).nown : 0)]
#define HQ_PERS_IDS(P_, DD)                                                                     \
    {                                                                                           \
        c_raw.x = (uint32_t)HQ_PERS_ID1(0, P_, DD); c_raw.y = (uint32_t)HQ_PERS_ID1(1, P_, DD); \
        c_raw.z = (uint32_t)HQ_PERS_ID1(2, P_, DD); c_raw.w = (uint32_t)HQ_PERS_ID1(3, P_, DD); \
    }
    /* element row of element thread tid of patch DD: clamped, unconditional */
#define HQ_PERS_ROW(DD)                                                                         \
    {                                                                                           \
        const int q_ = tid < (DD).npairs ? tid : 0;                                             \
        const int64_t gc_ = (DD).pair_off + (((DD).flags & HQ_PATCH_UNIFORM) ? 0 : q_);         \
        c_raw = *(reinterpret_cast<const hq_u32x4*>(pidx) + ((DD).pidx_off + q_));              \
        c_beta = pbeta[gc_]; c_c1 = pc1[gc_]; c_c2 = pc2[gc_];                                  \
    }
    /* 3-double n_t of service thread st's owned nodes of patch DD: clamped, unconditional */
#define HQ_PERS_NT3(N3, DD)                                                                     \
    _Pragma("unroll") for (int r_ = 0; r_ < 3; r_++) {                                          \
        const int j_ = st + r_ * HQ_PERS_STHREADS;                                              \
        const double* q_ = nt3 + 3 * ((int64_t)(DD).base + (j_ < (DD).nown ? j_ : 0));          \
        N3[r_][0] = q_[0]; N3[r_][1] = q_[1]; N3[r_][2] = q_[2];                                \
    }

    int pm = -1, p0 = HQ_SLOT_PATCH(slot), p1 = HQ_SLOT_PATCH(slot + W), p2 = HQ_SLOT_PATCH(slot + 2 * W);
    hq_patch_desc Dm = hq_patch_desc_or_empty(desc, -1);           /* patch k-1: its update is pending */
    hq_patch_desc D0 = hq_patch_desc_or_empty(desc, p0);
    hq_patch_desc D1 = hq_patch_desc_or_empty(desc, p1);
    hq_patch_desc D2 = hq_patch_desc_or_empty(desc, p2);
    /* carried from one iteration to the next, 4 dwords both roles share: element threads the
     * local node ids of their element of the CURRENT patch (+ beta, c1, c2), service threads the
     * gather ids of their local nodes of the NEXT patch */
    hq_u32x4 c_raw = { 0, 0, 0, 0 };
    double c_beta = 0.0, c_c1 = 0.0, c_c2 = 0.0;
    {   /* prologue: patch 0 into node buffer 0 */
        const int tid = tid0, st = tid0 - HQ_PERS_ETHREADS;
        for (int i = tid; i < 2 * nfacc; i += T) s_fb[i] = 0.0;
        if (tid < HQ_PERS_ETHREADS) {
            HQ_PERS_ROW(D0)
        } else {
            HQ_PERS_IDS(p1, D1)
        }
        if (tid < D0.nown + D0.nhalo) {
            const int64_t g = tid < D0.nown ? (int64_t)D0.base + tid
                                            : (int64_t)halo[(int64_t)p0 * hstride + (tid - D0.nown)];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                s_mem[3 * tid + d] = u1g[3 * g + d];
                s_mem[3 * nlmax + 3 * tid + d] = u2g[3 * g + d];
            }
        }
        __syncthreads();
    }

    for (int k = 0;; k++) {
        /* keep the per-patch address arithmetic inside the iteration: hipcc otherwise hoists
         * table + f(thread) for every table out of the loop and spills */
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int st = tid - HQ_PERS_ETHREADS;
        double* __restrict__ s_u1 = s_mem + (k & 1) * 6 * nlmax;          /* patch k */
        double* __restrict__ s_u2 = s_u1 + 3 * nlmax;
        double* __restrict__ s_f = s_fb + (k & 1) * nfacc;
        double* __restrict__ o_u1 = s_mem + ((k + 1) & 1) * 6 * nlmax;    /* patch k-1, then patch k+1 */
        double* __restrict__ o_u2 = o_u1 + 3 * nlmax;
        double* __restrict__ o_f = s_fb + ((k + 1) & 1) * nfacc;          /* accumulators of patch k-1 */
        const int p3 = HQ_SLOT_PATCH(slot + 3 * W);
        const hq_patch_desc D3 = hq_patch_desc_or_empty(desc, p3);

        /* patch k-1 with a source, hanging nodes or interface nodes: workgroup-wide, before its update */
        if (pm >= 0) {
            const bool has_src = F && src_ptr[pm + 1] > src_ptr[pm];
            const bool has_ds = ds_ptr && ds_ptr[pm + 1] > ds_ptr[pm];
            const bool has_if = if_ptr && if_ptr[pm + 1] > if_ptr[pm];
            if (has_src || has_ds || has_if || Dm.nacc > Dm.nown) {
                if (has_src) {                           /* compute_addforce_s, psolve.c:5917-5927 */
                    for (int i = src_ptr[pm] + tid; i < src_ptr[pm + 1]; i += T) {
                        int ln = src_ent[2 * i], li = src_ent[2 * i + 1];
                        for (int d = 0; d < 3; d++) atomicAdd(&o_f[3 * ln + d], F[3 * li + d] * dt2);
                    }
                    __syncthreads();
                }
                if (has_ds) {                            /* compute_adjust DISTRIBUTION, psolve.c:5942-5987 */
                    for (int i = ds_ptr[pm] + tid; i < ds_ptr[pm + 1]; i += T) {
                        const int src = ds_ent[3 * i], dst = ds_ent[3 * i + 1];
                        const double deps = (double)(unsigned)ds_ent[3 * i + 2];
                        for (int d = 0; d < 3; d++) atomicAdd(&o_f[3 * dst + d], o_f[3 * src + d] / deps);
                    }
                    __syncthreads();
                }
                if (has_if) {                            /* partial forces to the exchange, psolve.c:4301 */
                    for (int i = if_ptr[pm] + tid; i < if_ptr[pm + 1]; i += T) {
                        int ln = if_ent[2 * i];
                        double* o = iforce + 3 * (int64_t)if_ent[2 * i + 1];
                        o[0] = o_f[3 * ln]; o[1] = o_f[3 * ln + 1]; o[2] = o_f[3 * ln + 2];
                    }
                }
                for (int i = 3 * Dm.nown + tid; i < 3 * Dm.nacc; i += T) o_f[i] = 0.0;
                __syncthreads();
            }
        }

        HQ_STAMPT(0, p0, 0);
        HQ_STAMPT(HQ_PERS_ETHREADS, p0, 4);
        if (tid < HQ_PERS_ETHREADS) {
            /* ---- element waves ---- */
            /* the row requested last iteration is in registers by now (the compiler's wait for it
             * sits here, where this wave has nothing else in flight) */
            asm volatile("" : "+v"(c_raw), "+v"(c_beta), "+v"(c_c1), "+v"(c_c2));
            HQ_STAMPT(0, p0, 1);
            for (int q = tid; q < D0.npairs; q += HQ_PERS_ETHREADS) {
                if (q != tid) {                          /* patches with more than 768 elements: late row */
                    const int64_t gc = D0.pair_off + ((D0.flags & HQ_PATCH_UNIFORM) ? 0 : q);
                    c_raw = *(reinterpret_cast<const hq_u32x4*>(pidx) + (D0.pidx_off + q));
                    c_beta = pbeta[gc]; c_c1 = pc1[gc]; c_c2 = pc2[gc];
                }
                const hq_u32x4 raw = c_raw;
                const double beta = c_beta;
                int l[8];
                l[0] = raw.x & 0xffff; l[1] = raw.x >> 16;
                l[2] = raw.y & 0xffff; l[3] = raw.y >> 16;
                l[4] = raw.z & 0xffff; l[5] = raw.z >> 16;
                l[6] = raw.w & 0xffff; l[7] = raw.w >> 16;
                double X[8], Y[8], Z[8];
#pragma unroll
                for (int n = 0; n < 8; n++) {
                    const double* a = &s_u1[3 * l[n]];
                    const double* b = &s_u2[3 * l[n]];
                    double a0 = a[0], a1_ = a[1], a2_ = a[2];
                    X[n] = a0 + beta * (a0 - b[0]);
                    Y[n] = a1_ + beta * (a1_ - b[1]);
                    Z[n] = a2_ + beta * (a2_ - b[2]);
                }
                hq_element_force(X, Y, Z, c_c1, c_c2);
#pragma unroll
                for (int n = 0; n < 8; n++) {
                    if (l[n] < D0.nacc) {
                        atomicAdd(&s_f[3 * l[n] + 0], X[n]);
                        atomicAdd(&s_f[3 * l[n] + 1], Y[n]);
                        atomicAdd(&s_f[3 * l[n] + 2], Z[n]);
                    }
                }
            }
            HQ_STAMPT(0, p0, 2);
            HQ_PERS_ROW(D1)                              /* flies over the barrier */
        } else {
            /* ---- service waves ---- */
            /* the gather ids requested last iteration are in registers by now */
            asm volatile("" : "+v"(c_raw));
            const int32_t idn[HQ_PERS_NR] = { (int32_t)c_raw.x, (int32_t)c_raw.y, (int32_t)c_raw.z, (int32_t)c_raw.w };
            /* requests: n_t of patch k-1's nodes (3-double form), node data of patch k+1 */
            double n3[3][3];
            HQ_PERS_NT3(n3, Dm)
            double a1[HQ_PERS_NR][3], a2[HQ_PERS_NR][3];
            const int nl1 = D1.nown + D1.nhalo;
#pragma unroll
            for (int r = 0; r < HQ_PERS_NR; r++) {
                const int j = st + r * HQ_PERS_STHREADS;
                const int64_t g = j < D1.nown ? (int64_t)D1.base + j : (j < nl1 ? (int64_t)idn[r] : 0);
#pragma unroll
                for (int d = 0; d < 3; d++) { a1[r][d] = u1g[3 * g + d]; a2[r][d] = u2g[3 * g + d]; }
            }
            /* gather ids of patch k+2: requested now, used next iteration */
            HQ_PERS_IDS(p2, D2)
            HQ_STAMPT(HQ_PERS_ETHREADS, p0, 5);
            /* update of patch k-1 (solver_compute_displacement, psolve.c:4078-4106), accumulators re-zeroed */
            const bool iso = (Dm.flags & HQ_PATCH_ISO) != 0;
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int n = st + r * HQ_PERS_STHREADS;
                if (n < Dm.nown) {
                    double* out = ung + 3 * ((int64_t)Dm.base + n);
                    if (iso) {
#pragma unroll
                        for (int d = 0; d < 3; d++) {
                            double f = o_f[3 * n + d] + (n3[r][1] * o_u1[3 * n + d] - n3[r][2] * o_u2[3 * n + d]);
                            o_f[3 * n + d] = 0.0;
                            out[d] = f / n3[r][0];
                        }
                    } else {
                        const double* q = nt + 7 * ((int64_t)Dm.base + n);
                        double np[7];
#pragma unroll
                        for (int i = 0; i < 7; i++) np[i] = q[i];
#pragma unroll
                        for (int d = 0; d < 3; d++) {
                            double f = o_f[3 * n + d] + (np[1 + d] * o_u1[3 * n + d] - np[4 + d] * o_u2[3 * n + d]);
                            o_f[3 * n + d] = 0.0;
                            out[d] = f / np[0];
                        }
                    }
                }
            }
            HQ_STAMPT(HQ_PERS_ETHREADS, p0, 6);
            /* patch k+1 into the node buffer of patch k-1: every slot is written by the thread
             * that read it for the update above */
#pragma unroll
            for (int r = 0; r < HQ_PERS_NR; r++) {
                const int j = st + r * HQ_PERS_STHREADS;
                if (j < nl1) {
#pragma unroll
                    for (int d = 0; d < 3; d++) { o_u1[3 * j + d] = a1[r][d]; o_u2[3 * j + d] = a2[r][d]; }
                }
            }
            HQ_STAMPT(HQ_PERS_ETHREADS, p0, 7);
        }
        __syncthreads();
        HQ_STAMPT(0, p0, 3);
        if (p0 < 0) break;                               /* that was the update of the last patch */
        slot += W;
        pm = p0; p0 = p1; p1 = p2; p2 = p3;
        Dm = D0; D0 = D1; D1 = D2; D2 = D3;
    }
#undef HQ_SLOT_PATCH
#undef HQ_PERS_ID1
#undef HQ_PERS_IDS
#undef HQ_PERS_ROW
#undef HQ_PERS_NT3
}
