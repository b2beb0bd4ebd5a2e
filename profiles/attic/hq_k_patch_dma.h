/* Shelved experiment (round 1): LDS-DMA double-buffered persistent patch kernel.
 * Parity-green, 3.09 ms/step on the 64M box vs 2.90 for hq_k_patch_step: the halo gather needs
 * dword-granular global_load_lds (dwordx3 leaves a 4-byte hole per lane), ~92 wave-instructions
 * per patch at ~50 cycles each; see DESIGN.md s7.  Not compiled. */
/*
 * hq_k_patch_dma: the patch step as ONE persistent 1024-thread workgroup per CU with two LDS
 * node buffers.  The stamps of hq_k_patch_step say a workgroup has bulk loads in flight for only
 * ~40 % of its life (descriptor wait, element loop, barriers carry none), so with two workgroups
 * per CU the CU's memory pipe idles a third of the time and the HBM-bound memory phases (1.8 ms)
 * and the VALU/LDS-bound element loop (1.0 ms) add up instead of overlapping.  Here everything
 * patch k+1 needs is copied HBM -> LDS by LDS-DMA (global_load_lds: no registers) while all 16
 * waves run the element loop and update of patch k:
 *   node data   the owned run as dwordx4 pieces, the halo gather dword by dword
 *               (lane = (halo node, dword of its 24-byte record)): both lane-linear in LDS;
 *   n_t         3-double form of the owned nodes, dwordx4 pieces;
 *   halo ids    of patch k+2, dwordx4 (they become the gather addresses one iteration later).
 *
 * vmcnt retires in order and hipcc waits vmcnt(0) wherever a register with a load pending, or an
 * LDS location a DMA may be writing, is touched, so inside the span where the DMAs fly
 *   - the only ordinary loads are the pair rows of patch k+1, requested after the DMAs and
 *     consumed after the drain that ends the iteration (descriptors are scalar loads);
 *   - the LDS atomics and the re-zeroing of the accumulators are inline asm (the accumulators
 *     are disjoint from every DMA target, which the compiler cannot see);
 *   - the barrier between element loop and update is a raw s_barrier behind lgkmcnt(0) only
 *     (__syncthreads would wait vmcnt(0), i.e. for the DMA, cdna_hip_programming.md s5).
 * Patches with sources, hanging nodes, interface nodes or dashpot nodes take ordinary-load
 * paths that drain the DMA early: correct, just not overlapped, and few.
 */
#define HQ_DMA_THREADS 1024
#define HQ_DMA_WAVES 16
#define HQ_DMA_NLMAX 1024        /* owned + halo nodes per patch (node buffers, id lists)        */
#define HQ_DMA_PMAX 768          /* owned nodes (n_t buffers)                                    */
#define HQ_DMA_HROUNDS 3         /* halo gather: 6 dwords x nhalo <= 1024 lanes x 3 rounds       */
#define HQ_DMA_HMAX (HQ_DMA_THREADS * HQ_DMA_HROUNDS / 6)

#define HQ_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define HQ_AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define HQ_AS3(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ hq_patch_desc hq_patch_desc_or_empty(const hq_patch_desc* __restrict__ desc, int p)
{
    hq_patch_desc D = desc[p < 0 ? 0 : p];
    if (p < 0) { D.nown = 0; D.nhalo = 0; D.npairs = 0; D.nacc = 0; D.flags = 0; }
    return D;
}

__device__ __forceinline__ void hq_lds_add_f64x3(uint32_t addr, double x, double y, double z)
{
    asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:8\n\tds_add_f64 %0, %3 offset:16"
                 :: "v"(addr), "v"(x), "v"(y), "v"(z) : "memory");
}

/* LDS doubles: node buffers 2 x (u1[3 nlmax] | u2[3 nlmax]) | f[nfacc] | n_t 2 x [3 PMAX] | ids 2 x [NLMAX ints] */
static inline size_t hq_patch_dma_lds(int nlmax, int nfacc)
{
    return (12 * (size_t)nlmax + (size_t)nfacc + 6 * (size_t)HQ_DMA_PMAX + HQ_DMA_NLMAX) * sizeof(double);
}

__global__ void __launch_bounds__(HQ_DMA_THREADS)
hq_k_patch_dma(int32_t count, int32_t per_xcd, const int32_t* __restrict__ order, int32_t nlmax,
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
    double* __restrict__ s_f = s_mem + 12 * nlmax;
    double* __restrict__ s_ntb = s_f + nfacc;
    int32_t* __restrict__ s_idb = reinterpret_cast<int32_t*>(s_ntb + 6 * HQ_DMA_PMAX);
    const uint32_t f_lds = (uint32_t)(size_t)HQ_AS3(s_f);
    const int tid0 = threadIdx.x, T = HQ_DMA_THREADS;
    const int W = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7);
    const int end = min((xcd + 1) * per_xcd, count);
    int slot = xcd * per_xcd + (int)(blockIdx.x >> 3);
    if (slot >= end) return;

#define HQ_SLOT_PATCH(s) ((s) < end ? (order ? order[(s)] : (s)) : -1)
    /* LDS-DMA of NB bytes (a multiple of 8) that are contiguous in global memory: dwordx4, whose
     * LDS image is lane-linear at 16 bytes per lane; an 8-byte tail as two dwords */
#define HQ_DMA_RUN(SRC, DST, NB)                                                                \
    {                                                                                           \
        const int nch_ = (int)((NB) >> 4);                                                      \
        for (int c_ = wave; c_ * 64 < nch_; c_ += HQ_DMA_WAVES)                                 \
            if (c_ * 64 + lane < nch_)                                                          \
                __builtin_amdgcn_global_load_lds(HQ_AS1(reinterpret_cast<const char*>(SRC) + 16 * (c_ * 64 + lane)), \
                                                 HQ_AS3(reinterpret_cast<char*>(DST) + 1024 * c_), 16, 0, 0); \
        if (((NB) & 8) && wave == 0 && lane < 2)                                                \
            __builtin_amdgcn_global_load_lds(HQ_AS1(reinterpret_cast<const char*>(SRC) + 16 * nch_ + 4 * lane), \
                                             HQ_AS3(reinterpret_cast<char*>(DST) + 16 * nch_), 4, 0, 0); \
    }
    /* LDS-DMA of patch DD's node data into the node buffer at B1 (u1) / B2 (u2) and of its owned
     * nodes' 3-double n_t into NT.  Owned nodes are one contiguous run; halo nodes are gathered
     * dword by dword (lane = (halo node, dword of its 24-byte record): dwordx3 would leave a
     * 4-byte hole after every lane's 12 bytes, profiles/micro/glds_layout.hip), ids in ID[] */
#define HQ_DMA_NODES(ID, DD, B1, B2, NT)                                                        \
    {                                                                                           \
        const int64_t ob_ = 24 * (int64_t)(DD).base;                                            \
        const int nb_ = 24 * (DD).nown;                                                         \
        HQ_DMA_RUN(reinterpret_cast<const char*>(u1g) + ob_, B1, nb_)                           \
        HQ_DMA_RUN(reinterpret_cast<const char*>(u2g) + ob_, B2, nb_)                           \
        if ((DD).flags & HQ_PATCH_ISO) HQ_DMA_RUN(reinterpret_cast<const char*>(nt3) + ob_, NT, nb_) \
        _Pragma("unroll") for (int r_ = 0; r_ < HQ_DMA_HROUNDS; r_++) {                         \
            const int c_ = wave + HQ_DMA_WAVES * r_;                                            \
            const int g_ = c_ * 64 + lane;                                                      \
            if (g_ < 6 * (DD).nhalo) {                                                          \
                const int64_t o_ = 24 * (int64_t)ID[r_] + 4 * (g_ - 6 * (g_ / 6));              \
                __builtin_amdgcn_global_load_lds(HQ_AS1(reinterpret_cast<const char*>(u1g) + o_), \
                                                 HQ_AS3(reinterpret_cast<char*>(B1) + nb_ + 256 * c_), 4, 0, 0); \
                __builtin_amdgcn_global_load_lds(HQ_AS1(reinterpret_cast<const char*>(u2g) + o_), \
                                                 HQ_AS3(reinterpret_cast<char*>(B2) + nb_ + 256 * c_), 4, 0, 0); \
            }                                                                                   \
        }                                                                                       \
    }
    /* LDS-DMA of patch P_'s halo id list (NH entries) into IB */
#define HQ_DMA_IDLIST(P_, NH, IB)                                                               \
    for (int c_ = wave; c_ * 256 < (NH); c_ += HQ_DMA_WAVES)                                    \
        if (c_ * 256 + lane * 4 < (NH))                                                         \
            __builtin_amdgcn_global_load_lds(HQ_AS1(halo + (int64_t)(P_) * hstride + c_ * 256 + lane * 4), \
                                             HQ_AS3(reinterpret_cast<char*>(IB) + 1024 * c_), 16, 0, 0);
    /* this lane's gather ids for patch DD out of the LDS list IB */
#define HQ_IDS_FROM_LDS(ID, DD, IB)                                                             \
    _Pragma("unroll") for (int r_ = 0; r_ < HQ_DMA_HROUNDS; r_++) {                             \
        const int h_ = ((wave + HQ_DMA_WAVES * r_) * 64 + lane) / 6;                            \
        ID[r_] = h_ < (DD).nhalo ? (IB)[h_] : 0;                                                \
    }

    int p0 = HQ_SLOT_PATCH(slot), p1 = HQ_SLOT_PATCH(slot + W), p2 = HQ_SLOT_PATCH(slot + 2 * W);
    hq_patch_desc D0 = hq_patch_desc_or_empty(desc, p0);
    hq_patch_desc D1 = hq_patch_desc_or_empty(desc, p1);
    hq_patch_desc D2 = hq_patch_desc_or_empty(desc, p2);
    hq_pair_data cur;
    {   /* prologue: patch 0 into buffers 0, the id list of patch 1 into id buffer 1 */
        const int tid = tid0, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        for (int i = tid; i < nfacc; i += T) s_f[i] = 0.0;
        int32_t id0[HQ_DMA_HROUNDS];
#pragma unroll
        for (int r = 0; r < HQ_DMA_HROUNDS; r++) {
            const int h = ((wave + HQ_DMA_WAVES * r) * 64 + lane) / 6;
            id0[r] = h < D0.nhalo ? halo[(int64_t)p0 * hstride + h] : 0;
        }
        if (tid < D0.npairs) cur = hq_pair_load<false>(pidx, pc1, pc2, pbeta, D0.pair_off + tid);
        HQ_DMA_NODES(id0, D0, s_mem, s_mem + 3 * nlmax, s_ntb)
        if (p1 >= 0) { HQ_DMA_IDLIST(p1, D1.nhalo, s_idb + HQ_DMA_NLMAX) }
        __syncthreads();
    }

    for (int k = 0;; k++) {
        /* keep the per-patch address arithmetic inside the iteration: hipcc otherwise hoists
         * table + f(thread) for every table out of the loop and spills */
        HQ_STAMPD(0);
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        double* __restrict__ s_u1 = s_mem + (k & 1) * 6 * nlmax;
        double* __restrict__ s_u2 = s_u1 + 3 * nlmax;
        double* __restrict__ s_nt = s_ntb + (k & 1) * 3 * HQ_DMA_PMAX;
        double* __restrict__ n_u1 = s_mem + ((k + 1) & 1) * 6 * nlmax;
        double* __restrict__ n_nt = s_ntb + ((k + 1) & 1) * 3 * HQ_DMA_PMAX;

        /* 1. DMA: patch k+1's nodes and n_t (gather ids from the list that landed last
         *    iteration), patch k+2's id list */
        int32_t idn[HQ_DMA_HROUNDS];
        HQ_IDS_FROM_LDS(idn, D1, s_idb + ((k + 1) & 1) * HQ_DMA_NLMAX)
        HQ_DMA_NODES(idn, D1, n_u1, n_u1 + 3 * nlmax, n_nt)
        if (p2 >= 0) { HQ_DMA_IDLIST(p2, D2.nhalo, s_idb + (k & 1) * HQ_DMA_NLMAX) }
        __builtin_amdgcn_sched_barrier(0);
        /* 2. the only ordinary loads of the span: pair rows of patch k+1 (consumed after the
         *    drain), and the descriptor of patch k+3 (scalar) */
        hq_pair_data nxt;
        if (tid < D1.npairs) nxt = hq_pair_load<false>(pidx, pc1, pc2, pbeta, D1.pair_off + tid);
        const int p3 = HQ_SLOT_PATCH(slot + 3 * W);
        const hq_patch_desc D3 = hq_patch_desc_or_empty(desc, p3);
        __builtin_amdgcn_sched_barrier(0);

        HQ_STAMPD(1);
        /* 3. element loop of patch k on the current buffer: one element per thread (the planner
         *    keeps patches at <= 1024 elements) */
        if (tid < D0.npairs) {
            const uint4 raw = cur.raw;
            const double beta = cur.beta;
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
                double a0 = a[0], a1 = a[1], a2 = a[2];
                X[n] = a0 + beta * (a0 - b[0]);
                Y[n] = a1 + beta * (a1 - b[1]);
                Z[n] = a2 + beta * (a2 - b[2]);
            }
            hq_element_force(X, Y, Z, cur.c1, cur.c2);
#pragma unroll
            for (int n = 0; n < 8; n++)
                if (l[n] < D0.nacc) hq_lds_add_f64x3(f_lds + 24 * l[n], X[n], Y[n], Z[n]);
        }
        HQ_STAMPD(2);
        if (F) {                                         /* compute_addforce_s, psolve.c:5917-5927 */
            for (int i = src_ptr[p0] + tid; i < src_ptr[p0 + 1]; i += T) {
                int ln = src_ent[2 * i], li = src_ent[2 * i + 1];
                for (int d = 0; d < 3; d++) atomicAdd(&s_f[3 * ln + d], F[3 * li + d] * dt2);
            }
        }
        if (ds_ptr && ds_ptr[p0 + 1] > ds_ptr[p0]) {     /* compute_adjust DISTRIBUTION, psolve.c:5942-5987 */
            HQ_LDS_BARRIER();
            for (int i = ds_ptr[p0] + tid; i < ds_ptr[p0 + 1]; i += T) {
                const int src = ds_ent[3 * i], dst = ds_ent[3 * i + 1];
                const double deps = (double)(unsigned)ds_ent[3 * i + 2];
                for (int d = 0; d < 3; d++) atomicAdd(&s_f[3 * dst + d], s_f[3 * src + d] / deps);
            }
        }
        HQ_LDS_BARRIER();
        HQ_STAMPD(3);

        /* 4. interface partial forces (psolve.c:4301), then update + re-zero the accumulators */
        if (if_ptr && if_ptr[p0 + 1] > if_ptr[p0]) {
            for (int i = if_ptr[p0] + tid; i < if_ptr[p0 + 1]; i += T) {
                int ln = if_ent[2 * i];
                double* o = iforce + 3 * (int64_t)if_ent[2 * i + 1];
                o[0] = s_f[3 * ln]; o[1] = s_f[3 * ln + 1]; o[2] = s_f[3 * ln + 2];
            }
            HQ_LDS_BARRIER();
        }
        if (tid < D0.nown) {                             /* solver_compute_displacement, psolve.c:4078-4106 */
            const int n = tid;
            double np[7];
            if (D0.flags & HQ_PATCH_ISO) {
                np[0] = s_nt[3 * n]; np[1] = s_nt[3 * n + 1]; np[4] = s_nt[3 * n + 2];
                np[2] = np[3] = np[1];
                np[5] = np[6] = np[4];
            } else {
                const double* q = nt + 7 * ((int64_t)D0.base + n);
#pragma unroll
                for (int i = 0; i < 7; i++) np[i] = q[i];
            }
            double* out = ung + 3 * ((int64_t)D0.base + n);
            double f[3];
#pragma unroll
            for (int d = 0; d < 3; d++)
                f[d] = s_f[3 * n + d] + (np[1 + d] * s_u1[3 * n + d] - np[4 + d] * s_u2[3 * n + d]);
            asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:8\n\tds_write_b64 %0, %1 offset:16"
                         :: "v"(f_lds + 24 * n), "v"(0.0) : "memory");
#pragma unroll
            for (int d = 0; d < 3; d++) out[d] = f[d] / np[0];
        }
        for (int i = 3 * D0.nown + tid; i < 3 * D0.nacc; i += T) s_f[i] = 0.0;
        HQ_STAMPD(4);
        /* 5. drain: the DMAs have landed, every wave is done with buffer k and the accumulators */
        __syncthreads();
        HQ_STAMPD(5);
        HQ_STAMPD(6);
        if (p1 < 0) break;
        slot += W;
        p0 = p1; p1 = p2; p2 = p3;
        D0 = D1; D1 = D2; D2 = D3;
        cur = nxt;
    }
#undef HQ_SLOT_PATCH
#undef HQ_DMA_RUN
#undef HQ_DMA_NODES
#undef HQ_DMA_IDLIST
#undef HQ_IDS_FROM_LDS
}

