/* Shelved experiment (round 1): hq_k_patch_step made persistent -- two 512-thread workgroups per CU
 * looping over patches from the XCD's work queue, the next patch's descriptor and first-round halo
 * ids requested an iteration ahead.  Parity-green on all tests; 13.3 G elem/s on the 64M box
 * against 31.5 for hq_k_patch_pers: the loop-carried state pushes the 122-register body over 128
 * (38 VGPRs spilled around the element loop, scratch traffic as large as the LDS traffic) and the
 * ticket atomic's read-back waits for the id loads at the top of every iteration.  Not compiled. */
/*
 * hq_k_patch_pers2 (HQ_PATCH_PIPE=6): hq_k_patch_step made persistent -- two 512-thread workgroups
 * per CU, each looping over patches drawn from the XCD's work queue, with the next patch's
 * descriptor and first-round halo ids requested an iteration ahead.  A workgroup's iteration
 * still waits out one memory latency (its node data), which is what the other workgroup of
 * the CU computes through.
 */
#define HQ_NOSTAMP(k) do { } while (0)
__global__ void __launch_bounds__(512, 4)
hq_k_patch_pers2(int32_t npatches, int32_t per_xcd, const int32_t* __restrict__ order, int32_t nlmax, int32_t nfacc,
                const hq_patch_desc* __restrict__ desc,
                const uint4* __restrict__ pidx, const double* __restrict__ pc1,
                const double* __restrict__ pc2, const double* __restrict__ pbeta,
                const int32_t* __restrict__ halo, const double* __restrict__ u1g,
                const double* __restrict__ u2g, double* __restrict__ ung,
                const double* __restrict__ nt, const double* __restrict__ nt3,
                const int32_t* __restrict__ src_ptr,
                const int32_t* __restrict__ src_ent, const double* __restrict__ F, double dt2,
                const int32_t* __restrict__ if_ptr, const int32_t* __restrict__ if_ent,
                double* __restrict__ iforce, const int32_t* __restrict__ ds_ptr,
                const int32_t* __restrict__ ds_ent, int32_t hstride,
                 int32_t* __restrict__ tickets)
{
    extern __shared__ __align__(16) double s_mem[];
    double* __restrict__ s_u1 = s_mem;
    double* __restrict__ s_u2 = s_mem + 3 * nlmax;
    double* __restrict__ s_f = s_mem + 6 * nlmax;

    constexpr bool NT = false;
    constexpr int DIAG = 0;
    int32_t* __restrict__ s_tick = reinterpret_cast<int32_t*>(s_f + nfacc);    /* [2]: the slots drawn ahead */
    const int tid0 = threadIdx.x, T = 512;
    const int W = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7);
    const int end = min((xcd + 1) * per_xcd, npatches);
#define HQ_SLOT_PATCH(s) ((s) < end ? (order ? order[(s)] : (s)) : -1)
#define HQ_P2_EXIT()                                                                            \
    {                                                                                           \
        if (tid0 == 0 && atomicAdd(&tickets[8 + xcd], 1) == W - 1) {   /* last workgroup of the XCD out */ \
            tickets[xcd] = 0;                                                                   \
            tickets[8 + xcd] = 0;                                                               \
        }                                                                                       \
    }
    if (tid0 == 0) { for (int i = 0; i < 2; i++) s_tick[i] = xcd * per_xcd + atomicAdd(&tickets[xcd], 1); }
    __syncthreads();
    int p = HQ_SLOT_PATCH(__builtin_amdgcn_readfirstlane(s_tick[0]));
    int pn = HQ_SLOT_PATCH(__builtin_amdgcn_readfirstlane(s_tick[1]));
    if (p < 0) {
        HQ_P2_EXIT()
        return;
    }
    /* first-round halo ids: their address needs only the patch number */
    int32_t hid[4];
#pragma unroll
    for (int k = 0; k < 4; k++) hid[k] = halo[(int64_t)p * hstride + (k * T + tid0) / 3];
    hq_patch_desc D = desc[p];
    for (int it = 0;; it++) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));                        /* (address arithmetic stays inside the iteration) */
    const int32_t* __restrict__ hl = halo + (int64_t)p * hstride;
    /* what the NEXT patch needs first -- its descriptor and its first-round halo ids -- is asked
     * for now and used an iteration later; the slot after that is drawn (the value is read at the
     * end of the iteration) */
    const hq_patch_desc Dn = hq_patch_desc_or_empty(desc, pn);
    int32_t hidn[4];
#pragma unroll
    for (int k = 0; k < 4; k++) hidn[k] = halo[(int64_t)(pn < 0 ? 0 : pn) * hstride + (k * T + tid) / 3];
    int32_t drawn = 0;
    if (tid == 0) drawn = atomicAdd(&tickets[xcd], 1);
    const int own3 = D.nown * 3, halo3 = D.nhalo * 3;
    for (int i = own3 + tid; i < 3 * D.nacc; i += T) s_f[i] = 0.0;   /* hanging nodes on owned anchors */
    if (DIAG == 6 && D.nown > 0) HQ_NOSTAMP(1);

    /* HQ_PATCH_WFORM (uniform beta, owned <= nlmax / 2): the LDS image is w = u1 + beta (u1 - u2) of
     * all local nodes | u1 of the owned | u2 of the owned, which halves the gathers of the element loop */
    const bool wf = (D.flags & HQ_PATCH_WFORM) != 0;
    const double wbeta = wf ? pbeta[D.pair_off] : 0.0;
    const int o2off = 3 * (nlmax / 2);
    hq_pair_data cur;
    const int cstep = (D.flags & HQ_PATCH_UNIFORM) ? 0 : 1;     /* uniform patch: every row reads coefficient 0 */
    if (tid < D.npairs) cur = hq_pair_load<NT>(pidx, pc1, pc2, pbeta, D.pidx_off + tid, D.pair_off + cstep * tid);

    {   /* stage: owned nodes are one contiguous run of doubles, halo nodes a gather */
        const double* g1 = u1g + 3 * (int64_t)D.base;
        const double* g2 = u2g + 3 * (int64_t)D.base;
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
    HQ_NOSTAMP(2);
    __syncthreads();
    HQ_NOSTAMP(3);

    for (int q = tid; q < (DIAG == 1 ? 0 : D.npairs); q += T) {
        hq_pair_data nxt;
        if (q + T < D.npairs)
            nxt = hq_pair_load<NT>(pidx, pc1, pc2, pbeta, D.pidx_off + q + T, D.pair_off + cstep * (q + T));
        const uint4 raw = cur.raw;
        const double beta = cur.beta;
        int l[8];
        l[0] = raw.x & 0xffff; l[1] = raw.x >> 16;
        l[2] = raw.y & 0xffff; l[3] = raw.y >> 16;
        l[4] = raw.z & 0xffff; l[5] = raw.z >> 16;
        l[6] = raw.w & 0xffff; l[7] = raw.w >> 16;
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
            if (l[n] < D.nacc) {
                atomicAdd(&s_f[3 * l[n] + 0], X[n]);
                atomicAdd(&s_f[3 * l[n] + 1], Y[n]);
                atomicAdd(&s_f[3 * l[n] + 2], Z[n]);
            }
        }
        cur = nxt;
    }
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

    HQ_NOSTAMP(4);
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
    HQ_NOSTAMP(5);

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
        double* out = ung + 3 * ((int64_t)D.base + n);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double m2 = iso ? np[1] : np[1 + d], m1 = iso ? np[4] : np[4 + d];
            const double x1 = wf ? s_u2[3 * n + d] : s_u1[3 * n + d], x2 = wf ? s_u2[o2off + 3 * n + d] : s_u2[3 * n + d];
            double f = s_f[3 * n + d] + (m2 * x1 - m1 * x2);
            if (NT) __builtin_nontemporal_store(f / np[0], out + d);
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
    if (tid == 0) s_tick[it & 1] = xcd * per_xcd + drawn;          /* the slot of the iteration after the next */
    /* the LDS image and the accumulators are free again when everybody is through the update */
    __syncthreads();
    if (pn < 0) break;
    p = pn; D = Dn;
#pragma unroll
    for (int k = 0; k < 4; k++) hid[k] = hidn[k];
    pn = HQ_SLOT_PATCH(__builtin_amdgcn_readfirstlane(s_tick[it & 1]));
    }
    HQ_P2_EXIT()
#undef HQ_P2_EXIT
#undef HQ_SLOT_PATCH
}

#undef HQ_NOSTAMP

