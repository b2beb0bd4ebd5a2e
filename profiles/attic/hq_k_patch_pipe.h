/* Shelved experiment (round 1): 256-thread persistent register-pipelined patch kernel.
 * Parity-green, 12 % slower than hq_k_patch_step; see DESIGN.md s7.  Not compiled. */
/*
 * Persistent, software-pipelined form of hq_k_patch_step.  A workgroup walks the patches
 * slot, slot + W, slot + 2W, ... of its XCD's run; while it computes patch j the loads that
 * stage patch j+1 (owned run + halo gather, KO + KH doubles per array per thread) are
 * already in flight into registers, and the halo ids / descriptor of patch j+2 behind them,
 * so the memory system stays busy during the element loop (the non-pipelined kernel moves
 * no bytes while both resident workgroups compute).  T = 256 threads: 2 waves per SIMD,
 * 256 registers per lane to hold the in-flight patch.
 *
 * EXPERIMENT (opt-in, HQ_PATCH_PIPE=1), parity-green but 12 % slower than hq_k_patch_step on
 * the 64M box (3.49 vs 3.11 ms): the in-flight patch costs 256 registers per lane = 8 waves
 * per CU, and at 2 waves per SIMD the element loop loses more than the overlap gains.
 * AHEAD = false (only descriptor + halo ids one patch ahead, 512 threads) spills 79 VGPRs at
 * the 128-register budget and is not instantiated.  Kept as the starting point for a
 * loader-wave design (2 loader + 6 compute waves per 512-thread workgroup).
 */
template <int T, int KO, int KH, bool AHEAD, int WPS, int NR>
__global__ void __launch_bounds__(T, WPS)
hq_k_patch_pipe(int32_t npatches, int32_t per_xcd, const int32_t* __restrict__ order, int32_t nlmax,
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
                const int32_t* __restrict__ ds_ent)
{
    extern __shared__ __align__(16) double s_mem[];
    double* __restrict__ s_u1 = s_mem;
    double* __restrict__ s_u2 = s_mem + 3 * nlmax;
    double* __restrict__ s_f = s_mem + 6 * nlmax;

    const int tid = threadIdx.x;
    const int W = (int)(gridDim.x >> 3);                       /* workgroups per XCD */
    const int xcd = (int)(blockIdx.x & 7);
    const int end = min((xcd + 1) * per_xcd, npatches);
    int slot = xcd * per_xcd + (int)(blockIdx.x >> 3);
    if (slot >= end) return;

    int p = order ? order[slot] : slot;
    hq_patch_desc D = desc[p];
    double a1[KO], a2[KO], b1[KH], b2[KH];
    int hid[KH];
    hq_pair_data cur;

#define HQ_PIPE_IDS(DD)                                                                   \
    {                                                                                     \
        const int32_t* hl_ = halo + (DD).halo_off;                                        \
        _Pragma("unroll") for (int k = 0; k < KH; k++) {                                  \
            int i_ = k * T + tid;                                                         \
            hid[k] = (i_ < 3 * (DD).nhalo) ? hl_[i_ / 3] : 0;                             \
        }                                                                                 \
    }
#define HQ_PIPE_ISSUE(DD)                                                                 \
    {                                                                                     \
        const double* g1_ = u1g + 3 * (int64_t)(DD).base;                                 \
        const double* g2_ = u2g + 3 * (int64_t)(DD).base;                                 \
        _Pragma("unroll") for (int k = 0; k < KO; k++) {                                  \
            int i_ = k * T + tid;                                                         \
            if (i_ < 3 * (DD).nown) { a1[k] = g1_[i_]; a2[k] = g2_[i_]; }                 \
        }                                                                                 \
        _Pragma("unroll") for (int k = 0; k < KH; k++) {                                  \
            int i_ = k * T + tid;                                                         \
            if (i_ < 3 * (DD).nhalo) {                                                    \
                int64_t g_ = 3 * (int64_t)hid[k] + (i_ - 3 * (i_ / 3));                   \
                b1[k] = u1g[g_]; b2[k] = u2g[g_];                                         \
            }                                                                             \
        }                                                                                 \
        if (tid < (DD).npairs) cur = hq_pair_load<false>(pidx, pc1, pc2, pbeta, (DD).pair_off + tid); \
    }

    HQ_PIPE_IDS(D)
    if (AHEAD) HQ_PIPE_ISSUE(D)
    int slot_n = slot + W;
    bool has_n = slot_n < end;
    int pn = 0;
    hq_patch_desc Dn = D;
    if (has_n) {
        pn = order ? order[slot_n] : slot_n;
        Dn = desc[pn];
        if (AHEAD) HQ_PIPE_IDS(Dn)
    }

    for (;;) {
        const int own3 = D.nown * 3, halo3 = D.nhalo * 3;
        if (!AHEAD) {
            /* the ids of this patch's halo arrived during the previous patch: one latency
             * (the data itself) instead of three (descriptor -> ids -> data) */
            HQ_PIPE_ISSUE(D)
            if (has_n) HQ_PIPE_IDS(Dn)
        }
        /* registers -> LDS */
#pragma unroll
        for (int k = 0; k < KO; k++) {
            int i = k * T + tid;
            if (i < own3) { s_u1[i] = a1[k]; s_u2[i] = a2[k]; }
        }
#pragma unroll
        for (int k = 0; k < KH; k++) {
            int i = k * T + tid;
            if (i < halo3) { s_u1[own3 + i] = b1[k]; s_u2[own3 + i] = b2[k]; }
        }
        for (int i = tid; i < 3 * D.nacc; i += T) s_f[i] = 0.0;
        hq_pair_data mine = cur;
        __syncthreads();

        /* vmcnt retires in order: everything THIS patch still needs from memory (pair data of
         * the later rounds, nodal constants) is requested first, the next patch's staging loads
         * last, so no wait inside the element loop is ordered behind them */
        const bool iso = (D.flags & HQ_PATCH_ISO) != 0;
        hq_pair_data pr[NR - 1];
#pragma unroll
        for (int r = 1; r < NR; r++)
            if (r * T + tid < D.npairs) pr[r - 1] = hq_pair_load<false>(pidx, pc1, pc2, pbeta, D.pair_off + r * T + tid);
        double np[7];
        if (tid < D.nown) {
            if (iso) {
                const double* q = nt3 + 3 * ((int64_t)D.base + tid);
                np[0] = q[0]; np[1] = q[1]; np[4] = q[2];
                np[2] = np[3] = np[1];
                np[5] = np[6] = np[4];
            } else {
                const double* q = nt + 7 * ((int64_t)D.base + tid);
#pragma unroll
                for (int k = 0; k < 7; k++) np[k] = q[k];
            }
        }
        int slot_nn = slot_n + W;
        bool has_nn = has_n && slot_nn < end;
        int pnn = 0;
        hq_patch_desc Dnn = Dn;
        if (AHEAD && has_n) HQ_PIPE_ISSUE(Dn)
        if (has_nn) {
            pnn = order ? order[slot_nn] : slot_nn;
            Dnn = desc[pnn];
            if (AHEAD) HQ_PIPE_IDS(Dnn)
        }

#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int q = r * T + tid;
            if (r > 0) mine = pr[r - 1];
            if (q < D.npairs) {
                const uint4 raw = mine.raw;
                const double beta = mine.beta;
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
                hq_element_force(X, Y, Z, mine.c1, mine.c2);
#pragma unroll
                for (int n = 0; n < 8; n++) {
                    if (l[n] < D.nacc) {
                        atomicAdd(&s_f[3 * l[n] + 0], X[n]);
                        atomicAdd(&s_f[3 * l[n] + 1], Y[n]);
                        atomicAdd(&s_f[3 * l[n] + 2], Z[n]);
                    }
                }
            }
        }
        for (int q = NR * T + tid; q < D.npairs; q += T) {       /* patches with unusually many elements */
            mine = hq_pair_load<false>(pidx, pc1, pc2, pbeta, D.pair_off + q);
            const uint4 raw = mine.raw;
            const double beta = mine.beta;
            int l[8];
            l[0] = raw.x & 0xffff; l[1] = raw.x >> 16;
            l[2] = raw.y & 0xffff; l[3] = raw.y >> 16;
            l[4] = raw.z & 0xffff; l[5] = raw.z >> 16;
            l[6] = raw.w & 0xffff; l[7] = raw.w >> 16;
            double X[8], Y[8], Z[8];
            for (int n = 0; n < 8; n++) {
                const double* a = &s_u1[3 * l[n]];
                const double* b = &s_u2[3 * l[n]];
                double a0 = a[0], a1_ = a[1], a2_ = a[2];
                X[n] = a0 + beta * (a0 - b[0]);
                Y[n] = a1_ + beta * (a1_ - b[1]);
                Z[n] = a2_ + beta * (a2_ - b[2]);
            }
            hq_element_force(X, Y, Z, mine.c1, mine.c2);
            for (int n = 0; n < 8; n++) {
                if (l[n] < D.nacc) {
                    atomicAdd(&s_f[3 * l[n] + 0], X[n]);
                    atomicAdd(&s_f[3 * l[n] + 1], Y[n]);
                    atomicAdd(&s_f[3 * l[n] + 2], Z[n]);
                }
            }
        }
        if (F) {
            for (int k = src_ptr[p] + tid; k < src_ptr[p + 1]; k += T) {
                int ln = src_ent[2 * k], li = src_ent[2 * k + 1];
                for (int d = 0; d < 3; d++) atomicAdd(&s_f[3 * ln + d], F[3 * li + d] * dt2);
            }
        }
        if (ds_ptr && ds_ptr[p + 1] > ds_ptr[p]) {
            __syncthreads();
            for (int k = ds_ptr[p] + tid; k < ds_ptr[p + 1]; k += T) {
                const int src = ds_ent[3 * k], dst = ds_ent[3 * k + 1];
                const double deps = (double)(unsigned)ds_ent[3 * k + 2];
                for (int d = 0; d < 3; d++) atomicAdd(&s_f[3 * dst + d], s_f[3 * src + d] / deps);
            }
        }
        __syncthreads();

        for (int n = tid; n < D.nown; n += T) {
            if (n != tid) {
                if (iso) {
                    const double* q = nt3 + 3 * ((int64_t)D.base + n);
                    np[0] = q[0]; np[1] = q[1]; np[4] = q[2];
                    np[2] = np[3] = np[1];
                    np[5] = np[6] = np[4];
                } else {
                    const double* q = nt + 7 * ((int64_t)D.base + n);
#pragma unroll
                    for (int k = 0; k < 7; k++) np[k] = q[k];
                }
            }
            double* out = ung + 3 * ((int64_t)D.base + n);
#pragma unroll
            for (int d = 0; d < 3; d++) {
                double f = s_f[3 * n + d] + (np[1 + d] * s_u1[3 * n + d] - np[4 + d] * s_u2[3 * n + d]);
                out[d] = f / np[0];
            }
        }
        if (if_ptr) {
            for (int k = if_ptr[p] + tid; k < if_ptr[p + 1]; k += T) {
                int ln = if_ent[2 * k];
                double* o = iforce + 3 * (int64_t)if_ent[2 * k + 1];
                o[0] = s_f[3 * ln]; o[1] = s_f[3 * ln + 1]; o[2] = s_f[3 * ln + 2];
            }
        }
        if (!has_n) break;
        __syncthreads();                                       /* LDS is reused by the next patch */
        D = Dn; p = pn;
        Dn = Dnn; pn = pnn;
        has_n = has_nn;
        slot_n = slot_nn;
    }
#undef HQ_PIPE_IDS
#undef HQ_PIPE_ISSUE
}
