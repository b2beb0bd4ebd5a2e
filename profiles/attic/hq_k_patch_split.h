/* Round-2 experiment, measured and not kept (not compiled).  hq_k_patch_seed with the waves of the
 * workgroup in two roles: waves 0-7 do the element work of patch k (two elements per thread, no vector-
 * memory instruction in a lattice patch with uniform coefficients), waves 8-15 load patch k+1 (two nodes per
 * thread), write its LDS image and seeds and finish patch k.  Parity-green on every GPU parity test
 * (HQ_PATCH_PIPE=7).  64M box: 2.223 ms against 2.06 ms for hq_k_patch_seed; with the element work ablated
 * 1.998 ms, with the node loads ablated 1.984 ms, with both and the stores ablated 1.301 ms: each role's
 * path alone is as long as the whole iteration of the one-role kernel, so running them side by side gains
 * nothing -- eight waves do not keep the memory pipe as full as sixteen, and eight element waves in two
 * rounds leave the LDS queue's atomics in front of the second round's gathers. */
/*
 * hq_k_patch_split: the seeded single-barrier patch step (hq_k_patch_seed) with the work of a patch
 * split between two kinds of waves, so that the CU's memory pipe and its VALU / LDS pipes are busy
 * at the same time:
 *
 *   waves 0-7  (ELEMENT): gather, element arithmetic and atomics of patch k, two elements per thread
 *                         in two rounds (q = t and q = t + 512); no vector-memory instruction at all
 *                         in a lattice patch with uniform coefficients (the element rows of a lattice
 *                         patch are the same for every patch and live in registers, the coefficients
 *                         come through LDS from the loader waves);
 *   waves 8-15 (LOADER) : request u1, u2 and n_t of patch k+1 (two local nodes per thread), write its
 *                         LDS image and the accumulator seeds, and after the barrier finish patch k:
 *                         un = acc / m0 for their owned nodes.
 *
 * In hq_k_patch_pers / _seed all sixteen waves stand in the same phase: they block together on the
 * vector-memory queue while the node data of the next patch is requested (~4k cycles of a ~9k cycle
 * iteration, at the HBM rate of ~11 B/clk/CU), then compute together, then queue for the LDS atomics
 * together.  Here the loader waves' blocking costs nothing: the element waves compute meanwhile.
 * Buffers and the one barrier per patch are those of hq_k_patch_seed (two images, three accumulator
 * arrays).
 */
#define HQ_SPLIT_ET 512          /* element threads = loader threads */
#if HQ_EXP_IS(41) || HQ_EXP_IS(43)   /* ablation (results wrong): no node loads */
#define HQ_SPLIT_LD(p, i) (1e-3 * (double)((i) & 7))
#else
#define HQ_SPLIT_LD(p, i) (p)[i]
#endif

__global__ void __launch_bounds__(HQ_PERS_THREADS)
hq_k_patch_split(int32_t count, int32_t per_xcd, const int32_t* __restrict__ order, int32_t nrows,
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
                 int32_t* __restrict__ tickets, const uint16_t* __restrict__ lat_row, int64_t lat_pidx_off)
{
    extern __shared__ __align__(16) double s_mem[];
    /* LDS: image[2][2][3 nrows] | acc[3][nfacc] | coefficients[2][4] | ticket ring */
    double* __restrict__ s_fg = s_mem + 12 * nrows;
    double* __restrict__ s_coef = s_fg + 3 * nfacc;
    int32_t* __restrict__ s_tick = reinterpret_cast<int32_t*>(s_coef + 8);
    const int tid0 = threadIdx.x, T = HQ_PERS_THREADS;
    const int W = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7);
    const int end = min((xcd + 1) * per_xcd, count);
    const bool elem_role = tid0 < HQ_SPLIT_ET;          /* wave-uniform */
    HQ_WG_STAMP(0);
#define HQ_SLOT_PATCH(s) ((s) < end ? (order ? order[(s)] : (s)) : -1)
#define HQ_DRAW() (xcd * per_xcd + atomicAdd(&tickets[HQ_TICKET_STRIDE * xcd], 1))
    if (tid0 == 0) { for (int i = 0; i < 5; i++) s_tick[i] = HQ_DRAW(); }
    /* per-thread constants of a lattice patch: LDS rows of this thread's two local nodes (loader), element
     * rows of its two elements (element role) */
    const int lt0 = tid0 & (HQ_SPLIT_ET - 1);
    int lrowA = lat_row ? (int)lat_row[lt0] : lt0, lrowB = lat_row ? (int)lat_row[lt0 + HQ_SPLIT_ET] : lt0 + HQ_SPLIT_ET;
    hq_u32x4 lraw1 = { 0, 0, 0, 0 }, lraw2 = { 0, 0, 0, 0 };
    if (lat_pidx_off >= 0) {
        lraw1 = *(reinterpret_cast<const hq_u32x4*>(pidx) + (lat_pidx_off + lt0));
        lraw2 = *(reinterpret_cast<const hq_u32x4*>(pidx) + (lat_pidx_off + (lt0 + HQ_SPLIT_ET < HQ_LAT_NELEM ? lt0 + HQ_SPLIT_ET : 0)));
    }
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
    /* loader state carried from one iteration to the next: |mass_simple| of its two owned nodes of the current
     * patch, gather ids of its two local nodes of the next */
    double m0A = 1.0, m0B = 1.0;
    int32_t idA = 0, idB = 0;

    /* halo id of local node j of patch (P_, DD), clamped */
#define HQ_SPLIT_ID(P_, DD, j_) \
    halo[(int64_t)((P_) < 0 ? 0 : (P_)) * hstride + (((j_) >= (DD).nown && (j_) < (DD).nown + (DD).nhalo) ? (j_) - (DD).nown : 0)]
    /* n_t of local node j_ of patch DD into np_[0..6] (see hq_k_patch_seed) */
#define HQ_SPLIT_NT(DD, j_, np_)                                                                \
    {                                                                                           \
        const int64_t nn_ = (int64_t)(DD).base + (((j_) < (DD).nown && !((DD).flags & HQ_PATCH_NTSAME)) ? (j_) : 0); \
        const double* q3_ = nt3 + 3 * nn_;                                                      \
        np_[0] = q3_[0];                                                                        \
        if ((DD).flags & HQ_PATCH_ISO) { np_[1] = q3_[1]; np_[4] = q3_[2]; }                    \
        else {                                                                                  \
            const double* q7_ = nt + 7 * nn_;                                                   \
            _Pragma("unroll")                                                                   \
            for (int i_ = 1; i_ < 7; i_++) np_[i_] = q7_[i_];                                   \
        }                                                                                       \
    }
    /* image row and accumulator seed of local node j_ (LDS row row_) of patch DD */
#define HQ_SPLIT_WRITE(DD, j_, row_, x1, x2, np_, beta_, ib_, ab_)                              \
    {                                                                                           \
        hq_lds_double* iu1_ = (hq_lds_double*)s_mem + (ib_) * 6 * nrows;                        \
        hq_lds_double* iu2_ = iu1_ + 3 * nrows;                                                 \
        hq_lds_double* ac_ = (hq_lds_double*)s_fg + (ab_) * nfacc;                              \
        if ((j_) < (DD).nown + (DD).nhalo) {                                                    \
            if ((DD).flags & HQ_PATCH_WFORM) {                                                  \
                _Pragma("unroll")                                                               \
                for (int d = 0; d < 3; d++) iu1_[3 * (row_) + d] = x1[d] + (beta_) * (x1[d] - x2[d]); \
            } else {                                                                            \
                _Pragma("unroll")                                                               \
                for (int d = 0; d < 3; d++) { iu1_[3 * (row_) + d] = x1[d]; iu2_[3 * (row_) + d] = x2[d]; } \
            }                                                                                   \
        }                                                                                       \
        if ((j_) < (DD).nown) {                                                                 \
            const bool iso_ = ((DD).flags & HQ_PATCH_ISO) != 0;                                 \
            _Pragma("unroll")                                                                   \
            for (int d = 0; d < 3; d++) {                                                       \
                const double m2_ = iso_ ? np_[1] : np_[1 + d], m1_ = iso_ ? np_[4] : np_[4 + d]; \
                ac_[3 * (row_) + d] = np_[0] < 0.0 ? 0.0 : (m2_ * x1[d] - m1_ * x2[d]);         \
            }                                                                                   \
        } else if ((j_) < (DD).nacc) {              /* hanging nodes on owned anchors (id-ordered patches) */ \
            _Pragma("unroll")                                                                   \
            for (int d = 0; d < 3; d++) ac_[3 * (j_) + d] = 0.0;                                \
        }                                                                                       \
    }
    /* both nodes of this loader thread for patch DD (gather ids ia_, ib2_): request, then image + seed;
     * leaves |mass_simple| of the two nodes in (ma_, mb_) */
#define HQ_SPLIT_LOAD_PATCH(DD, ia_, ib2_, imgbuf_, accbuf_, ma_, mb_)                          \
    {                                                                                           \
        const int jA_ = lt, jB_ = lt + HQ_SPLIT_ET;                                             \
        const int64_t gA_ = jA_ < (DD).nown ? (int64_t)(DD).base + jA_ : (jA_ < (DD).nown + (DD).nhalo ? (int64_t)(ia_) : 0); \
        const int64_t gB_ = jB_ < (DD).nown ? (int64_t)(DD).base + jB_ : (jB_ < (DD).nown + (DD).nhalo ? (int64_t)(ib2_) : 0); \
        double xA1[3], xA2[3], xB1[3], xB2[3], npA[7], npB[7];                                  \
        _Pragma("unroll")                                                                       \
        for (int d = 0; d < 3; d++) { xA1[d] = HQ_SPLIT_LD(u1g, 3 * gA_ + d); xA2[d] = HQ_SPLIT_LD(u2g, 3 * gA_ + d); }   \
        _Pragma("unroll")                                                                       \
        for (int d = 0; d < 3; d++) { xB1[d] = HQ_SPLIT_LD(u1g, 3 * gB_ + d); xB2[d] = HQ_SPLIT_LD(u2g, 3 * gB_ + d); }   \
        HQ_SPLIT_NT(DD, jA_, npA)                                                               \
        HQ_SPLIT_NT(DD, jB_, npB)                                                               \
        const int64_t gc_ = (DD).pair_off;                                                      \
        const double wb_ = pbeta[gc_], wc1_ = pc1[gc_], wc2_ = pc2[gc_];                        \
        const int rowA_ = ((DD).flags & HQ_PATCH_LATTICE) ? lrowA : jA_;                        \
        const int rowB_ = ((DD).flags & HQ_PATCH_LATTICE) ? lrowB : jB_;                        \
        HQ_SPLIT_WRITE(DD, jA_, rowA_, xA1, xA2, npA, wb_, imgbuf_, accbuf_)                    \
        HQ_SPLIT_WRITE(DD, jB_, rowB_, xB1, xB2, npB, wb_, imgbuf_, accbuf_)                    \
        if (lt == 0) { s_coef[4 * (imgbuf_)] = wb_; s_coef[4 * (imgbuf_) + 1] = wc1_; s_coef[4 * (imgbuf_) + 2] = wc2_; } \
        ma_ = fabs(npA[0]); mb_ = fabs(npB[0]);                                                 \
    }

    {   /* prologue: zero the accumulators, then the loader waves bring patch 0 into image 0 / accumulators 0 */
        for (int i = tid0; i < 3 * nfacc; i += T) s_fg[i] = 0.0;
        __syncthreads();
        if (!elem_role) {
            const int lt = lt0;
            const int32_t i0A = HQ_SPLIT_ID(p0, D0, lt), i0B = HQ_SPLIT_ID(p0, D0, lt + HQ_SPLIT_ET);
            idA = HQ_SPLIT_ID(p1, D1, lt); idB = HQ_SPLIT_ID(p1, D1, lt + HQ_SPLIT_ET);
            HQ_SPLIT_LOAD_PATCH(D0, i0A, i0B, 0, 0, m0A, m0B)
            asm volatile("" : "+v"(idA), "+v"(idB), "+v"(m0A), "+v"(m0B));
        }
        asm volatile("" : "+v"(lrowA), "+v"(lrowB), "+v"(lraw1), "+v"(lraw2));
        __syncthreads();
    }

    int ab = 0;                                         /* accumulator array of the current patch: k % 3 */
    for (int k = 0;; k++) {
        int lt = lt0;
        asm volatile("" : "+v"(lt));                     /* (the per-patch address arithmetic stays inside the iteration) */
        hq_lds_double* __restrict__ s_u1 = (hq_lds_double*)s_mem + (k & 1) * 6 * nrows;
        hq_lds_double* __restrict__ s_u2 = s_u1 + 3 * nrows;
        hq_lds_double* __restrict__ s_f = (hq_lds_double*)s_fg + ab * nfacc;
        const int abn = ab == 2 ? 0 : ab + 1;
        const int slot3 = __builtin_amdgcn_readfirstlane(s_tick[(k + 3) & 7]);   /* drawn two iterations ago */
        const int p3 = HQ_SLOT_PATCH(slot3);
        const hq_patch_desc D3 = hq_patch_desc_or_empty(desc, p3);
        int32_t drawn = 0;
        double m0An = 1.0, m0Bn = 1.0;
        int32_t idAn = 0, idBn = 0;

        if (elem_role) {
            /* ---------------- element waves: patch k ---------------- */
            if (tid0 == 0) drawn = HQ_DRAW();
            const bool lat = (D0.flags & HQ_PATCH_LATTICE) != 0, uni = (D0.flags & HQ_PATCH_UNIFORM) != 0;
            const bool wf0 = (D0.flags & HQ_PATCH_WFORM) != 0;
#pragma unroll 1
            for (int r = 0; r < 2; r++) {
                const int q = lt + r * HQ_SPLIT_ET;
                if (__builtin_amdgcn_readfirstlane(r * HQ_SPLIT_ET) >= D0.npairs) break;
#if HQ_EXP_IS(40) || HQ_EXP_IS(43)   /* ablation (results wrong): no element work */
                const bool has_elem = q < 0;
#else
                const bool has_elem = q < D0.npairs;
#endif
                hq_u32x4 raw = r ? lraw2 : lraw1;
                double beta, c1, c2;
                if (!lat) raw = *(reinterpret_cast<const hq_u32x4*>(pidx) + (D0.pidx_off + (has_elem ? q : 0)));
                if (uni) {
                    const hq_lds_double* cf = (const hq_lds_double*)s_coef + 4 * (k & 1);
                    beta = cf[0]; c1 = cf[1]; c2 = cf[2];
                } else {
                    const int64_t gc = D0.pair_off + (has_elem ? q : 0);
                    beta = pbeta[gc]; c1 = pc1[gc]; c2 = pc2[gc];
                }
                if (has_elem) {
                    int l[8];
                    double X[8], Y[8], Z[8];
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
                    hq_element_force(X, Y, Z, c1, c2);
                    asm volatile("" : "+v"(raw));        /* the rows again from the packed element row */
                    HQ_PIDX_UNPACK(l, raw)
#pragma unroll
                    for (int n = 0; n < 8; n++) {
                        if (HQ_PIDX_HAS_ACC(raw, n)) {
                            hq_lds_double* a = hq_lds_row3(s_f, l[n]);
                            HQ_LDS_ADD(a + 0, X[n]);
                            HQ_LDS_ADD(a + 1, Y[n]);
                            HQ_LDS_ADD(a + 2, Z[n]);
                        }
                    }
                }
            }
            if (F) {                                     /* compute_addforce_s, psolve.c:5917-5927 */
                for (int i = src_ptr[p0] + lt; i < src_ptr[p0 + 1]; i += HQ_SPLIT_ET) {
                    int ln = src_ent[2 * i], li = src_ent[2 * i + 1];
                    for (int d = 0; d < 3; d++) HQ_LDS_ADD(&s_f[3 * ln + d], F[3 * li + d] * dt2);
                }
            }
        } else {
            /* ---------------- loader waves: patch k+1 in, gather ids of patch k+2 ---------------- */
            idAn = HQ_SPLIT_ID(p2, D2, lt); idBn = HQ_SPLIT_ID(p2, D2, lt + HQ_SPLIT_ET);
            HQ_SPLIT_LOAD_PATCH(D1, idA, idB, (k + 1) & 1, abn, m0An, m0Bn)
            asm volatile("" : "+v"(idAn), "+v"(idBn));
        }
        __syncthreads();
        if (tid0 == 0) s_tick[(k + 5) & 7] = drawn;      /* its old content was read at iteration k-6 */
        if (ds_ptr && ds_ptr[p0 + 1] > ds_ptr[p0]) {     /* compute_adjust DISTRIBUTION, psolve.c:5942-5987 */
            for (int i = ds_ptr[p0] + tid0; i < ds_ptr[p0 + 1]; i += T) {
                const int src = ds_ent[3 * i], dst = ds_ent[3 * i + 1];
                const double deps = (double)(unsigned)ds_ent[3 * i + 2];
                for (int d = 0; d < 3; d++) HQ_LDS_ADD(&s_f[3 * dst + d], s_f[3 * src + d] / deps);
            }
            __syncthreads();
        }
        if (!elem_role) {
            /* interface partial forces (psolve.c:4301: pure element force, their seed is 0), then the update of
             * this thread's owned nodes of patch k (solver_compute_displacement, psolve.c:4078-4106) */
            if (if_ptr && if_ptr[p0 + 1] > if_ptr[p0]) {
                for (int i = if_ptr[p0] + lt; i < if_ptr[p0 + 1]; i += HQ_SPLIT_ET) {
                    int ln = if_ent[2 * i];
                    double* o = iforce + 3 * (int64_t)if_ent[2 * i + 1];
                    o[0] = s_f[3 * ln]; o[1] = s_f[3 * ln + 1]; o[2] = s_f[3 * ln + 2];
                }
            }
            const bool lat0 = (D0.flags & HQ_PATCH_LATTICE) != 0;
            if (lt < D0.nown) {
                double* out = ung + 3 * ((int64_t)D0.base + lt);
                const hq_lds_double* __restrict__ acc = s_f + 3 * (lat0 ? lrowA : lt);
#if !(HQ_EXP_IS(42) || HQ_EXP_IS(43))   /* ablation (results wrong): no stores */
#pragma unroll
                for (int d = 0; d < 3; d++) out[d] = acc[d] / m0A;
#else
                if (acc[0] == 1.2345e-300) out[0] = m0A;
#endif
            }
            if (lt + HQ_SPLIT_ET < D0.nown) {
                double* out = ung + 3 * ((int64_t)D0.base + lt + HQ_SPLIT_ET);
                const hq_lds_double* __restrict__ acc = s_f + 3 * (lat0 ? lrowB : lt + HQ_SPLIT_ET);
#pragma unroll
                for (int d = 0; d < 3; d++) out[d] = acc[d] / m0B;
            }
            m0A = m0An; m0B = m0Bn; idA = idAn; idB = idBn;
        }
        if (p1 < 0) break;
        p0 = p1; p1 = p2; p2 = p3;
        D0 = D1; D1 = D2; D2 = D3;
        ab = abn;
    }
    HQ_PERS_EXIT()
#undef HQ_PERS_EXIT
#undef HQ_SLOT_PATCH
#undef HQ_DRAW
#undef HQ_SPLIT_ID
#undef HQ_SPLIT_NT
#undef HQ_SPLIT_WRITE
#undef HQ_SPLIT_LOAD_PATCH
}


