/*
 * hq_solver.h -- C-ABI of the MI355X (gfx950) explicit time-stepping engine.
 *
 * Drop-in boundary for the hot path of CMU-Quake/hercules quake/forward
 * (SURVEY.md s8b).  The reference has no plugin registry: the seam is a set
 * of C functions called from solver_run() (psolve.c:4241-4324) that work on
 * the caller-owned plain arrays of mesh_t (octor.h:166-179) and mysolver_t
 * (psolve.h:295-312).  Each entry point below names the reference function(s)
 * it replaces.  Plain pointers and sizes only; no HIP, torch or C++ types.
 *
 * Ownership : the caller owns every host buffer it passes; the context owns
 *             all device memory, streams and events.
 * Errors    : every call returns HQ_OK (0) or a negative hq_status; nothing
 *             aborts the process (the reference MPI_Abort()s, util.h:128).
 * Threading : one context per GPU / mesh partition; calls on one context must
 *             not overlap.  hq_run() only enqueues work; hq_sync(), hq_gather(),
 *             hq_download() wait for it.
 */
#ifndef HQ_SOLVER_H
#define HQ_SOLVER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HQ_API __attribute__((visibility("default")))

typedef struct hq_ctx hq_ctx;

typedef enum {
    HQ_OK            =  0,
    HQ_ERR_ARG       = -1,   /* bad argument / inconsistent description     */
    HQ_ERR_NOMEM     = -2,   /* host or device allocation failed            */
    HQ_ERR_DEVICE    = -3,   /* HIP runtime error (see hq_last_error)       */
    HQ_ERR_NODEVICE  = -4,   /* no gfx950 device / kernel image not loadable */
    HQ_ERR_COMM      = -5,   /* RCCL error                                  */
    HQ_ERR_STATE     = -6    /* call not valid in the context's state       */
} hq_status;

/* Element-kernel variants (hq_desc.variant). */
enum {
    HQ_VARIANT_AUTO    = 0,  /* = PATCH                                                        */
    HQ_VARIANT_SCATTER = 1,  /* element kernel + fp64 atomics, node kernel                     */
    HQ_VARIANT_PATCH   = 2   /* owner-computes patches, one fused kernel (+ interface kernels  */
                             /* on partitions); anchors of hanging nodes must be anchored      */
};

/*
 * One neighbour's shared-node list: messenger_t (psolve.h:230-249).
 * `mapping` holds local node ids in the order both sides agreed on
 * (ascending global id in the reference, psolve.c:4704-4863).
 */
typedef struct {
    int32_t        procid;
    int32_t        nodecount;
    const int32_t* mapping;
} hq_messenger;

/*
 * solver_float (psolve.h:60-64): the type of the caller's tm1 / tm2 arrays and n_t rows.  The reference's
 * -DSINGLE_PRECISION_SOLVER makes it float; the library built with -DHQ_SINGLE_PRECISION_SOLVER
 * (libhq_solver_f32.so, same sources) then takes and returns floats at these places and keeps the device-resident
 * state in floats (36 instead of 72 compulsory bytes per node and step); e_t, the source table and every sum inside
 * a kernel stay double, as the reference's own locals do.  A client compiles with the same macro and checks
 * hq_real_bytes() == sizeof(hq_real).  A separately named dtype with its own oracle build and tolerance
 * (tests/test_gpu_single_precision.py); never the headline.
 */
#ifdef HQ_SINGLE_PRECISION_SOLVER
typedef float hq_real;
#else
typedef double hq_real;
#endif

/*
 * One communication schedule: schedule_t (psolve.h:255-274).
 * c-list: ranks that OWN nodes I harbor -- I send them force contributions and
 *         receive displacements from them.
 * s-list: ranks that harbor nodes I OWN -- the reverse.
 */
typedef struct {
    int32_t             c_count;
    const hq_messenger* first_c;
    int32_t             s_count;
    const hq_messenger* first_s;
} hq_schedule;

/*
 * Everything solver_run() reads, as plain arrays.
 */
typedef struct {
    /* mesh_t */
    int32_t        lenum;        /* elements on this partition               */
    int32_t        nharbored;    /* nodes harbored (owned + shared copies)   */
    int32_t        ldnnum;       /* dangling (hanging) nodes owned           */
    const int32_t* lnid;         /* [lenum][8]  elem_t.lnid (octor.h:110-115) */
    const int32_t* node_xyz;     /* [nharbored][3] node_t.x,y,z in ticks (octor.h:133-147); optional
                                    (NULL): only used to place patch cuts on octree cubes */
    /* dnodeTable (octor.h:153-158) as CSR: node dn_ldnid[k] hangs on
     * anchors dn_lanid[dn_ptr[k] .. dn_ptr[k+1])  (deps = 2 or 4)            */
    const int32_t* dn_ldnid;
    const int32_t* dn_ptr;
    const int32_t* dn_lanid;
    /* mysolver_t */
    const double*  eTable;       /* [lenum][4]      e_t  c1..c4 (psolve.h:196-198) */
    const hq_real* nTable;       /* [nharbored][7]  n_t  (psolve.h:210-214)   */
    const hq_real* tm1;          /* [nharbored][3]  u(t)      or NULL = 0     */
    const hq_real* tm2;          /* [nharbored][3]  u(t-dt)   or NULL = 0     */
    hq_schedule    an_sched;     /* anchored-node schedule  (mysolver_t.an_sched) */
    hq_schedule    dn_sched;     /* dangling-node schedule  (mysolver_t.dn_sched) */
    /* Param */
    double         deltaT;       /* Param.theDeltaT (source force = F * dt^2) */
    int32_t        rank, nranks; /* Global.myID, Global.theGroupSize          */
    int32_t        variant;      /* HQ_VARIANT_*                              */
    int32_t        reserved;
    const int64_t* node_gnid;    /* [nharbored] node_t.gnid (octor.h:133-147); optional (NULL): only used by
                                    HQ_DEBUG_HALO, which falls back to a mix of node_xyz */
    /* ABI 5.  What solver_init built eTable FROM, optional (NULL / 0): edata_t of every element as solver_init left it
     * (psolve.h:95-97; mu_and_lambda may have rewritten Vp, psolve.c:3253-3263) and the constants it combined them with.
     * Where given, the kernel for meshes whose material differs from element to element (hq_k_brick_het) keeps 12 bytes per
     * element -- rho, Vs, Vp -- instead of the 24 of (c1, c2, beta), and rebuilds the caller's very doubles on the fly:
     * hq_create checks every element bit for bit against eTable first and keeps the 24 bytes wherever one differs.
     * The n_t rows of such units travel as {mass_simple, mass_simple - mass_minusaM}: mass_minusaM comes out EXACTLY,
     * mass2_minusaM as 2 m0 - (m0 - m1), equal to the caller's to 1e-15 relative (checked per node; a node that misses keeps
     * its unit unpacked) -- NOT bit for bit: the reference sums it in another order (psolve.c:3436-3471).  The difference is
     * of the class of the element sums' own order; tests/test_gpu_fullsize.py pins 10 000 steps against the unpacked form. */
    const float*  edata;         /* [lenum][4] edgesize (m), Vp, Vs, rho                                             */
    double mat_bbase;            /* Global.theBBase (compute_setab, psolve.c:5813-5876)                              */
    double mat_threshold_damping;/* Param.theThresholdDamping                                                        */
    double mat_threshold_vpvs;   /* Param.theThresholdVpVs                                                           */
} hq_desc;

typedef struct {
    int32_t variant;             /* variant actually built                    */
    int32_t npatches;            /* patch variant: number of patches          */
    int64_t patch_pairs;         /* patch variant: sum of per-patch elements  */
    int64_t device_bytes;        /* device memory owned by the context        */
    int32_t step;                /* next step to be executed                  */
    int32_t nranks;
    int32_t lattice_patches;     /* patch variant: patches with rows and lanes on one lattice (conflict-free LDS) */
    int32_t stencil_patches;     /* patch variant: patches stepped by hq_k_patch_stencil                          */
    int32_t ragged_patches;      /* patch variant: of the stencil patches, the lattice SUBSETS (faces, far-face cubes) */
    int32_t brick_units;         /* patch variant: workgroup units of hq_k_brick (tile column x planes)          */
    int64_t brick_nodes;         /* patch variant: nodes stepped by hq_k_brick (the rest belongs to the patches)  */
    int32_t brick_units_pernode; /* of brick_units: uniform coefficients, n_t rows of their own (hq_k_brick<true>)  */
    int32_t brick_units_het;     /* of brick_units: per-element coefficients (hq_k_brick_het)                      */
    int64_t pcie_h2d_bytes;      /* bytes the entry points moved host -> device since hq_create returned (source   */
    int64_t pcie_d2h_bytes;      /* windows, gather ids, uploads, host-staged halos) and device -> host (gathers,   */
                                 /* downloads, host-staged halos): what a run costs on PCIe between outputs         */
    int32_t transport;           /* 0 none, 1 RCCL, 2 IPC, 3 host-staged, 4 in-process group, 5 loopback (diagnostic)  */
    int32_t ipc_arena_coarse;    /* IPC: 1 if the receive arena had to be coarse-grained memory (ranks of one device only) */
    int32_t ipc_arena_kind;      /* IPC: 0 fine-grained, 1 uncached, 2 coarse-grained device memory                        */
    int32_t debug_halo;          /* 1: every halo record is checked on receipt (HQ_DEBUG_HALO / hq_options.debug_halo)     */
    int32_t brick_units_packed;  /* of brick_units_het: coefficients as three floats, n_t rows as two doubles (hq_desc.edata) */
    int32_t brick_units_ragged;  /* of the units with one n_t row: partly filled tiles (HQ_BK_RAGGED, hq_options.brick_ragged) */
    /* ABI 6 */
    int32_t brick_stream;        /* 1: the shell's patches run on the compute stream BESIDE the bricks, which have a stream of their own */
    int32_t brick_units_ragged_het; /* of brick_units_het: partly filled tiles (hq_k_brick_het<., RAGGED>, hq_options.brick_ragged_het) */
    /* Where a step's device time goes -- the library's share of the reference's per-phase report (print_timing_stat,
     * psolve.c:6041-6266: "Compute addforces e", "... schedule send data", ...), from HIP events, averaged over
     * `timed_steps` steps: every step of a context with hq_options.phase_timing = 1, every fourth step of an hq_run_timed batch.
     * microseconds per step; the phases overlap, so they do not add up to t_step_us. */
    int64_t timed_steps;
    double  t_step_us;           /* first kernel's start -> last kernel's end (compute streams and exchange chain)       */
    double  t_shell_us;          /* patch launches: the shell behind the bricks, or every node where there are no bricks
                                    (element force + nodal update + hanging nodes inside patches)                         */
    double  t_interior_us;       /* brick launches (element force + nodal update of the bulk)                            */
    double  t_chain_us;          /* exchange chain, its release (interface patches done) -> displacements shared: pack,
                                    transport, interface update, unpack, compute_adjust between ranks -- waits for the
                                    neighbours' records included                                                          */
    double  t_chain_exposed_us;  /* of it, what ran AFTER the step's compute kernels had ended (not hidden)                */
} hq_info;

/* Number of gfx950 devices visible (0 if none / no HIP runtime). */
HQ_API int hq_device_count(void);

/* Text of the last error on this thread ("" if none). */
HQ_API const char* hq_last_error(void);

/*
 * Build the device-resident solver state.
 * Replaces: the calloc()s of solver_init (psolve.c:3317-3325) and
 * stiffness_init (stiffness.c:101-105); consumes the eTable/nTable that
 * solver_init computed (psolve.c:3360-3473) and the schedules of
 * schedule_build (psolve.c:4704-4863).
 */
HQ_API int hq_create(const hq_desc* desc, int device, hq_ctx** out);

/*
 * Typed options (ABI 5).  The reference steers its solver through one explicit parameter struct (Param, psolve.c:193-284);
 * this is the library's: everything that selects kernels, plans or transports, per CONTEXT -- two contexts of one process
 * may differ, and nothing depends on the environment unless the caller wants it to.  hq_options_init fills every field
 * with "library default" (-1); hq_create_opts(desc, device, NULL, &ctx) == hq_create(desc, device, &ctx).
 * The environment: an HQ_* variable of a field's name overrides the field ONLY where the caller allows it --
 * hq_options.allow_env = 1, or allow_env = -1 in a process that sets HQ_ALLOW_ENV=1 (experiments, profiles/tools, the
 * test suite); a host program that passes allow_env = 0 cannot be steered by a stray variable.  Every setting is
 * resolved ONCE, at hq_create_opts; hq_get_options returns exactly that: what the context runs with (switches as 0 / 1,
 * -1 where the library's default applies).
 * `size` = sizeof(hq_options) of the caller: a newer library treats the fields an older client does not know as -1.
 */
typedef struct {
    uint64_t size;
    /* which kernels */
    int32_t no_bricks;           /* HQ_NO_BRICKS          1: no z-marching brick kernels, patches everywhere             */
    int32_t brick_cz;            /* HQ_BRICK_CZ           planes per brick unit (default: 32, shorter on small meshes)   */
    int32_t brick_minz;          /* HQ_BRICK_MINZ         shortest run of planes worth a tile column (4)                 */
    int32_t brick_minnodes;      /* HQ_BRICK_MINNODES     fewest nodes worth a tile column (512)                         */
    int32_t brick_no_het;        /* HQ_BRICK_NO_HET       1: no per-element-coefficient brick units (hq_k_brick_het)     */
    int32_t brick_no_ntsame;     /* HQ_BRICK_NO_NTSAME    1: n_t rows per node even where a unit shares one              */
    int32_t brick_by_component;  /* HQ_BRICK_BY_COMPONENT 1 / 0: the 100-register / 118-register form of hq_k_brick
                                                          (default: 100 on contexts with a transport)                   */
    int32_t brick_stream;        /* HQ_BRICK_STREAM       1: bricks on a stream of their own beside the patches          */
    int32_t brick_no_faces;      /* HQ_BRICK_NO_FACES     1: the domain's z faces stay with the patches instead of being stepped as
                                                          the first / last plane of the tile columns under them           */
    int32_t brick_half_tiles;    /* HQ_BRICK_HALF_TILES   (only with brick_ragged = 0) 0: no second planner round with 32-wide full tiles   */
    int32_t brick_no_pack;       /* HQ_BRICK_NO_PACK      1: per-element coefficients as 24-byte (c1, c2, beta) even where
                                                          hq_desc.edata would let them travel as 12 bytes                */
    int32_t patch_pipe;          /* HQ_PATCH_PIPE         6 hq_k_patch_seed (default), 4 hq_k_patch_pers, 0 hq_k_patch_step */
    int32_t patch_threads;       /* HQ_PATCH_THREADS      workgroup size of the patch kernels (512)                      */
    int32_t patch_pmax;          /* HQ_PATCH_PMAX         owned nodes per patch (768)                                    */
    int32_t patch_pmerge;        /* HQ_PATCH_PMERGE       neighbouring cubes are merged up to this (512)                 */
    int32_t patch_psplit;        /* HQ_PATCH_PSPLIT       a cube with more owned nodes is halved (pmax)                  */
    int32_t patch_nlmax;         /* HQ_PATCH_NLMAX        owned + halo nodes staged in LDS (1024)                        */
    int32_t patch_vmax;          /* HQ_PATCH_VMAX         extra accumulators for hanging nodes (set by the planner)      */
    int32_t patch_ragged;        /* HQ_PATCH_RAGGED       lattice-subset patches: 0 element form, 1 stencil form, 2 stencil
                                                          form except on the partition interface                        */
    int32_t patch_no_lattice;    /* HQ_PATCH_NO_LATTICE   1: no lattice patches                                          */
    int32_t patch_no_stencil;    /* HQ_PATCH_NO_STENCIL   1: lattice patches through the element kernels                 */
    int32_t patch_no_uniform;    /* HQ_PATCH_NO_UNIFORM   1: per-element coefficients even in uniform patches            */
    int32_t patch_no_iso;        /* HQ_PATCH_NO_ISO       1: 7-double n_t rows everywhere                                */
    int32_t patch_no_ntsame;     /* HQ_PATCH_NO_NTSAME    1: n_t rows per node even where a patch shares one             */
    int32_t patch_no_dedup;      /* HQ_PATCH_NO_DEDUP     1: every patch keeps its own connectivity rows                 */
    int32_t patch_wform;         /* HQ_PATCH_WFORM        0: hq_k_patch_pers keeps u1, u2 instead of w in LDS            */
    int32_t patch_merge_rounds;  /* HQ_PATCH_MERGE_ROUNDS one patch launch where the shell is at most this many rounds (1) */
    /* the exchange chain */
    int32_t overlap;             /* HQ_OVERLAP            1 / 0: the chain on its own stream beside the interior kernels
                                                          (default: yes between processes / GPUs, no inside one process) */
    int32_t no_overlap;          /* HQ_NO_OVERLAP         1: never create the exchange stream                            */
    int32_t reserve_cus;         /* HQ_RESERVE_CUS        CUs the interior launch leaves to the chain (8)                */
    int32_t cu_mask;             /* HQ_CU_MASK            1: compute stream with a CU mask that leaves reserve_cus free  */
    int32_t no_fused_share;      /* HQ_NO_FUSED_SHARE     1: the displacement sharing packed by its own kernel           */
    int32_t group_copies;        /* HQ_GROUP_COPIES       1: in-process groups exchange through peer copies              */
    int32_t debug_halo;          /* HQ_DEBUG_HALO         1: every halo record checked on receipt (-DDEBUG, psolve.c:5002-5007) */
    /* the IPC transport */
    int32_t ipc_arena;           /* HQ_IPC_ARENA          0 fine-grained, 1 uncached, 2 coarse-grained; default: first that exports */
    double  ipc_timeout_ms;      /* HQ_IPC_TIMEOUT_MS     a wait for a neighbour's records gives up after this (20 000)  */
    double  loopback_delay_us;   /* HQ_LOOPBACK_DELAY_US  diagnostic: loopback flags raised this late                    */
    /* messages */
    int32_t verbose;             /* HQ_PATCH_VERBOSE      1: where hq_create's time goes; 2: the brick planner's too     */
    int32_t quiet;               /* HQ_QUIET              1: no advice on stderr                                         */
    /* (added behind the others) */
    int32_t brick_ragged;        /* HQ_BRICK_RAGGED       0: no partly filled tile columns beside level interfaces and material
                                                          boundaries (the second planner round is then brick_half_tiles')    */
    int32_t brick_ragged_minfill;/* HQ_BRICK_RAGGED_MINFILL fewest nodes a plane of such a column owns, of 512 (128)        */
    /* (ABI 6) */
    int32_t allow_env;           /* 1: HQ_* environment variables override the fields above; 0: the environment is ignored;
                                    -1 (default): honoured only in a process that sets HQ_ALLOW_ENV=1 (experiments, tests) */
    int32_t brick_ragged_het;    /* HQ_BRICK_RAGGED_HET   0: no partly filled tiles of the per-element kernel (hq_k_brick_het<., RAGGED>):
                                                          beside level interfaces of a mesh whose material differs from element to
                                                          element the nodes then stay with the patches                       */
    int32_t reserved0;
    int32_t phase_timing;        /* HQ_PHASE_TIMING       1: every step records where its device time goes (six events per
                                                          step; hq_info.t_*_us) -- the library's share of print_timing_stat
                                                          (psolve.c:6041-6266); hq_run_timed batches always do          */
} hq_options;

HQ_API void hq_options_init(hq_options* opts, uint64_t size);
HQ_API int  hq_create_opts(const hq_desc* desc, int device, const hq_options* opts, hq_ctx** out);
/* what the context runs with (resolved at hq_create_opts); writes min(size, sizeof) bytes */
HQ_API int  hq_get_options(hq_ctx* ctx, hq_options* out, uint64_t size);

/* solver_delete (psolve.c:3627-3649). */
HQ_API int hq_destroy(hq_ctx* ctx);

/*
 * hq_info carries no size field of its own, so the call does: hq_get_info_sized writes min(size, sizeof(hq_info))
 * bytes, and the hq_get_info() of this header passes the caller's own sizeof -- a client compiled against an older
 * (shorter) hq_info is never written past its struct, a newer one gets what this library knows and zeros behind.
 * The exported SYMBOL hq_get_info is kept for binaries built against the headers that had no hq_get_info_sized; the
 * last of those ended its struct with brick_nodes, so the symbol writes those 56 bytes and never more.
 * hq_abi_version() == HQ_ABI_VERSION is the check a separately compiled client makes at start-up.
 */
#define HQ_ABI_VERSION 6
HQ_API int hq_abi_version(void);
/* sizeof(hq_real) of the library that is loaded: 8 (libhq_solver.so), 4 (libhq_solver_f32.so) */
HQ_API int hq_real_bytes(void);
HQ_API int hq_get_info_sized(hq_ctx* ctx, hq_info* info, uint64_t size);
HQ_API int hq_get_info(hq_ctx* ctx, hq_info* info);
#ifndef HQ_SOLVER_IMPLEMENTATION
#define hq_get_info(ctx, info) hq_get_info_sized((ctx), (info), (uint64_t)sizeof(hq_info))
#endif

/*
 * Multi-GPU: one RCCL communicator over the ranks that hold the partitions.
 * Rank 0 obtains a 128-byte id, the host code broadcasts it (MPI_Bcast in the
 * reference's world), every rank calls hq_comm_init.
 * Replaces: solver_run_init_comm / schedule_prepare (psolve.c:3785-3808).
 */
HQ_API int hq_comm_unique_id(void* id128);
HQ_API int hq_comm_init(hq_ctx* ctx, const void* id128);
/* Diagnostic: one grouped ncclRecv + ncclSend of `count` doubles from this rank to itself, through
 * the entry points and on the stream the halo exchange uses, checked on the host.  Lets a
 * single-GPU box exercise the RCCL binding (a communicator of one rank is enough). */
HQ_API int hq_comm_selftest(hq_ctx* ctx, int32_t count);

/*
 * Host-staged transport: the halo records travel through pinned host memory and the CALLER's transport -- in the
 * reference's world the MPI_Irecv / MPI_Isend / MPI_Waitall of schedule_senddata (psolve.c:5013-5033) on
 * comm_solver -- for systems without RCCL between the ranks' GPUs, or several ranks on one GPU.  At each of the (up
 * to four) exchanges of a step the engine packs on the device, copies the records to the host and calls
 * `fn(user, nrecv, recv_peer, recv_count, recv_buf, nsend, send_peer, send_count, send_buf, tag)`: counts in
 * doubles, one entry per neighbour with records, `tag` = 0..3 names the exchange (anchored-node contribution /
 * sharing, dangling-node contribution / sharing); `fn` must have received everything when it returns 0.
 * The exchange chain runs on its own stream as with RCCL: only that stream is waited for.
 */
typedef int (*hq_host_exchange_fn)(void* user, int32_t nrecv, const int32_t* recv_peer, const int64_t* recv_count,
                                   double* const* recv_buf, int32_t nsend, const int32_t* send_peer,
                                   const int64_t* send_count, const double* const* send_buf, int32_t tag);
HQ_API int hq_comm_init_host(hq_ctx* ctx, hq_host_exchange_fn fn, void* user);

/*
 * Device-to-device transport between PROCESSES of one node without RCCL: direct peer stores over xGMI (or inside one
 * GPU that several ranks share) into receive buffers exported through HIP IPC -- the "direct peer stores" design of
 * SURVEY s5 for schedule_senddata (psolve.c:4945-5079): pack (psolve.c:4985-5011) writes each record where its
 * receiver's unpack (:5035-5073) reads it, and the MPI_Waitall (:5033) becomes a wait on one epoch flag per sending
 * neighbour in the receiver's own memory.  No host hop, no collective, no library between the two GPUs.
 *   1. every rank: hq_comm_ipc_export(ctx, blob)            blob: HQ_IPC_BLOB_BYTES bytes
 *   2. the host all-gathers the blobs in rank order          (MPI_Allgather on comm_solver in the reference's world)
 *   3. every rank: hq_comm_init_ipc(ctx, all_blobs)          all_blobs: nranks x HQ_IPC_BLOB_BYTES
 * A rank whose neighbour stops sending does not spin for ever: a wait longer than HQ_IPC_TIMEOUT_MS (default 20 000)
 * gives up, and the next hq_sync returns HQ_ERR_COMM.
 * HQ_DEBUG_HALO (the reference's -DDEBUG exchange, psolve.c:5002-5007, 5058-5069) is carried: every record travels with
 * a check word -- its node's global id, mixed with the number of the exchange and the record's own three values -- in
 * an id arena beside the record arena; a record that is misrouted, stale (an earlier exchange's) or torn fails the
 * receiver's check and the next hq_sync returns HQ_ERR_COMM.  Every rank of the run must set it.  The receive arena is
 * fine-grained device memory, uncached device memory where the runtime will not export that, coarse-grained as the last
 * resort (ranks of one device only): hq_info.ipc_arena_kind says which.
 * Where the host driver exports device memory through dmabuf only (the MI355X pool this was built on), the process needs
 * HSA_ENABLE_IPC_MODE_LEGACY=0 in its environment BEFORE its first HIP call, or hipIpcGetMemHandle fails with "invalid
 * argument" (hq_comm_ipc_export then returns HQ_ERR_DEVICE with that text; RCCL needs the same setting).
 */
#define HQ_IPC_BLOB_BYTES 4096
HQ_API int hq_comm_ipc_export(hq_ctx* ctx, void* blob);
HQ_API int hq_comm_init_ipc(hq_ctx* ctx, const void* all_blobs);
/* the same with the number of blobs the caller holds (must be nranks): a short gather is refused, not read past its end */
HQ_API int hq_comm_init_ipc_n(hq_ctx* ctx, const void* all_blobs, int32_t nblobs);
/* Diagnostic (profiles/tools/rank_alone_trace.py): the IPC transport with this rank as its own only peer -- every record
 * it sends lands in its own receive buffers and raises its own flags.  The displacements of interface nodes are then
 * WRONG by construction; the step's kernels, streams and waits are exactly those of a rank whose neighbours answer
 * with zero latency. */
HQ_API int hq_comm_init_loopback(hq_ctx* ctx);

/*
 * In-process transport for hosts that drive several partitions from ONE process
 * (and for tests on a single GPU): ctxs[i] must be the context of rank i of n.
 * The halo records then travel by device-to-device copies ordered with HIP
 * events; hq_group_run enqueues `nsteps` steps for all members in lockstep
 * (hq_run on a linked context is not allowed to run alone).
 */
HQ_API int hq_group_link(hq_ctx** ctxs, int32_t n);
HQ_API int hq_group_run(hq_ctx** ctxs, int32_t n, int32_t nsteps);

/*
 * Source forces for steps [step0, step0+nsteps): F[nsteps][nloaded][3], the
 * payload of force_process.<rank> (quakesource.c:2453-2466).
 * Replaces: read_myForces (psolve.c:3651-3667) + compute_addforce_s (:5912-5928).
 * Steps outside the window apply no source.
 */
HQ_API int hq_set_source(hq_ctx* ctx, int32_t nloaded, const int32_t* loaded_lnid,
                         int32_t step0, int32_t nsteps, const double* F);

/*
 * Enqueue `nsteps` iterations of the solver_run loop body (psolve.c:4265-4319):
 * tm1/tm2 swap, source force, stiffness + Rayleigh damping force
 * (compute_addforce_effective stiffness.c:180-237 + damping_addforce
 * damping.c:29-103), force contribution exchange and hanging-node
 * distribution (psolve.c:4298-4301), nodal update (solver_compute_displacement
 * psolve.c:4072-4114), displacement sharing and hanging-node assignment
 * (psolve.c:4312-4315).  Asynchronous.
 */
HQ_API int hq_run(hq_ctx* ctx, int32_t nsteps);

/*
 * Wait for all enqueued work.  With HQ_DEBUG_HALO=1 in the environment at hq_create (the reference's
 * -DDEBUG build, psolve.c:5002-5007, 5058-5069) every halo record travels with the global identity of
 * its node and the receiver checks it: hq_sync then returns HQ_ERR_COMM if any record arrived for
 * another node than the schedule names.
 */
HQ_API int hq_sync(hq_ctx* ctx);

/*
 * solver_check_nan (psolve.c:3769-3782) on the device-resident tm1, tm2 (and force in the scatter
 * variant): *nonfinite = count of NaN / infinite values.  The reference aborts; here the caller decides.
 */
HQ_API int hq_check_finite(hq_ctx* ctx, int64_t* nonfinite);

/*
 * State as the NEXT loop iteration sees it after its swap -- what stations,
 * planes and checkpoints read (psolve.c:4271-4280): tm1 = u(step*dt),
 * tm2 = u((step-1)*dt).  Either output may be NULL.
 * hq_gather replaces the tm1[] reads of interpolate_station_displacements
 * (psolve.c:6679-6710) / planes; hq_download those of checkpoint_write
 * (io_checkpoint.c:98-112) and the 4D output (output.c:1265).
 */
HQ_API int hq_gather(hq_ctx* ctx, int32_t n, const int32_t* lnid, hq_real* tm1_out, hq_real* tm2_out);
/*
 * The same with tm3 = u((step-2)*dt), which the reference keeps when station accelerations are
 * printed (solver_compute_displacement psolve.c:4093-4101; read at :6762-6778).  The patch variant
 * has it for free -- it is the buffer the next step overwrites; zero before the second step and
 * after hq_upload, as the reference's calloc'ed tm3 is.  HQ_ERR_STATE in the scatter variant.
 */
HQ_API int hq_gather3(hq_ctx* ctx, int32_t n, const int32_t* lnid, hq_real* tm1_out, hq_real* tm2_out,
                      hq_real* tm3_out);
HQ_API int hq_download(hq_ctx* ctx, hq_real* tm1, hq_real* tm2);
/* checkpoint_read (io_checkpoint.c:134-236): overwrite the fields, set the step. */
HQ_API int hq_upload(hq_ctx* ctx, const hq_real* tm1, const hq_real* tm2, int32_t step);

/*
 * Single phases, for per-function parity tests against the reference loops
 * (scatter variant only; the patch variant fuses them):
 *   hq_phase_force : force += stiffness + damping element forces of the
 *                    current (tm1,tm2), after the swap and the source force
 *   hq_phase_update: solver_compute_displacement + zeroing of force
 *   hq_download_force: copy the force accumulator [nharbored][3]
 */
HQ_API int hq_phase_force(hq_ctx* ctx);
HQ_API int hq_phase_update(hq_ctx* ctx);
HQ_API int hq_download_force(hq_ctx* ctx, double* force);

/*
 * Timed run for bench.py: enqueues nsteps like hq_run between two HIP events
 * on the context's compute stream and additionally brackets every launch of
 * the dominant kernel with events.  Returns wall ms for the whole batch and
 * the average ms per launch of the dominant kernel.
 */
HQ_API int hq_run_timed(hq_ctx* ctx, int32_t nsteps, double* total_ms, double* kernel_ms_avg);

/*
 * Host-only self-check of the patch planner (needs no device): plans `desc` as hq_create would and
 * verifies the element rows, accumulate flags and node coverage of every patch.
 * report = {patches, lattice patches, (patch, element) pairs, distinct element-row blocks,
 *           LDS passes of the gathers, gather instructions (per 32-lane group), gather passes of the
 *           lattice patches (23 groups x 8 corners each when free of bank conflicts), faults}.
 */
HQ_API int hq_plan_check(const hq_desc* desc, int64_t report[8]);

/*
 * Host-only self-check of what hq_k_patch_stencil reads (needs no device; desc->node_xyz required): every patch-shape
 * table (lattice rows, element masks, boundary lists) against the mesh's connectivity, and the element-matrix blocks
 * of the boundary phase against the kernels' own element arithmetic.
 * report = {patches, patches with a table, full lattices among them, boundary nodes, element corners checked, faults}.
 */
HQ_API int hq_stencil_plan_check(const hq_desc* desc, int64_t report[6]);

/*
 * Host-only self-check of the brick planner (needs no device; desc->node_xyz required): plans the bricks -- the bulk
 * of uniformly refined, homogeneous regions, stepped by hq_k_brick on a tile-major node numbering of the device's own
 * -- as hq_create would and verifies them against the mesh's connectivity alone: permutation, coverage, the eight
 * equal elements and the dashpot-free n_t row of every brick node, and every neighbour the kernel will read.
 * report = {brick nodes, tile columns, units, units with one n_t row, units with per-element coefficients
 *           (hq_k_brick_het), neighbours checked, patch nodes, faults}.
 */
HQ_API int hq_brick_plan_check(const hq_desc* desc, int64_t report[8]);
/* ... with report[8] = units that own only part of their tile (ragged, one n_t row), report[9] = the nodes those own,
 * report[10] / [11] = the same for the ragged units of the per-element kernel; n >= 8 entries */
HQ_API int hq_brick_plan_check_n(const hq_desc* desc, int64_t* report, int32_t n);

/*
 * Host-only: the sixteen coefficients {p1[6], p2[6], q1[2], q2[2]} of the assembled 27-point stencil
 * S = c1 S1 + c2 S2 that hq_k_patch_stencil applies on uniform lattice patches -- the same operator
 * -(c1 K1 + c2 K2) that compute_addforce_effective + damping_addforce apply element by element
 * (stiffness.c:180-237, damping.c:29-103), assembled per node.  Diagonal blocks S[d][a][a]: class
 * (d_a != 0) + 2 x (number of the other two offsets that are non-zero); off-diagonal S[d][a][b] =
 * q[d_c != 0] sgn(d_a) sgn(d_b).  HQ_ERR_STATE if the symmetry check of the tables failed.
 */
HQ_API int hq_stencil_coefficients(double out[16]);

/* Name of the dominant kernel as it appears in rocprofv3 kernel traces. */
HQ_API const char* hq_dominant_kernel(hq_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* HQ_SOLVER_H */
