/*
 * hq_host.h -- C host side above include/hq_solver.h.
 *
 * Mirrors, for uniformly refined boxes, the host code that surrounds the hot
 * path in CMU-Quake/hercules quake/forward (the reference is C, so the host
 * side is C too):
 *
 *   hqh_box_create    mesh_generate + octor_extractmesh for a uniform box
 *                     (psolve.c:1920-2176, octor.c:5268-6650: elements in octree
 *                     pre-order, nodes in Z-order, block partition
 *                     octor.c:4939-4944, node ownership :5466-5475),
 *                     solver_init (psolve.c:3280-3510: eTable, nTable with
 *                     Rayleigh/mass damping and Lysmer dashpots) and
 *                     schedule_build (psolve.c:4704-4863)
 *   hqh_point_source  source_initnodalforce (quakesource.c:420-476) for one
 *                     double-couple point source
 *   hqh_stations      compute_csi_eta_dzeta (psolve.c:6378-6440)
 *   hqh_solver_run    solver_run (psolve.c:4241-4324): source window, step loop,
 *                     station / plane / checkpoint cadence, on an hq_ctx
 *   hqh_octbox_*      layered boxes on several octree levels with hanging nodes, whole
 *                     or cut into octor's per-rank tables; hqh_layered_column: the Vs rule
 *   hqh_etree_read, hqh_mesh_from_leaves   the reference's mesh.e -> octor's mesh tables
 *   hqh_forcefile_*, hqh_checkpoint_*, hqh_plane_*, hqh_station_*, hqh_wavefield_*   its file formats
 *
 *   hqh_cvm_*         a CVM etree as the mesher's material model (cvm_query)
 *
 * The distributed octree mesher, the slip-function source generator and the IO-PE pool stay in the
 * reference (out of scope).
 */
#ifndef HQ_HOST_H
#define HQ_HOST_H

#include <stdint.h>

#include "hq_solver.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hqh_box hqh_box;

enum { HQH_DAMP_NONE = 0, HQH_DAMP_RAYLEIGH = 1, HQH_DAMP_MASS = 2 };

typedef struct {
    int32_t       nx, ny, nz;        /* elements per axis; powers of two           */
    double        h;                 /* element edge, metres (edata_t.edgesize)    */
    int32_t       nlayers;           /* >= 1 horizontal layers                     */
    const double* layer_ztop;        /* [nlayers] top depth of each layer, [0]=0   */
    const float*  layer_vp;          /* [nlayers] edata_t.Vp                       */
    const float*  layer_vs;          /* [nlayers] edata_t.Vs                       */
    const float*  layer_rho;         /* [nlayers] edata_t.rho                      */
    double        deltaT;            /* Param.theDeltaT                            */
    double        freq;              /* Param.theFreq (simulation_wave_max_freq_hz)*/
    int32_t       damping;           /* HQH_DAMP_* (type_of_damping)               */
    double        threshold_damping; /* the_threshold_damping                      */
    double        threshold_vpvs;    /* the_threshold_Vp_over_Vs                   */
    int32_t       halfspace;         /* 1: z = 0 is a free surface (-DHALFSPACE)   */
    int32_t       rank, nranks;      /* this partition / number of partitions (<= 64) */
    int32_t       lateral_classes;   /* > 1: the element (i, j, k) belongs to class hash(i, j, k) mod this and its
                                      * Vp, Vs, rho are its layer's times 1 + lateral_amp (2 class / (classes - 1) - 1):
                                      * material that differs from element to element, as on a real CVM mesh
                                      * (psolve.c:3360-3409 reads every element's own edata_t); 0 | 1: depth only  */
    double        lateral_amp;       /* 0 <= amp < 1                                                              */
    int32_t       origin[3];         /* this box is a WINDOW of a larger one whose element (origin + (i, j, k)) it holds:
                                      * classes and layers are taken at the larger box's indices (bench.py's parity
                                      * windows); {0, 0, 0} otherwise                                             */
    int32_t       solver_float;      /* 0 | 8: the n_t rows as the reference's double build sums them; 4: as its
                                      * -DSINGLE_PRECISION_SOLVER build does (psolve.h:60-64: float fields, every `+=` of
                                      * psolve.c:3440-3471 rounds to float) -- still handed over as doubles, exact floats.
                                      * On a partition: as that build's N RANKS sum them -- each rank its own elements,
                                      * then solver_init's three mass exchanges (psolve.c:3498-3507: sharers' rows added
                                      * to the owner's in messenger order, hanging nodes' shares handed out by their
                                      * owners in between); every harbored copy holds its owner's row.  Bit for bit the
                                      * multi-rank float oracle's (tests/test_host_float_tables.py), which steps to the float
                                      * reference's 8-rank checkpoints bit for bit (tests/test_oracle_single_precision.py) */
} hqh_box_params;

typedef struct {
    int64_t total_elements, total_nodes;   /* whole box                            */
    int32_t lenum, nharbored, nowned;      /* this partition                       */
    int32_t nneighbors;                    /* ranks I exchange with                */
    int64_t shared_nodes;                  /* harbored nodes someone else owns or shares */
} hqh_box_info;

HQ_API int  hqh_box_create(const hqh_box_params* p, hqh_box** out);
HQ_API void hqh_box_destroy(hqh_box* box);
HQ_API int  hqh_box_get_info(const hqh_box* box, hqh_box_info* info);

/* Fill an hq_desc with pointers into arrays owned by `box` (valid until
 * hqh_box_destroy).  variant/tm1/tm2 are left for the caller. */
HQ_API int  hqh_box_desc(const hqh_box* box, hq_desc* desc);

/* Read-only views for tests: lnid [lenum][8], node_ijk [nharbored][3] (element
 * units), eTable [lenum][4], nTable [nharbored][7], owner [nharbored]. */
HQ_API const int32_t* hqh_box_lnid(const hqh_box* box);
HQ_API const int32_t* hqh_box_node_ijk(const hqh_box* box);
HQ_API const double*  hqh_box_etable(const hqh_box* box);
HQ_API const double*  hqh_box_ntable(const hqh_box* box);
HQ_API const int32_t* hqh_box_owner(const hqh_box* box);
/* the n_t rows of any of this library's tables as the float array a reference built with -DSINGLE_PRECISION_SOLVER holds
 * (hq_desc.nTable of libhq_solver_f32.so): out[rows][7]; exact where the tables were made with solver_float = 4 */
HQ_API int hqh_ntable_to_float(const double* ntable, int64_t rows, float* out);
/* edata_t of this partition's elements as solver_init reads them (psolve.c:3372-3385): out[lenum][3] = Vp, Vs, rho */
HQ_API int hqh_box_material(const hqh_box* box, float* out);

/*
 * Double-couple point source at (x,y,z) metres: finds the containing element
 * and returns the 8 equivalent nodal force vectors for unit moment
 * (source_initnodalforce).  *nloaded is 8 if the element is on this partition,
 * else 0 (each rank loads only its own elements' nodes, quakesource.c).
 */
HQ_API int hqh_point_source(const hqh_box* box, double x, double y, double z, double strike_deg,
                            double dip_deg, double rake_deg, int32_t* nloaded, int32_t lnid[8],
                            double pattern[24]);

/* Stations: containing element and trilinear weights; mine[s] = 1 if the
 * element is local.  ids [n][8], phi [n][8]. */
HQ_API int hqh_stations(const hqh_box* box, int32_t n, const double* xyz, int32_t* ids, double* phi,
                        int32_t* mine);

/* Called at every station print step with the interpolated displacements [n][3] -- or, with
 * station_derivs = 1 | 2, [n][6] | [n][9]: displacement, velocity (u1 - u2) / dt and acceleration
 * (u1 - 2 u2 + u3) / dt^2 as print_station_velocities / print_station_accelerations add them
 * (psolve.c:6737-6787). */
typedef void (*hqh_station_fn)(void* user, int32_t step, int32_t n, const double* disp);

typedef struct {
    /* source: a reference force file (force_file != NULL: nloaded/loaded_lnid must match its
     * header) or the synthetic F(step) = moment * ramp(step*dt) * pattern */
    const char* force_file;
    int32_t nloaded;
    const int32_t* loaded_lnid;
    const double*  pattern;          /* [nloaded][3] nodal force for unit moment  */
    double         moment;           /* N m                                        */
    double         rise_time;        /* s; smooth ramp 0.5 (1 - cos(pi t / T))      */
    int32_t        source_window;    /* steps of force table uploaded at a time    */
    /* stations */
    int32_t nstations;
    const int32_t* station_ids;      /* [n][8] */
    const double*  station_phi;      /* [n][8] */
    int32_t        station_rate;     /* output_stations_print_rate; 0 = never      */
    hqh_station_fn station_fn;
    void*          station_user;
    /* output planes (io_planes.c): every plane_rate steps, 3 doubles per grid point of every
     * plane appended to <plane_dir>/planedisplacements.<i> (Old_print_plane_displacements,
     * io_planes.c:252-270).  Points of all planes concatenated; plane_mine (may be NULL) = 0
     * where the containing element is on another partition: that value is left as last written */
    int32_t        nplanes;
    const int32_t* plane_npoints;    /* [nplanes] */
    const int32_t* plane_ids;        /* [sum npoints][8] */
    const double*  plane_phi;        /* [sum npoints][8] */
    const int32_t* plane_mine;       /* [sum npoints] or NULL */
    int32_t        plane_rate;       /* output_planes_print_rate; 0 = never */
    const char*    plane_dir;        /* output_planes_directory */
    /* checkpoints (solver_write_checkpoint psolve.c:3842-3851, checkpoint_write
     * io_checkpoint.c:29-127): at every step != step0 that is a multiple of checkpoint_rate the
     * state goes to <checkpoint_dir>/checkpoint.out0 and .out1 in turn; single partition only:
     * on a context of nranks > 1 hqh_solver_run returns HQ_ERR_STATE when checkpoint_rate > 0
     * (partitions call hqh_checkpoint_write themselves with their rank, rank 0 first) */
    int32_t        checkpoint_rate;  /* checkpointing_rate; 0 = never */
    const char*    checkpoint_dir;   /* checkpoint_path */
    /* 0: stations report displacement; 1: + velocity; 2: + velocity and acceleration (needs the
     * patch variant: hq_gather3) */
    int32_t        station_derivs;
    /* 4D wavefield files (solver_output_wavefield psolve.c:3858-3864, po_do_output
     * output.c:1358-1400): at every step that is a multiple of wavefield_rate the displacement
     * and / or velocity of the partition's owned nodes goes to its place in the file(s), which
     * hqh_wavefield_create made.  wavefield_count = 0: the whole mesh is this partition. */
    int32_t        wavefield_rate;        /* simulation_output_rate; 0 = never */
    const char*    wavefield_disp_file;   /* output_displacement_file or NULL */
    const char*    wavefield_vel_file;    /* output_velocity_file or NULL */
    int64_t        wavefield_total_nodes;
    int64_t        wavefield_base_gnid;   /* global id of the first owned node */
    int32_t        wavefield_first_owned; /* its local id */
    int32_t        wavefield_count;       /* owned nodes (contiguous in local and global order) */
} hqh_run_params;

/* The reference's 4D output file (out_hdr_t psolve.h:120-186, 136 bytes; then output_steps blocks
 * of total_nodes x 3 doubles in global node order). */
typedef struct {
    int64_t total_nodes, total_elements;
    double  domain_x, domain_y, domain_z;
    double  mesh_ticksize;                /* metres per octor tick */
    double  delta_t;
    int32_t output_rate;                  /* simulation_output_rate */
    int32_t total_time_steps;
} hqh_wavefield_info;
/* po_init_output_header + po_create_file (output.c:514-598).  quantity: 1 displacement, 2 velocity. */
HQ_API int hqh_wavefield_create(const char* path, const hqh_wavefield_info* info, int32_t quantity);
/* write_displacement / write_velocity (output.c:1230-1352) of output step `out_step` (0-based) for
 * `count` nodes from local id first_owned whose global ids start at base_gnid; tm1, tm2 the
 * partition's host arrays (tm2 unused for displacement). */
HQ_API int hqh_wavefield_write(const char* path, int64_t total_nodes, int32_t quantity, int32_t out_step,
                               int64_t base_gnid, int32_t first_owned, int32_t count, const double* tm1,
                               const double* tm2, double delta_t);

/* solver_run: steps [step0, step0 + nsteps) on `ctx` (a context made from `box`; or, _on, from
 * any mesh of `nharbored` nodes and time step deltaT). */
HQ_API int hqh_solver_run(hq_ctx* ctx, const hqh_box* box, const hqh_run_params* rp,
                          int32_t step0, int32_t nsteps);
HQ_API int hqh_solver_run_on(hq_ctx* ctx, double deltaT, int32_t nharbored, const hqh_run_params* rp,
                             int32_t step0, int32_t nsteps);

/*
 * Files in the reference's formats, so runs can be exchanged with psolve.
 *
 * force_process.<rank> (written by quakesource.c:2453-2466, read per step by
 * read_myForces psolve.c:3651-3667): int32 n; int32 lnid[n]; double F[steps][n][3].
 *   hqh_forcefile_info : n and the number of steps in the file; lnid (may be NULL) gets n ids
 *   hqh_forcefile_read : F for steps [step0, step0+nsteps); steps past the end read as 0
 *   hqh_forcefile_write: create such a file
 */
HQ_API int hqh_forcefile_info(const char* path, int32_t* nloaded, int32_t* nsteps, int32_t* lnid, int32_t lnid_cap);
HQ_API int hqh_forcefile_read(const char* path, int32_t step0, int32_t nsteps, double* F);
HQ_API int hqh_forcefile_write(const char* path, int32_t nloaded, const int32_t* lnid, int32_t nsteps, const double* F);

/*
 * checkpoint.out{0,1} / checkpoint.in (io_checkpoint.c:29-236): header
 * {groupsize, step, nharboredmax} ints; rank r's stripe at 12 + 2 r nharboredmax 24:
 * u((step-1) dt) then u(step dt), nharbored fvector_t each.
 *   hqh_checkpoint_write: rank 0 must have been called (creates the file) before the
 *                         others write their stripes (the reference barriers there);
 *   hqh_checkpoint_read : verifies the rank count, loads this rank's stripe into the
 *                         context and sets its step; returns the step in *step.
 */
HQ_API int hqh_checkpoint_write(hq_ctx* ctx, const char* path, int32_t step, int32_t rank, int32_t nranks,
                                int32_t nharbored, int32_t nharboredmax);
HQ_API int hqh_checkpoint_read(hq_ctx* ctx, const char* path, int32_t rank, int32_t nranks,
                               int32_t nharbored, int32_t* step);

/*
 * Output planes (io_planes.c:280-520).  A plane is a grid of n_strike x n_dip points from an
 * origin (domain coordinates, metres) along strike and down dip; point index =
 * iStrike * n_dip + iDownDip.  hqh_plane_points restates compute_global_coords
 * (geometrics.c:33-70, rake = 0); hqh_domain_coords the (longitude, latitude) -> domain x, y
 * map through the four surface corners (compute_domain_coords_linearinterp,
 * geometrics.c:178-244; x follows latitude).  Containing elements and weights of the points:
 * hqh_stations.
 */
typedef struct {
    double  origin[3];
    double  step_strike;
    int32_t n_strike;
    double  step_dip;
    int32_t n_dip;
    double  strike_deg, dip_deg;
} hqh_plane;

HQ_API int hqh_plane_points(const hqh_plane* pl, double* xyz);
HQ_API int hqh_domain_coords(double lon, double lat, const double lon_corners[4], const double lat_corners[4],
                             double len_x, double len_y, double* x, double* y);

/* One station line in the reference's text format (psolve.c:6727-6731). */
HQ_API int hqh_station_format(char* buf, int32_t cap, double time, const double disp[3]);
/* The same with the velocity / acceleration columns (psolve.c:6755-6787): vals = 3 (1 + derivs)
 * numbers as the station callback gets them. */
HQ_API int hqh_station_format_derivs(char* buf, int32_t cap, double time, const double* vals, int32_t derivs);
/* The first line of a station file (psolve.c:6636-6648; no newline: every data line starts with one). */
HQ_API int hqh_station_header(char* buf, int32_t cap, int32_t derivs);
/* Displacement, velocity, acceleration of one station from the 8 node rows of tm1, tm2, tm3 (each
 * [8][3]; tm2 / tm3 may be NULL below derivs 1 / 2), in the reference's order of operations. */
HQ_API int hqh_station_kinematics(const double* phi, const double* tm1, const double* tm2, const double* tm3,
                                  double dt, int32_t derivs, double* vals);

/*
 * Two-level layered box: the top nz_fine layers of elements of edge h over nz_coarse
 * layers of elements of edge 2h -- what the reference's Vs rule (quake_util.c:215-225)
 * makes of a soft layer over a stiff half-space; a 2:1 interface whose fine nodes that are
 * not coarse vertices are hanging nodes (octor node_setproperty, octor.c:3280-3860).
 * Produces octor's tables: elements in octree pre-order, nodes in Z-order,
 * dnodeTable with anchors in octor's list order (octor.c:6493-6612), eTable, and nTable
 * after the hanging-node mass distribution (psolve.c:3498-3507).
 *
 * With nranks > 1 the box is cut the way octor_partitiontree / octor_extractmesh leave it on
 * rank `rank`: a contiguous block of the pre-ordered leaves (octor.c:4939-4944); a node is
 * owned by the rank whose leaf contains its far-boundary-adjusted point (octor.c:5466-5475);
 * the rank harbors the vertices of its elements, the nodes it owns (direct sharing,
 * octor.c:5516-5793) and the anchors of the hanging nodes it owns (indirect sharing,
 * node_harboranchored octor.c:3916-4042, :5795-6040); dnodeTable lists the hanging nodes it
 * OWNS; an_sched / dn_sched are schedule_build's (psolve.c:4704-4863), the messengers in ITS
 * order (a new messenger at the head of its list; a node's sharers as octor's share list has
 * them: the ranks its owner met as neighbours, in that order -- the reference's multi-rank
 * runs are reproduced bit for bit only in it); nTable carries the summed masses of shared nodes
 * (what solver_init's exchange leaves, psolve.c:3498-3507).  Every rank builds the whole box
 * first and cuts its part out.
 */
typedef struct hqh_octbox hqh_octbox;

typedef struct {
    int32_t nx, ny;              /* fine elements per horizontal axis (even)            */
    int32_t nz_fine;             /* fine layers on top (even)                           */
    int32_t nz_coarse;           /* layers of edge 2h below                             */
    double  h;                   /* fine edge, metres                                   */
    float   vp_top, vs_top, rho_top;
    float   vp_bot, vs_bot, rho_bot;
    double  deltaT, freq;
    int32_t damping;
    double  threshold_damping, threshold_vpvs;
    int32_t halfspace;
    int32_t rank, nranks;        /* this partition / number of partitions (0, 0 or 1 = the whole
                                    box); nranks <= 64 */
    int32_t solver_float;        /* as hqh_box_params.solver_float */
} hqh_octbox_params;

HQ_API int  hqh_octbox_create(const hqh_octbox_params* p, hqh_octbox** out);

/*
 * The same on any number of octree levels: from the top, layers[L] layers of elements of edge
 * h * 2^L (L = 0 .. nlevels-1 <= 7), every element layer with its own material -- the mesh the
 * reference's mesher makes of a layered model whose Vs rule asks for coarser elements with depth
 * (pinned on its own three-level mesh, tests/golden/c5_three_level).  nx, ny in finest edges,
 * multiples of 2^(nlevels-1); every slab must end on a plane the next level's cells align to.
 */
typedef struct {
    int32_t        nx, ny;
    int32_t        nlevels;
    const int32_t* layers;       /* [nlevels] element layers of edge h * 2^L */
    double         h;            /* finest edge, metres */
    const float   *vp, *vs, *rho;/* [sum of layers[]]: one material per element layer, from the top */
    double         deltaT, freq;
    int32_t        damping;
    double         threshold_damping, threshold_vpvs;
    int32_t        halfspace;
    int32_t        rank, nranks;
    int32_t        solver_float; /* as hqh_box_params.solver_float */
} hqh_octlevels_params;

HQ_API int  hqh_octbox_create_levels(const hqh_octlevels_params* p, hqh_octbox** out);
/* With nranks > 1 and >= 4 M elements in the whole box (or HQH_OCTBOX_LOCAL=1; =0 forbids) only this rank's tables are
 * built -- from the sorted leaf keys, no whole-box arrays: equal table for table to the cut of the whole box, except
 * that the global node index (view 8, `gid`) is then -1 for every node. */

/*
 * The column of leaves the reference's mesher makes of a layered model (material a function of
 * depth only), from the top: octor_refinetree with toexpand = vsrule (psolve.c:2185-2210,
 * quake_util.c:215-225: split while edge > Vs / factor, factor = simulation_wave_max_freq_hz x
 * simulation_node_per_wavelength) on the material setrec gives a leaf (psolve.c:1307-1397: the
 * minimum-Vs sample of the points at 0.01, 1 and 1.99 half-edges; stop at the first sample with
 * Vs <= vscut, which is then raised to vscut at the same Vp/Vs), then octor_balancetree (2:1
 * between neighbours, new leaves re-sampled).  The column starts as `ncoarse` cells of edge h0.
 * Output: nleaves leaves from the top, edge[] in metres and their materials (capacity `cap`).
 */
typedef struct {
    int32_t       nlayers;
    const double* ztop;          /* [nlayers] top depth of each layer, metres, [0] = 0 */
    const float  *vp, *vs, *rho; /* [nlayers] */
} hqh_layered_model;

HQ_API int hqh_layered_column(const hqh_layered_model* m, double h0, int32_t ncoarse, double factor, double vscut,
                              int32_t cap, double* edge, float* vp, float* vs, float* rho, int32_t* nleaves);
HQ_API void hqh_octbox_destroy(hqh_octbox* box);
HQ_API int  hqh_octbox_desc(const hqh_octbox* box, hq_desc* desc);
HQ_API int  hqh_octbox_solver_run(hq_ctx* ctx, const hqh_octbox* box, const hqh_run_params* rp,
                                  int32_t step0, int32_t nsteps);
/* views: which = 0 lnid [E][8], 1 node_xyz [N][3] (fine-edge units), 2 dn_ldnid, 3 dn_ptr,
 * 4 dn_lanid (int32); 5 eTable [E][4], 6 nTable [N][7] (double); 7 owner [N], 8 global node
 * id [N] (int32; partitions only).  *count = entries. */
HQ_API const void* hqh_octbox_view(const hqh_octbox* box, int32_t which, int64_t* count);
/* messenger lists of a partition: sched 0 = an_sched, 1 = dn_sched; list 0 = c-list, 1 = s-list */
HQ_API int hqh_octbox_schedule(const hqh_octbox* box, int32_t sched, int32_t list, int32_t* count,
                               const hq_messenger** first);

/*
 * mesh.e: the mesh database the reference's mesher writes (mesh_output, psolve.c:2361-2562) -- an
 * etree (etree/etree.c, btree.c: 273-byte etree header, B-tree meta data, 4 KiB pages; leaf
 * entries = 13-byte locational key + payload) keyed by the octant address of every element, payload
 * mdata_t = { int64 nid[8]; float edgesize, Vp, Vs, rho }.  hqh_etree_read returns the leaves in
 * key (= octree pre-) order: ticks [n][3] (lower-left corner), level [n], and the raw payloads
 * [n][*value_size] (malloc'ed, caller frees).  Little-endian files of 3 dimensions only.
 */
HQ_API int hqh_etree_read(const char* path, int64_t* n, int32_t* value_size, uint32_t** ticks, int32_t** level,
                          void** values);

/*
 * An octree mesh from its leaves, in octor's conventions (octor_extractmesh, octor.c:5268-6650),
 * whole or one of nranks partitions: nodes = the distinct element vertices in Z-order of their far-boundary-adjusted
 * coordinates (octor.c:6100-6106, 6166); hanging nodes by node_setproperty's rules (touch count,
 * boundary position, alignment to the next coarser grid, octor.c:3280-3860) with anchors in the
 * order the dnode correlation leaves them (octor.c:6493-6612); eTable / nTable as solver_init
 * leaves them (psolve.c:3280-3510) incl. the hanging-node mass distribution.  Elements must be in
 * octree pre-order (as hqh_etree_read returns them).
 *   elem_ticks [E][3]  lower-left corners, ticks;  elem_edge [E] edge, ticks
 *   edata      [E][4]  edgesize (m), Vp, Vs, rho   (edata_t, psolve.h:95-97)
 *   far_ticks  [3]     domain extent (nodes on the far faces are adjusted inwards for ordering)
 * The result is an hqh_octbox (same accessors: hqh_octbox_desc, hqh_octbox_view, _destroy).
 */
typedef struct {
    double  deltaT, freq;
    int32_t damping;
    double  threshold_damping, threshold_vpvs;
    int32_t halfspace;
    int32_t rank, nranks;        /* cut into octor's per-rank tables as hqh_octbox_create does (0, 0 or 1: whole mesh) */
    int32_t solver_float;        /* as hqh_box_params.solver_float */
} hqh_init_params;

HQ_API int hqh_mesh_from_leaves(int64_t E, const uint32_t* elem_ticks, const uint32_t* elem_edge, const float* edata,
                                const uint32_t far_ticks[3], const hqh_init_params* ip, hqh_octbox** out);

/*
 * The leaves the reference's mesher makes of a LATERALLY varying material model (BASELINE config 5's mesh class):
 * octor_newtree + octor_refinetree (toexpand = vsrule on setrec's 27-sample minimum-Vs record, psolve.c:1307-1397,
 * 2185-2210, quake_util.c:215-225) + octor_balancetree (2:1 across faces and edges, octor.c:4398-4775), for a model
 * given on a regular grid -- what a CVM etree whose leaves share one level is to cvm_query (cvm.c:266-311).  Pinned on
 * the mesh the real reference made of such a model (tests/golden/c5_basin).  Output: E leaves in octree pre-order --
 * elem_ticks [E][3], elem_edge [E] (octor ticks: the root cube is 2^30), edata [E][4] (edgesize, Vp, Vs, rho) --
 * malloc'ed, release with hqh_free; far_ticks[3] the domain's far end point, *ticksize metres per tick.  Feed them to
 * hqh_mesh_from_leaves.
 */
typedef struct {
    int32_t      nx, ny, nz;     /* cells along the mesh's x (north), y (east) and z (depth) */
    double       cell;           /* cell edge, metres */
    const float *vp, *vs, *rho;  /* [nz][ny][nx] */
} hqh_grid_model;

typedef struct {
    double  domain[3];           /* metres: Param.theDomainX / Y / Z (region_length_north / east, depth) */
    double  factor;              /* simulation_wave_max_freq_hz x simulation_node_per_wavelength */
    double  vscut;               /* simulation_shear_velocity_min */
    int32_t max_level;           /* refuse to refine below this octree level (0: 16) */
} hqh_mesher_params;

HQ_API int  hqh_octree_generate(const hqh_grid_model* m, const hqh_mesher_params* p, int64_t* E, uint32_t** elem_ticks,
                                uint32_t** elem_edge, float** edata, uint32_t far_ticks[3], double* ticksize);
HQ_API void hqh_free(void* p);

/*
 * A CVM etree -- the material database the reference's mesher queries (cvmdb_input_file; quake/cvm/cvm.h:72 cvm_query,
 * cvm.c:266-311; payload cvmpayload_t = float Vp, Vs, density; control block cvm_getdbctl cvm.c:62-215) -- read with this
 * library's own etree reader, so that "the same input etree" needs neither the reference's mesher nor its etree library:
 *   hqh_cvm_open   leaves + payloads + the control block (region lengths, domain end point -> metres per tick)
 *   hqh_cvm_query  cvm_query: the payload of the leaf octant that holds (east, north, depth) in metres; -1 outside
 *   hqh_cvm_grid   the database on a regular grid of its finest leaves' size, in the MESH's axes [k][y][x] (setrec queries
 *                  east = mesh y, north = mesh x, psolve.c:1352): the hqh_grid_model of hqh_octree_generate
 * tests/test_host_partition.py: from the very databases oracle/make_cvm wrote for the golden runs, the reference's meshes
 * leaf for leaf; tests/test_gpu_parity.py: from the database to the reference's checkpoints on the GPU.
 */
typedef struct hqh_cvm hqh_cvm;
HQ_API int  hqh_cvm_open(const char* path, hqh_cvm** out);
HQ_API void hqh_cvm_close(hqh_cvm* c);
HQ_API int  hqh_cvm_info(const hqh_cvm* c, int64_t* nleaves, int32_t levels[2], double region_m[3], double* ticksize);
HQ_API int  hqh_cvm_query(const hqh_cvm* c, double east_m, double north_m, double depth_m, float payload[3]);
HQ_API int  hqh_cvm_grid(const hqh_cvm* c, int32_t dims[3], double* cell_m, float** vp, float** vs, float** rho);

/* Fill F[nsteps][nloaded][3] for steps [step0, step0+nsteps) of the ramp source. */
HQ_API void hqh_source_table(const hqh_run_params* rp, double dt, int32_t step0, int32_t nsteps, double* F);

#ifdef __cplusplus
}
#endif
#endif /* HQ_HOST_H */
