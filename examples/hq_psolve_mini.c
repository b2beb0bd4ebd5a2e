/*
 * hq_psolve_mini.c -- a complete host program in C on the two C-ABI libraries
 * (include/hq_host.h + include/hq_solver.h), the shape of the reference's main()
 * (psolve.c:7335-7568) for a uniformly refined homogeneous box:
 *
 *   mesh + solver_init     hqh_box_create
 *   source_init            hqh_point_source        (double-couple point source)
 *   output_stations_init   hqh_stations
 *   solver_init (device)   hq_create
 *   solver_run             hqh_solver_run          (stations every `rate` steps)
 *   checkpoint_write       hqh_checkpoint_write    (reference file layout)
 *
 * usage: hq_psolve_mini nx ny nz h dt freq nsteps outdir
 * writes outdir/station.<i> (reference text format) and outdir/checkpoint.out0.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hq_host.h"

#define NSTATIONS 3

typedef struct {
    FILE* fp[NSTATIONS];
    double dt;
} station_files;

static void print_stations(void* user, int32_t step, int32_t n, const double* disp)
{
    station_files* sf = (station_files*)user;
    char line[128];
    for (int32_t s = 0; s < n; s++) {
        hqh_station_format(line, sizeof line, step * sf->dt, disp + 3 * s);
        fputs(line, sf->fp[s]);
    }
}

int main(int argc, char** argv)
{
    if (argc != 9) {
        fprintf(stderr, "usage: %s nx ny nz h dt freq nsteps outdir\n", argv[0]);
        return 2;
    }
    hqh_box_params p;
    memset(&p, 0, sizeof p);
    p.nx = atoi(argv[1]); p.ny = atoi(argv[2]); p.nz = atoi(argv[3]);
    p.h = atof(argv[4]); p.deltaT = atof(argv[5]); p.freq = atof(argv[6]);
    int32_t nsteps = atoi(argv[7]);
    const char* outdir = argv[8];
    double ztop = 0.0;
    float vp = 6000.0f, vs = 3464.0f, rho = 2700.0f;
    p.nlayers = 1; p.layer_ztop = &ztop; p.layer_vp = &vp; p.layer_vs = &vs; p.layer_rho = &rho;
    p.damping = HQH_DAMP_RAYLEIGH; p.threshold_damping = 0.05; p.threshold_vpvs = 3.0;
    p.halfspace = 1; p.rank = 0; p.nranks = 1;

    hqh_box* box = NULL;
    if (hqh_box_create(&p, &box) != HQ_OK) { fprintf(stderr, "hqh_box_create failed\n"); return 1; }
    hqh_box_info info;
    hqh_box_get_info(box, &info);
    printf("Total elements: %lld\nTotal nodes: %lld\n", (long long)info.total_elements, (long long)info.total_nodes);

    double L = p.nx * p.h, Lz = p.nz * p.h;
    int32_t nloaded = 0, loaded[8];
    double pattern[24];
    if (hqh_point_source(box, L / 2, L / 2, Lz / 5, 0.0, 90.0, 0.0, &nloaded, loaded, pattern) != HQ_OK) return 1;

    double xyz[NSTATIONS * 3] = { L / 2, L / 2, 0.0, 0.6 * L, 0.6 * L, 0.0, 0.75 * L, 0.75 * L, Lz / 4 };
    int32_t ids[NSTATIONS * 8], mine[NSTATIONS];
    double phi[NSTATIONS * 8];
    if (hqh_stations(box, NSTATIONS, xyz, ids, phi, mine) != HQ_OK) return 1;

    hq_desc d;
    hqh_box_desc(box, &d);
    hq_ctx* ctx = NULL;
    if (hq_abi_version() != HQ_ABI_VERSION) { fprintf(stderr, "libhq_solver.so has another ABI version than this program\n"); return 1; }
    if (hq_create(&d, 0, &ctx) != HQ_OK) { fprintf(stderr, "hq_create: %s\n", hq_last_error()); return 1; }

    station_files sf;
    sf.dt = p.deltaT;
    char path[512];
    for (int s = 0; s < NSTATIONS; s++) {
        snprintf(path, sizeof path, "%s/station.%d", outdir, s);
        sf.fp[s] = fopen(path, "w");
        if (!sf.fp[s]) { fprintf(stderr, "cannot open %s\n", path); return 1; }
        fputs("#  Time(s)         X|(m)         Y-(m)         Z.(m)", sf.fp[s]);   /* psolve.c:6636 */
    }
    hqh_run_params rp;
    memset(&rp, 0, sizeof rp);
    rp.nloaded = nloaded; rp.loaded_lnid = loaded; rp.pattern = pattern;
    rp.moment = 1e15; rp.rise_time = 40 * p.deltaT; rp.source_window = 128;
    rp.nstations = NSTATIONS; rp.station_ids = ids; rp.station_phi = phi;
    rp.station_rate = 10; rp.station_fn = print_stations; rp.station_user = &sf;

    int rc = hqh_solver_run(ctx, box, &rp, 0, nsteps);
    if (rc != HQ_OK) { fprintf(stderr, "hqh_solver_run: %d %s\n", rc, hq_last_error()); return 1; }
    for (int s = 0; s < NSTATIONS; s++) fclose(sf.fp[s]);

    snprintf(path, sizeof path, "%s/checkpoint.out0", outdir);
    rc = hqh_checkpoint_write(ctx, path, nsteps, 0, 1, info.nharbored, info.nharbored);
    if (rc != HQ_OK) { fprintf(stderr, "checkpoint_write failed\n"); return 1; }
    hq_info hi;
    hq_get_info(ctx, &hi);
    printf("steps run: %d  kernel variant: %d  device bytes: %lld\n", hi.step, hi.variant, (long long)hi.device_bytes);
    /* what the run moved over PCIe: source windows in, station rows out at their cadence, one checkpoint at the end */
    printf("PCIe host->device: %lld bytes, device->host: %lld bytes, of which the final checkpoint %lld; per step between "
           "outputs: %.1f bytes\n", (long long)hi.pcie_h2d_bytes, (long long)hi.pcie_d2h_bytes,
           (long long)(48LL * info.nharbored),
           nsteps > 0 ? (double)(hi.pcie_h2d_bytes + hi.pcie_d2h_bytes - 48LL * info.nharbored) / nsteps : 0.0);
    hq_destroy(ctx);
    hqh_box_destroy(box);
    return 0;
}
