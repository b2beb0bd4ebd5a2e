/*
 * make_cvm.c -- TEST INFRASTRUCTURE (golden-fixture generation only).
 *
 * Writes a small layered material database in the reference's own etree/CVM
 * format, USING THE REFERENCE'S LIBRARIES (etree/, quake/cvm/cvm.c, compiled in
 * place by oracle/build_ref.sh), so that the real `psolve` can be run on a model
 * whose Vs contrast makes its mesher refine the top layer one level deeper --
 * i.e. a mesh with hanging nodes (SURVEY.md s8 a13, config 5 in miniature).
 *
 * Same region and schema as examples/simple/simple_case.e (1000 x 1000 x 500 m,
 * "float Vp; float Vs; float density;", 2048 level-4 octants, etree root 2^31
 * ticks); only the material differs: octant layers 0..nsoft-1 (62.5 m each) are
 * soft.
 *
 * usage: make_cvm out.e nsoft Vp_soft Vs_soft rho_soft Vp Vs rho
 *    or: make_cvm out.e layers n  k0 Vp Vs rho  k1 Vp Vs rho ...   (layer i starts at octant layer k_i)
 */
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cvm.h"
#include "etree.h"

static unsigned compact3(unsigned long long v, int shift)
{
    unsigned r = 0;
    for (int b = 0; b < 10; b++) r |= (unsigned)((v >> (3 * b + shift)) & 1ULL) << b;
    return r;
}

int main(int argc, char** argv)
{
    int nlay = 0, kstart[8];
    cvmpayload_t lay[8];
    if (argc >= 4 && strcmp(argv[2], "layers") == 0) {
        nlay = atoi(argv[3]);
        if (nlay < 1 || nlay > 8 || argc != 4 + 4 * nlay) { fprintf(stderr, "bad layer list\n"); return 2; }
        for (int l = 0; l < nlay; l++) {
            kstart[l] = atoi(argv[4 + 4 * l]);
            lay[l].Vp = (float)atof(argv[5 + 4 * l]); lay[l].Vs = (float)atof(argv[6 + 4 * l]);
            lay[l].rho = (float)atof(argv[7 + 4 * l]);
        }
    } else if (argc == 9) {
        nlay = 2;
        kstart[0] = 0; kstart[1] = atoi(argv[2]);
        lay[0].Vp = (float)atof(argv[3]); lay[0].Vs = (float)atof(argv[4]); lay[0].rho = (float)atof(argv[5]);
        lay[1].Vp = (float)atof(argv[6]); lay[1].Vs = (float)atof(argv[7]); lay[1].rho = (float)atof(argv[8]);
    } else {
        fprintf(stderr, "usage: %s out.e nsoft Vp_s Vs_s rho_s Vp Vs rho | out.e layers n k Vp Vs rho ...\n", argv[0]);
        return 2;
    }
    const int level = 4, nx = 16, ny = 16, nz = 8;
    const etree_tick_t edge = (etree_tick_t)1 << (31 - level);

    etree_t* ep = etree_open(argv[1], O_CREAT | O_TRUNC | O_RDWR, 0, sizeof(cvmpayload_t), 3);
    if (!ep) { fprintf(stderr, "etree_open failed\n"); return 1; }
    if (etree_registerschema(ep, "float Vp; float Vs; float density;") != 0) {
        fprintf(stderr, "%s\n", etree_strerror(etree_errno(ep))); return 1;
    }
    if (etree_beginappend(ep, 1.0) != 0) { fprintf(stderr, "%s\n", etree_strerror(etree_errno(ep))); return 1; }
    /* octants in locational-code (Z) order: z is the most significant axis of a triplet */
    for (unsigned long long code = 0; code < (1ULL << (3 * level)); code++) {
        unsigned i = compact3(code, 0), j = compact3(code, 1), k = compact3(code, 2);
        if (i >= (unsigned)nx || j >= (unsigned)ny || k >= (unsigned)nz) continue;
        etree_addr_t a;
        memset(&a, 0, sizeof a);
        a.x = i * edge; a.y = j * edge; a.z = k * edge;
        a.level = level;
        a.type = ETREE_LEAF;
        int L = 0;
        for (int l = 0; l < nlay; l++) if ((int)k >= kstart[l]) L = l;
        if (etree_append(ep, a, &lay[L]) != 0) {
            fprintf(stderr, "append: %s\n", etree_strerror(etree_errno(ep))); return 1;
        }
    }
    if (etree_endappend(ep) != 0) { fprintf(stderr, "%s\n", etree_strerror(etree_errno(ep))); return 1; }

    dbctl_t* ctl = cvm_newdbctl();
    ctl->create_model_name = strdup("Title:TWOLAYER");
    ctl->create_author = strdup("Author:hq-oracle");
    ctl->create_date = strdup("Date:generated");
    ctl->create_field_count = strdup("3");
    ctl->create_field_names = strdup("Vp(float);Vs(float);density(float)");
    ctl->region_origin_latitude_deg = 0; ctl->region_origin_longitude_deg = 0;
    ctl->region_length_east_m = 1000; ctl->region_length_north_m = 1000;
    ctl->region_depth_shallow_m = 0; ctl->region_depth_deep_m = 500;
    ctl->domain_endpoint_x = nx * edge; ctl->domain_endpoint_y = ny * edge; ctl->domain_endpoint_z = nz * edge;
    if (cvm_setdbctl(ep, ctl) != 0) return 1;
    if (etree_close(ep) != 0) return 1;
    return 0;
}
