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
 *    or: make_cvm out.e regions L  Vp Vs rho  n  <region> ...
 *        a LATERALLY varying model ("basin"): database octants of level L (2^L x 2^L x 2^(L-1) of them); the
 *        background material, then n regions, later ones overriding earlier ones:
 *          box i0 i1 j0 j1 k0 k1 Vp Vs rho     octants with i0 <= i < i1, j0 <= j < j1, k0 <= k < k1
 *          dip a b c Vp Vs rho                 octants whose centre lies above the plane
 *                                              k + 0.5 < a + b (i + 0.5) + c (j + 0.5)     (a sediment wedge)
 *          grad gx gy gz                       (no material) every octant's Vp and Vs times
 *                                              f = 1 + gx (i + 0.5) / nx + gy (j + 0.5) / ny + gz (k + 0.5) / nz, its density
 *                                              times 1 + (f - 1) / 2, evaluated in double and rounded to float: a model
 *                                              whose material differs from octant to octant (a velocity gradient)
 *        The reference's Vs rule then refines wherever the soft material is (psolve.c:1308 setrec, :2185 toexpand,
 *        quake_util.c:215 vsrule), so refinement interfaces get x-, y- and z-normal faces and staircase corners.
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
    int level = 4;
    enum { MAXREG = 16 };
    int nreg = -1, reg_kind[MAXREG], reg_box[MAXREG][6];
    double reg_dip[MAXREG][3];
    cvmpayload_t reg_mat[MAXREG], background;
    if (argc >= 8 && strcmp(argv[2], "regions") == 0) {
        level = atoi(argv[3]);
        if (level < 3 || level > 7) { fprintf(stderr, "bad level\n"); return 2; }
        background.Vp = (float)atof(argv[4]); background.Vs = (float)atof(argv[5]); background.rho = (float)atof(argv[6]);
        nreg = atoi(argv[7]);
        if (nreg < 0 || nreg > MAXREG) { fprintf(stderr, "bad region count\n"); return 2; }
        int a = 8;
        for (int r = 0; r < nreg; r++) {
            if (a < argc && strcmp(argv[a], "box") == 0 && a + 10 <= argc) {
                reg_kind[r] = 0;
                for (int q = 0; q < 6; q++) reg_box[r][q] = atoi(argv[a + 1 + q]);
                a += 7;
            } else if (a < argc && strcmp(argv[a], "dip") == 0 && a + 7 <= argc) {
                reg_kind[r] = 1;
                for (int q = 0; q < 3; q++) reg_dip[r][q] = atof(argv[a + 1 + q]);
                a += 4;
            } else if (a < argc && strcmp(argv[a], "grad") == 0 && a + 4 <= argc) {
                reg_kind[r] = 2;
                for (int q = 0; q < 3; q++) reg_dip[r][q] = atof(argv[a + 1 + q]);
                a += 4;
                continue;
            } else { fprintf(stderr, "bad region %d\n", r); return 2; }
            reg_mat[r].Vp = (float)atof(argv[a]); reg_mat[r].Vs = (float)atof(argv[a + 1]); reg_mat[r].rho = (float)atof(argv[a + 2]);
            a += 3;
        }
        if (a != argc) { fprintf(stderr, "trailing arguments\n"); return 2; }
    } else if (argc >= 4 && strcmp(argv[2], "layers") == 0) {
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
    const int nx = 1 << level, ny = 1 << level, nz = 1 << (level - 1);
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
        const cvmpayload_t* mat = &lay[L];
        cvmpayload_t graded;
        double grade = 1.0;
        if (nreg >= 0) {
            mat = &background;
            for (int r = 0; r < nreg; r++) {
                int in;
                if (reg_kind[r] == 2) {
                    grade = 1.0 + reg_dip[r][0] * (i + 0.5) / nx + reg_dip[r][1] * (j + 0.5) / ny + reg_dip[r][2] * (k + 0.5) / nz;
                    continue;
                }
                if (reg_kind[r] == 0)
                    in = (int)i >= reg_box[r][0] && (int)i < reg_box[r][1] && (int)j >= reg_box[r][2] && (int)j < reg_box[r][3] &&
                         (int)k >= reg_box[r][4] && (int)k < reg_box[r][5];
                else
                    in = k + 0.5 < reg_dip[r][0] + reg_dip[r][1] * (i + 0.5) + reg_dip[r][2] * (j + 0.5);
                if (in) mat = &reg_mat[r];
            }
            if (grade != 1.0) {
                graded.Vp = (float)((double)mat->Vp * grade);
                graded.Vs = (float)((double)mat->Vs * grade);
                graded.rho = (float)((double)mat->rho * (1.0 + (grade - 1.0) / 2.0));
                mat = &graded;
            }
        }
        if (etree_append(ep, a, mat) != 0) {
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
