// (round 5) what does first-touching a large host array cost on the GPU box's 256 cores, and does MADV_HUGEPAGE help?
// hq_create's host phases allocate and fill ~1 GB per 8 M elements; a rank's hq_create takes 4-5 s there and <2 s on 8 cores.
//   gcc -O2 -fopenmp -o page_faults page_faults.c && ./page_faults
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <omp.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(void)
{
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char buf[128] = "?";
    if (f) { if (!fgets(buf, sizeof buf, f)) buf[0] = 0; fclose(f); }
    printf("transparent_hugepage/enabled: %s", buf);
    const size_t bytes = (size_t)1 << 30;
    for (int huge = 0; huge < 2; huge++)
        for (int nth = 1; nth <= 256; nth *= 16) {
            char* p = (char*)malloc(bytes + (2 << 20));
            char* a = (char*)(((size_t)p + (2 << 20) - 1) & ~(size_t)((2 << 20) - 1));
            if (huge) madvise(a, bytes, MADV_HUGEPAGE);
            double t0 = now();
#pragma omp parallel for num_threads(nth) schedule(static)
            for (long i = 0; i < (long)(bytes / 4096); i++) a[(size_t)i * 4096] = 1;
            double t1 = now();
            printf("madvise(HUGEPAGE) %d  threads %3d  first touch of 1 GiB: %.3f s\n", huge, nth, t1 - t0);
            free(p);
        }
    return 0;
}
