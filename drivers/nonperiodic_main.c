/* Command-line driver with the reference's interface (msm/nonperiodic.c:3-29):
 *     nonperiodic datafile [p h a mu]
 * reads the particle file, sets every charge to 1 as the reference does (nonperiodic.c:9-10) and prints the open-boundary
 * energy 1/2 sum_{i != j} q_i q_j / r_ij in the reference's format.  With one argument the sum is the all-pairs sum on
 * the GPU (direct() of libmsm_amd.so, msm/msmLibrary.h:212, in place of the reference's double loop); with the four
 * extra arguments it is the open-boundary MSM (msm3d(), msm/msmLibrary.h:216) of order p, grid spacing h, cutoff a and
 * quasi-interpolation width mu. */
#include <stdio.h>
#include <stdlib.h>
#include "msmLibrary.h"

int main(int argc, char *argv[]) {
    double *r, *q, *f, Lx, Ly, Lz, u = 0.0;
    int N, i;
    if (argc != 2 && argc != 6) {
        printf("Usage: %s datafile [p h a mu]\n", argv[0]);
        return 1;
    }
    if (data_read(argv[1], &q, &r, &N, &Lx, &Ly, &Lz)) {
        fprintf(stderr, "%s\n", msm_amd_last_error());
        return 1;
    }
    for (i = 0; i < N; i++) q[i] = 1.0;
    f = (double *)calloc(3 * (size_t)N, sizeof(double));
    if (argc == 2) {
        direct(N, r, q, &u, f);
    } else {
        int p = atoi(argv[2]);
        msm3d(N, r, q, p, atof(argv[3]), atof(argv[4]), atoi(argv[5]), p, &u, f);
    }
    printf("u is %15.8e\n", u);
    free(f);
    free(q);
    free(r);
    return 0;
}
