/* Command-line driver with the reference's interface (msm/msmPeriodic.c:1423-1456):
 *     msmPeriodic datafile p mu abar
 * reads the particle file, runs msmPeriodicRunner() (GPU, through libmsm_amd.so),
 * prints the potential and writes msm.pot / msm.acc in the reference's formats. */
#include <stdio.h>
#include <stdlib.h>
#include "msmLibrary.h"

int main(int argc, char *argv[]) {
    double *r, *q, *f, Lx, Ly, Lz, u;
    int n, i, d;
    FILE *fp;
    if (argc != 5) {
        printf("Usage: %s datafile p mu abar\n", argv[0]);
        return 1;
    }
    if (data_read(argv[1], &q, &r, &n, &Lx, &Ly, &Lz)) {
        fprintf(stderr, "%s\n", msm_amd_last_error());
        return 1;
    }
    f = (double *)calloc((size_t)n * 3, sizeof(double));
    u = msmPeriodicRunner(r, q, f, n, atoi(argv[2]), atoi(argv[3]), atof(argv[4]), Lx, Ly, Lz);
    printf("%20.15e\n", u);
    fp = fopen("msm.pot", "w");
    fprintf(fp, "%20.15e\n", u);
    fclose(fp);
    fp = fopen("msm.acc", "w");
    fprintf(fp, "%d\n", n);
    for (d = 0; d < 3; d++)
        for (i = 0; i < n; i++) fprintf(fp, "%20.14e\n", f[3 * i + d]);
    fclose(fp);
    free(r); free(q); free(f);
    return 0;
}
