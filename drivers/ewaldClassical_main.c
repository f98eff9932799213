/* Command-line driver with the reference's interface (msm/ewaldClassical.c:125-133):
 *     ewaldClassical datafile alpha cutoffDirect cutoffFourier
 * reads the particle file and prints the Ewald energy computed on the GPU (ewaldClassicalRunner() of
 * libmsm_amd.so). */
#include <stdio.h>
#include <stdlib.h>
#include "ewaldClassical.h"

int main(int argc, char *argv[]) {
    if (argc != 5) {
        printf("Usage: %s datafile alpha cutoffDirect cutoffFourier\n", argv[0]);
        return 1;
    }
    ewaldClassicalRunner(argv[1], atof(argv[2]), atof(argv[3]), atof(argv[4]));
    return 0;
}
