#include <stdio.h>
int msmref_guarded_fclose(FILE *fp) { return fp ? fclose(fp) : 0; }
