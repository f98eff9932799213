#include <stdio.h>
#undef fclose /* this file is compiled with -include ref_fclose_guard.h too */
int msmref_guarded_fclose(FILE *fp) { return fp ? fclose(fp) : 0; }
