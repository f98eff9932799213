/* Build-time guard used only when compiling the reference's msmSimple.c into
 * oracle/_ref/: fclose(NULL) (msmSimple.c:1032, :1209) is undefined behaviour
 * and crashes on glibc.  Test infrastructure, not product code. */
#include <stdio.h>
int msmref_guarded_fclose(FILE *fp);
#define fclose msmref_guarded_fclose
