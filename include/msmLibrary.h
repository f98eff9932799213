/* Forwarder: lets a driver written against the reference's msm/msmLibrary.h
 * (`#include "msmLibrary.h"`) compile unchanged against libmsm_amd.so. */
#ifndef msmLibrary_h
#define msmLibrary_h
#include <stdlib.h>
#include <math.h>
#include "msm_amd.h"
#endif
