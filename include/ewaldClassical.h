/* Forwarder: lets a driver written against the reference's msm/ewaldClassical.h
 * (`#include "ewaldClassical.h"`) compile unchanged against libmsm_amd.so. */
#ifndef ewaldClassical_h
#define ewaldClassical_h
#include <stdio.h>
#include "msm_amd.h"
#endif
