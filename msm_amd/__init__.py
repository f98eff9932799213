"""msm_amd — MI355X-native multilevel summation (MSM) electrostatics.

The compute path is libmsm_amd.so (hand-written gfx950 kernels behind a C ABI,
see include/msm_amd.h).  This package only binds it.
"""
from . import capi
from .capi import MsmAmdError, load
from .engine import Engine, HostPlan, direct, ewald, ewaldClassical, msm3d, msmperiodic, plan_auto, plan_explicit, plan_open, plan_open_holds

__all__ = ["capi", "MsmAmdError", "load", "Engine", "HostPlan", "direct", "ewald", "ewaldClassical", "msm3d", "msmperiodic", "plan_auto", "plan_explicit", "plan_open", "plan_open_holds"]
