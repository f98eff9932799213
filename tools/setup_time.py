"""Cost of building a plan (position-independent setup): host stencil construction alone, and the whole msm_amd_create
with the stencils built on the host and on the GPU."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import msm_amd
lib = msm_amd.load()
torch.zeros(1, device="cuda:0")
for (N, p, abar) in ((1000000, 6, 4.0), (1000000, 4, 4.0), (1000000, 6, 6.0), (100000, 4, 4.0), (10000000, 6, 4.0)):
    desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), p, 10, abar, 0)
    t0 = time.perf_counter()
    hp = ctypes.c_void_p()
    lib.msm_amd_hostplan_create(ctypes.byref(desc), 0, ctypes.byref(hp))
    t1 = time.perf_counter()
    lib.msm_amd_hostplan_destroy(hp)
    res = {}
    for name, where in (("host", 0), ("gpu", 1)):
        lib.msm_amd_set_stencil_backend(where)
        ts = []
        for _ in range(4):
            t2 = time.perf_counter()
            eng = msm_amd.Engine(desc)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t2)
            eng.close()
        res[name] = ts
    print("N=%d p=%d abar=%g L=%d M=%d: host stencils alone %.1f ms; msm_amd_create with host stencils %s ms, with GPU stencils %s ms" %
          (N, p, abar, desc.L, desc.M[0], 1e3 * (t1 - t0), ["%.1f" % (1e3 * t) for t in res["host"]],
           ["%.1f" % (1e3 * t) for t in res["gpu"]]), flush=True)
