"""Host-pointer rates at N = 1M: msm_amd_eval_host (handle kept, 32 B/particle up and 24 B/particle down over PCIe
per evaluation) and the drop-in msmperiodic() (creates and destroys its handle, stencils included, on every call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msm_amd
from bench import make_system
N, p = 1000000, int(os.environ.get("ORDER", "6"))
r, q = make_system(N, 20261113)
desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), p, 10, 4.0, 0)
eng = msm_amd.Engine(desc)
eng.eval_host(r, q)
t0 = time.perf_counter()
for _ in range(5):
    en, acc = eng.eval_host(r, q)
dt = (time.perf_counter() - t0) / 5
print("eval_host: %.2f ms per evaluation = %.1f evals/s (p=%d)" % (dt * 1e3, 1 / dt, p))
eng.close()
if p == 4:   # the reference's msmperiodic() accepts p = 4 only
    for what in ("first call (builds its plan)", "next call (plan kept)", "next call (plan kept)"):
        t0 = time.perf_counter()
        st, u, f, _ = msm_amd.msmperiodic(r.copy(), q, 4, 4.0, 10, (1.0, 1.0, 1.0), mode=msm_amd.capi.MODE_SCALABLE)
        dt = time.perf_counter() - t0
        print("msmperiodic() %s: %.1f ms, utotal %.6f vs handle %.6f" % (what, dt * 1e3, u, en[6]))
