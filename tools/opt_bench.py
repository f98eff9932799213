"""A/B timing of engine options on the benchmark system: python tools/opt_bench.py key=value[,key=value] ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import msm_amd
from bench import make_system
N = 1000000
p = int(os.environ.get("ORDER", "6"))
r, q = make_system(N, 20261113)
desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), p, 10, 4.0, 0)
eng = msm_amd.Engine(desc)
rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
acc = torch.empty_like(rd)
for spec in sys.argv[1:] or [""]:
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    for _ in range(3):
        eng.eval(rd, qd, acc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.eval(rd, qd, acc)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    eng.set_profiling(True)
    eng.eval(rd, qd, acc)
    t = dict(eng.last_timings())
    try:
        t["pair_kernel"] = eng.last_kernel_ms("pair_kernel")
    except Exception:
        pass
    eng.set_profiling(False)
    print(spec or "(defaults)", "ms/eval %.3f" % ms, {k: round(float(v), 3) for k, v in t.items()})
