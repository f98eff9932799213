"""Stage timings of one configuration: python tools/config_timings.py N p abar [fine] (fine=2: half the grid spacing)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import msm_amd
from bench import make_system

N, p, abar = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
fine = int(sys.argv[4]) if len(sys.argv) > 4 else 1
A = (1.0, 1.0, 1.0)
r, q = make_system(N, 20261111 + 4)
lib = msm_amd.load()
d = msm_amd.plan_auto(N, A, p, 10, abar, 0)
if fine == 2:
    M = [2 * m for m in d.M]
    L = d.L + 1
    hL = (2.0 ** (L - 1)) * A[0] / M[0]
    beta = lib.choosebetaNew(1e-12, (2.0 ** L) * d.a, hL)
    kmax = int(np.ceil(lib.choosekmax(1e-12, beta, hL)))
    d = msm_amd.plan_explicit(N, A, M, p, 10, L, d.a, beta, kmax)
eng = msm_amd.Engine(d)
rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
acc = torch.empty_like(rd)
for _ in range(2):
    eng.eval(rd, qd, acc)
eng.set_profiling(True)
eng.eval(rd, qd, acc)
t = eng.last_timings()
print("N=%d p=%d abar=%g M=%s L=%d a=%.4f" % (N, p, abar, list(d.M), d.L, d.a), {k: round(float(v), 3) for k, v in t.items()},
      "pair_kernel", round(eng.last_kernel_ms("pair_kernel"), 3), flush=True)
