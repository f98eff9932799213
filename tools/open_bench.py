"""Open-boundary MSM (msm3d path) on N random charges in the unit cube: time per evaluation, stage times, and the error
against the direct sum of the same system on the GPU.  python tools/open_bench.py [N] [p] [abar] [L]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import msm_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 6
abar = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
L = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = np.random.default_rng(20261114)
r = rng.random((N, 3))
q = np.where(np.arange(N) % 2 == 0, 1.0, -1.0)
ht = (1.0 / N) ** (1.0 / 3.0)
h = 1.0 / 96 if N == 1000000 else ht   # 1M: the spacing of the periodic benchmark
desc = msm_amd.plan_open(N, r.min(axis=0), r.max(axis=0), p, 10, h, abar * ht, L)
t0 = time.perf_counter()
eng = msm_amd.Engine(desc)
t_create = time.perf_counter() - t0
rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
acc = torch.empty_like(rd)
for _ in range(3):
    en, _ = eng.eval(rd, qd, acc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.eval(rd, qd, acc, want_energies=False)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
eng.set_profiling(True)
eng.eval(rd, qd, acc)
stages = {k: round(float(v), 3) for k, v in eng.last_timings().items()}
eng.set_profiling(False)
print("N=%d p=%d abar=%g: grid %s, L=%d, cell %.4f; create %.1f ms; %.3f ms/eval; stages %s" %
      (N, p, abar, list(desc.M), desc.L, desc.A[0], 1e3 * t_create, ms, stages), flush=True)
t0 = time.perf_counter()
u, f = msm_amd.direct(r, q)
td = time.perf_counter() - t0
a = acc.cpu().numpy()
print("direct sum %.3f s (host arrays); energy msm %.9e direct %.9e rel %.2e; force rel rms %.2e" %
      (td, en[6], u, abs(en[6] - u) / abs(u), np.sqrt(((a - f) ** 2).sum() / (f ** 2).sum())), flush=True)
