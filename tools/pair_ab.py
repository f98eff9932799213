"""A/B of pair-kernel variants on the benchmark system: results against the default variant, pair-kernel time.
python tools/pair_ab.py [order] variant [variant ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import msm_amd
from bench import make_system
N = 1000000
p = int(sys.argv[1])
r, q = make_system(N, 20261113)
desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), p, 10, 4.0, 0)
eng = msm_amd.Engine(desc)
rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
acc = torch.empty_like(rd)
ref = None
for v in [int(x) for x in sys.argv[2:]]:
    eng.set_option("pair_variant", v)
    for _ in range(3):
        en, _ = eng.eval(rd, qd, acc)
    a = acc.cpu().numpy().copy()
    eng.set_profiling(True)
    ms = []
    for _ in range(10):
        eng.eval(rd, qd, acc)
        ms.append(eng.last_kernel_ms("pair_kernel"))
    eng.set_profiling(False)
    if ref is None:
        ref = (en.copy(), a)
    d = np.sqrt(((a - ref[1]) ** 2).sum() / (ref[1] ** 2).sum())
    print("variant", v, "pair kernel ms min %.4f mean %.4f" % (min(ms), sum(ms) / len(ms)), "acc rel rms diff vs first %.2e" % d,
          "energy diff %.2e" % (np.abs(en - ref[0]).max() / np.abs(ref[0][:6]).max()), flush=True)
