"""Per-wave timeline of the list-walking pair kernel on one rank of W (or the whole system, W = 1): when its waves
start and end, how many are resident per SIMD over time, how the work spreads over XCDs, CUs and SIMDs.
python tools/pair_timeline.py W [options]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import msm_amd
from msm_amd import capi
from msm_amd.slab import SlabRank, NullCollectives
from bench import make_system
from rank_sim_common import FakeDist

W = int(sys.argv[1]); N = int(os.environ.get("RANK_SIM_N", "1000000")); p = 6
r, q = make_system(N, 20261113)
desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), p, 10, 4.0, 0)
s = SlabRank(torch, FakeDist(), msm_amd, desc, r, q, W // 2, W, 0, perturb=True,
             collectives=NullCollectives(W // 2, W) if W > 1 else None)
for kv in filter(None, (sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else "").split(",")):
    k, v = kv.split("=")
    s.eng.set_option(k, int(v))
lib = capi.load()
slots = lib.msm_amd_pair_trace_slots(s.eng._h)
for _ in range(3): s.step(want_energies=False)
torch.cuda.synchronize()
buf = torch.zeros((slots, 4), dtype=torch.int64, device="cuda")
lib.msm_amd_set_pair_trace(s.eng._h, C.c_void_p(buf.data_ptr()))
s.step(want_energies=False)
torch.cuda.synchronize()
lib.msm_amd_set_pair_trace(s.eng._h, None)
t = buf.cpu().numpy()
t = t[t[:, 3] > 0]
t0 = t[:, 0].min()
start, end, kind = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01, t[:, 3]   # microseconds
hw = t[:, 2] & 0xFFFFFFFF; xcc = (t[:, 2] >> 32) & 0xF
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
unit = ((xcc * 8 + se) * 2 + sh) * 16 + cu          # one CU
usimd = unit * 4 + simd
work = kind == 3
print("W", W, "slots", slots, "waves traced", len(t), "evaluated", int(work.sum()), "halo-only", int((kind == 2).sum()), "empty", int((kind == 1).sum()))
print("kernel span %.1f us; first evaluated wave starts %.1f, last start %.1f, last end %.1f" % (end.max(), start[work].min(), start[work].max(), end[work].max()))
d = end[work] - start[work]
print("evaluated wave lifetime us: mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % (d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
dn = end[~work] - start[~work]
print("other waves lifetime us: mean %.2f max %.2f" % (dn.mean(), dn.max()))
print("distinct CUs", len(np.unique(unit)), "distinct SIMDs", len(np.unique(usimd)))
# resident evaluated waves per SIMD over time (chip average) and busy SIMDs
T = np.arange(0, end.max(), 5.0)
for x in T:
    live = work & (start <= x) & (end > x)
    per = np.bincount(usimd[live], minlength=usimd.max() + 1)
    print("  t=%6.1f us: resident evaluated waves %6d, SIMDs with >=1: %4d, with >=2: %4d, mean per busy SIMD %.2f" %
          (x, live.sum(), (per > 0).sum(), (per > 1).sum(), per[per > 0].mean() if (per > 0).any() else 0))
mid = 0.5 * end.max()
live = work & (start <= mid) & (end > mid)
print("at t=%.0f us: evaluated waves resident per CU, histogram (waves: CUs):" % mid,
      dict(zip(*np.unique(np.bincount(unit[live], minlength=unit.max() + 1), return_counts=True))))
print("   per SIMD (waves: SIMDs):", dict(zip(*np.unique(np.bincount(usimd[live], minlength=usimd.max() + 1), return_counts=True))))
print("   per XCD:", np.bincount(xcc[live]).tolist())
# gap between the end of a wave and the start of the next one on the same SIMD slot is not visible; the start rate is:
st = np.sort(start[work]); k = len(st) // 2
print("   evaluated-wave starts per us around the middle: %.1f" % (1000.0 / (st[k + 500] - st[k - 500])))
# work per XCD and per CU: evaluated wave-microseconds
for name, key in (("XCD", xcc), ("CU", unit), ("SIMD", usimd)):
    cnt = np.bincount(key[work]); cnt = cnt[cnt > 0]
    last = np.array([end[work & (key == k)].max() for k in np.unique(key[work])])
    print("%s: evaluated waves per unit min %d mean %.1f max %d; last end per unit: min %.1f mean %.1f max %.1f" %
          (name, cnt.min(), cnt.mean(), cnt.max(), last.min(), last.mean(), last.max()))
