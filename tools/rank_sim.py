"""Time the work of ONE rank of a W-rank slab decomposition on one GPU (collectives replaced by no-ops)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import msm_amd
from msm_amd.slab import SlabRank, NullCollectives
from bench import make_system, SlabProfiler

from rank_sim_common import FakeDist

N = int(os.environ.get("RANK_SIM_N", "1000000")); p = int(sys.argv[2]) if len(sys.argv) > 2 else 6   # RANK_SIM_N: particles of the whole system
W = int(sys.argv[1])
RANK = int(os.environ.get("RANK_SIM_RANK", str(W // 2)))   # which rank's work (rank 0 and the last one own planes that wrap around the period)
r, q = make_system(N, 20261113)
desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), p, 10, 4.0, 0)
# argv[4]: exchange mode (0 auto, 1 all-gather, 2 halo); argv[5] == "py": the stage-by-stage Python step instead of the C step
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
c_step = not (len(sys.argv) > 5 and sys.argv[5] == "py")
# RANK_SIM_STREAM=default: the engine on the legacy default stream (the step then moves to streams of its own and back);
# otherwise on a stream of the caller's, as a time-stepping code that owns its stream would create it
side = None if os.environ.get("RANK_SIM_STREAM", "side") == "default" else torch.cuda.Stream()
if side is not None:
    torch.cuda.set_stream(side)
# RANK_SIM_DELAY="halo,gather,reduce" (microseconds): the collectives hold their stream that long instead of returning at once
delay = [float(x) for x in os.environ.get("RANK_SIM_DELAY", "").split(",") if x]
if W > 1 and delay:
    from rank_sim_common import delay_collectives
    coll = delay_collectives(torch, RANK, W, *delay)
else:
    coll = NullCollectives(RANK, W) if W > 1 else None
s = SlabRank(torch, FakeDist(), msm_amd, desc, r, q, RANK, W, 0, perturb=True, mode=mode, collectives=coll, c_step=c_step, stream=side)
for kv in filter(None, (sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else "").split(",")):   # engine options: key=value,key=value
    k, v = kv.split("=")
    s.eng.set_option(k, int(v))
# 0.1 s of the same steps first: the GPU takes about 30 ms of load to reach its clocks (a rank of 8 timed over 20 steps
# behind 3 warm-up steps never left that ramp)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1:
    for _ in range(10): s.step(want_energies=False)
    torch.cuda.synchronize()
NT = 100
t0 = time.perf_counter()
for _ in range(NT): s.step(want_energies=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / NT
prof = SlabProfiler(torch, s).profile(5)
print("N", N, "M", list(desc.M), "W", W, "rank", RANK, "p", p, "mode", s.eng.slab_exchange_mode(W, mode) if W > 1 else "-", "c_step", s.coll is not None, "n_own", s.n_own, "n_tot", s.n_tot, "ms/step", round(dt * 1e3, 3), {k: round(v, 3) for k, v in prof.items()})
