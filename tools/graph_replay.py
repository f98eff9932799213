"""Does replaying one evaluation from a HIP graph beat queueing its launches?  (Engine on its own stream; the same
positions every time.)  python tools/graph_replay.py [order]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import msm_amd
from bench import make_system
N = 1000000
p = int(sys.argv[1]) if len(sys.argv) > 1 else 6
r, q = make_system(N, 20261113)
desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), p, 10, 4.0, 0)
st = torch.cuda.Stream()
eng = msm_amd.Engine(desc, stream=st.cuda_stream)
rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
acc = torch.empty_like(rd)
torch.cuda.synchronize()
with torch.cuda.stream(st):
    for _ in range(3):
        eng.eval(rd, qd, acc, want_energies=False)
st.synchronize()
ref = acc.clone()
t0 = time.perf_counter()
for _ in range(100):
    eng.eval(rd, qd, acc, want_energies=False)
st.synchronize()
eager = (time.perf_counter() - t0) / 100 * 1e3
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    eng.eval(rd, qd, acc, want_energies=False)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 100 * 1e3
print("order", p, "eager ms/eval %.4f   graph replay ms/eval %.4f   same bits: %s" % (eager, graph, bool(torch.equal(ref, acc))))
