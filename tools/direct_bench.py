"""Throughput of the open-boundary direct sum (msm_amd_direct_device) on one GPU."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import msm_amd
lib = msm_amd.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rng = np.random.default_rng(3)
r = torch.as_tensor(rng.random((N, 3)), device="cuda:0")
q = torch.as_tensor(np.where(np.arange(N) % 2 == 0, 1.0, -1.0), device="cuda:0")
acc = torch.zeros((N, 3), dtype=torch.float64, device="cuda:0")
work = torch.zeros(lib.msm_amd_direct_work_doubles(N), dtype=torch.float64, device="cuda:0")
u = C.c_double()
def run():
    assert lib.msm_amd_direct_device(N, r.data_ptr(), q.data_ptr(), 0, N, 0, acc.data_ptr(), work.data_ptr(), None, C.byref(u)) == 0
run()
t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
pairs = float(N) * N
print("N", N, "seconds", round(dt, 4), "Gpairs/s", round(pairs / dt / 1e9, 1), "TFLOP/s at 20 flop/pair", round(pairs * 20 / dt / 1e12, 2), "u", u.value)
