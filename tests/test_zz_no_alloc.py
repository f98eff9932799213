"""The allocation-free direct sum (msm_amd_direct_device_noalloc) against msm_amd_direct_device: same kernels, so the
accelerations and the slice energy must be bitwise equal; two calls in a row, the second on a smaller input, so that
what the first call left in the library's built-in scratch array cannot leak into the second."""
import ctypes as C

import numpy as np
import pytest

import msm_amd
from msm_amd import capi

pytestmark = pytest.mark.gpu


def _both(torch, lib, r, q, i0, i1, conv):
    dev = "cuda:0"
    n = q.shape[0]
    rd = torch.as_tensor(r, device=dev)
    qd = torch.as_tensor(q, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    # reference: the existing entry point with a caller-supplied work buffer
    acc_a = torch.full((i1 - i0, 3), float("nan"), dtype=torch.float64, device=dev)
    work = torch.empty(lib.msm_amd_direct_work_doubles(n), dtype=torch.float64, device=dev)
    u_a = C.c_double(0.0)
    capi.check(lib.msm_amd_direct_device(n, rd.data_ptr(), qd.data_ptr(), i0, i1, conv, acc_a.data_ptr(), work.data_ptr(), st,
                                         C.byref(u_a)), "direct_device")
    # the entry point under test: outputs allocated by the caller beforehand, nothing allocated inside
    acc_b = torch.full((i1 - i0, 3), float("nan"), dtype=torch.float64, device=dev)
    u_b = torch.full((1,), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    capi.check(lib.msm_amd_direct_device_noalloc(n, rd.data_ptr(), qd.data_ptr(), i0, i1, conv, acc_b.data_ptr(),
                                                 u_b.data_ptr(), st), "direct_device_noalloc")
    torch.cuda.synchronize()
    return u_a.value, acc_a.cpu().numpy(), float(u_b.cpu()[0]), acc_b.cpu().numpy()


def test_noalloc_direct_matches_direct_device_twice():
    import torch
    assert torch.cuda.is_available()
    lib = msm_amd.load()
    rng = np.random.default_rng(7)
    # first call: 20 000 particles, rows [1000, 19000) -> 36 workgroups of partials in the scratch array
    n1 = 20000
    r1 = rng.random((n1, 3))
    q1 = np.where(np.arange(n1) % 2 == 0, 1.0, -1.0) * (1.0 + rng.random(n1))
    ua, fa, ub, fb = _both(torch, lib, r1, q1, 1000, 19000, capi.ACC_GRADIENT)
    assert np.isfinite(fb).all() and np.isfinite(ub)
    assert ua == ub and np.array_equal(fa, fb)
    # second call, smaller: 3 000 particles, all rows, force convention -> 6 workgroups; stale partials of the first
    # call sit behind them in the array and must not be summed
    n2 = 3000
    r2 = rng.random((n2, 3)) * 2.0
    q2 = rng.standard_normal(n2)
    ua, fa, ub, fb = _both(torch, lib, r2, q2, 0, n2, capi.ACC_FORCE)
    assert np.isfinite(fb).all() and np.isfinite(ub)
    assert ua == ub and np.array_equal(fa, fb)
    # against the plain double sum on the host for a subset (the oracle of the direct sum is the sum itself)
    d = r2[:50, None, :] - r2[None, :, :]
    rr = np.sqrt((d ** 2).sum(-1))
    rr[np.arange(50), np.arange(50)] = np.inf
    grad = -(q2[None, :, None] * d / rr[..., None] ** 3).sum(1)
    ref = -q2[:50, None] * grad
    assert np.abs(fb[:50] - ref).max() <= 1e-12 * np.abs(ref).max()


def test_noalloc_direct_rejects_oversized_slice_and_bad_arguments():
    import torch
    lib = msm_amd.load()
    cap = lib.msm_amd_direct_noalloc_max_slice()
    assert cap == 8192 * 512
    one = torch.zeros(8, dtype=torch.float64, device="cuda:0")
    # sizes are checked before anything is launched: the pointers are never touched
    rc = lib.msm_amd_direct_device_noalloc(cap + 1000, one.data_ptr(), one.data_ptr(), 0, cap + 513, 0, one.data_ptr(),
                                           one.data_ptr(), None)
    assert rc == -5 and b"exceeds" in lib.msm_amd_last_error()
    assert lib.msm_amd_direct_device_noalloc(10, one.data_ptr(), one.data_ptr(), 0, 2, 0, one.data_ptr(), None, None) == -1
