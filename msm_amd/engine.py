"""Host-side mirror of the reference's operator interface on top of the C ABI.

`msmperiodic(...)` mirrors msmperiodic() of msm/msmLibrary.h:188 (same argument
meaning, same side effect on `r` in literal mode) and `Engine` is the
device-resident form: a plan built once, then energy + acceleration
evaluations on tensors that stay in HBM.  PyTorch is used for device memory
and streams only; all arithmetic happens in libmsm_amd.so.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import PlanDesc, MsmStats, check, load

_dp = C.POINTER(C.c_double)


def plan_auto(N, A, p=4, mu=10, abar=4.0, L=0):
    """The reference's parameter choices (msmPeriodic.c:147-263) for L grid levels (L=0: pick)."""
    d = PlanDesc()
    check(load().msm_amd_plan_auto(int(N), float(A[0]), float(A[1]), float(A[2]), int(p), int(mu), float(abar), int(L),
                                   C.byref(d)), "msm_amd_plan_auto")
    return d


def plan_open(N, lo, hi, p, mu, h, a, L=0):
    """Open-boundary plan (msm_amd_plan_open) for particles inside the box [lo, hi]."""
    d = PlanDesc()
    lo3 = (C.c_double * 3)(*[float(x) for x in lo])
    hi3 = (C.c_double * 3)(*[float(x) for x in hi])
    check(load().msm_amd_plan_open(int(N), lo3, hi3, int(p), int(mu), float(h), float(a), int(L), C.byref(d)),
          "msm_amd_plan_open")
    return d


def plan_open_holds(desc, lo, hi):
    """True while the open-boundary plan `desc` still serves particles inside the box [lo, hi] (msm_amd_plan_open_holds)."""
    lo3 = (C.c_double * 3)(*[float(x) for x in lo])
    hi3 = (C.c_double * 3)(*[float(x) for x in hi])
    return bool(load().msm_amd_plan_open_holds(C.byref(desc), lo3, hi3))


def plan_explicit(N, A, M, p, mu, L, a, beta, kmax, acc_convention=capi.ACC_GRADIENT):
    d = PlanDesc()
    d.N, d.p, d.mu, d.L, d.kmax, d.acc_convention = int(N), int(p), int(mu), int(L), int(kmax), int(acc_convention)
    for k in range(3):
        d.M[k] = int(M[k])
        d.A[k] = float(A[k])
    d.zplane0, d.nzplanes, d.n_owned = 0, int(M[2]), 0
    d.a, d.beta = float(a), float(beta)
    return d


class HostPlan:
    """Position-independent part of a plan (omega', stencils); needs no GPU."""

    def __init__(self, desc, literal=False):
        self.desc = desc
        self._h = C.c_void_p()
        check(load().msm_amd_hostplan_create(C.byref(desc), 1 if literal else 0, C.byref(self._h)),
              "msm_amd_hostplan_create")

    def close(self):
        if self._h:
            load().msm_amd_hostplan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stencil(self, level):
        """(values[z][y][x], first[3]) of grid level `level`; level == L is the Fourier top stencil."""
        size = (C.c_int * 3)()
        first = (C.c_int * 3)()
        check(load().msm_amd_hostplan_stencil_info(self._h, level, size, first), "stencil_info")
        out = np.zeros(size[0] * size[1] * size[2])
        check(load().msm_amd_hostplan_copy_stencil(self._h, level, out.ctypes.data_as(_dp)), "copy_stencil")
        return out.reshape(size[2], size[1], size[0]), list(first)

    def write_msmsimple_cache(self, directory=""):
        """msmSimple's two stencil cache files (msm/msmSimple.c:1018-1031, 1196-1209) for this L = 1 plan."""
        check(load().msm_amd_hostplan_write_msmsimple_cache(self._h, str(directory).encode()), "write_msmsimple_cache")

    def wprime(self):
        n = C.c_int(0)
        check(load().msm_amd_hostplan_copy_wprime(self._h, None, C.byref(n)), "copy_wprime")
        w = np.zeros(n.value)
        check(load().msm_amd_hostplan_copy_wprime(self._h, w.ctypes.data_as(_dp), C.byref(n)), "copy_wprime")
        return w

    def scalars(self):
        s = np.zeros(8)
        check(load().msm_amd_hostplan_scalars(self._h, s.ctypes.data_as(_dp)), "hostplan_scalars")
        return {"csr": s[0], "self_image_sum": s[1], "gamma0": s[2], "chisum": s[3], "J": s[4:8].copy()}


class Engine:
    """A plan plus its device buffers on one GPU; evaluations reuse them (no allocation per call)."""

    def __init__(self, desc, device=0, literal=False, stream=None):
        import torch  # device memory and streams only
        if not torch.cuda.is_available():
            raise capi.MsmAmdError("msm_amd needs a HIP device; none is visible and there is no CPU path")
        self.torch = torch
        self.desc = desc
        self.device = device
        self.literal = literal
        self._h = C.c_void_p()
        sp = C.c_void_p(stream) if stream else None
        fn = load().msm_amd_create_literal if literal else load().msm_amd_create
        check(fn(C.byref(desc), device, sp, C.byref(self._h)), "msm_amd_create")
        self.N = desc.N
        self.L = desc.L

    def close(self):
        if self._h:
            load().msm_amd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- whole evaluation ----
    def _dev(self, x, n):
        t = self.torch
        if not isinstance(x, t.Tensor):
            x = t.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device="cuda:%d" % self.device)
        assert x.dtype == t.float64 and x.is_cuda and x.is_contiguous() and x.numel() == n
        return x

    def eval(self, r, q, acc=None, want_energies=True):
        """r (N,3) and q (N,) float64 CUDA tensors (or arrays, then uploaded).  Returns (energies[7] or None, acc)."""
        t = self.torch
        r = self._dev(r, 3 * self.N)
        q = self._dev(q, self.N)
        if acc is None:
            acc = t.empty((self.N, 3), dtype=t.float64, device=r.device)
        en = np.zeros(capi.U_COUNT) if want_energies else None
        check(load().msm_amd_eval(self._h, r.data_ptr(), q.data_ptr(), acc.data_ptr(),
                                  en.ctypes.data_as(_dp) if want_energies else None), "msm_amd_eval")
        return en, acc

    def eval_enqueue(self, r, q, acc, energies_pinned=None):
        """Queue one evaluation without synchronising.  energies_pinned: a pinned CPU float64 tensor of 7 entries
        that receives the energies once the stream has been synchronised (None: no copy)."""
        check(load().msm_amd_eval_enqueue(self._h, r.data_ptr(), q.data_ptr(), acc.data_ptr(),
                                          energies_pinned.data_ptr() if energies_pinned is not None else None),
              "msm_amd_eval_enqueue")

    def eval_reuse(self, r, q, skin, acc=None, want_energies=True):
        """msm_amd_eval_reuse: an evaluation that keeps its neighbour lists while no particle has moved more than skin / 2
        since they were built (checked, and rebuilt if need be, on the device)."""
        t = self.torch
        r = self._dev(r, 3 * self.N)
        q = self._dev(q, self.N)
        if acc is None:
            acc = t.empty((self.N, 3), dtype=t.float64, device=r.device)
        en = np.zeros(capi.U_COUNT) if want_energies else None
        check(load().msm_amd_eval_reuse(self._h, r.data_ptr(), q.data_ptr(), acc.data_ptr(),
                                        en.ctypes.data_as(_dp) if want_energies else None, float(skin)), "msm_amd_eval_reuse")
        return en, acc

    def reuse_stats(self):
        """(evaluations through eval_reuse, list rebuilds among them)"""
        ev, rb = C.c_long(0), C.c_long(0)
        check(load().msm_amd_reuse_stats(self._h, C.byref(ev), C.byref(rb)), "msm_amd_reuse_stats")
        return ev.value, rb.value

    def eval_host(self, r, q):
        r = np.ascontiguousarray(r, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64)
        acc = np.zeros((self.N, 3))
        en = np.zeros(capi.U_COUNT)
        check(load().msm_amd_eval_host(self._h, r.ctypes.data_as(_dp), q.ctypes.data_as(_dp), acc.ctypes.data_as(_dp),
                                       en.ctypes.data_as(_dp)), "msm_amd_eval_host")
        return en, acc

    # ---- stages ----
    def stage_bin(self, r, q):
        self._r = self._dev(r, 3 * self.N)
        self._q = self._dev(q, self.N)
        check(load().msm_amd_stage_bin(self._h, self._r.data_ptr(), self._q.data_ptr()), "stage_bin")

    def stage_pairs(self, acc):
        check(load().msm_amd_stage_pairs(self._h, acc.data_ptr()), "stage_pairs")

    def stage_anterpolate(self):
        check(load().msm_amd_stage_anterpolate(self._h), "stage_anterpolate")

    def stage_grid(self):
        check(load().msm_amd_stage_grid(self._h), "stage_grid")

    def stage_interpolate(self, acc):
        check(load().msm_amd_stage_interpolate(self._h, acc.data_ptr()), "stage_interpolate")

    def stage_energies(self):
        en = np.zeros(capi.U_COUNT)
        check(load().msm_amd_stage_energies(self._h, en.ctypes.data_as(_dp)), "stage_energies")
        return en

    # ---- slab ranks ----
    def tensor_device(self):
        return "cuda:%d" % self.device

    def set_counts(self, n_total, n_owned):
        check(load().msm_amd_set_counts(self._h, int(n_total), int(n_owned)), "set_counts")
        self.N = int(n_total)

    def set_planes(self, zplane0, nzplanes):
        check(load().msm_amd_set_planes(self._h, int(zplane0), int(nzplanes)), "set_planes")
        self.desc.zplane0, self.desc.nzplanes = int(zplane0), int(nzplanes)

    def stage_raw_sums(self):
        """Queue the raw sums [pair, q0.e0 (owned planes), qL.efour, sum q, sum q^2]; they stay on the device."""
        check(load().msm_amd_stage_raw_sums(self._h, None), "stage_raw_sums")

    def raw_sums_tensor(self):
        return self._wrap(load().msm_amd_raw_sums_device_ptr(self._h), (5,))

    def combine_energies(self, sums):
        s = np.ascontiguousarray(sums, dtype=np.float64)
        en = np.zeros(capi.U_COUNT)
        check(load().msm_amd_combine_energies(self._h, s.ctypes.data_as(_dp), en.ctypes.data_as(_dp)), "combine_energies")
        return en

    def slab_step(self, coll, r, q, acc, mode=capi.SLAB_AUTO):
        """msm_amd_slab_step: one evaluation of a slab rank with its exchange steps through `coll` (a pointer to a
        capi.Collectives, or None for a handle that owns the whole grid); queued, nothing waits for the GPU."""
        check(load().msm_amd_slab_step(self._h, coll, r.data_ptr(), q.data_ptr(), acc.data_ptr(), int(mode)), "msm_amd_slab_step")

    def slab_energies(self, coll):
        en = np.zeros(capi.U_COUNT)
        check(load().msm_amd_slab_energies(self._h, coll, en.ctypes.data_as(_dp)), "msm_amd_slab_energies")
        return en

    def slab_exchange_mode(self, world, mode=capi.SLAB_AUTO):
        m = load().msm_amd_slab_exchange_mode(self._h, int(world), int(mode))
        if m < 0:
            check(m, "msm_amd_slab_exchange_mode")
        return m

    def _wrap(self, ptr, shape):
        """Zero-copy float64 torch view of device memory owned by the handle."""
        class _Wrap:
            pass
        w = _Wrap()
        w.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
        return self.torch.as_tensor(w, device="cuda:%d" % self.device)

    # ---- state ----
    def grid_dims(self, level):
        d = (C.c_int * 3)()
        check(load().msm_amd_grid_dims(self._h, level, d), "grid_dims")
        return list(d)

    def grid(self, which, level=0):
        d = self.grid_dims(self.L - 1 if which == capi.GRID_EFOUR else level)
        out = np.zeros(d[0] * d[1] * d[2])
        check(load().msm_amd_copy_grid(self._h, which, level, out.ctypes.data_as(_dp)), "copy_grid")
        return out.reshape(d[2], d[1], d[0])

    def set_grid(self, which, level, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        check(load().msm_amd_set_grid(self._h, which, level, v.ctypes.data_as(_dp)), "set_grid")

    def grid_tensor(self, which, level=0):
        """Zero-copy torch view of an internal grid (used by the slab all-gather)."""
        d = self.grid_dims(self.L - 1 if which == capi.GRID_EFOUR else level)
        ptr = load().msm_amd_grid_device_ptr(self._h, which, level)
        return self._wrap(ptr, (d[2], d[1], d[0]))

    def stencil(self, level):
        size = (C.c_int * 3)()
        first = (C.c_int * 3)()
        check(load().msm_amd_stencil_info(self._h, level, size, first), "stencil_info")
        out = np.zeros(size[0] * size[1] * size[2])
        check(load().msm_amd_copy_stencil(self._h, level, out.ctypes.data_as(_dp)), "copy_stencil")
        return out.reshape(size[2], size[1], size[0]), list(first)

    def write_msmsimple_cache(self, directory=""):
        check(load().msm_amd_write_msmsimple_cache(self._h, str(directory).encode()), "write_msmsimple_cache")

    def op_restrict(self, level, fine, coarse):
        check(load().msm_amd_op_restrict(self._h, level, fine.data_ptr(), coarse.data_ptr()), "op_restrict")

    def op_prolong_add(self, level, coarse, fine):
        check(load().msm_amd_op_prolong_add(self._h, level, coarse.data_ptr(), fine.data_ptr()), "op_prolong_add")

    def op_convolve(self, level, q, e):
        check(load().msm_amd_op_convolve(self._h, level, q.data_ptr(), e.data_ptr()), "op_convolve")

    def set_profiling(self, on=True):
        load().msm_amd_set_profiling(self._h, 1 if on else 0)

    def set_option(self, key, value):
        check(load().msm_amd_set_option(self._h, key.encode(), int(value)), "set_option")

    def last_kernel_ms(self, name):
        ms = C.c_double()
        check(load().msm_amd_last_kernel_ms(self._h, name.encode(), C.byref(ms)), "last_kernel_ms")
        return ms.value

    def last_timings(self):
        ms = np.zeros(10)
        check(load().msm_amd_last_timings(self._h, ms.ctypes.data_as(_dp)), "last_timings")
        names = ("bin", "pairs", "anterpolation", "restriction", "convolution", "top", "prolongation", "interpolation",
                 "reductions", "total")
        return dict(zip(names, ms))


def direct(r, q):
    """direct() of msm/msmLibrary.h:212 through the C ABI: open-boundary all-pairs energy and potential gradient.

    Returns (u: float, acc: (N,3) array)."""
    lib = load()
    r = np.ascontiguousarray(r, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    n = q.shape[0]
    f = np.zeros((n, 3))
    u = C.c_double(0.0)
    lib.direct(n, r.ctypes.data_as(_dp), q.ctypes.data_as(_dp), C.byref(u), f.ctypes.data_as(_dp))
    return u.value, f


def msm3d(r, q, p, h, a, mu, s=None):
    """msm3d() of msm/msmLibrary.h:216 through the C ABI: open-boundary MSM energy and potential gradient.

    Returns (u: float, acc: (N,3) array, stats)."""
    lib = load()
    r = np.ascontiguousarray(r, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    n = q.shape[0]
    f = np.zeros((n, 3))
    u = C.c_double(0.0)
    st = lib.msm3d(n, r.ctypes.data_as(_dp), q.ctypes.data_as(_dp), int(p), float(h), float(a), int(mu),
                   int(p if s is None else s), C.byref(u), f.ctypes.data_as(_dp))
    return u.value, f, st


def ewaldClassical(r, q, A, alpha, cutoffDirect, cutoffFourier):
    """ewaldClassical() of msm/ewaldClassical.h:6 through the C ABI (host arrays): the total energy."""
    lib = load()
    r = np.ascontiguousarray(r, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    return lib.ewaldClassical(r.ctypes.data_as(_dp), q.ctypes.data_as(_dp), q.shape[0], float(A[0]), float(A[1]), float(A[2]),
                              float(alpha), float(cutoffDirect), float(cutoffFourier))


def ewald(r, q, A, alpha, rcut, kcut, i0=0, i1=None, literal=False, want_grad=True):
    """Ewald potential and gradient at particles [i0, i1) on the GPU (msm_amd_ewald_device).

    r (N,3), q (N,): float64 CUDA tensors or host arrays.  Returns (energies, phi, grad): energies is the array
    [real, reciprocal, self, background, total] of the slice's share 1/2 sum q_i phi_i, phi and grad are CUDA tensors."""
    import torch
    lib = load()
    dev = r.device if isinstance(r, torch.Tensor) and r.is_cuda else torch.device("cuda:0")
    rd = torch.as_tensor(r, dtype=torch.float64, device=dev).contiguous()
    qd = torch.as_tensor(q, dtype=torch.float64, device=dev).contiguous()
    n = qd.shape[0]
    i1 = n if i1 is None else i1
    nw = lib.msm_amd_ewald_work_doubles(n, float(kcut), i1 - i0)
    if nw <= 0:
        raise capi.MsmAmdError("msm_amd_ewald_work_doubles: bad argument")
    work = torch.empty(nw, dtype=torch.float64, device=dev)
    phi = torch.empty(i1 - i0, dtype=torch.float64, device=dev)
    grad = torch.empty((i1 - i0, 3), dtype=torch.float64, device=dev) if want_grad else None
    Aa = (C.c_double * 3)(*[float(a) for a in A])
    u5 = np.zeros(5)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream().cuda_stream
        check(lib.msm_amd_ewald_device(n, rd.data_ptr(), qd.data_ptr(), Aa, float(alpha), float(rcut), float(kcut),
                                       capi.EWALD_LITERAL if literal else capi.EWALD_FULL, i0, i1, phi.data_ptr(),
                                       grad.data_ptr() if want_grad else None, work.data_ptr(), st, u5.ctypes.data_as(_dp)),
              "ewald")
    return u5, phi, grad


def msmperiodic(r, q, p, abar, mu, A, mode=None, quiet=True):
    """msmperiodic() of msm/msmLibrary.h:188 through the C ABI.

    r (N,3) float64 array (shifted in place by +0.5 in literal mode, as the reference does), q (N,).
    Returns (stats: MsmStats, uappr: float, fappr: (N,3) array)."""
    lib = load()
    r = np.ascontiguousarray(r, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    n = q.shape[0]
    f = np.zeros((n, 3))
    u = C.c_double(0.0)
    old = lib.msm_amd_get_mode()
    if mode is not None:
        lib.msm_amd_set_mode(mode)
    lib.msm_amd_set_quiet(1 if quiet else 0)
    try:
        st = lib.msmperiodic(n, r.ctypes.data_as(_dp), q.ctypes.data_as(_dp), int(p), float(abar), int(mu), C.byref(u),
                             f.ctypes.data_as(_dp), float(A[0]), float(A[1]), float(A[2]))
    finally:
        lib.msm_amd_set_mode(old)
    return st, u.value, f, r
