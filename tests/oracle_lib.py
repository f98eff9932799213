"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (msm_amd) never does.
"""
import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmsmref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class LiteralResult(C.Structure):
    _fields_ = [("N", C.c_int), ("p", C.c_int), ("mu", C.c_int), ("L", C.c_int), ("kmax", C.c_int),
                ("Mx", C.c_int), ("My", C.c_int), ("Mz", C.c_int),
                ("slmin", C.c_int * 3), ("slmax", C.c_int * 3), ("slsize", C.c_int * 3), ("sLn", C.c_int * 3),
                ("chisize", C.c_int), ("pad_", C.c_int),
                ("abar", C.c_double), ("Ax", C.c_double), ("Ay", C.c_double), ("Az", C.c_double),
                ("htilde", C.c_double), ("hx", C.c_double), ("hy", C.c_double), ("hz", C.c_double),
                ("a", C.c_double), ("aL", C.c_double), ("beta", C.c_double), ("csr", C.c_double),
                ("chisum", C.c_double),
                ("u_short_real", C.c_double), ("u_short_self", C.c_double), ("u_short_csr", C.c_double),
                ("u_long_real", C.c_double), ("u_long_four", C.c_double), ("u_long_self", C.c_double),
                ("u_total", C.c_double), ("netcharge", C.c_double), ("netgridcharge", C.c_double),
                ("sl", _dp), ("sL", _dp), ("qgrd", _dp), ("edir", _dp), ("etop", _dp),
                ("cvecx", _dp), ("cvecy", _dp), ("cvecz", _dp), ("chi", _dp),
                ("fshort", _dp), ("flong", _dp), ("facc", _dp), ("rshift", _dp)]


class SpecParams(C.Structure):
    _fields_ = [("N", C.c_int), ("p", C.c_int), ("mu", C.c_int), ("L", C.c_int), ("kmax", C.c_int),
                ("nthreads", C.c_int), ("M", C.c_int * 3), ("A", C.c_double * 3),
                ("a", C.c_double), ("beta", C.c_double), ("eps", C.c_double)]


def _arr(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO):
        raise RuntimeError("oracle/liboracle.so is missing: run `make -C oracle` (or __graft_entry__.build())")
    L = C.CDLL(ORACLE_SO)
    L.or_mylog2.restype = C.c_double
    L.or_mylog2.argtypes = [C.c_double]
    L.or_chooseM.restype = C.c_int
    L.or_chooseM.argtypes = [C.c_double, C.c_double]
    L.or_chooseL.restype = C.c_int
    L.or_chooseL.argtypes = [C.c_int] * 4
    L.or_chooseL_formula.restype = C.c_int
    L.or_chooseL_formula.argtypes = [C.c_int] * 4 + [_ip, _ip]
    L.or_choosebetaNew.restype = C.c_double
    L.or_choosebetaNew.argtypes = [C.c_double] * 3
    L.or_choosekmax.restype = C.c_double
    L.or_choosekmax.argtypes = [C.c_double] * 3
    L.or_minOf3.restype = C.c_int
    L.or_minOf3.argtypes = [C.c_int] * 3
    L.or_bspline.restype = C.c_double
    L.or_bspline.argtypes = [C.c_int, C.c_double]
    L.or_bspline_prime.restype = C.c_double
    L.or_bspline_prime.argtypes = [C.c_int, C.c_double]
    L.or_gamma.restype = C.c_double
    L.or_gamma.argtypes = [C.c_double, C.c_int]
    L.or_gamma_lit.restype = C.c_double
    L.or_gamma_lit.argtypes = [C.c_double, C.c_int]
    L.or_gamma_prime.restype = C.c_double
    L.or_gamma_prime.argtypes = [C.c_double, C.c_int]
    L.or_wprime.restype = C.c_int
    L.or_wprime.argtypes = [C.c_int, C.c_int, _dp]
    L.or_jcoef.restype = None
    L.or_jcoef.argtypes = [C.c_int, _dp]
    L.or_short_range_periodic.restype = C.c_double
    L.or_short_range_periodic.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_double, _dp] + [C.c_double] * 3
    L.or_literal_run.restype = C.c_int
    L.or_literal_run.argtypes = [C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_int] + [C.c_double] * 3 + \
        [C.POINTER(LiteralResult)]
    L.or_literal_free.restype = None
    L.or_literal_free.argtypes = [C.POINTER(LiteralResult)]
    L.or_spec_auto_params.restype = C.c_int
    L.or_spec_auto_params.argtypes = [C.c_int] + [C.c_double] * 3 + [C.c_int, C.c_int, C.c_double, C.c_int,
                                                                    C.POINTER(SpecParams)]
    L.or_spec_plan_create.restype = C.c_void_p
    L.or_spec_plan_create.argtypes = [C.POINTER(SpecParams)]
    L.or_spec_set_image_range.restype = None
    L.or_spec_set_image_range.argtypes = [C.c_int]
    L.or_spec_set_open_boundary.restype = None
    L.or_spec_set_open_boundary.argtypes = [C.c_int]
    L.or_spec_plan_destroy.restype = None
    L.or_spec_plan_destroy.argtypes = [C.c_void_p]
    L.or_spec_stencil.restype = _dp
    L.or_spec_stencil.argtypes = [C.c_void_p, C.c_int, _ip, _ip]
    L.or_spec_stencil_top.restype = _dp
    L.or_spec_stencil_top.argtypes = [C.c_void_p, _ip]
    L.or_spec_wprime.restype = _dp
    L.or_spec_wprime.argtypes = [C.c_void_p, _ip]
    for nm in ("or_spec_grid_q", "or_spec_grid_e"):
        f = getattr(L, nm)
        f.restype = _dp
        f.argtypes = [C.c_void_p, C.c_int, _ip]
    L.or_spec_grid_efour.restype = _dp
    L.or_spec_grid_efour.argtypes = [C.c_void_p, _ip]
    L.or_spec_eval.restype = C.c_int
    L.or_spec_eval.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.or_spec_pairs.restype = None
    L.or_spec_pairs.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
    L.or_spec_pairs_ex.restype = None
    L.or_spec_pairs_ex.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp]
    L.or_spec_anterpolate.restype = None
    L.or_spec_anterpolate.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.or_spec_restrict.restype = None
    L.or_spec_restrict.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    L.or_spec_prolong_add.restype = None
    L.or_spec_prolong_add.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    L.or_spec_convolve.restype = None
    L.or_spec_convolve.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    L.or_spec_convolve_top.restype = None
    L.or_spec_convolve_top.argtypes = [C.c_void_p, _dp, _dp]
    L.or_spec_interpolate.restype = None
    L.or_spec_interpolate.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.or_ewald.restype = C.c_double
    L.or_ewald.argtypes = [C.c_int, _dp, _dp] + [C.c_double] * 5 + [C.c_int, C.c_int, _dp, C.c_int]
    L.or_direct.restype = C.c_double
    L.or_direct.argtypes = [C.c_int, _dp, _dp, C.c_int, _dp, C.c_int]
    L.or_ewald_classical.restype = C.c_double
    L.or_ewald_classical.argtypes = [_dp, _dp, C.c_int] + [C.c_double] * 6
    _lib = L
    return L


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def wprime(p, mu):
    w = np.zeros(mu + p // 2 + 1)
    rc = lib().or_wprime(p, mu, w.ctypes.data_as(_dp))
    assert rc == 0
    return w


def literal_run(r, q, p, abar, mu, A):
    """Run the literal mirror of msmperiodic(); returns a dict of numpy copies."""
    L = lib()
    r, rp = _arr(r)
    q, qp = _arr(q)
    res = LiteralResult()
    rc = L.or_literal_run(len(q), rp, qp, p, abar, mu, A[0], A[1], A[2], C.byref(res))
    out = {"rc": rc}
    if rc == 0:
        for k, _t in LiteralResult._fields_:
            v = getattr(res, k)
            if isinstance(v, (int, float)):
                out[k] = v
        out["slmin"] = list(res.slmin)
        out["slsize"] = list(res.slsize)
        n = len(q)
        M3 = res.Mx * res.My * res.Mz
        ns = res.slsize[0] * res.slsize[1] * res.slsize[2]
        nL = res.sLn[0] * res.sLn[1] * res.sLn[2]
        for k, cnt in (("sl", ns), ("sL", nL), ("qgrd", M3), ("edir", M3), ("etop", M3), ("fshort", 3 * n),
                       ("flong", 3 * n), ("facc", 3 * n), ("rshift", 3 * n), ("chi", res.chisize),
                       ("cvecx", res.kmax + 1)):
            out[k] = np.ctypeslib.as_array(getattr(res, k), shape=(cnt,)).copy() if cnt else np.zeros(0)
    L.or_literal_free(C.byref(res))
    return out


class SpecPlan:
    """Specification engine plan (see oracle/msm_spec.c)."""

    def __init__(self, prm, open_boundary=False):
        self.prm = prm
        lib().or_spec_set_open_boundary(1 if open_boundary else 0)
        try:
            self.h = lib().or_spec_plan_create(C.byref(prm))
        finally:
            lib().or_spec_set_open_boundary(0)
        if not self.h:
            raise RuntimeError("or_spec_plan_create failed")
        self.N = prm.N
        self.L = prm.L

    @classmethod
    def auto(cls, N, A, p=4, mu=10, abar=4.0, L=1, nthreads=0):
        prm = SpecParams()
        rc = lib().or_spec_auto_params(N, A[0], A[1], A[2], p, mu, abar, L, C.byref(prm))
        if rc:
            raise RuntimeError("or_spec_auto_params failed (M not divisible by 2^(L-1)?)")
        prm.nthreads = nthreads
        return cls(prm)

    @classmethod
    def explicit(cls, N, A, M, p, mu, L, a, beta, kmax, nthreads=0, open_boundary=False):
        prm = SpecParams()
        prm.N, prm.p, prm.mu, prm.L, prm.kmax, prm.nthreads = N, p, mu, L, kmax, nthreads
        for d in range(3):
            prm.M[d] = M[d]
            prm.A[d] = A[d]
        prm.a, prm.beta, prm.eps = a, beta, 1e-12
        return cls(prm, open_boundary=open_boundary)

    def close(self):
        if self.h:
            lib().or_spec_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dims(self, l):
        return [self.prm.M[d] >> l for d in range(3)]

    def stencil(self, l):
        size = (C.c_int * 3)()
        first = (C.c_int * 3)()
        p = lib().or_spec_stencil(self.h, l, size, first)
        n = size[0] * size[1] * size[2]
        return np.ctypeslib.as_array(p, shape=(n,)).copy().reshape(size[2], size[1], size[0]), list(first)

    def stencil_top(self):
        size = (C.c_int * 3)()
        p = lib().or_spec_stencil_top(self.h, size)
        n = size[0] * size[1] * size[2]
        return np.ctypeslib.as_array(p, shape=(n,)).copy().reshape(size[2], size[1], size[0])

    def grid(self, which, l=0):
        d = (C.c_int * 3)()
        if which == "efour":
            p = lib().or_spec_grid_efour(self.h, d)
        else:
            p = getattr(lib(), "or_spec_grid_" + which)(self.h, l, d)
        n = d[0] * d[1] * d[2]
        return np.ctypeslib.as_array(p, shape=(n,)).copy().reshape(d[2], d[1], d[0])

    def eval(self, r, q):
        r, rp = _arr(r)
        q, qp = _arr(q)
        n = self.N
        acc = np.zeros(3 * n)
        fu = np.zeros(3 * n)
        al = np.zeros(3 * n)
        en = np.zeros(7)
        sec = np.zeros(8)
        lib().or_spec_eval(self.h, rp, qp, acc.ctypes.data_as(_dp), fu.ctypes.data_as(_dp), al.ctypes.data_as(_dp),
                           en.ctypes.data_as(_dp), sec.ctypes.data_as(_dp))
        return {"acc": acc.reshape(n, 3), "fshort_unw": fu.reshape(n, 3), "along": al.reshape(n, 3),
                "energies": en, "seconds": sec}

    def pairs(self, r, q):
        r, rp = _arr(r)
        q, qp = _arr(q)
        n = self.N
        aw = np.zeros(3 * n)
        au = np.zeros(3 * n)
        u = C.c_double(0)
        lib().or_spec_pairs(self.h, rp, qp, aw.ctypes.data_as(_dp), au.ctypes.data_as(_dp), C.byref(u))
        return aw.reshape(n, 3), au.reshape(n, 3), u.value

    def pairs_phi(self, r, q):
        r, rp = _arr(r)
        q, qp = _arr(q)
        n = self.N
        aw = np.zeros(3 * n)
        phi = np.zeros(n)
        u = C.c_double(0)
        lib().or_spec_pairs_ex(self.h, rp, qp, aw.ctypes.data_as(_dp), None, C.byref(u), phi.ctypes.data_as(_dp))
        return aw.reshape(n, 3), phi

    def anterpolate(self, r, q):
        r, rp = _arr(r)
        q, qp = _arr(q)
        d = self.dims(0)
        g = np.zeros(d[0] * d[1] * d[2])
        lib().or_spec_anterpolate(self.h, rp, qp, g.ctypes.data_as(_dp))
        return g.reshape(d[2], d[1], d[0])

    def restrict(self, l, fine):
        fine, fp = _arr(fine)
        d = self.dims(l + 1)
        c = np.zeros(d[0] * d[1] * d[2])
        lib().or_spec_restrict(self.h, l, fp, c.ctypes.data_as(_dp))
        return c.reshape(d[2], d[1], d[0])

    def prolong_add(self, l, coarse, fine):
        coarse, cp = _arr(coarse)
        fine = np.array(fine, dtype=np.float64, copy=True)
        lib().or_spec_prolong_add(self.h, l, cp, fine.ctypes.data_as(_dp))
        return fine

    def convolve(self, l, q):
        q, qp = _arr(q)
        e = np.zeros(q.size)
        lib().or_spec_convolve(self.h, l, qp, e.ctypes.data_as(_dp))
        return e.reshape(q.shape)

    def convolve_top(self, q):
        q, qp = _arr(q)
        e = np.zeros(q.size)
        lib().or_spec_convolve_top(self.h, qp, e.ctypes.data_as(_dp))
        return e.reshape(q.shape)

    def interpolate(self, r, e):
        r, rp = _arr(r)
        e, ep = _arr(e)
        al = np.zeros(3 * self.N)
        lib().or_spec_interpolate(self.h, rp, ep, al.ctypes.data_as(_dp))
        return al.reshape(self.N, 3)


def ewald(r, q, A, alpha, rcut, kcut, nsub=None, nthreads=0):
    r, rp = _arr(r)
    q, qp = _arr(q)
    n = len(q)
    nsub = n if nsub is None else nsub
    acc = np.zeros(3 * nsub)
    u = lib().or_ewald(n, rp, qp, A[0], A[1], A[2], alpha, rcut, kcut, nsub, acc.ctypes.data_as(_dp), nthreads)
    return u, acc.reshape(nsub, 3)


def direct(r, q, nthreads=0):
    """Open-boundary direct sum (msm/nonperiodic.c:12-22 plus the gradient): (u, acc)."""
    r, rp = _arr(r)
    q, qp = _arr(q)
    n = len(q)
    acc = np.zeros(3 * n)
    u = lib().or_direct(n, rp, qp, n, acc.ctypes.data_as(_dp), nthreads)
    return u, acc.reshape(n, 3)


def neutral_system(N, A=(1.0, 1.0, 1.0), seed=20261111):
    """Uniform random positions in [0,A)^3, charges +1/-1 alternating (exactly neutral)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r = rng.random((N, 3)) * np.asarray(A)
    q = np.where(np.arange(N) % 2 == 0, 1.0, -1.0)
    return r, q
