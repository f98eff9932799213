"""CPU-side checks of the product library: it loads, exports every symbol the header declares, keeps the
reference's msm_stats layout, restates the host helpers exactly, builds the same stencils as the oracle, and
refuses to compute without a GPU (no fallback).  No device work is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def msm():
    import msm_amd
    msm_amd.load()
    return msm_amd


def test_library_exports_every_declared_symbol(msm):
    hdr = open(os.path.join(ROOT, "include", "msm_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(msm_amd_\w+|msmperiodic|msmPeriodicRunner|mylog2|choose\w+|minOf3Integers|data_read|"
                           r"short_range_periodic|print_stats|direct|msm3d|ewaldClassical\w*|msm4g_linkedlist_\w+)\s*\(", hdr))
    names -= {"msm_amd_plan_desc", "msm_amd_handle", "msm_amd_hostplan"}
    assert len(names) >= 55
    lib = C.CDLL(msm.capi.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding covers the same set
    assert set(msm.capi.SIGNATURES) == names, set(msm.capi.SIGNATURES) ^ names


def test_msm_stats_layout(msm):
    S = msm.capi.MsmStats
    assert C.sizeof(S) == 960
    # offsets of the reference's struct (msm/msmLibrary.h:46-110), measured against the real header
    want = {"N": 0, "h": 16, "beta": 72, "csr": 96, "chinx": 104, "cvecx": 120, "clenx": 144, "chisum": 160, "chi": 168,
            "sLsize": 176, "M0": 192, "Mx": 204, "Mlx": 240, "slsizex": 360, "Ax": 480, "size": 492, "rminx": 616,
            "volume": 664, "short_range_interactions": 680, "netcharge": 696, "netgridcharge": 704,
            "cputime_initialization": 784, "cputime_total": 880, "potential_total": 888, "potential_long_fourier": 952}
    for k, off in want.items():
        assert getattr(S, k).offset == off, (k, getattr(S, k).offset, off)


def test_helpers_bit_exact(msm):
    lib = msm.load()
    h = O.golden("helpers.json")
    for row in h["rows"]:
        M = lib.chooseM(row["htilde"], row["A"])
        assert M == row["M"]
        hx, a, L = row["A"] / M, row["abar"] * row["htilde"], row["L"]
        b = lib.choosebetaNew(1e-12, (2.0 ** L) * a, (2.0 ** (L - 1)) * hx)
        assert b == row["beta"]
        assert lib.choosekmax(1e-12, b, (2.0 ** (L - 1)) * hx) == row["kmax_double"]
        assert lib.chooseL(M, M, M, row["N"]) == 1   # msmLibrary.c:64-66
    for x, y in h["extra"]["mylog2"]:
        assert lib.mylog2(x) == y
    for a, b, c, m in h["extra"]["minOf3Integers"]:
        assert lib.minOf3Integers(a, b, c) == m


def test_short_range_periodic_host(msm):
    lib = msm.load()
    fx = O.golden("msmPeriodic_changaN8.json")
    r = np.array(fx["system"]["r"]) + 0.5
    q = np.array(fx["system"]["q"])
    f1, f2 = np.zeros(24), np.zeros(24)
    dp = O._dp
    for i in range(8):
        for j in range(8):
            u1 = lib.short_range_periodic(i, j, r.ctypes.data_as(dp), q.ctypes.data_as(dp), 3, 2.0,
                                          f1.ctypes.data_as(dp), 1.0, 1.0, 1.0)
            u2 = O.lib().or_short_range_periodic(i, j, r.ctypes.data_as(dp), q.ctypes.data_as(dp), 3, 2.0,
                                                 f2.ctypes.data_as(dp), 1.0, 1.0, 1.0)
            assert u1 == u2
    np.testing.assert_allclose(f1, f2, rtol=1e-14, atol=1e-20)


def test_linked_list_and_data_read(msm, tmp_path):
    lib = msm.load()
    ll = lib.msm4g_linkedlist_new()
    for k in (5, 7, 9):
        lib.msm4g_linkedlist_addvalue(ll, k, k * 1.5)
    lib.msm4g_linkedlist_add(ll, 11)
    assert lib.msm4g_linkedlist_search(ll, 9) == 2 and lib.msm4g_linkedlist_search(ll, 4) == -1
    assert lib.msm4g_linkedlist_getvalue(ll, 1) == 10.5 and lib.msm4g_linkedlist_getvalue(ll, 9) == -1
    lib.msm4g_linkedlist_destroy(ll)
    p = tmp_path / "sys.txt"
    p.write_text("2.0 3.0 4.0\n2\n 0.1 0.2 0.3 1.0\n 0.4 0.5 0.6 -1.0\n")
    q, r = O._dp(), O._dp()
    n = C.c_int()
    L = [C.c_double() for _ in range(3)]
    assert lib.data_read(str(p).encode(), C.byref(q), C.byref(r), C.byref(n), *[C.byref(x) for x in L]) == 0
    assert n.value == 2 and [x.value for x in L] == [2.0, 3.0, 4.0]
    assert [r[i] for i in range(6)] == [0.1, 0.2, 0.3, 0.4, 0.5, 0.6] and [q[0], q[1]] == [1.0, -1.0]
    assert lib.data_read(b"/nonexistent/file", C.byref(q), C.byref(r), C.byref(n), *[C.byref(x) for x in L]) == 1


@pytest.mark.parametrize("N,L,p,mu", [(512, 1, 4, 10), (4096, 2, 4, 10), (4096, 2, 6, 6), (64, 1, 4, 10), (30000, 3, 4, 10)])
def test_host_plan_matches_oracle(msm, N, L, p, mu):
    """omega', J, every level stencil and the Fourier stencil: the product's separable construction against the
    oracle's direct sums."""
    A = (1.0, 1.0, 1.0)
    d = msm.plan_auto(N, A, p, mu, 4.0, L)
    hp = msm.HostPlan(d)
    sp = O.SpecPlan.auto(N, A, p, mu, 4.0, L, nthreads=4)
    assert (d.beta, d.kmax, d.a, list(d.M)) == (sp.prm.beta, sp.prm.kmax, sp.prm.a, list(sp.prm.M))
    assert np.array_equal(hp.wprime(), O.wprime(p, mu))
    tol = 2e-11 if p == 6 else 1e-12
    for l in range(L):
        K, f = hp.stencil(l)
        Ko, fo = sp.stencil(l)
        assert f == fo and K.shape == Ko.shape
        assert np.abs(K - Ko).max() <= tol * np.abs(Ko).max()
    K, _ = hp.stencil(L)
    Ko = sp.stencil_top()
    assert np.abs(K - Ko).max() <= 1e-12 * np.abs(Ko).max()
    J = np.zeros(4)
    O.lib().or_jcoef(p, J.ctypes.data_as(O._dp))
    assert np.array_equal(hp.scalars()["J"], J)
    assert hp.scalars()["gamma0"] == O.lib().or_gamma(0.0, p)


def test_literal_host_plan_matches_reference_stencil(msm):
    """Literal mode's level-0 stencil against the reference's own `Stencil[0]` dump (printed with %25.16f)."""
    for name in ("changaN8", "CsClN16", "changaN64", "changaN512"):
        fx = O.golden("msmPeriodic_%s.json" % name)
        s = fx["system"]
        d = msm.plan_auto(s["N"], s["L"], 4, 10, 4.0, 1)
        hp = msm.HostPlan(d, literal=True)
        K, first = hp.stencil(0)
        ref = np.array(fx["stencil0"])
        assert K.size == ref.size
        assert np.abs(K.ravel() - ref).max() <= 1e-11 * np.abs(ref).max()


def test_plan_auto_levels(msm):
    d = msm.plan_auto(1000000, (1.0, 1.0, 1.0), 4, 10, 4.0, 0)
    assert list(d.M) == [96, 96, 96] and d.L == 4          # 96 -> 48 -> 24 -> 12
    d = msm.plan_auto(100000, (1.0, 1.0, 1.0), 4, 10, 4.0, 0)
    assert list(d.M) == [48, 48, 48] and d.L == 3          # 48 -> 24 -> 12
    with pytest.raises(msm.MsmAmdError):
        msm.plan_auto(1000, (1.0, 1.0, 1.0), 4, 10, 4.0, 6)   # 16 is not divisible by 32
    with pytest.raises(msm.MsmAmdError):
        msm.HostPlan(msm.plan_explicit(10, (1, 1, 1), (8, 8, 8), 5, 10, 1, 0.3, 3.0, 5))


def test_no_cpu_fallback(msm):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    d = msm.plan_auto(100, (1.0, 1.0, 1.0), 4, 10, 4.0, 1)
    h = C.c_void_p()
    rc = msm.load().msm_amd_create(C.byref(d), 0, None, C.byref(h))
    assert rc == -2 and not h.value   # MSM_AMD_ENODEV
    assert b"no CPU path" in msm.load().msm_amd_last_error()
    with pytest.raises(msm.MsmAmdError):
        msm.Engine(d)
    # the host-pointer entry points of the other drivers fail loudly too: no result, the reason on stderr
    r = np.random.default_rng(1).random((8, 3))
    q = np.where(np.arange(8) % 2 == 0, 1.0, -1.0)
    u = msm.ewaldClassical(r, q, (1.0, 1.0, 1.0), 6.0, 10.0, 4.0)
    assert np.isnan(u) and b"no CPU path" in msm.load().msm_amd_last_error()
    u, f = msm.direct(r, q)
    assert u == 0.0 and not f.any() and b"no CPU path" in msm.load().msm_amd_last_error()
    # the multi-rank entry points: null handles are refused, and a communicator cannot be made without a device
    lib = msm.load()
    assert lib.msm_amd_slab_step(None, None, None, None, None, 0) == -1 and b"null argument" in lib.msm_amd_last_error()
    assert lib.msm_amd_slab_exchange_mode(None, 8, 0) < 0
    assert lib.msm_amd_slab_energies(None, None, (C.c_double * 16)()) == -1
    # the profiling hook of the pair kernel refuses a null handle as well
    assert lib.msm_amd_set_pair_trace(None, None) == -1 and lib.msm_amd_pair_trace_slots(None) == 0
    ident = C.create_string_buffer(128)
    if lib.msm_amd_rccl_unique_id(ident) == 0:   # librccl is present in the image
        from msm_amd.capi import Collectives
        coll = C.POINTER(Collectives)()
        assert lib.msm_amd_rccl_create(ident, 0, 1, C.byref(coll)) < 0 and not coll
        assert b"ncclCommInitRank" in lib.msm_amd_last_error()


def test_plan_open(msm):
    """msm_amd_plan_open: the cell keeps the grid spacing, is a whole number of coarsest cells, holds the particles in
    its middle with room for the spline supports and the widest compact stencil, and at least three cutoffs."""
    lo, hi = (-3.0, 2.5, 10.0), (-2.0, 3.1, 11.3)
    for (N, p, h, a, L) in ((4000, 4, 1 / 16, 0.25, 2), (1000000, 6, 1 / 96, 0.04, 0), (50, 6, 0.5, 2.0, 1), (20000, 6, 1 / 28, 0.125, 3)):
        d = msm.plan_open(N, lo, hi, p, 10, h, a, L)
        assert d.open_boundary == 1 and d.kmax == 0 and (L == 0 or d.L == L)
        hc = h * 2 ** (d.L - 1)
        for k in range(3):
            E = hi[k] - lo[k]
            assert d.M[k] % (2 ** (d.L - 1)) == 0
            assert abs(d.A[k] - d.M[k] * h) < 1e-12 * d.A[k]
            room = d.A[k] - E
            assert room >= (p + 2) * hc - 1e-12 and d.A[k] >= 3 * a
            if d.L >= 2:
                assert room >= (p / 2) * hc + 2 ** (d.L - 1) * a
            assert abs((lo[k] - d.origin[k]) - (d.origin[k] + d.A[k] - hi[k])) < 1e-9
            assert d.extent[k] == E
        if L == 0:
            assert 8 <= max(d.M) >> (d.L - 1) <= 32 or d.L == 1
    with pytest.raises(msm.MsmAmdError):
        msm.plan_open(10, lo, hi, 5, 10, 0.1, 0.4, 1)
    # one level and a long cutoff: still no pair within the cutoff through the wrap
    d = msm.plan_open(1000, (0, 0, 0), (4, 4, 4), 4, 10, 0.1, 1.0, 1)
    assert min(d.A) - 4.0 >= 1.0
    # a coarsest grid too large for the all-pairs top level is refused when the plan is built
    with pytest.raises(msm.MsmAmdError):
        msm.HostPlan(msm.plan_open(100000, (0, 0, 0), (1, 1, 1), 4, 10, 1 / 64, 4 / 64, 1))
    # an open plan builds on the host like a periodic one: no images, no Fourier level
    d = msm.plan_open(300, (0, 0, 0), (1, 1, 1), 4, 10, 1 / 6, 4 / 6, 2)
    hp = msm.HostPlan(d)
    K, first = hp.stencil(d.L - 1)
    ML = [m >> (d.L - 1) for m in d.M]
    assert list(K.shape) == [2 * ML[2] - 1, 2 * ML[1] - 1, 2 * ML[0] - 1] and first == [-(ML[0] - 1), -(ML[1] - 1), -(ML[2] - 1)]
    assert not hp.stencil(d.L)[0].any() and hp.scalars()["csr"] == 0.0 and hp.scalars()["self_image_sum"] == 0.0
    sp = O.SpecPlan.explicit(300, list(d.A), list(d.M), 4, 10, d.L, d.a, 1.0, 0, nthreads=4, open_boundary=True)
    for l in range(d.L):
        Ko, fo = sp.stencil(l)
        Kh, fh = hp.stencil(l)
        assert fh == fo and np.abs(Kh - Ko).max() <= 1e-12 * np.abs(Ko).max()


from test_oracle_golden import SIMPLE, simple_case


@pytest.mark.parametrize("fn,sysname,copies", SIMPLE)
def test_msmsimple_stencil_cache_files(msm, fn, sysname, copies, tmp_path):
    """The cache files msm_amd writes for msmSimple carry msmSimple's own names, its format, and the values the real
    msmSimple wrote into its files (fixtures), to the tolerance of msmSimple's image sums (1e-8 relative change,
    msmSimple.c:16, i.e. 2e-11 of the largest entry on these systems)."""
    fx, r, q, A = simple_case(fn, sysname, copies)
    sc, args = fx["scalars"], fx["args"]
    mu, nu, M = args[2], args[3], args[4:7]
    d = msm.plan_explicit(len(q), A, M, nu, mu, 1, sc["a"], sc["beta"], int(sc["kmax"]))
    hp = msm.HostPlan(d)
    hp.write_msmsimple_cache(tmp_path)
    f1 = tmp_path / ("stencil_l00_Mx%02d_My%02d_Mz%02d_nu%02d_mu%02d.dat" % (M[0], M[1], M[2], nu, mu))
    f2 = tmp_path / ("stencil_fourier_Mx%02d_My%02d_Mz%02d_nu%02d.dat" % (M[0], M[1], M[2], nu))
    for f, key, tol in ((f1, "Kl", 2e-11), (f2, "Kfourier", 1e-12)):
        lines = f.read_text().splitlines()
        ref = np.array(fx[key])
        assert len(lines) == ref.size and all(len(l) == 25 for l in lines)
        got = np.array([float(l) for l in lines])          # msmSimple reads them back with fscanf("%lf")
        assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    with pytest.raises(msm.MsmAmdError):
        msm.HostPlan(msm.plan_auto(4096, (1.0, 1.0, 1.0), 4, 10, 4.0, 2)).write_msmsimple_cache(tmp_path)


def test_real_msmsimple_runs_on_our_cache_files(msm, tmp_path):
    """The real msmSimple (built from the reference's sources into oracle/_ref when they are present; skipped
    otherwise), started in a directory that holds the cache files msm_amd wrote, reads them instead of computing its
    stencils and prints the energies of its own full run (fixture)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "msmSimple")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/msmSimple not built (no reference sources on this machine)")
    fn, sysname, copies = SIMPLE[-1]      # changaN64, M = 4: the longest stencil computation among the fixtures
    fx, r, q, A = simple_case(fn, sysname, copies)
    sc, args = fx["scalars"], fx["args"]
    mu, nu, M = args[2], args[3], args[4:7]
    d = msm.plan_explicit(len(q), A, M, nu, mu, 1, sc["a"], sc["beta"], int(sc["kmax"]))
    msm.HostPlan(d).write_msmsimple_cache(tmp_path)
    before = {f.name: f.read_bytes() for f in tmp_path.iterdir()}
    assert len(before) == 2
    res = subprocess.run([exe, str(args[0]), repr(float(args[1])), str(mu), str(nu), str(M[0]), str(M[1]), str(M[2]), "1"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-500:]
    for name, data in before.items():
        assert (tmp_path / name).read_bytes() == data          # read, not rewritten
    got = {}
    for ln in res.stdout.splitlines():
        m = re.match(r"^(\S[^:]*?)\s*:\s*([-+0-9.eEnaif]+)\s*$", ln)
        if m and m.group(1).strip() not in got:
            try:
                got[m.group(1).strip()] = float(m.group(2))
            except ValueError:
                pass
    scale = abs(sc["ushort_real"]) + abs(sc["ulong_self"])
    for k in ("ulong_real", "ulong_four", "utotal"):
        assert abs(got[k] - sc[k]) <= 2e-10 * max(abs(sc[k]), 1e-3 * scale), (k, got[k], sc[k])
    assert got["time_stencilKl"] < 0.05 * max(sc["time_stencilKl"], 1e-3) or got["time_stencilKl"] < 0.01


# ---------------------------------------------------------------------------
# the header as the reference's drivers see it
# ---------------------------------------------------------------------------
def test_generated_tables_header_is_current(tmp_path):
    """include/msm_tables.h (TAUP4 / TAUP4D / ERFBETA, B-spline pieces, JN, OMEGA: msm/msmLibrary.h:18-20,113-185) is
    what tools/gen_tables.cpp prints from the library's own spline_pieces / two_scale / compute_wprime / gamma_poly."""
    import subprocess
    exe = tmp_path / "gen_tables"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "msm_amd", "csrc"),
                    "-DMSM_AMD_NO_TABLES", os.path.join(ROOT, "tools", "gen_tables.cpp"),
                    os.path.join(ROOT, "msm_amd", "csrc", "msm_host.cpp"), "-o", str(exe)], check=True, timeout=300)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "include", "msm_tables.h")).read()


def test_tables_header_values(msm, tmp_path):
    """The generated header through a C compiler: the spline pieces against the Cox-de Boor recursion of
    msmPeriodic.c:10-19 and a finite difference, JN against binomials, OMEGA against the library's omega' (which
    tests/test_oracle_golden.py pins to the reference's table), TAUP4 / TAUP4D against the series of msmPeriodic.c:22-34."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text(r'''
#include "msmLibrary.h"
double piece(int p, int der, int v, double t) {
    const struct MasterBaseFunction *b = p == 4 ? (der ? &CubicBSplineD : &CubicBSpline) : (der ? &QuinticBSplineD : &QuinticBSpline);
    return b->p == p ? b->region[v](t) : -1e300;
}
double jn(int i, int j) { return JN[i][j]; }
double omega(int i, int j) { return OMEGA[i][j]; }
double taup4(double x) { return TAUP4(x); }
double taup4d(double x) { return TAUP4D(x); }
double erfbeta(double b, double x) { return ERFBETA(b, x); }
''')
    so = tmp_path / "t.so"
    subprocess.run(["gcc", "-O1", "-std=c99", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(so), "-lm"],
                   check=True, timeout=120)
    t = C.CDLL(str(so))
    for f in (t.piece, t.jn, t.omega, t.taup4, t.taup4d, t.erfbeta):
        f.restype = C.c_double
    t.piece.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double]
    t.jn.argtypes = t.omega.argtypes = [C.c_int, C.c_int]
    t.taup4.argtypes = t.taup4d.argtypes = [C.c_double]
    t.erfbeta.argtypes = [C.c_double, C.c_double]

    def bspline(k, u):   # cardinal B-spline of order k on [0, k)
        if k == 1:
            return 1.0 if 0 <= u < 1 else 0.0
        return u / (k - 1) * bspline(k - 1, u) + (k - u) / (k - 1) * bspline(k - 1, u - 1.0)
    for p in (4, 6):
        for v in range(p):
            lo = p / 2 - 1 - v            # region[v] covers [lo, lo + 1]
            for f in np.linspace(0.01, 0.99, 23):
                x = lo + f
                assert abs(t.piece(p, 0, v, x) - bspline(p, x + p / 2)) < 5e-15
                d = bspline(p - 1, x + p / 2) - bspline(p - 1, x + p / 2 - 1)   # B_p' = B_{p-1}(u) - B_{p-1}(u - 1)
                assert abs(t.piece(p, 1, v, x) - d) < 5e-15
    from math import comb, erf, pi, sqrt
    for i, p in enumerate((4, 6)):
        for j in range(6):
            assert t.jn(i, j) == (comb(p, p // 2 + j) / 2 ** (p - 1) if j <= p // 2 else 0.0)
        w = msm.HostPlan(msm.plan_explicit(64, (1.0, 1.0, 1.0), (4, 4, 4), p, 10, 1, 1.0, 2.5, 5)).wprime()
        assert len(w) == 10 + p // 2 + 1
        for j in range(14):
            assert t.omega(i, j) == (w[j] if j < len(w) else 0.0)
        ref = O.wprime(p, 10)
        assert np.abs(np.array([t.omega(i, j) for j in range(len(ref))]) - ref).max() < 3e-15 * abs(ref[0])
    for x in np.linspace(0.0, 1.0, 11):
        y = x * x - 1.0
        series = 1.0 - y / 2 + 3.0 * y * y / 8 - 5.0 * y ** 3 / 16      # binomial series of (1 + y)^(-1/2), 4 terms
        assert abs(t.taup4(x) - series) < 4e-16 * 2.1875
        dseries = (-0.5 + 0.75 * y - 15.0 * y * y / 16) * 2 * x
        assert abs(t.taup4d(x) - dseries) < 1e-15 * 4.375
    assert t.taup4(0.0) == 35.0 / 16
    assert t.erfbeta(1.3, 0.0) == 2.0 * 1.3 / sqrt(pi) and abs(t.erfbeta(1.3, 0.5) - erf(0.65) / 0.5) < 1e-16


REF_SRC = "/root/reference/msm"


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference's sources are not on this machine")
def test_unmodified_reference_drivers_build_against_include_and_link_the_library(tmp_path):
    """msmExpected.c, nonperiodic.c, ewaldClassical.c and msmPeriodic.c, compiled where they lie with -Iinclude (so
    that "msmLibrary.h" / "ewaldClassical.h" resolve to this repository's headers) and linked to libmsm_amd.so instead
    of msmLibrary.o, reproduce what the real builds printed (fixtures).  They need no GPU: msmPeriodic.c brings its
    own msmperiodic() and takes chooseM / choosebetaNew / choosekmax / chooseL / short_range_periodic / data_read /
    print_stats / the linked list from the library."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from gen_golden import parse_periodic_stdout
    exe = {}
    for name in ("msmExpected", "nonperiodic", "ewaldClassical", "msmPeriodic"):
        exe[name] = tmp_path / name
        res = subprocess.run(["gcc", "-O2", "-std=c99", "-I" + os.path.join(ROOT, "include"), "-o", str(exe[name]),
                              os.path.join(REF_SRC, name + ".c"), "-L" + os.path.join(ROOT, "msm_amd"), "-lmsm_amd",
                              "-Wl,-rpath," + os.path.join(ROOT, "msm_amd"), "-lm"], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]

    def data(name):
        return os.path.join(REF_SRC, "data", "benchmark_%s.txt" % name)

    def run(name, *args):
        res = subprocess.run([str(exe[name])] + [str(a) for a in args], cwd=tmp_path, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-500:]
        return res
    for sysname in ("NaClN8", "changaN8", "CsClN16"):
        fx = O.golden("msmExpected_%s.json" % sysname)
        got = {}
        for ln in run("msmExpected", data(sysname), *fx["args"][1:]).stdout.splitlines():
            m = re.match(r"^(.{0,20}?)\s*:\s*([-+0-9.eEnaif]+)\s*$", ln)
            if m:
                got[m.group(1).strip()] = float(m.group(2))
        assert set(got) == set(fx["scalars"])
        scale = max(abs(v) for k, v in fx["scalars"].items() if k.startswith("("))
        for k, v in fx["scalars"].items():
            assert abs(got[k] - v) <= 1e-13 * max(abs(v), scale if "u" in k else 0.0), (sysname, k, got[k], v)
    for sysname in ("CsClN2", "NaClN8"):
        fx = O.golden("msmPeriodic_%s.json" % sysname)
        res = run("msmPeriodic", data(sysname), 4, 10, 4)
        got = parse_periodic_stdout(res.stdout.splitlines())
        assert got["scalars"] == fx["scalars"] and got["stencil0"] == fx["stencil0"] and got["gridcharge0"] == fx["gridcharge0"]
        assert set(got["stats"]) == set(fx["stats"])
        for k, v in fx["stats"].items():     # print_stats (msmLibrary.c:290-376) comes from the library here
            assert k.startswith("time") or got["stats"][k] == v, (sysname, k, got["stats"][k], v)
        assert float(res.stdout.splitlines()[-1]) == fx["potential"]
    fx = O.golden("nonperiodic_changaN8.json")
    assert run("nonperiodic", data("changaN8")).stdout.strip() == "u is %15s" % fx["printed"]
    fx = O.golden("ewaldClassical_NaClN8.json")
    assert float(run("ewaldClassical", data("NaClN8"), *fx["args"][1:]).stdout.strip().split()[-1]) == fx["total"]


def test_bench_refuses_more_ranks_than_devices():
    """bench.py --gpus N starts N ranks or fails: with fewer devices than ranks (none here) it exits non-zero with a
    message and prints no result line; a launcher whose WORLD_SIZE disagrees with --gpus is refused too."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the devices")
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MSM_AMD_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0
    assert "needs 2 devices" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0"), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "must agree" in r.stderr and "{" not in r.stdout


def test_generated_tables_against_the_reference_header():
    """include/msm_tables.h is generated from this library's own routines, not transcribed; where /root/reference is
    present (this container, not the GPU box) its OMEGA and JN entries are held against msm/msmLibrary.h:150-185: the
    cubic-spline row and JN are the same doubles, the quintic row (which the reference's msmperiodic never reads, p being
    4 there: msmPeriodic.c:100-104) agrees to a few units in the last place - the two programs factorise the same
    symbol in a different order of operations."""
    ref = "/root/reference/msm/msmLibrary.h"
    if not os.path.exists(ref):
        pytest.skip("the reference is not on this machine")

    def table(path, name):
        txt = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
        body = txt[txt.index(name):]
        body = body[body.index("{"):body.index("};")]
        return [float(eval(t)) for t in re.findall(r"-?[0-9][0-9.e+\-/]*", body)]

    ours = os.path.join(ROOT, "include", "msm_tables.h")
    om_r, om_o = np.array(table(ref, "OMEGA[2]")), np.array(table(ours, "OMEGA[2]"))
    assert om_r.shape == om_o.shape == (28,)
    assert np.array_equal(om_r[:14], om_o[:14])
    ulp = np.abs(om_r[14:] - om_o[14:]) / np.spacing(np.abs(om_r[14:]))
    assert ulp.max() <= 32, ulp
    jn_r, jn_o = np.array(table(ref, "JN[2]")), np.array(table(ours, "JN[2]"))
    assert np.array_equal(jn_r, jn_o)
