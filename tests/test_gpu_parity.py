"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the reference's golden outputs.

Tolerances (FP64 everywhere): energies 1e-12 relative to the largest energy component of the system;
accelerations 1e-12 relative RMS (sqrt(sum|f-f_ref|^2 / sum|f_ref|^2)), with a floor of one pair term for
systems whose accelerations cancel by symmetry; grids 1e-12 relative to the grid's largest entry.
"""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

TOL = 1e-12


def rel_rms(a, b, floor=0.0):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    den = max(np.sqrt((b ** 2).sum()), floor * np.sqrt(b.size))
    return np.sqrt(((a - b) ** 2).sum()) / max(den, 1e-300)


def grid_err(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def torch():
    import torch as t
    return t


@pytest.fixture(scope="module")
def msm():
    import msm_amd
    msm_amd.load()
    return msm_amd


def spec_for(desc, nthreads=8):
    return O.SpecPlan.explicit(desc.N, list(desc.A), list(desc.M), desc.p, desc.mu, desc.L, desc.a, desc.beta, desc.kmax,
                               nthreads=nthreads)


def system(name):
    fx = O.golden("msmPeriodic_%s.json" % name)
    s = fx["system"]
    return fx, np.array(s["r"]).reshape(-1, 3), np.array(s["q"]), s["L"]


CASES = [  # (N, box, p, mu, L)
    (3000, (1.0, 1.0, 1.0), 4, 10, 1),     # cells path, compact stencil? M=16: full range
    (3000, (1.0, 1.0, 1.0), 4, 10, 2),
    (3000, (1.0, 1.0, 1.0), 6, 6, 2),
    (20000, (1.0, 1.0, 1.0), 4, 10, 2),    # M=32: compact 15^3 stencil at level 0 -> tiled kernel
    (20000, (1.3, 0.9, 1.1), 4, 10, 2),    # non-cubic cell
    (20000, (1.0, 1.0, 1.0), 6, 6, 2),     # quintic spline on a grid large enough for the tiled anterpolation
    (300, (1.0, 1.0, 1.0), 4, 10, 1),      # small box: all-images pair kernel
    (64, (2.0, 2.0, 2.0), 6, 4, 1),
]


@pytest.mark.parametrize("N,A,p,mu,L", CASES)
def test_stagewise_parity(msm, torch, N, A, p, mu, L):
    r, q = O.neutral_system(N, A, seed=100 + N + p + L)
    r[0] = [0.0, 0.0, 0.0]                       # on the cell corner
    r[1] = [A[0] * 1.25, -0.3 * A[1], A[2] * 2]  # outside the primary cell: must be wrapped
    desc = msm.plan_auto(N, A, p, mu, 4.0, L)
    sp = spec_for(desc)
    eng = msm.Engine(desc)
    dev = "cuda:0"
    rd = torch.as_tensor(r, device=dev)
    qd = torch.as_tensor(q, device=dev)

    # stencils (host side, but part of the product)
    for l in range(L):
        K, first = eng.stencil(l)
        Ko, fo = sp.stencil(l)
        assert first == fo and K.shape == Ko.shape
        assert grid_err(K, Ko) < (1e-11 if p == 6 else 1e-12)
    assert grid_err(eng.stencil(L)[0], sp.stencil_top()) < 1e-12

    # short range
    eng.stage_bin(rd, qd)
    acc_s = torch.zeros((N, 3), dtype=torch.float64, device=dev)
    eng.stage_pairs(acc_s)
    aw, au, ureal = sp.pairs(r, q)
    assert rel_rms(acc_s.cpu().numpy(), aw) < TOL

    # anterpolation
    eng.stage_anterpolate()
    g0 = eng.grid(msm.capi.GRID_Q, 0)
    g0o = sp.anterpolate(r, q)
    assert grid_err(g0, g0o) < TOL
    assert abs(g0.sum() - q.sum()) < 1e-9 * max(np.abs(q).sum(), 1)

    # single grid operators on the oracle's grids
    grids_q = [g0o]
    for l in range(L - 1):
        fine = torch.as_tensor(grids_q[l], device=dev)
        d = eng.grid_dims(l + 1)
        coarse = torch.zeros((d[2], d[1], d[0]), dtype=torch.float64, device=dev)
        co = sp.restrict(l, grids_q[l])
        got = []
        for fused in (0, 1, 4):    # three separable passes / one thread per node / tiles through LDS: the same bits
            eng.set_option("transfer_fused", fused)
            coarse.zero_()
            eng.op_restrict(l, fine, coarse)
            got.append(coarse.cpu().numpy().copy())
            assert grid_err(got[-1], co) < TOL
        assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])
        eng.set_option("transfer_fused", 4)
        grids_q.append(co)
    for l in range(L):
        qg = torch.as_tensor(grids_q[l], device=dev)
        e = torch.zeros_like(qg)
        eo = sp.convolve(l, grids_q[l])
        for tiled, small in ((0, 0), (1, 0), (0, 1)):   # thread-per-node, LDS-tiled, wave-per-node
            eng.set_option("conv_tiled", tiled)
            eng.set_option("conv_small", small)
            e.zero_()
            eng.op_convolve(l, qg, e)
            assert grid_err(e.cpu().numpy(), eo) < TOL, ("level", l, "tiled", tiled, "small", small)
        eng.set_option("conv_tiled", 1)
        eng.set_option("conv_small", 1)
    qg = torch.as_tensor(grids_q[L - 1], device=dev)
    e = torch.zeros_like(qg)
    eng.op_convolve(L, qg, e)
    assert grid_err(e.cpu().numpy(), sp.convolve_top(grids_q[L - 1])) < TOL
    if L > 1:
        rng = np.random.default_rng(3)
        d1, d0 = eng.grid_dims(1), eng.grid_dims(0)
        c = rng.standard_normal((d1[2], d1[1], d1[0]))
        f = rng.standard_normal((d0[2], d0[1], d0[0]))
        got = []
        for fused in (0, 1, 4):
            eng.set_option("transfer_fused", fused)
            fd = torch.as_tensor(f, device=dev)
            eng.op_prolong_add(0, torch.as_tensor(c, device=dev), fd)
            got.append(fd.cpu().numpy().copy())
            assert grid_err(got[-1], sp.prolong_add(0, c, f)) < TOL
        assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])
        eng.set_option("transfer_fused", 4)

    # whole grid stage, interpolation, energies
    eng.stage_grid()
    ref = sp.eval(r, q)
    for l in range(L):
        assert grid_err(eng.grid(msm.capi.GRID_E, l), sp.grid("e", l)) < TOL
    acc_l = torch.zeros((N, 3), dtype=torch.float64, device=dev)
    eng.stage_interpolate(acc_l)
    assert rel_rms(acc_l.cpu().numpy(), ref["along"]) < TOL
    en = eng.stage_energies()
    scale = np.abs(ref["energies"][:6]).max()
    assert np.abs(en - ref["energies"]).max() < TOL * scale, (en, ref["energies"])

    # and the one-call form
    en2, acc = eng.eval(rd, qd)
    assert np.abs(en2 - ref["energies"]).max() < TOL * scale
    assert rel_rms(acc.cpu().numpy(), ref["acc"]) < TOL
    en3, acc3 = eng.eval_host(r, q)
    assert np.array_equal(en3, en2) and np.array_equal(acc3, acc.cpu().numpy())  # deterministic
    eng.close()
    sp.close()


def test_pair_kernel_variants_agree(msm, torch):
    N, A = 30000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=77)
    desc = msm.plan_auto(N, A, 4, 10, 4.0, 2)
    eng = msm.Engine(desc)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    out = []
    for v in (1, 2, 3, 4, 5, 6, 7, 8, 9):  # 1-4: cell lists; 5: cluster pairs with lists; 6: list-free; 7: lists, every pair from both sides; 8: lists walked per particle; 9: the same with survivors split by the half of the cluster they are near
        eng.set_option("pair_variant", v)
        en, acc = eng.eval(rd, qd)
        out.append((en.copy(), acc.cpu().numpy().copy()))
    for k in (1, 2, 3, 4, 5, 6, 7, 8):
        assert rel_rms(out[k][1], out[0][1]) < 1e-13
        assert abs(out[k][0][0] - out[0][0][0]) < 1e-13 * abs(out[0][0][0])
    sp = spec_for(desc)
    aw, au, ureal = sp.pairs(r, q)
    for v in (1, 2, 3, 4, 5, 6, 7, 8, 9):
        eng.set_option("pair_variant", v)
        eng.stage_bin(rd, qd)
        a2 = torch.zeros((N, 3), dtype=torch.float64, device="cuda:0")
        eng.stage_pairs(a2)
        assert rel_rms(a2.cpu().numpy(), aw) < TOL


STENCIL_PLANS = [
    # N, A, p, mu, abar, L, literal
    (512, (1.0, 1.0, 1.0), 4, 10, 4.0, 1, True),      # the reference's own regime: one level, images -5..5
    (64, (1.0, 1.0, 1.0), 4, 10, 4.0, 1, True),
    (3000, (1.0, 1.2, 0.9), 4, 10, 4.0, 2, False),
    (3000, (1.0, 1.0, 1.0), 6, 4, 4.0, 2, False),
    (100000, (1.0, 1.0, 1.0), 4, 10, 4.0, 3, False),  # 48 -> 24 -> 12
    (1000000, (1.0, 1.0, 1.0), 6, 10, 4.0, 4, False), # the benchmark's plan: 96 -> 48 -> 24 -> 12
    (200000, (1.0, 1.0, 1.0), 6, 10, 6.0, 0, False),  # wide stencils
]


@pytest.mark.parametrize("N,A,p,mu,abar,L,literal", STENCIL_PLANS)
def test_device_stencils_match_host(msm, N, A, p, mu, abar, L, literal):
    """Stencils generated on the GPU (msm_amd_create's default) against the host builder: every level and the
    Fourier top level.  Same sums in the same order, so the two can differ only where the device library rounds
    erf / erfc / exp differently (an ulp or two).  Below the coarsest level the kernel is a polynomial (IEEE
    operations only): 1e-14 of the largest entry.  Top level: 1e-13 of the largest entry, which is the sum of the
    absolute values of all terms.  Coarsest level: its table holds values up to T = 2 beta / sqrt(pi) + gamma(0) / a_L
    (erf(beta r) / r near r = 0) and the three quasi-interpolation passes amplify a rounding difference by
    (sum |omega'|)^3, so the bar is 8 ulp * T * (sum |omega'|)^3 - the uncertainty the host values carry themselves.
    The evaluation's own 1e-12 parity tests run on the device-built stencils (the default)."""
    from msm_amd import capi
    lib = msm.load()
    desc = msm.plan_auto(N, A, p, mu, abar, L)
    host = msm.HostPlan(desc, literal=literal)
    old = lib.msm_amd_get_stencil_backend()
    try:
        lib.msm_amd_set_stencil_backend(capi.STENCILS_DEVICE)
        assert lib.msm_amd_get_stencil_backend() == capi.STENCILS_DEVICE
        eng = msm.Engine(desc, literal=literal)
    finally:
        lib.msm_amd_set_stencil_backend(old)
    for level in range(desc.L + 1):
        kd, fd = eng.stencil(level)
        kh, fh = host.stencil(level)
        assert kd.shape == kh.shape and tuple(fd) == tuple(fh)
        if level == desc.L - 1:
            w = host.wprime()
            amp = (abs(w[0]) + 2 * np.abs(w[1:]).sum()) ** 3
            T = 2 * desc.beta / np.sqrt(np.pi) + host.scalars()["gamma0"] / (desc.a * 2 ** (desc.L - 1))
            tol = 8 * np.finfo(float).eps * T * amp
        else:
            tol = (1e-13 if level == desc.L else 1e-14) * np.abs(kh).max()
        assert np.abs(kd - kh).max() <= tol, "level %d of %d" % (level, desc.L)
    eng.close()
    host.close()


@pytest.mark.parametrize("clustered", [False, True])
def test_cluster_sorts_agree(msm, torch, clustered, monkeypatch):
    """The two-level sort (column buckets ranked in LDS) and the sort with one global counter per fine bin store the
    particles in the same slots: energies and accelerations are bitwise equal.  The clustered system puts a
    third of the particles into a few columns, so that their buckets take the long-bucket path of the column pass."""
    N, A = 60000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=5)
    if clustered:
        rng = np.random.default_rng(3)
        k = N // 3
        r[:k, 0] = 0.40 + 0.02 * rng.random(k)
        r[:k, 1] = 0.70 + 0.02 * rng.random(k)
    desc = msm.plan_auto(N, A, 6, 10, 4.0, 2)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    out = []
    for mode in ("1", "0"):
        monkeypatch.setenv("MSM_AMD_SORT", mode)
        eng = msm.Engine(desc)
        en, acc = eng.eval(rd, qd)
        out.append((np.array(en), acc.cpu().numpy().copy()))
        eng.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    if not clustered:
        sp = spec_for(desc)
        ref = sp.eval(r, q)
        assert rel_rms(out[0][1], ref["acc"]) < TOL


@pytest.mark.parametrize("p,M", [(4, (28, 36, 26)), (6, (26, 24, 41)), (4, (24, 24, 24)), (6, (32, 40, 48))])
def test_anterpolation_tiled(msm, torch, p, M):
    """LDS-tiled anterpolation (grids >= 24 per axis) against the oracle and against the gather kernel, on grid
    sizes that are not multiples of the tile edge, with particles on cell faces and on the period boundary."""
    N, A = 15000, (1.0, 1.2, 0.9)
    r, q = O.neutral_system(N, A, seed=5 + p)
    r[:50] = np.floor(r[:50] / np.array(A) * np.array(M)) * np.array(A) / np.array(M)   # on grid nodes
    r[50] = [0.0, 0.0, 0.0]
    r[51] = [A[0] * (1 - 1e-17), A[1] * (1 - 1e-16), A[2]]
    d = msm.plan_explicit(N, A, M, p, 6, 1, 0.12, 20.0, 8)
    eng = msm.Engine(d)
    sp = spec_for(d)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    want = sp.anterpolate(r, q)
    got = {}
    # tiled with sums per column in registers (order 6), tiled with one LDS update per particle and plane, gather
    for key, (tiled, cols) in {1: (1, 1), 2: (1, 0), 0: (0, 0)}.items():
        eng.set_option("anterp_tiled", tiled)
        eng.set_option("anterp_cols", cols)
        eng.stage_bin(rd, qd)
        eng.stage_anterpolate()
        got[key] = eng.grid(msm.capi.GRID_Q, 0)
        assert grid_err(got[key], want) < TOL, key
    assert grid_err(got[1], got[0]) < 1e-13 and grid_err(got[2], got[0]) < 1e-13
    eng.set_option("anterp_tiled", 1)
    eng.set_option("anterp_cols", 1)
    eng.stage_bin(rd, qd)
    eng.stage_anterpolate()
    assert np.array_equal(eng.grid(msm.capi.GRID_Q, 0), got[1])   # reproducible
    eng.close()
    sp.close()


@pytest.mark.parametrize("abar,p", [(3.61, 4), (4.03, 6), (4.5, 4), (5.0, 6), (5.5, 4), (6.0, 4)])   # 17, 19, 21, 23, 25, 29 taps
def test_wide_stencils(msm, torch, abar, p):
    """Stencils wider than the LDS tile's full halo (17..23 taps per axis) are applied in several staging
    passes by the multi-level convolution kernel: grids and totals against the oracle."""
    N, A = 20000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=31)
    desc = msm.plan_auto(N, A, p, 10, abar, 2)
    eng = msm.Engine(desc)
    sizes = [eng.stencil(l)[0].shape[0] for l in range(desc.L)]
    assert sizes[0] >= 17, sizes
    sp = spec_for(desc)
    ref = sp.eval(r, q)
    en, acc = eng.eval(r, q)
    for l in range(desc.L):
        assert grid_err(eng.grid(msm.capi.GRID_E, l), sp.grid("e", l)) < TOL, (l, sizes)
    assert np.abs(en - ref["energies"]).max() < TOL * np.abs(ref["energies"][:6]).max()
    assert rel_rms(acc.cpu().numpy(), ref["acc"]) < TOL
    eng.close()
    sp.close()


@pytest.mark.parametrize("a_over_h,p,taps", [(4.0, 6, 15), (4.0, 4, 15), (4.6, 6, 19)])
def test_slab_convolution_kernel_on_whole_grids(msm, torch, a_over_h, p, taps):
    """The convolution kernel of the slab ranks (tiles of 8 planes, the stencil columns shared out over four groups of
    waves) applied to whole grids, with and without the mirror columns taken together: every level against the oracle."""
    N, A, M, L = 20000, (1.0, 1.0, 1.0), (48, 48, 48), 2
    r, q = O.neutral_system(N, A, seed=41)
    lib = msm.load()
    a = a_over_h / M[0]
    hL = (2.0 ** (L - 1)) * A[0] / M[0]
    beta = lib.choosebetaNew(1e-12, (2.0 ** L) * a, hL)
    kmax = int(np.ceil(lib.choosekmax(1e-12, beta, hL)))
    desc = msm.plan_explicit(N, A, M, p, 10, L, a, beta, kmax)
    eng = msm.Engine(desc)
    assert eng.stencil(0)[0].shape[0] == taps
    sp = spec_for(desc)
    ref = sp.eval(r, q)
    for sym in (2, 0):
        eng.set_option("conv_groups", 2)
        eng.set_option("conv_sym", sym)
        en, acc = eng.eval(r, q)
        for l in range(desc.L):
            assert grid_err(eng.grid(msm.capi.GRID_E, l), sp.grid("e", l)) < TOL, (l, sym)
        assert np.abs(en - ref["energies"]).max() < TOL * np.abs(ref["energies"][:6]).max()
        assert rel_rms(acc.cpu().numpy(), ref["acc"]) < TOL
    eng.close()
    sp.close()


@pytest.mark.parametrize("a_over_h,p,width", [(6.3, 4, 25), (7.6, 6, 31), (8.8, 4, 35)])
def test_very_wide_stencils(msm, torch, a_over_h, p, width):
    """Stencils beyond 23 taps per axis (a grid of half the usual spacing, or a long cutoff) are cut into cubes that
    the multi-level convolution kernel applies one after the other: grids and totals against the oracle."""
    N, A, M, L = 12000, (1.0, 1.0, 1.0), (48, 48, 48), 2
    r, q = O.neutral_system(N, A, seed=37)
    lib = msm.load()
    a = a_over_h / M[0]
    hL = (2.0 ** (L - 1)) * A[0] / M[0]
    beta = lib.choosebetaNew(1e-12, (2.0 ** L) * a, hL)
    kmax = int(np.ceil(lib.choosekmax(1e-12, beta, hL)))
    desc = msm.plan_explicit(N, A, M, p, 10, L, a, beta, kmax)
    eng = msm.Engine(desc)
    assert eng.stencil(0)[0].shape[0] == width
    sp = spec_for(desc)
    ref = sp.eval(r, q)
    en, acc = eng.eval(r, q)
    for l in range(desc.L):
        assert grid_err(eng.grid(msm.capi.GRID_E, l), sp.grid("e", l)) < TOL, l
    assert np.abs(en - ref["energies"]).max() < TOL * np.abs(ref["energies"][:6]).max()
    assert rel_rms(acc.cpu().numpy(), ref["acc"]) < TOL
    eng.close()
    sp.close()


def test_eval_enqueue(msm, torch):
    """The non-synchronising entry point: several evaluations queued back to back, energies in pinned slots."""
    N, A = 30000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=3)
    eng = msm.Engine(msm.plan_auto(N, A, 4, 10, 4.0, 2))
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    en, acc = eng.eval(rd, qd)
    ring = torch.zeros((3, 7), dtype=torch.float64).pin_memory()
    accs = [torch.zeros_like(acc) for _ in range(3)]
    for k in range(3):
        eng.eval_enqueue(rd, qd, accs[k], ring[k])
    torch.cuda.synchronize()
    for k in range(3):
        assert np.array_equal(ring[k].numpy(), en) and torch.equal(accs[k], acc)
    eng.close()


def test_conventions(msm, torch):
    """ACC_LITERAL = msmLibrary.c:168-176 / msmPeriodic.c:1158 weighting, ACC_FORCE = -q_i * gradient."""
    N, A = 2500, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=9)
    q = q * (1.0 + 0.3 * np.arange(N) / N)  # unequal magnitudes so that the conventions differ
    base = msm.plan_auto(N, A, 4, 10, 4.0, 1)
    sp = spec_for(base)
    ref = sp.eval(r, q)
    fs = ref["acc"] - ref["along"]
    want = {msm.capi.ACC_GRADIENT: ref["acc"],
            msm.capi.ACC_LITERAL: 0.5 * (q[:, None] * ref["fshort_unw"] + fs) + q[:, None] * ref["along"],
            msm.capi.ACC_FORCE: -q[:, None] * ref["acc"]}
    for conv, w in want.items():
        d = msm.plan_auto(N, A, 4, 10, 4.0, 1)
        d.acc_convention = conv
        eng = msm.Engine(d)
        en, acc = eng.eval(r, q)
        assert rel_rms(acc.cpu().numpy(), w) < TOL
        eng.close()


ALL = ["CsClN2", "NaClN8", "NaNaN8", "changaN8", "CsClN16", "NaClN64", "NaNaN64", "changaN64", "CsClN128",
       "NaClN512", "NaNaN512", "changaN512"]
LABELS = (("potential_short_real", "(A) ushort_real"), ("potential_short_self", "(B) ushort_self"),
          ("potential_short_csr", "(C) ushort_csr"), ("potential_long_real", "(D) ulong_real"),
          ("potential_long_fourier", "(E) ulong_four"), ("potential_long_self", "(F) ulong_self"),
          ("potential_total", "utotal"))


@pytest.mark.parametrize("name", ALL)
def test_dropin_msmperiodic_matches_reference(msm, name):
    """msmperiodic() on the GPU == the reference's msmPeriodic on every system of msm/data (p=4 mu=10 abar=4):
    parameters, (A)..(F), utotal, msm.acc / msm.acclong / msm.accshort, and the +0.5 shift of the caller's r."""
    fx, r, q, A = system(name)
    r_in = r.copy()
    st, u, f, r_after = msm.msmperiodic(r_in, q, 4, 4.0, 10, A, mode=msm.capi.MODE_LITERAL, quiet=True)
    assert np.array_equal(r_after, r + 0.5)  # msmPeriodic.c:75-76
    s = fx["stats"]
    assert st.Mx == int(s["Mx My Mz"].split()[0]) and st.kmax == int(s["kmax"]) and st.L == 1
    assert st.beta == float(s["beta"]) and st.a == float(s["a"])
    comps = [abs(fx["scalars"][lab]) for _, lab in LABELS[:6]]
    scale = max(comps)
    for field, lab in LABELS:
        ref = fx["scalars"][lab]
        assert abs(getattr(st, field) - ref) <= TOL * scale, (lab, getattr(st, field), ref)
    assert abs(u - fx["potential"]) <= TOL * scale
    lit = O.literal_run(r, q, 4, 4.0, 10, A)
    floor = np.abs(q).max() / lit["a"] ** 2   # one pair term
    assert rel_rms(f, np.array(fx["acc"]).reshape(-1, 3), floor) < TOL
    assert rel_rms(f, lit["facc"].reshape(-1, 3), floor) < TOL


def test_dropin_keeps_its_plan(msm):
    """msmperiodic() reuses the plan of its last call when the next call needs the same one, builds a new one when it
    does not, and gives the same numbers either way."""
    lib = msm.load()
    N, A = 6000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=12)
    out = []
    lib.msm_amd_release_cached_plan()
    for k in range(3):
        if k == 2:
            lib.msm_amd_release_cached_plan()
        st, u, f, _ = msm.msmperiodic(r.copy(), q, 4, 4.0, 10, A, mode=msm.capi.MODE_SCALABLE)
        out.append((u, f.copy(), lib.msm_amd_plan_builds()))
    assert out[0][0] == out[1][0] == out[2][0]
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][1], out[2][1])
    assert out[1][2] == out[0][2]                # the second call built nothing
    assert out[2][2] == out[1][2] + 1            # the third, after the release, built the plan again
    r2, q2 = O.neutral_system(5000, A, seed=13)   # a different system: a different plan
    st, u2, f2, _ = msm.msmperiodic(r2.copy(), q2, 4, 4.0, 10, A, mode=msm.capi.MODE_SCALABLE)
    eng = msm.Engine(msm.plan_auto(5000, A, 4, 10, 4.0, 0))
    en, acc = eng.eval_host(r2, q2)
    assert abs(u2 - en[6]) <= 1e-12 * np.abs(en[:6]).max() and rel_rms(f2, acc) < 1e-12
    eng.close()
    assert lib.msm_amd_last_call_mode() == msm.capi.MODE_SCALABLE   # what the last msmperiodic() call did
    lib.msm_amd_release_cached_plan()


def test_dropin_mode_from_the_environment(msm, tmp_path):
    """A driver that never calls msm_amd_set_mode() chooses the mode with MSM_AMD_MODE; in auto mode the library says
    once, on stderr, that a system above 4096 particles took the scalable path."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import msm_amd, oracle_lib as O
        lib = msm_amd.capi.load()
        r, q = O.neutral_system(%%d, (1.0, 1.0, 1.0), seed=3)
        st, u, f, _ = msm_amd.msmperiodic(r.copy(), q, 4, 4.0, 10, (1.0, 1.0, 1.0), mode=None, quiet=%%d)
        print("MODE", lib.msm_amd_last_call_mode(), repr(float(u)))
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    def run(n, env_mode, quiet=1):
        env = dict(os.environ)
        env.pop("MSM_AMD_MODE", None)
        if env_mode:
            env["MSM_AMD_MODE"] = env_mode
        p = subprocess.run([sys.executable, "-c", code % (n, quiet)], env=env, capture_output=True, text=True, timeout=110,
                           cwd=str(tmp_path))
        assert p.returncode == 0, p.stderr[-2000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("MODE")][0].split()
        return int(line[1]), float(line[2]), p.stderr
    m_auto, u_auto, err = run(512, None)
    assert m_auto == msm.capi.MODE_LITERAL and "scalable mode" not in err
    m_sc, u_sc, _ = run(512, "scalable")
    assert m_sc == msm.capi.MODE_SCALABLE and u_sc != u_auto   # (the literal index conventions are not the scalable ones)
    m_big, _, err = run(5000, None, quiet=0)
    assert m_big == msm.capi.MODE_SCALABLE and "scalable mode" in err


def test_dropin_stagewise_literal(msm, torch):
    """Literal mode stage by stage against the literal mirror: grid charges, edir, etop, fshort, flong."""
    for name in ("changaN64", "CsClN16", "changaN512"):
        fx, r, q, A = system(name)
        lit = O.literal_run(r, q, 4, 4.0, 10, A)
        d = msm.plan_auto(len(q), A, 4, 10, 4.0, 1)
        d.acc_convention = msm.capi.ACC_LITERAL
        eng = msm.Engine(d, literal=True)
        K, first = eng.stencil(0)
        assert first == lit["slmin"]
        assert grid_err(K.ravel(), lit["sl"]) < 1e-11
        rs = r + 0.5
        eng.stage_bin(torch.as_tensor(rs, device="cuda:0"), torch.as_tensor(q, device="cuda:0"))
        eng.stage_anterpolate()
        assert grid_err(eng.grid(msm.capi.GRID_Q, 0).ravel(), lit["qgrd"]) < TOL
        eng.stage_grid()
        assert grid_err(eng.grid(msm.capi.GRID_E, 0).ravel(), lit["edir"]) < 1e-11
        assert grid_err(eng.grid(msm.capi.GRID_EFOUR, 0).ravel(), lit["etop"]) < 1e-11
        fs = torch.zeros((len(q), 3), dtype=torch.float64, device="cuda:0")
        fl = torch.zeros_like(fs)
        eng.stage_pairs(fs)
        eng.stage_interpolate(fl)
        floor = np.abs(q).max() / lit["a"] ** 2
        assert rel_rms(fs.cpu().numpy(), lit["fshort"].reshape(-1, 3), floor) < TOL
        assert rel_rms(fl.cpu().numpy(), lit["flong"].reshape(-1, 3), floor * 1e-3) < 1e-10
        eng.close()


SIMPLE = [("msmSimple_changaN8_nu4_mu10_M2.json", "changaN8", 1), ("msmSimple_changaN8_nu6_mu10_M2.json", "changaN8", 1),
          ("msmSimple_changaN8_nu4_mu10_M3.json", "changaN8", 1), ("msmSimple_changaN8_nu6_mu6_M4.json", "changaN8", 1),
          ("msmSimple_CsClN16_nu4_mu10_M3.json", "CsClN16", 1), ("msmSimple_changaN64_nu4_mu10_M4.json", "changaN8", 2)]


@pytest.mark.parametrize("fn,sysname,copies", SIMPLE)
def test_matches_msmSimple_reference(msm, fn, sysname, copies):
    """HIP path == the reference's msmSimple (2-level, long-range forces, v=4/6, odd and even grids).
    msmSimple's own image sums stop at 1e-8 relative change (msmSimple.c:16,1060): 2e-10 on grid terms."""
    from test_oracle_golden import simple_case
    fx, r, q, A = simple_case(fn, sysname, copies)
    sc, args = fx["scalars"], fx["args"]
    mu, nu, M = args[2], args[3], args[4:7]
    d = msm.plan_explicit(len(q), A, M, nu, mu, 1, sc["a"], sc["beta"], int(sc["kmax"]))
    eng = msm.Engine(d)
    en, acc = eng.eval(r, q)
    scale = abs(sc["ushort_real"]) + abs(sc["ulong_self"])
    for i, k in enumerate(["ushort_real", "ushort_self", "ushort_csr", "ulong_real", "ulong_four", "ulong_self", "utotal"]):
        tol = 1e-12 if k in ("ushort_real", "ushort_self", "ushort_csr", "ulong_self") else 2e-10
        assert abs(en[i] - sc[k]) <= tol * max(abs(sc[k]), scale * 1e-3), (k, en[i], sc[k])
    ft = np.array(fx["acc"]["ftotal"]).reshape(-1, 3)
    assert rel_rms(acc.cpu().numpy(), ft, np.abs(q).max() / sc["a"] ** 2) < 2e-10
    eng.close()


def test_edge_cases(msm, torch):
    # a single particle: only self terms
    d = msm.plan_explicit(1, (1.0, 1.0, 1.0), (4, 4, 4), 4, 10, 1, 0.9, 3.0, 6)
    eng = msm.Engine(d)
    sp = spec_for(d)
    r, q = np.array([[0.3, 0.9, 0.5]]), np.array([2.0])
    en, acc = eng.eval(r, q)
    ref = sp.eval(r, q)
    assert np.abs(en - ref["energies"]).max() < TOL * np.abs(ref["energies"][:6]).max()
    assert np.abs(acc.cpu().numpy() - ref["acc"]).max() < 1e-10 * np.abs(q).max() / 0.81
    eng.close()
    # strongly clustered input: most cells empty, one cell crowded (several i-passes per cell)
    N, A = 4000, (1.0, 1.0, 1.0)
    rng = np.random.default_rng(4)
    r = 0.5 + 0.02 * rng.standard_normal((N, 3))
    r[: N // 4] = rng.random((N // 4, 3))
    q = np.where(np.arange(N) % 2 == 0, 1.0, -1.0)
    d = msm.plan_explicit(N, A, (16, 16, 16), 4, 10, 2, 0.22, 6.0, 11)
    eng = msm.Engine(d)
    sp = spec_for(d)
    ref = sp.eval(r, q)
    for v in (1, 2, 3, 4, 5, 6, 7, 9, 8):   # the crowded region overflows the cluster lists: the list-free kernel takes over
        eng.set_option("pair_variant", v)
        en, acc = eng.eval(r, q)
        assert np.abs(en - ref["energies"]).max() < TOL * np.abs(ref["energies"][:6]).max(), v
        assert rel_rms(acc.cpu().numpy(), ref["acc"]) < TOL, v
    eng.close()
    # invalid plans fail loudly
    bad = msm.plan_explicit(10, (1.0, 1.0, 1.0), (6, 6, 6), 4, 10, 3, 0.3, 3.0, 5)  # 6 not divisible by 4
    with pytest.raises(msm.MsmAmdError):
        msm.Engine(bad)
    bad = msm.plan_explicit(10, (1.0, 1.0, 1.0), (8, 8, 8), 5, 10, 1, 0.3, 3.0, 5)  # p = 5
    with pytest.raises(msm.MsmAmdError):
        msm.Engine(bad)


@pytest.mark.parametrize("p", [6, 4])
def test_full_size_properties(msm, torch, p):
    """N = 1M, L = 4 (the benchmark configuration, p = 6, and its cubic variant): the whole evaluation against the CPU
    oracle, and properties that hold at any size."""
    N, A = 1000000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=20261111 + 1)
    desc = msm.plan_auto(N, A, p, 10, 4.0, 0)
    assert list(desc.M) == [96, 96, 96] and desc.L == 4
    eng = msm.Engine(desc)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    en, acc = eng.eval(rd, qd)
    acc = acc.cpu().numpy()
    assert np.isfinite(en).all() and np.isfinite(acc).all()
    # the oracle on the same million particles (a few seconds on the box's cores): every energy component to 1e-12 of
    # the largest one - the total is what cancellation leaves of components sixty times larger, so relative to itself it
    # is held to 1e-10 - and the accelerations to 1e-12 relative RMS
    ref = spec_for(desc, nthreads=16).eval(r, q)
    scale = np.abs(ref["energies"][:6]).max()
    assert np.abs(en - ref["energies"]).max() <= TOL * scale, (en, ref["energies"])
    assert abs(en[6] - ref["energies"][6]) <= 1e-10 * abs(ref["energies"][6])
    assert rel_rms(acc, ref["acc"]) < TOL
    sub = np.random.default_rng(3).choice(N, 2000, replace=False)       # and no single particle is off either
    assert np.abs(acc[sub] - ref["acc"][sub]).max() <= 1e-11 * np.abs(ref["acc"]).max()
    # charge conservation through anterpolation and every restriction
    for l in range(desc.L):
        assert abs(eng.grid(msm.capi.GRID_Q, l).sum() - q.sum()) < 1e-8
    # the pair sum is antisymmetric: sum_i q_i acc_short_i = 0; the grid part obeys it to interpolation accuracy
    eng.stage_bin(rd, qd)
    a_s = torch.zeros((N, 3), dtype=torch.float64, device="cuda:0")
    eng.stage_pairs(a_s)
    tot = (qd[:, None] * a_s).sum(0).cpu().numpy()
    assert np.abs(tot).max() < 1e-9 * float((qd[:, None] * a_s).abs().sum())
    # bitwise reproducible
    en2, acc2 = eng.eval(rd, qd)
    assert np.array_equal(en, en2) and np.array_equal(acc, acc2.cpu().numpy())
    # parts of the grid branch on a second stream beside the pair branch (1: all of it, 2: the anterpolation, 3: and the
    # coarse levels, 4: as 3, from the end of the list build): same kernels, same order of every sum
    for mode in (1, 2, 3, 4):
        eng.set_option("overlap", mode)
        for _ in range(3):
            en2, acc2 = eng.eval(rd, qd)
            assert np.array_equal(en, en2) and np.array_equal(acc, acc2.cpu().numpy()), mode
    eng.set_option("overlap", 0)
    # the pair kernel variants agree at full size
    for v in (1, 2, 3, 4, 6, 7, 9, 8, 5):
        eng.set_option("pair_variant", v)
        en3, acc3 = eng.eval(rd, qd)
        assert rel_rms(acc3.cpu().numpy(), acc) < 1e-13, v
    # invariance under a lattice translation of some particles and under relabelling
    r2 = r.copy()
    r2[::7] += np.array([1.0, -2.0, 3.0])
    en4, acc4 = eng.eval(torch.as_tensor(r2, device="cuda:0"), qd)
    assert abs(en4[6] - en3[6]) < 1e-9 * np.abs(en3[:6]).max()
    perm = np.random.default_rng(0).permutation(N)
    en5, acc5 = eng.eval(torch.as_tensor(r[perm], device="cuda:0"), torch.as_tensor(q[perm], device="cuda:0"))
    assert abs(en5[6] - en3[6]) < 1e-10 * np.abs(en3[:6]).max()
    assert rel_rms(acc5.cpu().numpy(), acc3.cpu().numpy()[perm]) < 1e-10
    # sampled check of the accelerations against the Ewald sum on a subset (accuracy of the method itself)
    eng.close()


@pytest.mark.parametrize("p,M", [(4, 32), (6, 32), (4, 48), (6, 48)])
def test_two_scale_identities(msm, torch, p, M):
    """The level transfers pinned by the B-spline two-scale relation Phi_2h(x) = sum_j J_j Phi_h(x - j h)
    (msmLibrary.h:152-153), which no reference run exercises (chooseL is 1 there):
      restriction of the charges anterpolated on the grid M  ==  the charges anterpolated on the grid M/2,
      interpolation on M of a prolongated potential  ==  interpolation on M/2 of the potential itself
    (values and gradients), both to rounding and with no oracle in between.  The coarse grid of M = 32 takes the
    small-grid spline kernels, every other grid the tiled ones."""
    N, A = 20000, (1.0, 1.1, 0.9)
    rng = np.random.default_rng(11)
    r = rng.random((N, 3)) * np.array(A)
    q = rng.standard_normal(N)
    a = 0.12
    fine = msm.Engine(msm.plan_explicit(N, A, (M, M, M), p, 4, 2, a, 8.0, 4))
    coarse = msm.Engine(msm.plan_explicit(N, A, (M // 2,) * 3, p, 4, 1, a, 8.0, 4))
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    fine.stage_bin(rd, qd)
    fine.stage_anterpolate()
    coarse.stage_bin(rd, qd)
    coarse.stage_anterpolate()
    qf = torch.as_tensor(fine.grid(msm.capi.GRID_Q, 0), device="cuda:0")
    qc = torch.zeros((M // 2,) * 3, dtype=torch.float64, device="cuda:0")
    fine.op_restrict(0, qf, qc)
    ref = coarse.grid(msm.capi.GRID_Q, 0)
    assert np.abs(qc.cpu().numpy() - ref).max() < 1e-13 * np.abs(ref).max()
    ec = rng.standard_normal((M // 2,) * 3)
    ef = torch.zeros((M, M, M), dtype=torch.float64, device="cuda:0")
    fine.op_prolong_add(0, torch.as_tensor(ec, device="cuda:0"), ef)
    fine.set_grid(msm.capi.GRID_E, 0, ef.cpu().numpy())
    coarse.set_grid(msm.capi.GRID_E, 0, ec)
    af = torch.zeros((N, 3), dtype=torch.float64, device="cuda:0")
    ac = torch.zeros_like(af)
    fine.stage_interpolate(af)
    coarse.stage_interpolate(ac)
    assert rel_rms(af.cpu().numpy(), ac.cpu().numpy()) < 1e-13
    fine.close()
    coarse.close()


def test_accuracy_sweep_point(msm, torch):
    """One point of the cutoff x order x spacing sweep (abar = 5, quintic spline, the reference's grid choice at 100k
    particles: 48^3, three levels) against the converged Ewald sum of the same system on the GPU: force error and
    energy error at the level the sweep tables report for this point (3.5e-5, 2.1e-4), and better than abar = 4."""
    N, A = 100000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=20261111 + 4)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    u5, _, grad = msm.ewald(rd, qd, A, 22.4, 0.25, 40)
    grad = grad.cpu().numpy()
    err = {}
    for abar in (4.0, 5.0):
        desc = msm.plan_auto(N, A, 6, 10, abar, 3)
        assert list(desc.M) == [48, 48, 48]
        eng = msm.Engine(desc)
        en, acc = eng.eval(rd, qd)
        err[abar] = (rel_rms(acc.cpu().numpy(), grad), abs(en[6] - u5[4]) / abs(u5[4]))
        eng.close()
    assert err[5.0][0] < 6e-5 and err[5.0][1] < 1e-3, err
    assert err[5.0][0] < 0.5 * err[4.0][0], err


def test_accuracy_vs_ewald_100k(msm, torch):
    """Config 2 (100k, cubic B-spline, 3 levels): relative energy error and force RMS error against a
    full Ewald sum evaluated for a random subset of particles."""
    N, A = 100000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=20261111 + 2)
    desc = msm.plan_auto(N, A, 4, 10, 4.0, 3)
    assert list(desc.M) == [48, 48, 48]
    eng = msm.Engine(desc)
    en, acc = eng.eval(r, q)
    acc = acc.cpu().numpy()
    nsub = 200
    # put the subset first (or_ewald evaluates the first nsub particles against all N)
    uE, aE = O.ewald(r, q, A, alpha=20.0, rcut=0.25, kcut=28, nsub=nsub, nthreads=16)
    ferr = rel_rms(acc[:nsub], aE)
    assert ferr < 2e-3, ferr
    eng.close()


def open_system(N, seed, box=(1.0, 1.0, 1.0), shift=(0.0, 0.0, 0.0)):
    rng = np.random.default_rng(seed)
    r = rng.random((N, 3)) * np.array(box) + np.array(shift)
    q = np.where(np.arange(N) % 2 == 0, 1.0, -1.0)
    return r, q


OPEN_CASES = [  # N, box of the particles, shift, p, mu, h, a, L
    (2000, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), 4, 10, 1 / 12, 4 / 12, 1),
    (4000, (1.0, 1.0, 1.0), (-3.0, 2.5, 10.0), 4, 10, 1 / 16, 4 / 16, 2),     # away from the origin
    (6000, (1.0, 0.6, 1.3), (0.2, 0.0, -0.4), 6, 10, 1 / 20, 4 / 20, 2),      # non-cubic cloud, quintic spline
    (20000, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), 6, 6, 1 / 28, 3.5 / 28, 3),     # tiled spline kernels, three levels
]


@pytest.mark.parametrize("N,box,shift,p,mu,h,a,L", OPEN_CASES)
def test_open_boundary_parity(msm, torch, N, box, shift, p, mu, h, a, L):
    """Open-boundary MSM (msm3d of msm/msmLibrary.h:216, defined nowhere in the reference: parity unpinned by reference
    output) against the CPU oracle's open mode to 1e-12, stage by stage and as one evaluation, and against the
    direct sum of msm/nonperiodic.c:12-22 within the method's accuracy."""
    r, q = open_system(N, 40 + N, box, shift)
    desc = msm.plan_open(N, r.min(axis=0), r.max(axis=0), p, mu, h, a, L)
    assert desc.open_boundary == 1 and desc.L == L
    org = np.array(list(desc.origin))
    sp = O.SpecPlan.explicit(N, list(desc.A), list(desc.M), p, mu, L, a, 1.0, 0, nthreads=8, open_boundary=True)
    rs = r - org                                   # the oracle works in the cell's own coordinates
    assert rs.min() > 0 and (rs.max(axis=0) < np.array(list(desc.A))).all()
    eng = msm.Engine(desc)
    for l in range(L):
        K, first = eng.stencil(l)
        Ko, fo = sp.stencil(l)
        assert first == fo and K.shape == Ko.shape
        assert grid_err(K, Ko) < (1e-11 if p == 6 else 1e-12)
    dev = "cuda:0"
    rd, qd = torch.as_tensor(r, device=dev), torch.as_tensor(q, device=dev)
    eng.stage_bin(rd, qd)
    eng.stage_anterpolate()
    g0o = sp.anterpolate(rs, q)
    assert grid_err(eng.grid(msm.capi.GRID_Q, 0), g0o) < TOL
    ref = sp.eval(rs, q)
    # coarsest level alone: every node against every node, no wrap-around
    qL = sp.grid("q", L - 1)
    e = torch.zeros(qL.shape, dtype=torch.float64, device=dev)
    eng.op_convolve(L - 1, torch.as_tensor(qL, device=dev), e)
    assert grid_err(e.cpu().numpy(), sp.convolve(L - 1, qL)) < TOL
    eng.stage_grid()
    for l in range(L):
        assert grid_err(eng.grid(msm.capi.GRID_E, l), sp.grid("e", l)) < TOL
    en, acc = eng.eval(rd, qd)
    scale = np.abs(ref["energies"][:6]).max()
    assert np.abs(en - ref["energies"]).max() < TOL * scale, (en, ref["energies"])
    assert en[1] == 0.0 and en[2] == 0.0 and en[4] == 0.0      # no self-image, csr or Fourier terms
    acc = acc.cpu().numpy()
    assert rel_rms(acc, ref["acc"]) < TOL
    # physics: the direct sum
    u, f = msm.direct(r, q)
    # (at a = 4 h the periodic engine's errors against Ewald are 5e-3 / 7e-4 in the energy and 4e-4 / 1.4e-4 in the
    # forces for p = 4 / 6; the energy is a small difference of large terms)
    assert abs(en[6] - u) < (1e-2 if p == 4 else 3e-3) * abs(u)
    assert rel_rms(acc, f) < (2e-3 if p == 4 else 5e-4)
    eng.close()
    sp.close()


def test_msm3d_dropin(msm):
    """msm3d() through the C ABI on host arrays: same numbers as the handle layer on the plan it builds, caller's
    arrays untouched, direct sum within the method's accuracy; a bad s is refused like a bad p in msmperiodic."""
    N, p, mu, h, a = 30000, 6, 10, 1 / 32, 4 / 32
    r, q = open_system(N, 9, (1.0, 1.0, 0.8), (5.0, -1.0, 0.3))
    r0 = r.copy()
    u, f, st = msm.msm3d(r, q, p, h, a, mu)
    assert np.array_equal(r, r0)
    ud, fd = msm.direct(r, q)
    assert abs(u - ud) < 3e-3 * abs(ud) and rel_rms(f, fd) < 5e-4
    assert st.N == N and st.p == p and st.L >= 2 and st.potential_total == u
    desc = msm.plan_open(N, r.min(axis=0), r.max(axis=0), p, mu, h, a, 0)
    assert st.L == desc.L and st.Mx == desc.M[0]
    eng = msm.Engine(desc)
    en, acc = eng.eval_host(r, q)
    assert en[6] == u and np.array_equal(acc, f)
    eng.close()
    u2, f2, st2 = msm.msm3d(r, q, p, h, a, mu, s=3)
    assert u2 == 0.0 and st2.potential_total == 0.0


@pytest.mark.parametrize("name", ["changaN8", "changaN64", "changaN512"])
def test_msm3d_on_reference_systems(msm, name):
    """msm3d on the systems the real msm/nonperiodic was run on (charges forced to 1 as nonperiodic.c:9-10 does):
    energy against the reference's printed value and gradient against direct(), within the method's accuracy; edge
    cases: two particles, one particle, coincident bounding-box faces."""
    fx = O.golden("nonperiodic_%s.json" % name)
    s = fx["system"]
    r = np.array(s["r"]).reshape(-1, 3)
    q = np.ones(s["N"])
    ext = (r.max(axis=0) - r.min(axis=0)).max()
    h = ext / max(2.0, s["N"] ** (1.0 / 3.0))
    u, f, st = msm.msm3d(r, q, 6, h, 4.0 * h, 10)
    ud, fd = msm.direct(r, q)
    assert abs(u - fx["u"]) < 2e-3 * abs(fx["u"]) and abs(ud - fx["u"]) <= 0.6e-8 * abs(fx["u"])
    assert rel_rms(f, fd) < 2e-3
    # two particles closer than the cutoff, then further apart than the cutoff; a single particle
    for d in (0.3, 3.0):
        r2 = np.array([[0.0, 0.0, 0.0], [d, 0.0, 0.0]])
        q2 = np.array([1.0, -2.0])
        u2, f2, _ = msm.msm3d(r2, q2, 6, 0.25, 1.0, 10)
        assert abs(u2 - (-2.0 / d)) < 2e-3 * (2.0 / d)
        assert abs(f2[0, 0] - (-2.0 / d ** 2)) < 3e-3 * (2.0 / d ** 2) and abs(f2[1, 0] - (-1.0 / d ** 2)) < 3e-3 / d ** 2
    u1, f1, _ = msm.msm3d(np.array([[0.4, 0.5, 0.6]]), np.array([3.0]), 4, 0.25, 1.0, 10)
    # a single particle: what is left is the grid's error on the self term -q^2 gamma(0) / (2 a) = -9.84
    assert abs(u1) < 3e-3 * 9.84 and np.abs(f1).max() < 1e-2


def test_open_boundary_large(msm, torch):
    """Open-boundary MSM at 300k particles (three levels, 32^3 coarsest grid) against the direct sum on the GPU."""
    N, p, mu = 300000, 6, 10
    r, q = open_system(N, 11)
    ht = (1.0 / N) ** (1.0 / 3.0)
    desc = msm.plan_open(N, r.min(axis=0), r.max(axis=0), p, mu, ht, 4.0 * ht, 0)
    eng = msm.Engine(desc)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    en, acc = eng.eval(rd, qd)
    en2, acc2 = eng.eval(rd, qd)
    assert np.array_equal(en, en2) and torch.equal(acc, acc2)
    u, f = msm.direct(r, q)
    assert abs(en[6] - u) < 1e-2 * abs(u)     # the total is what is left of components forty times larger
    assert rel_rms(acc.cpu().numpy(), f) < 3e-4
    eng.close()


@pytest.mark.parametrize("name", ["changaN8", "changaN64", "changaN512", "NaClN64", "CsClN128"])
def test_direct_matches_reference_nonperiodic(msm, name):
    """direct() (msm/msmLibrary.h:212) on the GPU == the reference's msm/nonperiodic output (charges forced to 1
    there, energy printed with %15.8e) and == the CPU oracle to rounding on the system's own charges."""
    fx = O.golden("nonperiodic_%s.json" % name)
    s = fx["system"]
    r, q = np.array(s["r"]).reshape(-1, 3), np.array(s["q"])
    u1, _ = msm.direct(r, np.ones(s["N"]))
    assert "%15.8e" % u1 == "%15s" % fx["printed"]
    u, acc = msm.direct(r, q)
    uo, acco = O.direct(r, q)
    assert abs(u - uo) <= 1e-13 * np.abs(q).sum() ** 2 / np.abs(r).max()
    assert rel_rms(acc, acco) < TOL


def test_direct_large(msm, torch):
    """20k random charges: every lane of the all-pairs kernel is live; slices of i (the multi-GPU split) add up."""
    import ctypes as C
    N = 20000
    r, q = O.neutral_system(N, (3.0, 2.0, 1.0), seed=11)
    r[7] = r[3]                                    # an exact duplicate contributes nothing (and no NaN)
    u, acc = msm.direct(r, q)
    assert np.isfinite(u) and np.isfinite(acc).all()
    lib = msm.load()
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    work = torch.zeros(lib.msm_amd_direct_work_doubles(N), dtype=torch.float64, device="cuda:0")
    parts, usum = [], 0.0
    for i0, i1 in ((0, 7000), (7000, 7001), (7001, N)):
        a = torch.zeros((i1 - i0, 3), dtype=torch.float64, device="cuda:0")
        us = C.c_double()
        rc = lib.msm_amd_direct_device(N, rd.data_ptr(), qd.data_ptr(), i0, i1, msm.capi.ACC_GRADIENT, a.data_ptr(),
                                       work.data_ptr(), None, C.byref(us))
        assert rc == 0
        parts.append(a.cpu().numpy())
        usum += us.value
    assert np.array_equal(np.concatenate(parts), acc)
    assert abs(usum - u) <= 1e-12 * abs(u)
    # against the CPU oracle on a perturbed copy without the coincidence
    r2 = r.copy()
    r2[7] += 0.01
    u2, acc2 = msm.direct(r2, q)
    uo2, acco2 = O.direct(r2, q)
    assert abs(u2 - uo2) <= 1e-12 * np.abs(acco2).max()
    assert rel_rms(acc2, acco2) < TOL


@pytest.mark.parametrize("name", ["NaClN8", "changaN8", "NaClN64", "changaN64", "changaN512"])
def test_ewald_classical_matches_reference_output(msm, torch, name):
    """ewaldClassical() on the GPU against the total the reference's own program printed for the same file and
    arguments (tests/golden/ewaldClassical_*.json), and against the CPU restatement.  The changa totals are what
    is left of components eight orders of magnitude larger, so the tolerance is 1e-12 of the largest component."""
    fx = O.golden("ewaldClassical_%s.json" % name)
    _, alpha, cd, cf = fx["args"]
    _, r, q, A = system(name)
    u = msm.ewaldClassical(r, q, A, alpha, cd, cf)
    u5, _, _ = msm.ewald(r, q, A, alpha, cd, cf, literal=True, want_grad=False)
    assert u == u5[4]
    scale = np.abs(u5[:4]).max()
    assert abs(u - fx["total"]) <= 1e-12 * scale, (u, fx["total"], u5)
    r = np.ascontiguousarray(r)
    uo = O.lib().or_ewald_classical(r.ctypes.data_as(O._dp), q.ctypes.data_as(O._dp), len(q), A[0], A[1], A[2],
                                    float(alpha), float(cd), float(cf))
    assert abs(u - uo) <= 1e-12 * scale


def test_ewald_full_sum(msm, torch):
    """The converged Ewald sum (nearest image, cube of wave vectors) against the CPU oracle: energy components,
    potential gradient at every particle, a slice of particles, independence of the splitting parameter, and
    the pair of a charge with its own images in a non-cubic cell."""
    N, A = 1500, (1.0, 1.3, 0.8)
    r, q = O.neutral_system(N, A, seed=5)
    q = q * (1.0 + 0.25 * np.sin(np.arange(N)))          # not neutral: the background term takes part
    alpha, rcut, kcut = 14.0, 0.4, 26
    uo, ao = O.ewald(r, q, A, alpha, rcut, kcut, nthreads=8)
    u5, phi, grad = msm.ewald(r, q, A, alpha, rcut, kcut)
    assert abs(u5[4] - uo) < 1e-12 * abs(uo)
    assert abs(u5[:4].sum() - u5[4]) < 1e-12 * abs(u5).max()
    assert abs(0.5 * float((torch.as_tensor(q, device="cuda:0") * phi).sum()) - u5[4]) < 1e-12 * abs(u5).max()
    assert rel_rms(grad.cpu().numpy(), ao) < 1e-12
    # a slice sees the same potentials, and the slices' energies add up
    ua, pa, ga = msm.ewald(r, q, A, alpha, rcut, kcut, 0, 700)
    ub, pb, gb = msm.ewald(r, q, A, alpha, rcut, kcut, 700, N)
    assert torch.equal(torch.cat([pa, pb]), phi) and torch.equal(torch.cat([ga, gb]), grad)
    assert abs(ua[4] + ub[4] - u5[4]) < 1e-12 * abs(u5).max()
    # two converged splittings (erfc(alpha rcut) and exp(-(pi k/alpha)^2) below 1e-14 along the longest edge), same sum
    u5a, _, grada = msm.ewald(r, q, A, 14.0, 0.4, 34)
    u5b, _, gradb = msm.ewald(r, q, A, 15.5, 0.4, 38)
    assert abs(u5b[4] - u5a[4]) < 1e-11 * abs(u5a[4]) and rel_rms(gradb.cpu().numpy(), grada.cpu().numpy()) < 1e-11
    # bad arguments are refused
    with pytest.raises(msm.MsmAmdError):
        msm.ewald(r, q, A, alpha, 0.41, kcut)            # beyond half the shortest edge (0.8 / 2)


def test_ewald_agrees_with_msm_at_scale(msm, torch):
    """At a size the CPU oracle cannot reach: the GPU Ewald sum against the MSM evaluation of the same system
    (error of the MSM at abar = 6, p = 6 on a grid of half the usual spacing, far above the Ewald error)."""
    N, A = 200000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=9)
    alpha = 5.6 / 0.12
    u5, phi, grad = msm.ewald(r, q, A, alpha, 0.12, int(np.ceil(5.6 * alpha / np.pi)), 0, 2048)
    d0 = msm.plan_auto(N, A, 6, 10, 6.0, 0)
    eng = msm.Engine(d0)
    en, acc = eng.eval(r, q)
    err = rel_rms(acc[:2048].cpu().numpy(), grad.cpu().numpy())
    assert err < 2e-4, err
    eng.close()


def test_command_line_drivers(msm, tmp_path):
    """The restated mains of msm/msmPeriodic.c:1423 and msm/ewaldClassical.c:125, linked against the library: same
    argv, same stdout (the potential with %20.15e), same msm.pot / msm.acc files, on a file in the reference's format."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = [os.path.join(root, "drivers", n) for n in ("msmPeriodic", "ewaldClassical", "nonperiodic")]
    if not all(os.path.exists(e) for e in exe):   # normally built beforehand by __graft_entry__.build()
        subprocess.run(["make", "-C", os.path.join(root, "msm_amd", "csrc"), "driver"], capture_output=True, timeout=300)
    if not all(os.path.exists(e) for e in exe):
        pytest.fail("drivers are not built: make -C msm_amd/csrc driver")
    fx, r, q, A = system("NaClN8")
    data = tmp_path / "NaClN8.ini"
    with open(data, "w") as f:
        f.write("%.17g %.17g %.17g\n%d\n" % (A[0], A[1], A[2], len(q)))
        for i in range(len(q)):
            f.write("%.17g %.17g %.17g %.17g\n" % (r[i, 0], r[i, 1], r[i, 2], q[i]))
    out = subprocess.run([exe[0], str(data), "4", "10", "4"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    u = float(out.stdout.strip().split()[-1])
    scale = max(abs(v) for v in fx["scalars"].values())
    assert abs(u - fx["potential"]) <= TOL * scale
    assert abs(float(open(tmp_path / "msm.pot").read()) - fx["potential"]) <= TOL * scale
    acc = np.loadtxt(tmp_path / "msm.acc", skiprows=1).reshape(3, -1).T
    assert np.abs(acc - np.array(fx["acc"]).reshape(-1, 3)).max() < 1e-10
    fe = O.golden("ewaldClassical_NaClN8.json")
    out = subprocess.run([exe[1], str(data)] + [str(a) for a in fe["args"][1:]], cwd=tmp_path, capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    assert abs(float(out.stdout.strip().split()[-1]) - fe["total"]) <= 1e-11 * abs(fe["total"])
    assert "Classical" in out.stderr
    # msm/nonperiodic.c: same stdout line as the real program printed for this system; and its MSM form within the method
    fn = O.golden("nonperiodic_changaN64.json")
    sn = fn["system"]
    rn = np.array(sn["r"]).reshape(-1, 3)
    datan = tmp_path / "changaN64.ini"
    with open(datan, "w") as f:
        f.write("%.17g %.17g %.17g\n%d\n" % (sn["L"][0], sn["L"][1], sn["L"][2], sn["N"]))
        for i in range(sn["N"]):
            f.write("%.17g %.17g %.17g %.17g\n" % (rn[i, 0], rn[i, 1], rn[i, 2], sn["q"][i]))
    out = subprocess.run([exe[2], str(datan)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "u is %15s" % fn["printed"]
    h = float((rn.max(axis=0) - rn.min(axis=0)).max() / 4.0)
    out = subprocess.run([exe[2], str(datan), "6", repr(h), repr(4.0 * h), "10"], cwd=tmp_path, capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    assert abs(float(out.stdout.strip().split()[-1]) - fn["u"]) < 2e-3 * abs(fn["u"]), (out.stdout, out.stderr)


@pytest.mark.parametrize("name", ["CsClN2", "NaClN8", "CsClN16", "changaN64"])
def test_driver_prints_what_the_reference_prints(msm, tmp_path, name):
    """drivers/msmPeriodic (GPU engine behind the reference's main) against the stdout of the real msmPeriodic: the
    Stencil[0] dump, the level-0 grid charges, the two ulong_real_g1 / ulong_real_gLR lines (msmPeriodic.c:1340-1341)
    and every line of print_stats (msmLibrary.c:290-376), label for label; only the time fields may differ."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    from gen_golden import parse_periodic_stdout
    exe = os.path.join(root, "drivers", "msmPeriodic")
    assert os.path.exists(exe), "drivers are not built: make -C msm_amd/csrc driver"
    fx, r, q, A = system(name)
    data = tmp_path / (name + ".ini")
    with open(data, "w") as f:
        f.write("%.17g %.17g %.17g\n%d\n" % (A[0], A[1], A[2], len(q)))
        for i in range(len(q)):
            f.write("%.17g %.17g %.17g %.17g\n" % (r[i, 0], r[i, 1], r[i, 2], q[i]))
    out = subprocess.run([exe, str(data), "4", "10", "4"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = parse_periodic_stdout(out.stdout.splitlines())
    assert set(got["scalars"]) == set(fx["scalars"]) and set(got["stats"]) == set(fx["stats"])
    scale = max(abs(fx["scalars"][lab]) for _, lab in LABELS[:6])
    for k, v in fx["scalars"].items():
        # the two diagnostic sums are 1/r-like image sums a thousand times larger than what is left of their difference
        tol = TOL * max(scale, abs(v)) if "real_g" in k else TOL * scale
        assert abs(got["scalars"][k] - v) <= tol, (k, got["scalars"][k], v)
    for k, v in fx["stats"].items():
        if k.startswith("time"):
            continue
        a, b = got["stats"][k].split(), v.split()
        assert len(a) == len(b), (k, got["stats"][k], v)
        for x, y in zip(a, b):
            try:
                fy = float(y)
            except ValueError:
                assert x == y, (k, got["stats"][k], v)
                continue
            # %d and %25.16e fields agree to the last digit that matters; %f fields print six decimals
            assert abs(float(x) - fy) <= max(1e-12 * abs(fy), 1.5e-6 if "." in y and "e" not in y else 0.0), (k, x, y)
    assert grid_err(np.array(got["stencil0"]), np.array(fx["stencil0"])) < 1e-11
    # (on the neutral lattices the grid charges cancel to rounding noise: the bar is relative to one particle's charge)
    assert np.abs(np.array(got["gridcharge0"]) - np.array(fx["gridcharge0"])).max() < TOL * np.abs(q).max()
    assert abs(float(out.stdout.splitlines()[-1]) - fx["potential"]) <= TOL * scale
    for fn in ("msm.acclong", "msm.accshort", "msm.acc", "msm.pot"):
        assert (tmp_path / fn).exists()


@pytest.mark.parametrize("p", [4, 6])
def test_prolongation_chain_is_two_prolongations(msm, torch, p):
    """Two prolongations per launch (k_prolong_chain) leave in the finer grid the bits of two separate launches; the
    potentials of the level in between keep their own convolution only."""
    N, A = 100000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=21)
    desc = msm.plan_auto(N, A, p, 10, 4.0, 4)            # 48 -> 24 -> 12 -> 6
    assert desc.L == 4
    eng = msm.Engine(desc)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    out = {}
    for chain in (1, 0):
        eng.set_option("transfer_chain", chain)
        en, acc = eng.eval(rd, qd)
        out[chain] = (np.array(en), acc.cpu().numpy().copy(), [eng.grid(msm.capi.GRID_E, l).copy() for l in range(4)])
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    for l in (0, 1, 3):
        assert np.array_equal(out[0][2][l], out[1][2][l]), l
    assert not np.array_equal(out[0][2][2], out[1][2][2])   # level 2 without / with the part prolonged from level 3
    # and the separate launches are what the oracle computes, level by level
    sp = spec_for(desc)
    sp.eval(r, q)
    for l in range(4):
        assert grid_err(out[0][2][l], sp.grid("e", l)) < TOL, l
    eng.close()


def test_ten_million_properties(msm, torch):
    """N = 10M, quintic spline, the reference's parameter choices: grid 256^3, six levels, 19-tap stencils - the size of
    the accuracy sweep, where the oracle's whole evaluation is out of reach.  What is asserted at this size: finite and
    bitwise repeatable; charge conserved on every level; the short-range accelerations of the particles of a sub-box
    equal the oracle's pair sum over that sub-box and its neighbourhood to 1e-12; and the accelerations of a random
    subset of 20 000 particles agree with the converged Ewald sum of all 10M to the accuracy of the method at these
    parameters (6.8e-5 in the sweep table; asserted below 1.5e-4)."""
    N, A, p = 10000000, (1.0, 1.0, 1.0), 6
    r, q = O.neutral_system(N, A, seed=20261111 + 5)
    desc = msm.plan_auto(N, A, p, 10, 4.0, 0)
    assert list(desc.M) == [256, 256, 256] and desc.L == 6
    eng = msm.Engine(desc)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    en, acc = eng.eval(rd, qd)
    assert np.isfinite(en).all() and bool(torch.isfinite(acc).all())
    for l in range(desc.L):
        assert abs(eng.grid(msm.capi.GRID_Q, l).sum() - q.sum()) < 1e-7, l
    en2, acc2 = eng.eval(rd, qd)
    assert np.array_equal(en, en2) and torch.equal(acc, acc2)
    del acc2
    # short-range part on a sub-box: every partner of a particle of the inner cube lies in the outer one, which is far
    # from the cell's faces, so the oracle's periodic pair sum over the outer cube's particles alone is the same sum
    a_s = torch.zeros((N, 3), dtype=torch.float64, device="cuda:0")
    eng.stage_bin(rd, qd)
    eng.stage_pairs(a_s)
    lo, w = 0.43, 0.06
    outer = np.nonzero(((r > lo - desc.a) & (r < lo + w + desc.a)).all(axis=1))[0]
    inner = ((r[outer] > lo) & (r[outer] < lo + w)).all(axis=1)
    assert inner.sum() > 1500 and len(outer) < 30000
    sp = O.SpecPlan.explicit(len(outer), list(desc.A), [8, 8, 8], p, 2, 1, desc.a, 4.0, 2, nthreads=16)  # (only its pair sum is used)
    aw, _, _ = sp.pairs(r[outer], q[outer])
    got = a_s[torch.as_tensor(outer, device="cuda:0")].cpu().numpy()
    assert rel_rms(got[inner], aw[inner]) < TOL
    assert np.abs(got[inner] - aw[inner]).max() <= 1e-11 * np.abs(aw[inner]).max()
    del a_s
    # accuracy against the Ewald sum at a subset (the subset in front, both parts sorted by cell for the Ewald kernels)
    nsub = 20000
    first = np.random.default_rng(5).permutation(N)
    nb = 32
    cell = np.minimum((r[first] * nb).astype(np.int64), nb - 1)
    key = (cell[:, 2] * nb + cell[:, 1]) * nb + cell[:, 0] + (np.arange(N) >= nsub) * nb ** 3
    order = first[np.argsort(key, kind="stable")]
    sub = order[:nsub]
    rcut = 0.25
    alpha = 5.6 / rcut
    _, _, grad = msm.ewald(rd[torch.as_tensor(order, device="cuda:0")], qd[torch.as_tensor(order, device="cuda:0")], A, alpha, rcut,
                           int(np.ceil(5.6 * alpha / np.pi)), 0, nsub)
    ferr = rel_rms(acc[torch.as_tensor(sub, device="cuda:0")].cpu().numpy(), grad.cpu().numpy())
    assert ferr < 1.5e-4, ferr
    eng.close()


def test_open_boundary_one_million(msm, torch):
    """The non-periodic MSM (msm3d, msm/msmLibrary.h:216) at 1M particles - the fourth benchmark configuration - against
    the all-pairs direct sum of the same system on the GPU (direct, msm/msmLibrary.h:212; what msm/nonperiodic.c
    computes): force error and energy error of the method at a = 4 h, quintic spline, and bitwise repeatable."""
    N, p, mu = 1000000, 6, 10
    r, q = open_system(N, 13)
    ht = (1.0 / N) ** (1.0 / 3.0)
    desc = msm.plan_open(N, r.min(axis=0), r.max(axis=0), p, mu, ht, 4.0 * ht, 0)
    assert desc.open_boundary == 1 and max(desc.M) >= 128
    eng = msm.Engine(desc)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    en, acc = eng.eval(rd, qd)
    en2, acc2 = eng.eval(rd, qd)
    assert np.isfinite(en).all() and np.array_equal(en, en2) and torch.equal(acc, acc2)
    for l in range(desc.L):
        assert abs(eng.grid(msm.capi.GRID_Q, l).sum() - q.sum()) < 1e-8, l
    u, f = msm.direct(r, q)
    assert abs(en[6] - u) < 1e-2 * abs(u)
    assert rel_rms(acc.cpu().numpy(), f) < 3e-4
    eng.close()


@pytest.mark.parametrize("grid", [(96, 96, 96), (48, 64, 40)])
def test_anterpolation_weight_records_keep_the_bits(msm, torch, grid):
    """The column kernel of the anterpolation with the particles' weights in its staged records (default) and with their
    offsets, from which every lane evaluates its own polynomial pieces: the same products in the same order, so the grid
    charges, and with them everything downstream, agree bit for bit; a crowded region (long runs, many batches) and a
    grid of unequal, non-cubic sizes included."""
    N, A = 150000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=91)
    rng = np.random.default_rng(5)
    r[:8000] = (0.62 + 0.01 * rng.standard_normal((8000, 3))) % 1.0
    lib = msm.load()
    L, a = 3, 4.0 / max(grid)
    hL = (2.0 ** (L - 1)) * A[0] / min(grid)
    beta = lib.choosebetaNew(1e-12, (2.0 ** L) * a, hL)
    kmax = int(np.ceil(lib.choosekmax(1e-12, beta, hL)))
    desc = msm.plan_explicit(N, A, grid, 6, 10, L, a, beta, kmax)
    eng = msm.Engine(desc)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    en1, acc1 = eng.eval(rd, qd)
    acc1 = acc1.clone()
    q1 = eng.grid_tensor(0, 0).clone()
    eng.set_option("anterp_cols_weights", 0)
    en0, acc0 = eng.eval(rd, qd)
    q0 = eng.grid_tensor(0, 0).clone()
    assert float(q1.abs().max()) > 0
    assert torch.equal(q0, q1)
    assert np.array_equal(en0, en1) and torch.equal(acc0, acc1)
    eng.close()


@pytest.mark.parametrize("p,conv", [(6, 0), (4, 2)])
def test_slot_order_accelerations_keep_the_bits(msm, torch, p, conv):
    """A whole evaluation hands the short-range accelerations from the pair kernels to the interpolation kernel in slot
    order (one scatter into the caller's array instead of a scattered write and a scattered read-modify-write); with
    the option off, and stage by stage, the same two terms are added per component: the same bits, gradient and force
    conventions, a crowded region that overflows its neighbour lists included."""
    N, A = 120000, (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=77)
    rng = np.random.default_rng(3)
    r[:6000] = (0.31 + 0.004 * rng.standard_normal((6000, 3))) % 1.0     # list overflow: the list-free kernel writes slots too
    desc = msm.plan_auto(N, A, p, 10, 4.0, 3)
    desc.acc_convention = conv
    eng = msm.Engine(desc)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    en1, acc1 = eng.eval(rd, qd)
    acc1 = acc1.clone()
    eng.set_option("acc_slots", 0)
    en0, acc0 = eng.eval(rd, qd)
    assert np.array_equal(en0, en1) and torch.equal(acc0, acc1)
    eng.set_option("acc_slots", 1)
    for v in (9, 5, 6, 8):                    # split queues, per-cluster walk, list-free, default
        eng.set_option("pair_variant", v)
        en_a, acc_a = eng.eval(rd, qd)
        acc_a = acc_a.clone()
        eng.set_option("acc_slots", 0)
        en_b, acc_b = eng.eval(rd, qd)
        eng.set_option("acc_slots", 1)
        assert np.array_equal(en_a, en_b) and torch.equal(acc_a, acc_b), v
    # the reductions in one launch (the workgroup that ends last finishes) and in two: the same sums in the same order,
    # evaluation after evaluation (the ticket counter is back at zero every time)
    eng.set_option("energy_fused", 1)
    for _ in range(20):
        en_f, _ = eng.eval(rd, qd)
        assert np.array_equal(en_f, en1)
    eng.set_option("energy_fused", 0)
    en_t, _ = eng.eval(rd, qd)
    assert np.array_equal(en_t, en1)
    # stage by stage (the caller's array after every stage)
    acc2 = torch.zeros_like(acc1)
    eng.stage_bin(rd, qd)
    eng.stage_pairs(acc2)
    eng.stage_anterpolate()
    eng.stage_grid()
    eng.stage_interpolate(acc2)
    assert torch.equal(acc2, acc1)
    eng.close()


@pytest.mark.parametrize("p,conv,blob", [(6, 0, False), (4, 2, True)])
def test_eval_reuse_matches_eval(msm, torch, p, conv, blob):
    """msm_amd_eval_reuse over a short trajectory: neighbour lists built with cutoff a + skin are kept while no particle
    has moved more than skin / 2 (checked on the device) and rebuilt by the same call otherwise.  Every step equals
    msm_amd_eval of the same positions to 1e-12 (energies of the largest component, accelerations relative RMS; the sums
    run in another order, so not bit for bit); the rebuild count is what the displacements call for; a caller that wraps
    a particle by a cell edge moves nothing; the same trajectory on a second handle gives the same bits; a step that moves
    every particle by just under skin / 2 brings pairs from beyond the cutoff deep into it without a rebuild.  The second case
    has a crowded ball whose clusters overflow their lists (list-free kernel over the frozen bins) and the force
    convention."""
    N, A = 120000, (1.0, 1.0, 1.0)
    r0, q = O.neutral_system(N, A, seed=91)
    rng = np.random.default_rng(17)
    if blob:
        r0[:5000] = (0.62 + 0.004 * rng.standard_normal((5000, 3))) % 1.0
    desc = msm.plan_auto(N, A, p, 10, 4.0, 3)
    desc.acc_convention = conv
    skin = 0.1 * desc.a
    fresh, kept, again = msm.Engine(desc), msm.Engine(desc), msm.Engine(desc)
    qd = torch.as_tensor(q, device="cuda:0")
    # what a different order of the same sums costs on this system (the crowded ball adds terms of 1e7 that cancel): the
    # list-walking kernels per cluster and per particle against each other
    other = msm.Engine(desc)
    other.set_option("pair_variant", 5)
    _, a5 = other.eval(torch.as_tensor(r0, device="cuda:0"), qd)
    _, a8 = fresh.eval(torch.as_tensor(r0, device="cuda:0"), qd)
    tol = max(TOL, 4.0 * rel_rms(a5.cpu().numpy(), a8.cpu().numpy()))
    other.close()
    assert tol < 2e-11
    r = r0.copy()
    results = []
    for step in range(10):
        if step > 0:
            r = r + (rng.random((N, 3)) - 0.5) * 0.008 * desc.a            # small moves: the lists stay
        if step == 1:
            # every particle 0.0495 a along +x or -x, just under skin / 2 = 0.05 a: pairs that were up to 0.099 a beyond
            # the cutoff at the rebuild are now inside it, where they weigh up to 1e-5 of a force - lists built without the
            # skin would show
            r[:, 0] = r0[:, 0] + 0.0495 * desc.a * np.where(rng.random(N) < 0.5, -1.0, 1.0)
            r[:, 1:] = r0[:, 1:]
        if step == 2:
            r = r0 + (rng.random((N, 3)) - 0.5) * 0.008 * desc.a
        if step == 6:
            r[::97] += 0.3 * desc.a * (rng.random((len(r[::97]), 3)) - 0.5)   # some particles jump: this call rebuilds
        if step == 3:
            r[::11, 0] += 1.0                                                # the caller wraps coordinates: nothing moved
        if step == 4:
            r[::13, 2] -= 2.0
        rd = torch.as_tensor(r, device="cuda:0")
        en0, acc0 = fresh.eval(rd, qd)
        en1, acc1 = kept.eval_reuse(rd, qd, skin)
        scale = np.abs(en0[:6]).max()
        assert np.abs(en1 - en0).max() <= tol * scale, (step, en1, en0)
        assert rel_rms(acc1.cpu().numpy(), acc0.cpu().numpy()) < tol, step
        assert float((acc1 - acc0).abs().max()) <= 10 * tol * float(acc0.abs().max()), step
        results.append((en1, acc1.clone()))
        en2, acc2 = again.eval_reuse(rd, qd, skin)
        assert np.array_equal(en1, en2) and torch.equal(acc1, acc2), step
    evals, rebuilds = kept.reuse_stats()
    assert evals == 10 and rebuilds == 2, (evals, rebuilds)
    # a new skin rebuilds; a skin beyond the cutoff, a slab rank or the literal mode are refused
    kept.eval_reuse(rd, qd, 0.05 * desc.a)
    assert kept.reuse_stats() == (11, 3)
    with pytest.raises(msm.MsmAmdError):
        kept.eval_reuse(rd, qd, 1.5 * desc.a)
    for e in (fresh, kept, again):
        e.close()


@pytest.mark.parametrize("N,A,p,L", [(5000, (1.0, 1.1, 0.9), 4, 2), (30000, (1.3, 1.0, 1.1), 6, 2)])
def test_eval_reuse_small_and_skewed_plans(msm, torch, N, A, p, L):
    """msm_amd_eval_reuse on plans that take the other code paths: a grid too small for the tiled grid transfer (the
    columns of the sort are then not aligned with grid cells and a second, grid-cell sort feeds the spline kernels) and a
    cell with three different edges; a random walk of the particles with one jump, against msm_amd_eval to 1e-12."""
    r, q = O.neutral_system(N, A, seed=5)
    rng = np.random.default_rng(23)
    desc = msm.plan_auto(N, A, p, 10, 4.0, L)
    skin = 0.08 * desc.a
    fresh, kept = msm.Engine(desc), msm.Engine(desc)
    qd = torch.as_tensor(q, device="cuda:0")
    for step in range(6):
        if step:
            r = r + (rng.random((N, 3)) - 0.5) * 0.01 * desc.a
        if step == 3:
            r[::7] += 0.4 * desc.a
        rd = torch.as_tensor(r, device="cuda:0")
        en0, acc0 = fresh.eval(rd, qd)
        en1, acc1 = kept.eval_reuse(rd, qd, skin)
        assert np.abs(en1 - en0).max() <= TOL * np.abs(en0[:6]).max(), step
        assert rel_rms(acc1.cpu().numpy(), acc0.cpu().numpy()) < TOL, step
    assert kept.reuse_stats() == (6, 2)
    fresh.close()
    kept.close()


def test_random_plans_against_the_oracle(msm, torch):
    """A fixed batch of tests/fuzz_parity.py: random particle counts, cells of unequal edges, both spline orders, cutoffs
    of 3 to 5 grid spacings, one to three levels, on uniform systems, crowded balls, sheets, and particles on the faces of
    the cell and outside it - one evaluation each against the oracle, 1e-12 (1e-10 for the accelerations of the crowded
    balls, whose closest pairs are 1e-5 apart and closer: the last bit of a wrapped coordinate is 1e-11 of their force)."""
    import fuzz_parity
    failures = fuzz_parity.run(36, 7, quiet=True)
    assert not failures, "\n".join(failures)
