"""The CPU oracle against the reference's own outputs (tests/golden/*, produced by
oracle/gen_golden.py from the real reference).  No GPU needed."""
import numpy as np
import pytest

import oracle_lib as O

ALL = ["CsClN2", "NaClN8", "NaNaN8", "changaN8", "CsClN16", "NaClN64", "NaNaN64", "changaN64", "CsClN128",
       "NaClN512", "NaNaN512", "changaN512"]
LABELS = (("u_short_real", "(A) ushort_real"), ("u_short_self", "(B) ushort_self"), ("u_short_csr", "(C) ushort_csr"),
          ("u_long_real", "(D) ulong_real"), ("u_long_four", "(E) ulong_four"), ("u_long_self", "(F) ulong_self"),
          ("u_total", "utotal"))


def system(name):
    fx = O.golden("msmPeriodic_%s.json" % name)
    s = fx["system"]
    return fx, np.array(s["r"]).reshape(-1, 3), np.array(s["q"]), s["L"]


def test_helpers_bit_exact():
    h = O.golden("helpers.json")
    L = O.lib()
    for row in h["rows"]:
        M = L.or_chooseM(row["htilde"], row["A"])
        assert M == row["M"]
        hx = row["A"] / M
        a = row["abar"] * row["htilde"]
        lv = row["L"]
        beta = L.or_choosebetaNew(1e-12, (2.0 ** lv) * a, (2.0 ** (lv - 1)) * hx)
        assert beta == row["beta"]
        assert L.or_choosekmax(1e-12, beta, (2.0 ** (lv - 1)) * hx) == row["kmax_double"]
        assert L.or_chooseL(M, M, M, row["N"]) == row["chooseL"] == 1
    for x, y in h["extra"]["mylog2"]:
        assert L.or_mylog2(x) == y
    for a, b, c, m in h["extra"]["minOf3Integers"]:
        assert L.or_minOf3(a, b, c) == m


def test_wprime_table():
    # omega' for p=4, mu=10 is the table of msmLibrary.h:156-170 (values as printed there)
    ref = [3.4641016151377539e+00, -1.7320508075688767e+00, 6.7949192431122685e-01, -2.3978297178318450e-01,
           7.9713982076653617e-02, -2.5502951436845282e-02, 7.9437840692459776e-03, -2.4260315207973015e-03,
           7.2976833805943851e-04, -2.1633348486544432e-04, 6.1410409127506203e-05, -1.4537930745802521e-05,
           1.9569521248038610e-06]
    assert np.array_equal(O.wprime(4, 10), np.array(ref))
    ref6 = [1.2379121245354655e+01, -9.3765192523414775e+00, 5.8087547854721668e+00, -3.2655236531195819e+00,
            1.7352829708690274e+00, -8.8893259491018195e-01, 4.4379214314187693e-01, -2.1736815004215146e-01,
            1.0479655626770509e-01, -4.9312127870361926e-02, 2.1608174371344265e-02, -7.8774221940891460e-03,
            1.9814890638090049e-03, -2.4354138541380615e-04]
    np.testing.assert_allclose(O.wprime(6, 10), np.array(ref6), rtol=1e-14, atol=0)  # table agrees to ~3e-15 only
    # mu = 2 values recorded from calculatewprime(2) of the reference
    np.testing.assert_allclose(O.wprime(4, 2), [3.4641016151377539, -1.7235390029754101, 0.64273441009183618,
                                                 -0.17702963774851166, 0.025783423063208574], rtol=1e-15)


def test_two_scale_and_bspline():
    J = np.zeros(4)
    O.lib().or_jcoef(4, J.ctypes.data_as(O._dp))
    assert list(J[:3]) == [6. / 8, 4. / 8, 1. / 8]
    O.lib().or_jcoef(6, J.ctypes.data_as(O._dp))
    assert list(J) == [20. / 32, 15. / 32, 6. / 32, 1. / 32]
    for p in (4, 6):
        for t in np.linspace(0, 0.999, 7):
            w = [O.lib().or_bspline(p, t + (p // 2 - 1) - v + p // 2) for v in range(p)]
            assert abs(sum(w) - 1.0) < 1e-14  # partition of unity
        for rho in np.linspace(0, 1.2, 9):
            assert abs(O.lib().or_gamma(rho, p) - O.lib().or_gamma_lit(rho, p)) < 4e-15
    assert O.lib().or_gamma(0.0, 4) == 35. / 16


@pytest.mark.parametrize("name", ALL)
def test_literal_mirror_matches_reference(name):
    """or_literal_run == the real msmPeriodic on every system of msm/data (p=4, mu=10, abar=4)."""
    if name.endswith("N512") and name != "changaN512":
        pytest.skip("same code path as changaN512; skipped to keep the CPU suite short")
    fx, r, q, A = system(name)
    res = O.literal_run(r, q, 4, 4.0, 10, A)
    assert res["rc"] == 0
    st = fx["stats"]
    assert res["Mx"] == int(st["Mx My Mz"].split()[0])
    assert res["kmax"] == int(st["kmax"])
    assert res["beta"] == float(st["beta"])
    assert res["a"] == float(st["a"])
    for mine, label in LABELS:
        ref = fx["scalars"][label]
        assert abs(res[mine] - ref) <= 2e-15 * max(abs(ref), 1e-300) + 1e-300, (label, res[mine], ref)
    # dumps are printed with %25.16f / %25.16e / %20.14e: compare at the printed precision
    sten = np.array(fx["stencil0"])
    assert sten.size == res["sl"].size
    if sten.size:
        assert np.abs(res["sl"] - sten).max() <= 1.01e-16 + 1e-16
    g = np.array(fx["gridcharge0"])
    assert g.size == res["qgrd"].size
    if g.size:
        np.testing.assert_allclose(res["qgrd"], g, rtol=2e-16, atol=1e-30)
    for key, mine in (("acc", "facc"), ("acclong", "flong"), ("accshort", "fshort")):
        ref = np.array(fx[key])
        # on the symmetric lattices the accelerations are cancellation noise of O(1) pair terms
        fscale = np.abs(ref).max() + 10.0 * np.abs(q).max() / res["a"] ** 2
        np.testing.assert_allclose(res[mine], ref, rtol=1e-14, atol=1e-14 * fscale)


def test_literal_rejects_p6():
    _, r, q, A = system("changaN8")
    assert O.literal_run(r, q, 6, 4.0, 10, A)["rc"] == 1  # msmPeriodic.c:100-104


SIMPLE = [("msmSimple_changaN8_nu4_mu10_M2.json", "changaN8", 1), ("msmSimple_changaN8_nu6_mu10_M2.json", "changaN8", 1),
          ("msmSimple_changaN8_nu4_mu10_M3.json", "changaN8", 1), ("msmSimple_changaN8_nu6_mu6_M4.json", "changaN8", 1),
          ("msmSimple_CsClN16_nu4_mu10_M3.json", "CsClN16", 1), ("msmSimple_NaClN8_nu4_mu10_M2.json", "NaClN8", 1),
          ("msmSimple_changaN64_nu4_mu10_M4.json", "changaN8", 2)]


def simple_case(fn, sysname, copies):
    fx = O.golden(fn)
    _, r, q, A = system(sysname)
    if copies > 1:  # msmSimple builds its N=64 system from 2x2x2 copies of changaN8 (msmSimple.c:2513-2538)
        rr, qq = [], []
        for cx in range(copies):
            for cy in range(copies):
                for cz in range(copies):
                    rr.append(r + np.array([cx, cy, cz]) * np.array(A))
                    qq.append(q)
        r, q, A = np.concatenate(rr), np.concatenate(qq), [a * copies for a in A]
    return fx, r, q, A


@pytest.mark.parametrize("fn,sysname,copies", SIMPLE)
def test_spec_engine_matches_msmSimple(fn, sysname, copies):
    """The specification engine (L=1) == the reference's 2-level msmSimple, energies and accelerations,
    for v=4 and v=6, even and odd grids.  msmSimple stops its image sums at a relative change of 1e-8
    (TOL_DIRECT, msmSimple.c:16), which bounds the agreement of the grid terms at ~1e-10."""
    fx, r, q, A = simple_case(fn, sysname, copies)
    sc, args = fx["scalars"], fx["args"]
    mu, nu, M = args[2], args[3], args[4:7]
    pl = O.SpecPlan.explicit(len(q), A, M, nu, mu, 1, sc["a"], sc["beta"], int(sc["kmax"]))
    out = pl.eval(r, q)
    en = out["energies"]
    scale = abs(sc["ushort_real"]) + abs(sc["ulong_self"])
    for i, k in enumerate(["ushort_real", "ushort_self", "ushort_csr", "ulong_real", "ulong_four", "ulong_self", "utotal"]):
        tol = 1e-13 if k in ("ushort_real", "ushort_self", "ushort_csr", "ulong_self") else 2e-10
        assert abs(en[i] - sc[k]) <= tol * max(abs(sc[k]), scale * 1e-3), (k, en[i], sc[k])
    K, first = pl.stencil(0)
    Kl = np.array(fx["Kl"]).reshape(M[0], M[1], M[2]).transpose(2, 1, 0)
    assert np.abs(K - Kl).max() <= 2e-11 * np.abs(Kl).max()
    KT = pl.stencil_top()
    Mmin = [-((m - 1) // 2) for m in M]
    idx = [np.array([(Mmin[d] + i) % M[d] for i in range(M[d])]) for d in range(3)]
    KL = np.array(fx["Kfourier"]).reshape(M[0], M[1], M[2]).transpose(2, 1, 0)
    assert np.abs(KT[np.ix_(idx[2], idx[1], idx[0])] - KL).max() <= 1e-13 * np.abs(KL).max()
    acc = fx["acc"]
    fs = np.array(acc["fshort"]).reshape(-1, 3)
    fl = np.array(acc["flong_real"]).reshape(-1, 3) + np.array(acc["flong_four"]).reshape(-1, 3)
    # floor of the scale: one pair term, so that all-cancelling lattices are compared at rounding level
    fscale = max(np.abs(fs).max(), np.abs(fl).max(), np.abs(q).max() / sc["a"] ** 2)
    assert np.abs((out["acc"] - out["along"]) - fs).max() <= 1e-12 * fscale
    assert np.abs(out["along"] - fl).max() <= 2e-10 * fscale


@pytest.mark.parametrize("name", ["changaN8", "NaClN8", "NaNaN8", "CsClN16", "changaN64"])
def test_spec_vs_literal_and_expected(name):
    """With the reference's own parameter choices the specification engine reproduces the terms of the
    literal path that do not depend on its index conventions ((A),(B),(C),(F), the level-0 stencil) to rounding,
    and agrees with the grid-free msmExpected on (D)+(E) within the discretisation error of the method."""
    fx, r, q, A = system(name)
    lit = O.literal_run(r, q, 4, 4.0, 10, A)
    O.lib().or_spec_set_image_range(5)  # the reference sums images -5..5 only (msmPeriodic.c:536-538)
    try:
        pl = O.SpecPlan.auto(len(q), A, 4, 10, 4.0, 1)
    finally:
        O.lib().or_spec_set_image_range(0)
    assert pl.prm.beta == lit["beta"] and pl.prm.kmax == lit["kmax"] and pl.prm.a == lit["a"]
    out = pl.eval(r, q)
    en = out["energies"]
    for i, k in ((0, "u_short_real"), (1, "u_short_self"), (2, "u_short_csr"), (5, "u_long_self")):
        assert abs(en[i] - lit[k]) <= 1e-13 * max(abs(lit[k]), abs(lit["u_short_real"]) * 1e-3), k
    K, first = pl.stencil(0)
    assert list(first) == lit["slmin"] and list(K.shape[::-1]) == lit["slsize"]
    # the reference evaluates far images as 1/r - erf(beta r)/r (cancellation noise ~1e-17 each, 2e7 terms)
    assert np.abs(K.ravel() - lit["sl"]).max() <= 1e-11 * np.abs(lit["sl"]).max()
    ex = O.golden("msmExpected_%s.json" % name)["scalars"]
    for i, k in ((0, "(A) ushort_real"), (2, "(C) ushort_csr"), (5, "(F) ulong_self")):
        assert abs(en[i] - ex[k]) <= 1e-12 * max(abs(ex[k]), abs(ex["(A) ushort_real"]) * 1e-3), k
    exact_long = ex["(D) ulong_real"] + ex["(E) ulong_four"]
    scale = abs(ex["(A) ushort_real"]) + abs(ex["(F) ulong_self"])
    assert abs((en[3] + en[4]) - exact_long) <= 2e-3 * scale
    # short-range accelerations: the literal path's (q_i+q_j)/2 weighting (msmLibrary.c:168-176)
    fs_lit = 0.5 * (q[:, None] * out["fshort_unw"] + (out["acc"] - out["along"]))
    ref = lit["fshort"].reshape(-1, 3)
    assert np.abs(fs_lit - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300) + 1e-15


def test_spec_multilevel_consistency():
    """L=1, 2, 3 on the same system: total energy and accelerations agree within the method's accuracy and
    converge towards the Ewald sum; grid charge is conserved by anterpolation and restriction."""
    r, q = O.neutral_system(600, (1.0, 1.0, 1.0), seed=5)
    uE, aE = O.ewald(r, q, (1.0, 1.0, 1.0), alpha=8.0, rcut=0.9, kcut=10, nthreads=8)
    errs = []
    for L in (1, 2):
        pl = O.SpecPlan.explicit(600, (1.0, 1.0, 1.0), (16, 16, 16), 4, 10, L, 4.0 / 16 * 1.25, 0.0, 0)
        # parameters the reference way for this grid: a = abar*htilde, beta/kmax from the helpers
        prm = O.SpecParams()
        O.lib().or_spec_auto_params(600, 1.0, 1.0, 1.0, 4, 10, 4.0, 1, prm)
        pl.close()
        M = 16
        a = 4.0 * (1.0 / 600) ** (1.0 / 3.0)
        hx = 1.0 / M
        beta = O.lib().or_choosebetaNew(1e-12, (2.0 ** L) * a, (2.0 ** (L - 1)) * hx)
        kmax = int(np.ceil(abs(O.lib().or_choosekmax(1e-12, beta, (2.0 ** (L - 1)) * hx))))
        pl = O.SpecPlan.explicit(600, (1.0, 1.0, 1.0), (M, M, M), 4, 10, L, a, beta, kmax, nthreads=8)
        out = pl.eval(r, q)
        g0 = pl.grid("q", 0)
        assert abs(g0.sum() - q.sum()) < 1e-10
        if L > 1:
            assert abs(pl.grid("q", 1).sum() - q.sum()) < 1e-10
        errs.append((abs(out["energies"][6] - uE) / abs(uE),
                     np.sqrt(((out["acc"] - aE) ** 2).sum() / (aE ** 2).sum())))
        pl.close()
    for eu, ef in errs:
        assert eu < 5e-3 and ef < 5e-2, errs


def test_ewald_classical_matches_reference():
    for name, alpha, cd, cf in (("NaClN8", 6, 10, 30), ("changaN8", 12, 10, 30), ("NaClN64", 6, 10, 30),
                                ("changaN64", 12, 10, 30)):
        fx = O.golden("ewaldClassical_%s.json" % name)
        _, r, q, A = system(name)
        r = np.ascontiguousarray(r)
        u = O.lib().or_ewald_classical(r.ctypes.data_as(O._dp), q.ctypes.data_as(O._dp), len(q), A[0], A[1], A[2],
                                       float(alpha), float(cd), float(cf))
        assert abs(u - fx["total"]) <= 1e-13 * abs(fx["total"])
    # and the full Ewald sum agrees with it where the missing image loop does not matter (NaCl, alpha*L large)
    _, r, q, A = system("NaClN64")
    u2, _ = O.ewald(r, q, A, 1.5, 3.99, 13, nthreads=8)  # a different splitting must give the same total
    assert abs(u2 - O.golden("ewaldClassical_NaClN64.json")["total"]) < 1e-6 * abs(u2)


@pytest.mark.parametrize("name", ["changaN8", "changaN64", "changaN512", "NaClN64", "CsClN128"])
def test_direct_oracle_matches_reference_nonperiodic(name):
    """The oracle's open-boundary direct sum against the real msm/nonperiodic (prints %15.8e; charges forced to 1)."""
    fx = O.golden("nonperiodic_%s.json" % name)
    s = fx["system"]
    r = np.array(s["r"]).reshape(-1, 3)
    u, acc = O.direct(r, np.ones(s["N"]))
    assert abs(u - fx["u"]) <= 0.6e-8 * abs(fx["u"])       # the printed precision
    assert "%15.8e" % u == "%15s" % fx["printed"]
    # the gradient is the derivative of the energy: central difference on one coordinate
    eps = 1e-6 * np.abs(r).max()
    rp, rm = r.copy(), r.copy()
    rp[1, 0] += eps
    rm[1, 0] -= eps
    du = (O.direct(rp, np.ones(s["N"]))[0] - O.direct(rm, np.ones(s["N"]))[0]) / (2 * eps)
    assert abs(du - acc[1, 0]) <= 1e-6 * max(abs(du), np.abs(acc).max() * 1e-3)


@pytest.mark.parametrize("p,h,abar,L", [(4, 1 / 8, 4.0, 2), (6, 1 / 8, 4.0, 2), (6, 1 / 16, 6.0, 3)])
def test_spec_open_boundary_vs_direct(p, h, abar, L):
    """The oracle's open-boundary mode (msm3d of msm/msmLibrary.h:216, defined nowhere in the reference, so no reference
    output pins it) against the oracle's direct sum, which the real msm/nonperiodic pins: errors at the level of the
    method and falling with spline order and cutoff; the result does not depend on the number of levels beyond that."""
    N, mu = 400, 10
    rng = np.random.default_rng(1)
    r = rng.random((N, 3))
    q = np.where(np.arange(N) % 2 == 0, 1.0, -1.0)
    u, acc = O.direct(r, q)
    a, hc = abar * h, h * 2 ** (L - 1)
    need = max((p + 2) * hc, (p / 2) * hc + 2 ** (L - 1) * a + 2 * h)
    M = int(np.ceil((1 + need) / h))
    M += (-M) % (2 ** (L - 1))
    A = M * h
    sp = O.SpecPlan.explicit(N, [A] * 3, [M] * 3, p, mu, L, a, 1.0, 0, nthreads=8, open_boundary=True)
    out = sp.eval(r + (A - 1) / 2, q)
    en = out["energies"]
    assert en[1] == 0.0 and en[2] == 0.0 and en[4] == 0.0
    du = abs(en[6] - u) / abs(u)
    df = np.sqrt(((out["acc"] - acc) ** 2).sum() / (acc ** 2).sum())
    bar_u, bar_f = {(4, 4.0): (5e-3, 1e-3), (6, 4.0): (2e-4, 5e-4), (6, 6.0): (5e-5, 5e-5)}[(p, abar)]
    assert du < bar_u and df < bar_f, (du, df)
    sp.close()
