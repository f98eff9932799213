"""Random small systems through the drop-in msmperiodic() in literal mode against the oracle's literal mirror of the
reference's msmperiodic() (msm/msmPeriodic.c:52; the mirror is bit-exact against the real program on its own systems):
random particle counts, cell edges, cutoffs and quasi-interpolation orders.   python tests/fuzz_literal.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib as O


def run(ncases, seed):
    import msm_amd
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(ncases):
        N = int(rng.choice([2, 3, 8, 27, 64, 150, 400]))
        edge = float(rng.uniform(1.0, 12.0))
        A = (edge, edge, edge)      # (the reference's systems are cubes, and its grid-size rule reads one edge)
        p = 4
        mu = int(rng.choice([2, 4, 6, 10]))
        abar = float(rng.uniform(2.5, 5.5))
        r = rng.random((N, 3)) * edge
        q = rng.uniform(-1.0, 1.0, N)
        q -= q.mean()
        lit = O.literal_run(r, q, p, abar, mu, A)
        if lit["rc"] != 0:
            print("case %3d: the mirror refuses N %d A %.2f mu %d abar %.2f (rc %d)" % (case, N, edge, mu, abar, lit["rc"]))
            continue
        try:
            st, u, f, _ = msm_amd.msmperiodic(r.copy(), q, p, abar, mu, A, mode=msm_amd.capi.MODE_LITERAL, quiet=True)
        except msm_amd.MsmAmdError as e:
            print("case %3d FAIL: the library refuses what the mirror computes: %s" % (case, str(e)[:200]))
            bad += 1
            continue
        comps = [abs(lit[k]) for k in ("u_short_real", "u_short_self", "u_short_csr", "u_long_real", "u_long_four", "u_long_self")]
        scale = max(comps)
        de = abs(u - lit["u_total"]) / max(scale, 1e-300)
        fl = lit["facc"].reshape(-1, 3)
        floor = np.abs(q).max() / lit["a"] ** 2
        den = max(np.sqrt((fl ** 2).sum()), floor * np.sqrt(fl.size))
        da = float(np.sqrt(((f - fl) ** 2).sum()) / den)
        ok = de < 1e-12 and da < 1e-12 and st.Mx == lit["Mx"] and st.kmax == lit["kmax"] and st.beta == lit["beta"]
        bad += not ok
        print("case %3d %s N %3d A %5.2f mu %2d abar %.2f -> M %d kmax %d: energy %.1e acc %.1e" %
              (case, "ok  " if ok else "FAIL", N, edge, mu, abar, st.Mx, st.kmax, de, da), flush=True)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
