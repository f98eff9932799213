"""Random trajectories through msm_amd_eval_reuse against msm_amd_eval of the same positions (GPU against GPU): random
plans, skins from 0 to 0.3 cutoffs, steps that move the particles by random fractions of the skin (some beyond half of it,
which forces a rebuild), wrapped coordinates, a crowded ball now and then.   python tests/fuzz_reuse.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib as O


def run(ncases, seed):
    import torch
    import msm_amd
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(ncases):
        N = int(rng.choice([3000, 20000, 60000, 150000]))
        A = tuple(float(x) for x in rng.uniform(0.9, 1.5, 3)) if rng.random() < 0.5 else (1.0, 1.0, 1.0)
        p = int(rng.choice([4, 6])); abar = float(rng.uniform(3.0, 5.0)); L = int(rng.choice([0, 2, 3]))
        frac = float(rng.choice([0.0, 0.02, 0.05, 0.1, 0.3]))
        blob = rng.random() < 0.25
        try:
            desc = msm_amd.plan_auto(N, A, p, 10, abar, L)
        except msm_amd.MsmAmdError:
            continue
        skin = frac * desc.a
        if min(A) < 3.0 * (desc.a + skin) * 1.001:
            continue
        r, q = O.neutral_system(N, A, seed=3000 + case)
        if blob:
            k = N // 8
            r[:k] = (np.array(A) * 0.4 + 0.01 * min(A) * rng.standard_normal((k, 3))) % np.array(A)
        fresh, kept = msm_amd.Engine(desc), msm_amd.Engine(desc)
        qd = torch.as_tensor(q, device="cuda:0")
        worst_e = worst_a = 0.0
        try:
            for step in range(8):
                if step:
                    amp = float(rng.choice([0.05, 0.3, 0.6, 1.5])) * max(skin, 0.01 * desc.a)   # per-coordinate range
                    r = r + (rng.random((N, 3)) - 0.5) * amp
                    if rng.random() < 0.3:
                        r[::7] += np.array(A) * rng.integers(-2, 3, 3)      # the caller wraps some particles
                rd = torch.as_tensor(r, device="cuda:0")
                en0, acc0 = fresh.eval(rd, qd)
                en1, acc1 = kept.eval_reuse(rd, qd, skin)
                scale = np.abs(en0[:6]).max()
                worst_e = max(worst_e, float(np.abs(en1 - en0).max() / scale))
                worst_a = max(worst_a, float(torch.sqrt(((acc1 - acc0) ** 2).sum() / (acc0 ** 2).sum())))
            ev, rb = kept.reuse_stats()
            tol = 2e-11 if blob else 1e-12      # (another order of sums with terms that cancel: see test_eval_reuse_matches_eval)
            ok = worst_e < tol and worst_a < tol
            msg = ""
        except msm_amd.MsmAmdError as e:
            ok, ev, rb, msg = False, -1, -1, str(e)[:200]
        fresh.close(); kept.close()
        bad += not ok
        print("case %3d %s N %6d A %s p %d abar %.2f L %d skin %.2f a blob %d: energy %.1e acc %.1e, %d rebuilds in %d %s" %
              (case, "ok  " if ok else "FAIL", N, tuple(round(x, 2) for x in A), p, abar, desc.L, frac, blob, worst_e, worst_a, rb, ev, msg), flush=True)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
