"""Random open-boundary plans (msm3d of msm/msmLibrary.h:216) against the CPU oracle's open mode: random clouds (boxes of
unequal edges, anywhere in space), spline orders, grid spacings, cutoffs and level counts.
python tests/fuzz_open.py [cases] [seed]"""
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
        N = int(rng.choice([300, 2000, 8000, 25000]))
        box = tuple(float(x) for x in rng.uniform(0.5, 1.5, 3))
        shift = tuple(float(x) for x in rng.uniform(-20, 20, 3)) if rng.random() < 0.5 else (0.0, 0.0, 0.0)
        p = int(rng.choice([4, 6])); mu = int(rng.choice([6, 10])); L = int(rng.choice([1, 2, 3]))
        h = 1.0 / float(rng.choice([10, 12, 16, 20, 28])); a = float(rng.uniform(3.0, 4.8)) * h
        pts = rng.random((N, 3))
        if rng.random() < 0.3:      # most of the cloud in one corner
            pts[: N // 2] = pts[: N // 2] * 0.15
        r = pts * np.array(box) + np.array(shift)
        # no two particles closer than a fifth of the mean spacing: the force of a pair 1e-4 apart carries the last bit of a
        # coordinate of size 20 as 1e-11 of itself, on either side of the comparison
        from scipy.spatial import cKDTree
        dmin = 0.2 * (box[0] * box[1] * box[2] / N) ** (1.0 / 3.0)
        drop = {j for _, j in cKDTree(r).query_pairs(dmin)}
        r = r[[i for i in range(N) if i not in drop]]
        N = len(r)
        q = np.where(np.arange(N) % 2 == 0, 1.0, -1.0) * rng.uniform(0.5, 1.5, N)
        try:
            desc = msm_amd.plan_open(N, r.min(axis=0), r.max(axis=0), p, mu, h, a, L)
        except msm_amd.MsmAmdError as e:
            print("case %3d no plan: %s" % (case, str(e)[:100]))
            continue
        sp = O.SpecPlan.explicit(N, list(desc.A), list(desc.M), p, mu, L, a, 1.0, 0, nthreads=8, open_boundary=True)
        ref = sp.eval(r - np.array(list(desc.origin)), q)
        aw = sp.pairs(r - np.array(list(desc.origin)), q)[0]
        sp.close()
        eng = msm_amd.Engine(desc)
        rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
        en, acc = eng.eval(rd, qd)
        acc = acc.cpu().numpy()
        eng.stage_bin(rd, qd)
        acc_s = torch.zeros((N, 3), dtype=torch.float64, device="cuda:0")
        eng.stage_pairs(acc_s)
        acc_s = acc_s.cpu().numpy()
        eng.close()
        scale = np.abs(ref["energies"][:6]).max()
        de = float(np.abs(en - ref["energies"]).max() / scale)
        da = float(np.sqrt(((acc - ref["acc"]) ** 2).sum() / (ref["acc"] ** 2).sum()))
        ok = de < 1e-12 and da < 1e-12
        bad += not ok
        if not ok:
            den = np.sqrt((ref["acc"] ** 2).sum())
            worst = np.argsort(-np.abs(acc - ref["acc"]).max(axis=1))[:3]
            print("      short-range part %.1e, the rest %.1e (over the norm of the whole); worst particles: diff %s of %s" %
                  (np.sqrt(((acc_s - aw) ** 2).sum()) / den, np.sqrt((((acc - acc_s) - (ref["acc"] - aw)) ** 2).sum()) / den,
                   np.abs(acc - ref["acc"]).max(axis=1)[worst].tolist(), np.abs(ref["acc"]).max(axis=1)[worst].tolist()))
        print("case %3d %s N %6d box %s shift %s p %d mu %d h 1/%d a/h %.2f L %d -> M %s: energy %.1e acc %.1e" %
              (case, "ok  " if ok else "FAIL", N, tuple(round(x, 2) for x in box), tuple(round(x, 1) for x in shift), p, mu, round(1 / h),
               a / h, L, list(desc.M), de, da), flush=True)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
