"""Random slab decompositions against the one-GPU evaluation: ranks share cuda:0 (gloo, host-staged collectives), random
grids, cutoffs, spline orders, level counts and rank counts, uniform systems and crowded balls on a slab face.  A search
for cases the fixed cases of test_gpu_slab.py miss.   python tests/fuzz_slab.py [cases] [seed]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import test_gpu_slab as T


def run(ncases, seed):
    import torch.multiprocessing as mp
    import msm_amd
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(ncases):
        M = int(rng.choice([32, 48, 64, 96]))
        world = int(rng.choice([w for w in (2, 3, 4, 6, 8) if M % w == 0 and (M // w) % 2 == 0]))
        p = int(rng.choice([4, 6]))
        L = int(rng.choice([l for l in (2, 3, 4) if M % (2 ** (l - 1)) == 0 and M // 2 ** (l - 1) >= 8]))
        a = float(rng.uniform(3.0, 5.0)) / M
        if a > 1.0 / 3.2:
            a = 1.0 / 3.2
        N = int(rng.choice([8000, 30000, 60000]))
        kind = "blob" if rng.random() < 0.25 else "explicit"
        mode = int(rng.choice([0, 0, 1]))
        spec = (kind, N, p, L, M, a)
        with tempfile.TemporaryDirectory() as tmp:
            try:
                mp.spawn(T._worker, args=(world, T._free_port(), spec, mode, True, tmp), nprocs=world, join=True)
            except Exception as e:
                print("case %2d ERROR world %d %s mode %d: %s" % (case, world, spec, mode, str(e)[-400:]), flush=True)
                bad += 1
                continue
            Np, desc = T._plan(msm_amd, spec)
            r, q = T._system(Np, spec)
            eng = msm_amd.Engine(desc)
            en1, acc1 = eng.eval(r, q)
            acc1 = acc1.cpu().numpy()
            eng.close()
            acc = np.zeros((Np, 3)); seen = np.zeros(Np, dtype=int)
            scale = np.abs(en1[:6]).max()
            de, used = 0.0, None
            for k in range(world):
                z = np.load(os.path.join(tmp, "rank%d.npz" % k))
                acc[z["idx"]] = z["acc"]; seen[z["idx"]] += 1
                de = max(de, float(np.abs(z["en"] - en1).max() / scale))
                used = int(z["used"])
            da = float(np.sqrt(((acc - acc1) ** 2).sum() / (acc1 ** 2).sum()))
            ok = (seen == 1).all() and de < 1e-12 and da < (1e-10 if kind == "blob" else 1e-12)
            bad += not ok
            print("case %2d %s world %d %s asked %d used %s: energy %.1e acc %.1e" % (case, "ok  " if ok else "FAIL", world, spec, mode, used, de, da), flush=True)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
