"""Accuracy / cost sweep of the periodic MSM against a converged Ewald sum (cutoff abar x spline order p x grid).

    python tools/accuracy_sweep.py [N] [nsub]

One charge-neutral random system; the reference is the Ewald sum of the same system on the GPU
(msm_amd_ewald_device: potential gradient at every particle and the total energy).  For every sweep point the GPU
evaluates the whole system; reported are the force RMS error sqrt(sum|f-f_ref|^2 / sum|f_ref|^2) over all particles,
the relative error of the total energy, and the time per evaluation.  With nsub < N the reference is evaluated at a
random subset of nsub particles against all N (potential gradient only: no total energy), which is what makes the
10M-particle sweep affordable on one GPU; the force error is then over the subset.
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import msm_amd
from bench import make_system


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    nsub = min(N, int(sys.argv[2])) if len(sys.argv) > 2 else N
    A = (1.0, 1.0, 1.0)
    r, q = make_system(N, 20261111 + 4)
    if nsub < N:   # the subset first (a random sample: the generator's particles are independent of their index)
        first = np.random.default_rng(5).permutation(N)
        r, q = r[first], q[first]
    lib = msm_amd.load()
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    # Ewald parameters: erfc(alpha rcut) and exp(-(pi k / alpha)^2) both below 1e-13; particles sorted by cell so
    # that the particles of a wave share their partners
    rcut = 0.25
    alpha = 5.6 / rcut
    nb = 32
    cell = [(rd[:, d] * nb).long().clamp_(0, nb - 1) for d in range(3)]
    key = (cell[2] * nb + cell[1]) * nb + cell[0]
    if nsub < N:   # subset and rest sorted separately, the subset stays in front
        key = key + (torch.arange(N, device=key.device) >= nsub).long() * nb ** 3
    perm = torch.argsort(key)
    torch.cuda.synchronize()
    t0 = time.time()
    u5, _, gs = msm_amd.ewald(rd[perm], qd[perm], A, alpha, rcut, int(np.ceil(5.6 * alpha / np.pi)), 0, nsub)
    torch.cuda.synchronize()
    aE = torch.empty((nsub, 3), dtype=torch.float64, device=gs.device)
    aE[perm[:nsub]] = gs
    aE = aE.cpu().numpy()
    uE = float(u5[4])
    have_energy = nsub == N
    print("# N=%d Ewald reference on the GPU %.2f s, total energy %.12e" % (N, time.time() - t0, uE), flush=True)
    rows = []
    for fine in (1, 2):
        for p in (4, 6):
            for abar in (3.0, 4.0, 5.0, 6.0):
                d0 = msm_amd.plan_auto(N, A, p, 10, abar, 0)
                if fine == 2 and N > 2000000 and abar > 4.0:
                    continue   # stencils of 39 and 47 taps on a 512^3 grid: seconds per evaluation
                if fine == 1:
                    d = d0
                else:   # same cutoff on a grid of half the spacing, one more level
                    M = [2 * m for m in d0.M]
                    L = d0.L + 1
                    hL = (2.0 ** (L - 1)) * A[0] / M[0]
                    beta = lib.choosebetaNew(1e-12, (2.0 ** L) * d0.a, hL)
                    kmax = int(np.ceil(lib.choosekmax(1e-12, beta, hL)))
                    d = msm_amd.plan_explicit(N, A, M, p, 10, L, d0.a, beta, kmax)
                try:
                    eng = msm_amd.Engine(d)
                except msm_amd.MsmAmdError as e:
                    print("# skipped abar=%g p=%d fine=%d: %s" % (abar, p, fine, e), flush=True)
                    continue
                en, acc = eng.eval(rd, qd)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    eng.eval(rd, qd)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / 5 * 1e3
                a_gpu = acc[:nsub].cpu().numpy()
                ferr = float(np.sqrt(((a_gpu - aE) ** 2).sum() / (aE ** 2).sum()))
                uerr = float(abs(en[6] - uE) / abs(uE)) if have_energy else None
                rows.append({"abar": abar, "p": p, "M": list(d.M), "L": d.L, "a": d.a, "force_rms_err": ferr,
                             "energy_rel_err": uerr, "ms_per_eval": ms})
                print("abar=%g p=%d M=%d L=%d  force RMS err %.3e  energy rel err %s  %.3f ms/eval"
                      % (abar, p, d.M[0], d.L, ferr, ("%.3e" % uerr) if uerr is not None else "n/a", ms), flush=True)
                eng.close()
    out = os.path.join(ROOT, "gpurun_out", "accuracy_sweep_N%d.json" % N)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump({"N": N, "nsub": nsub, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
