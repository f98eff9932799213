"""Accuracy / cost sweep of the periodic MSM against a full Ewald sum (cutoff abar x spline order p x grid).

    python tools/accuracy_sweep.py [N] [nsub]

One charge-neutral random system; the reference values are a converged Ewald sum (CPU, oracle/ewald_oracle.c:
real and reciprocal parts with image loops) for the first nsub particles.  For every sweep point the GPU
evaluates the whole system; reported are the force RMS error sqrt(sum|f-f_ref|^2 / sum|f_ref|^2) over the subset,
the relative error of the subset's energy share u_i = 1/2 q_i phi_i ... (not available per particle from the
engine, so the total energy is compared with the Ewald total instead), and the time per evaluation.
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import msm_amd
import oracle_lib as O


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    nsub = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    A = (1.0, 1.0, 1.0)
    r, q = O.neutral_system(N, A, seed=20261111 + 4)
    lib = msm_amd.load()
    ht = (1.0 / N) ** (1.0 / 3.0)
    t0 = time.time()
    # Ewald parameters: erfc(alpha rcut) and exp(-(pi k / alpha)^2) both below 1e-13
    alpha = 5.6 / 0.25
    uE, aE = O.ewald(r, q, A, alpha=alpha, rcut=0.25, kcut=int(np.ceil(5.6 * alpha / np.pi)), nsub=N if N <= 20000 else nsub,
                     nthreads=16)
    have_energy = N <= 20000
    aE = aE[:nsub]
    print("# N=%d nsub=%d Ewald reference %.1f s (energy reference: %s)" % (N, nsub, time.time() - t0, have_energy), flush=True)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    rows = []
    for fine in (1, 2):
        for p in (4, 6):
            for abar in (3.0, 4.0, 5.0, 6.0):
                d0 = msm_amd.plan_auto(N, A, p, 10, abar, 0)
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
