#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REAL REFERENCE.

Test infrastructure only.  Needs oracle/_ref/ (built by `make -C oracle ref`
from the sources under /root/reference; that tree does not travel to the GPU
box, the JSON fixtures written here do).

For every data file under /root/reference/msm/data this runs

    msmPeriodic <file> 4 10 4        (reference driver, msmPeriodic.c:1423)

in a scratch directory and records: the input system, every scalar printed by
print_stats (msmLibrary.c:290), the `Stencil[0]` dump (msmPeriodic.c:600-610),
the level-0 grid charges (msmPeriodic.c:723-733), and the three acceleration
files msm.acc / msm.acclong / msm.accshort (msmPeriodic.c:1369-1381,1445-1450).

It also records msmSimple (the 2-level engine with long-range forces, v=4/6),
msmExpected (grid-free energy components), ewaldClassical totals, and a sweep
of the parameter helpers chooseM/choosebetaNew/choosekmax through
_ref/libmsmref.so.

Usage:  python oracle/gen_golden.py [--jobs 6] [--only periodic|simple|expected|ewald|helpers|nonperiodic]
"""
import argparse
import ctypes
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor, as_completed

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
DATA = "/root/reference/msm/data"
OUT = os.path.join(HERE, "..", "tests", "golden")


def read_system(path):
    with open(path) as f:
        tok = f.read().split()
    L = [float(tok[0]), float(tok[1]), float(tok[2])]
    n = int(tok[3])
    vals = [float(t) for t in tok[4:4 + 4 * n]]
    r = []
    q = []
    for i in range(n):
        r += vals[4 * i:4 * i + 3]
        q.append(vals[4 * i + 3])
    return {"L": L, "N": n, "r": r, "q": q}


def read_acc(path):
    with open(path) as f:
        tok = f.read().split()
    n = int(tok[0])
    v = [float(t) for t in tok[1:1 + 3 * n]]
    # file order: all fx, all fy, all fz  -> AoS [3*i+d]
    out = [0.0] * (3 * n)
    for d in range(3):
        for i in range(n):
            out[3 * i + d] = v[d * n + i]
    return out


def parse_periodic_stdout(lines):
    """What msmPeriodic prints (msmPeriodic.c:600-610 stencil dump, :723-733 grid charges, :1340-1341 and
    msmLibrary.c:290-376 labelled values) -> {"stencil0", "gridcharge0", "scalars", "stats"}."""
    i = 0
    stencil = []
    grid = []
    scalars = {}
    stats = {}
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("Stencil["):
            i += 1
            while i < len(lines) and " x " in lines[i]:
                stencil.append(float(lines[i].split()[-1]))
                i += 1
            continue
        if ln.startswith("grid charge at first level"):
            i += 1
            while i < len(lines) and ":" not in lines[i] and lines[i].strip():
                grid.append(float(lines[i].split()[0]))
                i += 1
            continue
        m = re.match(r"^(.{0,20}?)\s*:\s*([-+0-9.eEnaif]+)\s*$", ln)
        if m:
            scalars[m.group(1).strip()] = float(m.group(2))
        elif len(ln) > 31 and ln[30] == " ":
            key = ln[:30].strip()
            val = ln[31:].strip()
            if key and key not in ("chi", "cvecx", "cvecy", "cvecz"):
                stats[key] = val
        i += 1
    return {"stencil0": stencil, "gridcharge0": grid, "scalars": scalars, "stats": stats}


def run_periodic(name, p=4, mu=10, abar=4):
    src = os.path.join(DATA, "benchmark_%s.txt" % name)
    tmp = tempfile.mkdtemp(prefix="msmref_")
    t0 = time.time()
    try:
        res = subprocess.run([os.path.join(REF, "msmPeriodic"), src, str(p), str(mu), repr(float(abar))],
                             cwd=tmp, capture_output=True, text=True, check=True)
        lines = res.stdout.splitlines()
        fx = {"program": "msmPeriodic", "args": [name, p, mu, abar], "system": read_system(src)}
        fx.update(parse_periodic_stdout(lines))
        fx["potential"] = float(lines[-1])
        fx["acc"] = read_acc(os.path.join(tmp, "msm.acc"))
        fx["acclong"] = read_acc(os.path.join(tmp, "msm.acclong"))
        fx["accshort"] = read_acc(os.path.join(tmp, "msm.accshort"))
        fx["wall_seconds"] = time.time() - t0
        return "msmPeriodic_%s.json" % name, fx
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_simple(testid, name, abar, mu, nu, M):
    tmp = tempfile.mkdtemp(prefix="msmref_")
    t0 = time.time()
    try:
        res = subprocess.run([os.path.join(REF, "msmSimple"), str(testid), repr(float(abar)), str(mu), str(nu),
                              str(M[0]), str(M[1]), str(M[2]), "1"],
                             cwd=tmp, capture_output=True, text=True, check=True)
        scal = {}
        # display_results is printed twice when debug=1 (msmSimple.c:285,2018); keep the first block
        for ln in res.stdout.splitlines():
            m = re.match(r"^(\S[^:]*?)\s*:\s*([-+0-9.eEnaif]+)\s*$", ln)
            if m and m.group(1).strip() not in scal:
                try:
                    scal[m.group(1).strip()] = float(m.group(2))
                except ValueError:
                    pass
        fx = {"program": "msmSimple", "args": [testid, abar, mu, nu] + list(M), "name": name, "scalars": scal}
        for fn, key in (("simple.acc", "acc"), ("simple.acc_expected", "acc_expected")):
            with open(os.path.join(tmp, fn)) as f:
                tok = f.read().split()
            n = int(tok[0])
            v = [float(t) for t in tok[1:]]
            # rows: for d in x,y,z: for i: fshort flong_real flong_four ftotal
            cols = {"fshort": [0.0] * (3 * n), "flong_real": [0.0] * (3 * n), "flong_four": [0.0] * (3 * n),
                    "ftotal": [0.0] * (3 * n)}
            for d in range(3):
                for i in range(n):
                    row = v[(d * n + i) * 4:(d * n + i) * 4 + 4]
                    for c, k in enumerate(("fshort", "flong_real", "flong_four", "ftotal")):
                        cols[k][3 * i + d] = row[c]
            fx[key] = cols
            fx["N"] = n
        for fn in os.listdir(tmp):
            if fn.startswith("stencil_") and fn.endswith(".dat"):
                with open(os.path.join(tmp, fn)) as f:
                    vals = [float(t) for t in f.read().split()]
                fx["Kfourier" if "fourier" in fn else "Kl"] = vals
        fx["wall_seconds"] = time.time() - t0
        return "msmSimple_%s_nu%d_mu%d_M%d.json" % (name, nu, mu, M[0]), fx
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_expected(name, pmax, kmax, p=4, mu=10, abar=4):
    src = os.path.join(DATA, "benchmark_%s.txt" % name)
    t0 = time.time()
    res = subprocess.run([os.path.join(REF, "msmExpected"), src, str(p), str(mu), repr(float(abar)), str(pmax),
                          str(kmax)], capture_output=True, text=True, check=True)
    scal = {}
    for ln in res.stdout.splitlines():
        m = re.match(r"^(.+?)\s*:\s*([-+0-9.eEnaif]+)\s*$", ln)
        if m:
            scal[m.group(1).strip()] = float(m.group(2))
    return "msmExpected_%s.json" % name, {"program": "msmExpected", "args": [name, p, mu, abar, pmax, kmax],
                                          "scalars": scal, "wall_seconds": time.time() - t0}


def run_ewald(name, alpha, cd, cf):
    src = os.path.join(DATA, "benchmark_%s.txt" % name)
    res = subprocess.run([os.path.join(REF, "ewaldClassical"), src, repr(float(alpha)), repr(float(cd)),
                          repr(float(cf))], capture_output=True, text=True, check=True)
    return "ewaldClassical_%s.json" % name, {"program": "ewaldClassical", "args": [name, alpha, cd, cf],
                                             "total": float(res.stdout.split()[-1])}


def run_nonperiodic(name):
    """msm/nonperiodic.c: open-boundary direct energy with every charge forced to 1; prints `u is %15.8e`."""
    src = os.path.join(DATA, "benchmark_%s.txt" % name)
    res = subprocess.run([os.path.join(REF, "nonperiodic"), src], capture_output=True, text=True, check=True)
    m = re.search(r"u is\s+([-+0-9.eE]+)", res.stdout)
    return "nonperiodic_%s.json" % name, {"program": "nonperiodic", "args": [name], "u": float(m.group(1)),
                                          "printed": m.group(1), "system": read_system(src)}


def run_helpers():
    lib = ctypes.CDLL(os.path.join(REF, "libmsmref.so"))
    lib.chooseM.restype = ctypes.c_int
    lib.chooseM.argtypes = [ctypes.c_double, ctypes.c_double]
    lib.choosebetaNew.restype = ctypes.c_double
    lib.choosebetaNew.argtypes = [ctypes.c_double] * 3
    lib.choosekmax.restype = ctypes.c_double
    lib.choosekmax.argtypes = [ctypes.c_double] * 3
    lib.chooseL.restype = ctypes.c_int
    lib.chooseL.argtypes = [ctypes.c_int] * 4
    lib.mylog2.restype = ctypes.c_double
    lib.mylog2.argtypes = [ctypes.c_double]
    lib.minOf3Integers.restype = ctypes.c_int
    lib.minOf3Integers.argtypes = [ctypes.c_int] * 3
    rows = []
    for N in (2, 8, 16, 27, 64, 100, 128, 216, 512, 1000, 4096, 10000, 30000, 100000, 1000000, 10000000):
        for A in (1.0, 2.0, 3.5, 8.0):
            for abar in (3.0, 4.0, 5.0):
                vol = A ** 3
                ht = (vol / N) ** (1.0 / 3.0)
                M = lib.chooseM(ht, A)
                if M <= 0:
                    continue
                hx = A / M
                a = abar * ht
                for L in (1, 2, 3, 4):
                    aL = (2.0 ** L) * a
                    hL = (2.0 ** (L - 1)) * hx
                    beta = lib.choosebetaNew(1e-12, aL, hL)
                    km = lib.choosekmax(1e-12, beta, hL)
                    rows.append({"N": N, "A": A, "abar": abar, "L": L, "htilde": ht, "M": M, "beta": beta,
                                 "kmax_double": km, "chooseL": lib.chooseL(M, M, M, N)})
    extra = {"mylog2": [[x, lib.mylog2(x)] for x in (0.3, 1.0, 2.0, 7.5, 1e6)],
             "minOf3Integers": [[a, b, c, lib.minOf3Integers(a, b, c)]
                                for (a, b, c) in ((1, 2, 3), (3, 2, 1), (2, 1, 3), (5, 5, 5), (-1, 0, 1))]}
    return "helpers.json", {"program": "libmsmref.so helpers", "rows": rows, "extra": extra}


def _call(job):
    fn, args = job
    return globals()[fn](*args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=6)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    jobs = []
    if a.only in ("", "periodic"):
        # longest first
        for nm in ("changaN512", "NaClN512", "NaNaN512", "CsClN128", "changaN64", "NaClN64", "NaNaN64", "CsClN16",
                   "changaN8", "NaClN8", "NaNaN8", "CsClN2"):
            jobs.append(("run_periodic", (nm,)))
    if a.only in ("", "simple"):
        jobs += [("run_simple", (20, "changaN64", 4, 10, 4, (4, 4, 4))),
                 ("run_simple", (19, "changaN8", 4, 10, 4, (2, 2, 2))),
                 ("run_simple", (19, "changaN8", 4, 10, 6, (2, 2, 2))),
                 ("run_simple", (19, "changaN8", 4, 10, 4, (3, 3, 3))),
                 ("run_simple", (19, "changaN8", 3, 6, 6, (4, 4, 4))),
                 ("run_simple", (7, "NaClN8", 4, 10, 4, (2, 2, 2))),
                 ("run_simple", (14, "CsClN16", 4, 10, 4, (3, 3, 3))),
                 ("run_simple", (22, "simpleN2", 4, 2, 4, (2, 2, 2)))]
    if a.only in ("", "expected"):
        jobs += [("run_expected", ("changaN512", 3, 10)),
                 ("run_expected", ("changaN64", 4, 10)),
                 ("run_expected", ("changaN8", 5, 10)),
                 ("run_expected", ("NaClN8", 5, 10)),
                 ("run_expected", ("NaNaN8", 5, 10)),
                 ("run_expected", ("CsClN16", 5, 10))]
    if a.only in ("", "ewald"):
        jobs += [("run_ewald", ("NaClN8", 6, 10, 30)), ("run_ewald", ("changaN8", 12, 10, 30)),
                 ("run_ewald", ("NaClN64", 6, 10, 30)), ("run_ewald", ("changaN64", 12, 10, 30)),
                 ("run_ewald", ("changaN512", 24, 10, 60))]
    if a.only in ("", "helpers"):
        jobs.append(("run_helpers", ()))
    if a.only in ("", "nonperiodic"):
        jobs += [("run_nonperiodic", (nm,)) for nm in ("changaN8", "changaN64", "changaN512", "NaClN64", "CsClN128")]
    with ProcessPoolExecutor(max_workers=a.jobs) as ex:
        futs = {ex.submit(_call, j): j for j in jobs}
        for fu in as_completed(futs):
            j = futs[fu]
            try:
                fn, fx = fu.result()
            except Exception as e:  # keep going; report
                print("FAILED", j, e, file=sys.stderr, flush=True)
                continue
            with open(os.path.join(OUT, fn), "w") as f:
                json.dump(fx, f, separators=(",", ":"))
            print("wrote", fn, "%.1fs" % fx.get("wall_seconds", 0.0), flush=True)


if __name__ == "__main__":
    main()
