#!/usr/bin/env python3
"""MSM energy+force evaluations per second at N = 1M periodic on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one whole evaluation of the hot path on particles already resident in HBM: binning ->
short-range pair sum -> anterpolation -> restrictions -> per-level grid convolutions -> Fourier top level ->
prolongations -> interpolation -> energy reductions (and, for N > 1 ranks, the level-0 charge all-gather and
the energy all-reduce).  Position-independent setup (beta, kmax, omega', stencils, allocation, communicator)
is outside the timed region.  Consecutive steps evaluate different (displaced) copies of the system, made
beforehand, so nothing can be cached.  In front of the W warm-up steps the same evaluation runs --spinup-steps times
(default 60, untimed): the GPU needs about 30 ms of load to reach its clocks.  Evaluations are queued back to back (msm_amd_eval_enqueue): each one's energies are copied asynchronously into its own slot
of a pinned host ring, and the host synchronises once at the end of the timed region.

Workload (the configuration the headline metric is quoted on): 1,000,000 charge-neutral
particles in a periodic unit cube, quintic B-spline (p = 6) unless --order 4, abar = 4, mu = 10, four grid
levels 96 -> 48 -> 24 -> 12 plus the Fourier top level.  N > 1 keeps the total work fixed (strong scaling):
the same million particles are split into z slabs.

--gpus N > 1 started without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself: the parent, which
never touches the GPU, runs one child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set,
forwards rank 0's JSON line and exits non-zero if any rank did.  Under torch.distributed.run the flag must equal
WORLD_SIZE.  N ranks need N devices: the run fails otherwise instead of measuring fewer (MSM_AMD_BENCH_BACKEND=gloo is a
functional rehearsal with ranks sharing devices and is labelled as such in the line).

`list_reuse` in the line is a separate measurement of msm_amd_eval_reuse (neighbour lists kept between evaluations) on the
same steps; it is never `value`.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (k_pair_clusters_filtered, the short-range pair
sum): algorithmic flops (ordered in-cutoff pairs x 34 flop at p=4 / 42 at p=6) over that kernel's mean
duration, measured with a pair of HIP events recorded around it on the stream it runs on.  `traffic` is the
HBM-side byte count of one launch from rocprofv3 PMC passes (the newest profiles/r*_pair_traffic.json; null if none
is there).  `cpu_baseline` is the CPU oracle (a port of the same mathematics, oracle/msm_spec.c) timed on
this box's host cores on one evaluation of the same system (`reference_literal` inside it: the reference's own
algorithm on one core at the sizes it can run); `parity` compares the GPU result with it.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP64_PEAK_TFLOPS = 78.6   # gfx950 vector FP64: 256 CU x 128 flop/clk x 2.4 GHz
FLOP_PER_PAIR = {4: 34, 6: 42}


def splitmix64(seed, n):
    """The first n outputs of SplitMix64 seeded with `seed` (Steele, Lea, Flood 2014): state += 0x9E3779B97F4A7C15, then
    two xor-shift-multiply rounds; the i-th state is seed + (i + 1) * gamma, so the stream vectorises."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def make_system(N, seed):
    """N particles uniform in the unit cube (53-bit fractions of a SplitMix64 stream), charges +1 / -1 alternating:
    exactly neutral."""
    r = ((splitmix64(seed, 3 * N) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(N, 3)
    q = np.where(np.arange(N) % 2 == 0, 1.0, -1.0)
    return r, q


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup-steps", type=int, default=60,
                    help="untimed evaluations in front of the warm-up steps, the same on every rank: the GPU takes about 30 ms of "
                         "load to reach its clocks (the pair kernel of the first evaluation after a pause takes 0.81 ms, that of "
                         "the thirtieth 0.72), and the steps before the timed region leave it idle while the system is made")
    ap.add_argument("--particles", type=int, default=1000000)
    ap.add_argument("--order", type=int, default=6, choices=(4, 6))
    ap.add_argument("--levels", type=int, default=0, help="grid levels (0: the reference's formula, capped)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pair-variant", type=int, default=-1)
    ap.add_argument("--no-accuracy", action="store_true", help="skip the Ewald reference (error of the MSM result)")
    ap.add_argument("--opts", default="", help="engine options for A/B runs, key=value,key=value (msm_amd_set_option)")
    ap.add_argument("--no-list-reuse", action="store_true", help="skip the separate msm_amd_eval_reuse measurement")
    ap.add_argument("--reuse-skin", type=float, default=0.05, help="skin of the list-reuse measurement, in units of the cutoff a")
    ap.add_argument("--reference-binary", action="store_true",
                    help="cpu_baseline: also run the real msmPeriodic (oracle/_ref, when it was built) at N = 8 and 64 (about 10 s)")
    ap.add_argument("--literal-n", type=int, default=0,
                    help="cpu_baseline: also time the literal mirror on a synthetic system of this many particles "
                         "(4096 takes about three minutes on one core)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; they must agree" % (args.gpus, world))

    import torch
    import msm_amd
    from msm_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path in msm_amd)")
    if os.environ.get("MSM_AMD_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks need %d devices, %d visible" % (world, world, torch.cuda.device_count()))
    # MSM_AMD_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives staged through the host): a functional check, never a measurement
    backend = os.environ.get("MSM_AMD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend=backend)

    N, p, mu, abar = args.particles, args.order, 10, 4.0
    A = (1.0, 1.0, 1.0)
    r_all, q_all = make_system(N, 20261111 + 2)
    desc = msm_amd.plan_auto(N, A, p, mu, abar, args.levels)

    if world == 1:
        eng = msm_amd.Engine(desc, device=local_rank)
        if args.pair_variant >= 0:
            eng.set_option("pair_variant", args.pair_variant)
        stepper = SingleGpu(torch, eng, r_all, q_all, dev)
    else:
        from msm_amd.slab import SlabRank
        # as a time-stepping code would: the rank's work on a stream of its own (kept current from here on), the engine
        # created on it.  On the legacy default stream the two-stream step has to move to streams of its own and back,
        # two hand-overs between streams per evaluation (about 20 us of 0.39 ms on a rank of 8).
        side = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(side)
        stepper = SlabRank(torch, dist, msm_amd, desc, r_all, q_all, rank, world, local_rank, perturb=True, stream=side)
        eng = stepper.eng
        if args.pair_variant >= 0:
            eng.set_option("pair_variant", args.pair_variant)
        stepper.profile = SlabProfiler(torch, stepper).profile

    for kv in filter(None, args.opts.split(",")):
        k, v = kv.split("=")
        eng.set_option(k, int(v))

    exchange = ""
    multi = None
    if world > 1:
        m = eng.slab_exchange_mode(world)
        exchange = {1: "level-0 charge all-gather", 2: "level-0 charge halo exchange + level-1 all-gather"}.get(m, str(m))
        exchange += " (collectives: %s)" % type(stepper.coll).__name__
        # what the communicator itself reports (not what the launcher asked for), and what every rank holds
        rccl_ranks = stepper.coll.comm_info()[0] if hasattr(stepper.coll, "comm_info") else None
        mine = torch.tensor([stepper.n_own, stepper.n_tot, rccl_ranks if rccl_ranks is not None else -1], dtype=torch.int64,
                            device=dev if backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = [[int(x) for x in t.cpu()] for t in every]
        multi = {"rccl_ranks": rccl_ranks if all(e[2] == every[0][2] for e in every) else [e[2] for e in every],
                 "collectives": type(stepper.coll).__name__, "exchange_mode": exchange,
                 "per_rank": [{"n_owned": e[0], "n_total": e[1]} for e in every]}
        if sum(e[0] for e in every) != N:
            raise SystemExit("bench.py: the ranks own %d particles of %d" % (sum(e[0] for e in every), N))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ref_eval = None
    if world == 1 and rank == 0 and not (args.no_cpu_baseline and args.no_accuracy):
        ref_eval = stepper.eval_unperturbed()   # untimed: the same positions the CPU oracle and the Ewald sum get

    # evaluations are queued back to back on every rank too: the reduced sums stay on the device until they are read
    run_step = stepper.step if world == 1 else (lambda: stepper.step(want_energies=False))
    spinup = max(0, args.spinup_steps) * world   # (a rank's step is shorter: the same time under load on every rank count)
    for _ in range(spinup):
        run_step()
    for _ in range(args.warmup):
        run_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    energies = stepper.last_energies

    # per-kernel durations with HIP events on the stream the kernels run on (separate, untimed pass)
    nprof = max(3, min(10, args.steps))
    acc_ms = stepper.profile(nprof)

    # A separate measurement, never `value`: the same steps through msm_amd_eval_reuse, which keeps its neighbour lists
    # (built with cutoff a + skin) while no particle has moved more than skin / 2 - checked on the device in every call.
    reuse = None
    if world == 1 and not args.no_list_reuse:
        reuse = stepper.measure_reuse(args.reuse_skin * desc.a, args.warmup, args.steps, nprof)
        reuse["skin_over_a"] = args.reuse_skin
        ps = reuse.get("pair_kernel_ms", 0.0) * 1e-3
        if ps > 0:
            reuse["pair_kernel_tflops"] = N * (4.0 * math.pi / 3.0) * abar ** 3 * FLOP_PER_PAIR[p] / ps / 1e12
            reuse["pair_kernel_frac_of_peak"] = reuse["pair_kernel_tflops"] / FP64_PEAK_TFLOPS

    # ordered in-cutoff pairs handled by one launch (of rank 0): its owned particles x the neighbours within the cutoff
    pairs = (stepper.n_own if world > 1 else N) * (4.0 * math.pi / 3.0) * abar ** 3
    flops = pairs * FLOP_PER_PAIR[p]
    pair_s = (acc_ms.get("pair_kernel") or acc_ms.get("pairs") or acc_ms.get("total", 0.0)) * 1e-3
    achieved = flops / pair_s / 1e12 if pair_s > 0 else 0.0

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, cpu_out = cpu_baseline(desc, r_all, q_all, args.reference_binary, args.literal_n)
        en_gpu, acc_gpu = ref_eval
        en_cpu = cpu_out["energies"]
        scale = float(np.abs(en_cpu[:6]).max())
        da = acc_gpu - cpu_out["acc"]
        parity = {"system": "the unperturbed benchmark system (the timed steps displace it)",
                  "energy_max_abs_diff_over_largest_component": float(np.abs(en_gpu - en_cpu).max() / scale),
                  "largest_component": scale,
                  "utotal_gpu": float(en_gpu[6]), "utotal_cpu": float(en_cpu[6]),
                  "utotal_rel_diff": float(abs(en_gpu[6] - en_cpu[6]) / abs(en_cpu[6])),
                  "acc_rel_rms_diff": float(np.sqrt((da ** 2).sum() / (cpu_out["acc"] ** 2).sum())),
                  "tolerance": 1e-12,
                  "tolerance_note": "1e-12 of the largest energy component and 1e-12 relative RMS of the accelerations; the "
                                    "total is what cancellation leaves of components up to sixty times larger, so relative "
                                    "to itself it is held to 1e-10"}
    accuracy = None
    if rank == 0 and world == 1 and not args.no_accuracy:
        accuracy = ewald_accuracy(torch, msm_amd, r_all, q_all, A, ref_eval[0], ref_eval[1], dev)
    traffic = None
    tpath = next((t for t in (os.path.join(ROOT, "profiles", "r%02d_pair_traffic.json" % k) for k in (3, 2, 1)) if os.path.exists(t)),
                 "")
    if world == 1 and N == 1000000 and os.path.exists(tpath):  # the counters were taken on the default workload
        with open(tpath) as f:
            traffic = json.load(f).get("order%d" % p, {}).get("hbm_bytes_per_launch")

    if rank == 0:
        out = {
            "metric": "MSM energy+force evals/sec at N=1M periodic",
            "value": args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "spinup_steps": spinup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "periodic unit cube, %d charge-neutral random particles, B-spline order p=%d, "
                                   "abar=4, mu=10, grid %s, %d levels + Fourier top (kmax=%d)"
                                   % (N, p, "x".join(str(m) for m in desc.M), desc.L, desc.kmax),
                       "particles": N, "order": p, "levels": desc.L, "grid": list(desc.M),
                       "parallelism": "single GPU" if world == 1 else "z-slabs x%d, %s" % (world, exchange),
                       **({"engine_options": args.opts} if args.opts else {}),
                       **({} if backend == "nccl" else {"rehearsal": "ranks share devices, %s collectives through the host: not a measurement" % backend})},
            "roofline": {"kernel": "k_pair_clusters_filtered (short-range pair sum)%s" % (" of rank 0" if world > 1 else ""),
                         "bound": "valu_fp64", "achieved": achieved,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                         "traffic": traffic, "kernel_ms": pair_s * 1e3,
                         "sustained_peak": 64.0, "frac_of_sustained": achieved / 64.0,
                         "sustained_peak_source": "measured issue rate of v_fma_f64 (tools/micro/valu_rate.hip: one per 2.04 ns per "
                                                  "SIMD with four waves resident = 64 TFLOP/s for the chip); `frac` stays "
                                                  "against the data-sheet peak",
                         "note": "FP64 vector FMA bound (the FP64 MFMA peak is the same 78.6 TF); algorithmic flops = "
                                 "N*(4pi/3)*abar^3 ordered pairs x %d flop" % FLOP_PER_PAIR[p]},
            "kernel_ms": acc_ms,
            **({"multi_gpu": multi} if multi is not None else {}),
            **({"list_reuse": reuse} if reuse is not None else {}),
            "energies_last_timed_step": dict(zip(capi.U_NAMES, [float(x) for x in energies])) if energies is not None else None,
            "energies_note": "of the last timed step, whose positions are the benchmark system displaced by up to 5e-5; the "
                             "unperturbed system is the one in `parity` and `accuracy`",
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["parity"] = parity
        if accuracy is not None:
            out["accuracy"] = accuracy
        print(json.dumps(out))
    if dist is not None:
        # the library's communicator goes before the process group it was negotiated over
        barrier()
        if getattr(stepper, "coll", None) is not None:
            stepper.coll.close()
        dist.destroy_process_group()


def launch_ranks(n):
    """Start n ranks of this script (one process per GPU, rendezvous on 127.0.0.1), wait for them, print rank 0's JSON line.
    Runs in a process that never initialises the GPU (torch.cuda.device_count() does not) and replaces no process image:
    the ranks are children.  Returns the exit status: 0 only if every rank returned 0 and rank 0 printed its line."""
    import socket
    import subprocess
    import threading
    import torch
    have = torch.cuda.device_count()
    shared = os.environ.get("MSM_AMD_BENCH_BACKEND", "nccl") != "nccl"
    if have < 1 or (have < n and not shared):
        print("bench.py: --gpus %d needs %d devices, %d visible (no run with fewer ranks is made in its place)" % (n, n, have),
              file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for k in range(n):
        env = dict(os.environ, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if k == 0 else sys.stderr, text=True))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.read().splitlines()), daemon=True)
    reader.start()
    status = 0
    pending = set(range(n))
    while pending:
        for k in sorted(pending):
            rc = procs[k].poll()
            if rc is None:
                continue
            pending.discard(k)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                print("bench.py: rank %d exited with status %d; stopping the other ranks" % (k, rc), file=sys.stderr)
                for j in pending:
                    procs[j].terminate()      # (the exact children started above, by handle)
        time.sleep(0.05)
    reader.join(timeout=10)
    out = [ln for ln in lines if ln.startswith("{")]
    for ln in lines:
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if status == 0 and not out:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        status = 1
    if status == 0:
        print(out[-1])
    return status


def ewald_accuracy(torch, msm_amd, r_all, q_all, A, en_msm, acc_msm, dev):
    """Error of the MSM evaluation against a converged Ewald sum of the same system, computed on the GPU
    (msm_amd_ewald_device, all particles; outside the timed region)."""
    rcut = 0.25 * min(A)
    alpha = 5.6 / rcut                                   # erfc(alpha rcut) ~ 2e-15
    kcut = int(math.ceil(5.6 * alpha * max(A) / math.pi))  # exp(-(pi k / alpha)^2) ~ 2e-14 at the edge of the cube
    r = torch.as_tensor(r_all, device=dev)
    q = torch.as_tensor(q_all, device=dev)
    # neighbours in space next to each other in memory: the 64 particles of a wave then share their in-range partners
    nb = 32
    cell = [(r[:, d] / A[d] * nb).long().clamp_(0, nb - 1) for d in range(3)]
    perm = torch.argsort((cell[2] * nb + cell[1]) * nb + cell[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    u5, _, grad_sorted = msm_amd.ewald(r[perm], q[perm], A, alpha, rcut, kcut)
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    grad = torch.empty_like(grad_sorted)
    grad[perm] = grad_sorted
    d = torch.as_tensor(acc_msm, device=dev) - grad
    return {"reference": "Ewald sum on the GPU (alpha=%g, rcut=%g, |k_d|<=%d), all %d particles" % (alpha, rcut, kcut, len(q_all)),
            "force_rms_rel_error": float(torch.sqrt((d ** 2).sum() / (grad ** 2).sum())),
            "energy_rel_error": float(abs(en_msm[6] - u5[4]) / abs(u5[4])),
            "utotal_ewald": float(u5[4]), "utotal_msm": float(en_msm[6]), "reference_seconds": secs}


class SingleGpu:
    POSITION_SETS = 16

    def __init__(self, torch, eng, r, q, dev):
        self.torch = torch
        self.eng = eng
        self.r = torch.as_tensor(r, device=dev)
        self.q = torch.as_tensor(q, device=dev)
        self.acc = torch.empty_like(self.r)
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        # Consecutive steps evaluate different inputs: a few displaced copies of the system (up to 5e-5 per coordinate)
        # made here, before the timed region, and evaluated in turn - inputs resident in HBM, no displacement kernel
        # between the evaluations.
        delta = (torch.rand(self.r.shape, generator=g, dtype=torch.float64, device=dev) - 0.5) * 1e-4
        K = self.POSITION_SETS
        self.r_sets = torch.stack([self.r + delta * (((k + 1) / K) * (1.0 if k % 2 == 0 else -1.0)) for k in range(K)])
        self.ring = torch.zeros((64, 7), dtype=torch.float64).pin_memory()
        self.nstep = 0

    def step(self):
        # queued without a host synchronisation: every evaluation's energies are copied to their own slot of a
        # pinned host ring, and the caller synchronises once (bench: at the end of the timed region)
        slot = self.ring[self.nstep % self.ring.shape[0]]
        self.eng.eval_enqueue(self.r_sets[self.nstep % self.POSITION_SETS], self.q, self.acc, slot)
        self.nstep += 1

    @property
    def last_energies(self):
        self.torch.cuda.synchronize()
        return self.ring[(self.nstep - 1) % self.ring.shape[0]].numpy().copy() if self.nstep else None

    def eval_unperturbed(self):
        en, acc = self.eng.eval(self.r, self.q, self.acc)
        return np.array(en), acc.cpu().numpy().copy()

    def measure_reuse(self, skin, warmup, steps, nprof):
        """The same displaced copies of the system through msm_amd_eval_reuse: evaluations queued back to back, one
        synchronisation at the end; then the stage split of a few profiled steps."""
        t = self.torch
        k0 = self.nstep
        for k in range(max(1, warmup)):                 # (the first call builds the lists)
            self.eng.eval_reuse(self.r_sets[(k0 + k) % self.POSITION_SETS], self.q, skin, self.acc, want_energies=False)
        t.cuda.synchronize()
        _, rb0 = self.eng.reuse_stats()
        t0 = time.perf_counter()
        for k in range(steps):
            self.eng.eval_reuse(self.r_sets[(k0 + k) % self.POSITION_SETS], self.q, skin, self.acc, want_energies=False)
        t.cuda.synchronize()
        dt = time.perf_counter() - t0
        _, rb1 = self.eng.reuse_stats()
        # the last step against msm_amd_eval of the same positions
        rs = self.r_sets[(k0 + steps - 1) % self.POSITION_SETS]
        en_r, acc_r = self.eng.eval_reuse(rs, self.q, skin)
        acc_r = acc_r.clone()
        en_e, acc_e = self.eng.eval(rs, self.q)
        scale = float(np.abs(en_e[:6]).max())
        out = {"evals_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "rebuilds_in_timed_steps": rb1 - rb0,
               "vs_msm_amd_eval": {"energy_max_abs_diff_over_largest_component": float(np.abs(en_r - en_e).max() / scale),
                                   "acc_rel_rms_diff": float(t.sqrt(((acc_r - acc_e) ** 2).sum() / (acc_e ** 2).sum()))},
               "note": "msm_amd_eval_reuse: neighbour lists per group of 4 particles, built with cutoff a + skin and kept while no "
                       "particle has moved more than skin / 2 (checked on the device by every call, which rebuilds them itself "
                       "when needed); equal to msm_amd_eval to rounding, not bit for bit. A separate figure: `value` is "
                       "msm_amd_eval, which assumes nothing about the previous call."}
        self.eng.set_profiling(True)
        acc = {}
        for k in range(nprof):
            self.eng.eval_reuse(self.r_sets[k % self.POSITION_SETS], self.q, skin, self.acc, want_energies=False)
            t.cuda.synchronize()
            tm = dict(self.eng.last_timings())
            try:
                tm["pair_kernel"] = self.eng.last_kernel_ms("pair_kernel")
            except Exception:
                pass
            for kk, v in tm.items():
                acc[kk] = acc.get(kk, 0.0) + v / nprof
        self.eng.set_profiling(False)
        out["kernel_ms"] = acc
        out["pair_kernel_ms"] = acc.get("pair_kernel", 0.0)
        return out

    def profile(self, n):
        # the engine brackets every stage, and the pair kernel by itself, with hipEventRecord on its own stream
        self.eng.set_profiling(True)
        acc = {}
        for _ in range(n):
            self.step()
            self.torch.cuda.synchronize()
            t = dict(self.eng.last_timings())
            try:
                t["pair_kernel"] = self.eng.last_kernel_ms("pair_kernel")
            except Exception:
                pass
            for k, v in t.items():
                acc[k] = acc.get(k, 0.0) + v / n
        self.eng.set_profiling(False)
        return acc


class SlabProfiler:
    """Stage durations of a slab rank: events recorded on the current (= the engine's, null) stream."""

    def __init__(self, torch, slab):
        self.torch, self.slab = torch, slab

    def profile(self, n):
        t = self.torch
        acc = {}
        for _ in range(n):
            evs = []

            def mark(name):
                e = t.cuda.Event(enable_timing=True)
                e.record()
                evs.append((name, e))
            eng = self.slab.eng
            if hasattr(eng, "set_profiling"):
                eng.set_profiling(True)      # brackets the pair kernel with its own events
            self.slab.step(marks=mark)
            t.cuda.synchronize()
            for (name, e0), (_, e1) in zip(evs[:-1], evs[1:]):
                acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1) / n
            acc["total"] = acc.get("total", 0.0) + evs[0][1].elapsed_time(evs[-1][1]) / n
            if hasattr(eng, "set_profiling"):
                try:
                    acc["pair_kernel"] = acc.get("pair_kernel", 0.0) + eng.last_kernel_ms("pair_kernel") / n
                except Exception:
                    pass
                eng.set_profiling(False)
        return acc


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(desc, r, q, reference_binary=False, literal_n=0):
    """The CPU oracle (oracle/msm_spec.c: the same mathematics as the GPU path, C + OpenMP) on this box's cores, and
    the reference's own algorithm on one core in the only regime it can run (N <= a few thousand)."""
    import oracle_lib as O
    import msm_amd
    # the cores this job may use: the box exports its share as OMP_NUM_THREADS (os.cpu_count() is the whole machine)
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    sp = O.SpecPlan.explicit(desc.N, list(desc.A), list(desc.M), desc.p, desc.mu, desc.L, desc.a, desc.beta, desc.kmax,
                             nthreads=cores)
    t0 = time.perf_counter()
    out = sp.eval(r, q)
    dt = time.perf_counter() - t0
    names = ("pairs", "anterpolation", "restriction", "convolution", "top", "prolongation", "interpolation", "total")
    # the same port at the size of the second benchmark configuration (100k particles, cubic spline, three levels), on
    # one core - the reference is single-threaded - and on all of them
    small = {}
    N2 = 100000
    r2, q2 = make_system(N2, 20261111 + 1)
    d2 = msm_amd.plan_auto(N2, (1.0, 1.0, 1.0), 4, 10, 4.0, 3)
    for nt in (1, cores):
        sp2 = O.SpecPlan.explicit(d2.N, list(d2.A), list(d2.M), d2.p, d2.mu, d2.L, d2.a, d2.beta, d2.kmax, nthreads=nt)
        t1 = time.perf_counter()
        o2 = sp2.eval(r2, q2)
        d = time.perf_counter() - t1
        small["cores_%d" % nt] = {"cores": nt, "seconds": d, "evals_per_s": 1.0 / d,
                                  "stage_seconds": dict(zip(names, [float(x) for x in o2["seconds"]]))}
        sp2.close()
    return {"value": 1.0 / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "host": {"cpu_model": cpu_model(), "cpus_of_the_machine": os.cpu_count(), "cpus_used": cores},
            "sample": "1 evaluation of the same %d-particle system (all stages)" % desc.N,
            "note": "an unoptimised port: its anterpolation alone takes more than half of the time",
            "stage_seconds": dict(zip(names, [float(x) for x in out["seconds"]])),
            "utotal": float(out["energies"][6]),
            "port_100k_order4_3levels": small,
            "reference_literal": reference_literal_timings(reference_binary, literal_n)}, out


def reference_literal_timings(reference_binary=False, literal_n=0):
    """The reference's algorithm itself, single-threaded as the reference is (msmPeriodic.c:52, "time total" of
    msmLibrary.c:337), on the systems it can execute: the literal mirror oracle/msm_literal.c (stage for stage the
    reference's arithmetic; its image sums are tabulated, so its stencil set-up is faster than the reference's) at
    N = 8, 64, 512 of msm/data and, with --literal-n, on a synthetic system of that size (4096: the largest the
    algorithm is meaningful for; minutes); with --reference-binary and the reference built into oracle/_ref, the real
    msmPeriodic at N = 8 and 64 with its own clock() split between stencil set-up and everything else."""
    import subprocess
    import tempfile
    import oracle_lib as O
    res = {"cores": 1, "mirror": {}, "reference_binary": "not requested (--reference-binary)"}
    for name in ("changaN8", "changaN64", "changaN512"):
        fx = O.golden("msmPeriodic_%s.json" % name)
        sysd = fx["system"]
        rr, qq = np.array(sysd["r"]).reshape(-1, 3), np.array(sysd["q"])
        t0 = time.perf_counter()
        lit = O.literal_run(rr, qq, 4, 4.0, 10, sysd["L"])
        dt = time.perf_counter() - t0
        res["mirror"][name] = {"N": sysd["N"], "seconds": dt, "evals_per_s": 1.0 / dt, "utotal": float(lit["u_total"]),
                               "reference_recorded_seconds": float(fx["stats"]["time total"])}
    if literal_n > 0:
        rr, qq = make_system(literal_n, 20261111 + 0)
        t0 = time.perf_counter()
        lit = O.literal_run(rr, qq, 4, 4.0, 10, (1.0, 1.0, 1.0))
        dt = time.perf_counter() - t0
        res["mirror"]["synthetic_N%d" % literal_n] = {"N": literal_n, "seconds": dt, "evals_per_s": 1.0 / dt,
                                                      "utotal": float(lit["u_total"]),
                                                      "system": "SplitMix64 seed 20261111, unit cube, charges +1 / -1"}
    exe = os.path.join(ROOT, "oracle", "_ref", "msmPeriodic")
    if reference_binary and os.path.exists(exe):
        res["reference_binary"] = {}
        for name in ("changaN8", "changaN64"):
            fx = O.golden("msmPeriodic_%s.json" % name)
            sysd = fx["system"]
            rr, qq = np.array(sysd["r"]).reshape(-1, 3), np.array(sysd["q"])
            with tempfile.TemporaryDirectory() as tmp:
                path = os.path.join(tmp, name + ".txt")
                with open(path, "w") as f:
                    f.write("%.17g %.17g %.17g\n%d\n" % (sysd["L"][0], sysd["L"][1], sysd["L"][2], sysd["N"]))
                    for i in range(sysd["N"]):
                        f.write("%.17g %.17g %.17g %.17g\n" % (rr[i, 0], rr[i, 1], rr[i, 2], qq[i]))
                try:
                    t0 = time.perf_counter()
                    out = subprocess.run([exe, path, "4", "10", "4"], cwd=tmp, capture_output=True, text=True, timeout=120)
                    dt = time.perf_counter() - t0
                except (subprocess.TimeoutExpired, OSError) as e:
                    res["reference_binary"][name] = {"error": str(e)}
                    continue
            st = {}
            for ln in out.stdout.splitlines():
                if len(ln) > 31 and ln[:30].strip().startswith("time "):
                    st[ln[:30].strip()] = float(ln[31:].strip())
            total, sten = st.get("time total", dt), st.get("time stencill", 0.0) + st.get("time stencilL", 0.0)
            res["reference_binary"][name] = {"N": sysd["N"], "wall_seconds": dt, "time_total": total, "stencil_setup": sten,
                                             "without_stencil_setup": total - sten,
                                             "evals_per_s_with_setup": 1.0 / total,
                                             "evals_per_s_without_setup": 1.0 / max(total - sten, 1e-9)}
    elif reference_binary:
        res["reference_binary"] = "oracle/_ref/msmPeriodic is not built on this machine"
    return res


if __name__ == "__main__":
    main()
