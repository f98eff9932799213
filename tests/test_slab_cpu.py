"""The multi-GPU (z-slab) path without a GPU: partition logic, halo completeness and the two collectives are
exercised with world_size-2/3 gloo process groups.  The compute stages are supplied by a CPU stand-in built on
the oracle (test infrastructure — the product's SlabRank drives libmsm_amd.so); what is under test is
msm_amd/slab.py: that every particle is owned once, that the halo is wide enough for the pair sum and for the
anterpolation gather of the owned planes, and that all-gather + all-reduce reassemble the single-rank result."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O


def test_partition_is_exact():
    from msm_amd import slab
    rng = np.random.default_rng(0)
    r = rng.random((5000, 3)) * 3.0 - 1.0   # also outside the cell
    for world in (1, 2, 4, 8):
        seen = np.zeros(len(r), dtype=int)
        for k in range(world):
            own, hal = slab.select_slab(r, 1.0, 96, world, k, 0.05)
            seen[own] += 1
            assert not set(own) & set(hal)
            z = r[hal, 2] - np.floor(r[hal, 2])
            lo, hi = k / world, (k + 1) / world
            dist = np.minimum(np.mod(lo - z, 1.0), np.mod(z - hi, 1.0))
            assert (dist <= 0.05 + 1e-15).all()
        assert (seen == 1).all()
    with pytest.raises(ValueError):
        slab.slab_planes(96, 7, 0)


class OracleEngine:
    """CPU stand-in with the Engine interface, built from the oracle's stage functions."""

    def __init__(self, d):
        import torch
        self.torch = torch
        self.d = d
        self.M = list(d.M)
        self.z0, self.nz = d.zplane0, d.nzplanes
        self.q0 = torch.zeros((self.M[2], self.M[1], self.M[0]), dtype=torch.float64)
        self.sums = torch.zeros(5, dtype=torch.float64)
        self.sp = None

    def tensor_device(self):
        return "cpu"

    def set_counts(self, n_total, n_owned):
        self.n_total, self.n_owned = n_total, n_owned
        d = self.d
        if self.sp is not None:
            self.sp.close()
        self.sp = O.SpecPlan.explicit(n_total, list(d.A), list(d.M), d.p, d.mu, d.L, d.a, d.beta, d.kmax, nthreads=2)

    def grid_tensor(self, which, level=0):
        return self.q0

    def raw_sums_tensor(self):
        return self.sums

    def stage_bin(self, r, q):
        self.r, self.q = r.numpy().copy(), q.numpy().copy()

    def stage_pairs(self, acc):
        aw, phi = self.sp.pairs_phi(self.r, self.q)
        n = self.n_owned
        acc[:n] = self.torch.as_tensor(aw[:n])
        self.pair_sum = 0.5 * (self.q[:n] * phi[:n]).sum()

    def stage_anterpolate(self):
        g = self.sp.anterpolate(self.r, self.q)   # complete only on the owned planes (halo width)
        self.q0[self.z0:self.z0 + self.nz] = self.torch.as_tensor(g[self.z0:self.z0 + self.nz])

    def stage_grid(self):
        sp, L = self.sp, self.d.L
        qs = [self.q0.numpy().copy()]
        for l in range(L - 1):
            qs.append(sp.restrict(l, qs[l]))
        es = [sp.convolve(l, qs[l]) for l in range(L)]
        self.efour = sp.convolve_top(qs[L - 1])
        es[L - 1] = es[L - 1] + self.efour
        for l in range(L - 2, -1, -1):
            es[l] = sp.prolong_add(l, es[l + 1], es[l])
        self.e0, self.qs = es[0], qs

    def stage_interpolate(self, acc):
        al = self.sp.interpolate(self.r, self.e0)
        n = self.n_owned
        acc[:n] += self.torch.as_tensor(al[:n])

    def stage_raw_sums(self):
        z = slice(self.z0, self.z0 + self.nz)
        n = self.n_owned
        self.sums[:] = self.torch.as_tensor([self.pair_sum, (self.qs[0][z] * self.e0[z]).sum(),
                                             (self.qs[-1] * self.efour).sum(), self.q[:n].sum(),
                                             (self.q[:n] ** 2).sum()])

    def combine_energies(self, s):
        d = self.d
        vol = d.A[0] * d.A[1] * d.A[2]
        g0 = O.lib().or_gamma(0.0, d.p)
        ulf = 0.5 * s[2]
        ulr = 0.5 * s[1] - ulf
        # the self-image term vanishes here (a < cell edge)
        en = [s[0], 0.0, -0.5 * s[3] ** 2 * np.pi / (d.beta ** 2 * vol), ulr, ulf, -0.5 * s[4] * g0 / d.a]
        return np.array(en + [sum(en)])


def _worker(rank, world, port, N, M, L, p, out_dir):
    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd.slab import SlabRank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=42)
    q = q * (1.0 + 0.1 * np.sin(np.arange(N)))   # net charge != 0: exercises the (C) term
    a = 0.16
    desc = msm_amd.plan_explicit(N, (1.0, 1.0, 1.0), (M, M, M), p, 6, L, a, 9.0, 8)
    sr = SlabRank(torch, dist, msm_amd, desc, r, q, rank, world, engine_factory=OracleEngine, perturb=False)
    en = sr.step()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), en=en, idx=sr.own_idx, acc=sr.owned_acc().numpy())
    dist.barrier()
    dist.destroy_process_group()


def moved(r):
    """Displacements large enough to carry many particles across slab boundaries (and some out of the cell)."""
    i = np.arange(len(r))
    return r + np.stack([0.03 * np.cos(0.7 * i), 0.05 * np.sin(1.3 * i), 0.11 * np.sin(0.37 * i)], axis=1)


def _exchange_worker(rank, world, port, N, M, L, p, out_dir):
    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd.slab import SlabRank, select_slab, halo_width
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=42)
    q = q * (1.0 + 0.1 * np.sin(np.arange(N)))
    desc = msm_amd.plan_explicit(N, (1.0, 1.0, 1.0), (M, M, M), p, 6, L, 0.16, 9.0, 8)
    sr = SlabRank(torch, dist, msm_amd, desc, r, q, rank, world, engine_factory=OracleEngine, perturb=False, headroom=1.5)
    sr.step()
    rn = moved(r)
    sr.r[:sr.n_own] = torch.as_tensor(rn[sr.own_idx])     # what an integrator does: new positions of the owned particles
    sr.exchange()
    own, hal = select_slab(rn, 1.0, M, world, rank, halo_width(desc))
    assert set(sr.own_idx) == set(own) and set(sr.halo_idx) == set(hal)
    assert np.array_equal(sr.r.numpy(), rn[np.concatenate([sr.own_idx, sr.halo_idx])])
    assert np.array_equal(sr.q.numpy(), q[np.concatenate([sr.own_idx, sr.halo_idx])])
    en = sr.step()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), en=en, idx=sr.own_idx, acc=sr.owned_acc().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,p,L", [(2, 4, 2), (3, 6, 1)])
def test_exchange_migrates_particles_and_rebuilds_halos(tmp_path, world, p, L):
    """After the particles have moved, SlabRank.exchange() (two all-to-all collectives) leaves every rank with
    exactly the owned and halo particles select_slab gives for the new positions, and the evaluation equals the
    single-rank result on the new positions."""
    import torch.multiprocessing as mp
    N, M = 1500, 12
    mp.spawn(_exchange_worker, args=(world, _free_port(), N, M, L, p, str(tmp_path)), nprocs=world, join=True)
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=42)
    q = q * (1.0 + 0.1 * np.sin(np.arange(N)))
    rn = moved(r)
    sp = O.SpecPlan.explicit(N, (1.0, 1.0, 1.0), (M, M, M), p, 6, L, 0.16, 9.0, 8, nthreads=4)
    ref = sp.eval(rn, q)
    acc = np.zeros((N, 3))
    seen = np.zeros(N, dtype=int)
    for k in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % k))
        acc[z["idx"]] = z["acc"]
        seen[z["idx"]] += 1
        scale = np.abs(ref["energies"][:6]).max()
        assert np.abs(z["en"] - ref["energies"]).max() < 1e-12 * scale, (k, z["en"], ref["energies"])
    assert (seen == 1).all()
    err = np.sqrt(((acc - ref["acc"]) ** 2).sum() / (ref["acc"] ** 2).sum())
    assert err < 1e-12, err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,p,L", [(2, 4, 2), (3, 6, 1)])
def test_slab_reassembles_single_rank_result(tmp_path, world, p, L):
    import torch.multiprocessing as mp
    import msm_amd
    N, M = 1500, 12
    mp.spawn(_worker, args=(world, _free_port(), N, M, L, p, str(tmp_path)), nprocs=world, join=True)
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=42)
    q = q * (1.0 + 0.1 * np.sin(np.arange(N)))
    sp = O.SpecPlan.explicit(N, (1.0, 1.0, 1.0), (M, M, M), p, 6, L, 0.16, 9.0, 8, nthreads=4)
    ref = sp.eval(r, q)
    acc = np.zeros((N, 3))
    seen = np.zeros(N, dtype=int)
    for k in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % k))
        acc[z["idx"]] = z["acc"]
        seen[z["idx"]] += 1
        scale = np.abs(ref["energies"][:6]).max()
        assert np.abs(z["en"] - ref["energies"]).max() < 1e-12 * scale, (k, z["en"], ref["energies"])
    assert (seen == 1).all()
    err = np.sqrt(((acc - ref["acc"]) ** 2).sum() / (ref["acc"] ** 2).sum())
    assert err < 1e-12, err


def test_direct_rows_partition():
    from msm_amd.slab import direct_rows
    for N in (1, 7, 1000, 1000003):
        for world in (1, 2, 3, 8):
            cuts = [direct_rows(world, k, N) for k in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == N
            assert all(cuts[k][1] == cuts[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_open_slab_partition_is_exact_and_balanced():
    """Open-boundary slabs: plane cuts by particle count; every particle owned once; halo copies are exactly the
    particles within the halo distance of the slab (no periodic images)."""
    import msm_amd
    from msm_amd.slab import open_slab_edges, select_open_slab
    rng = np.random.default_rng(9)
    N = 20000
    r = rng.random((N, 3)) * np.array([1.0, 0.7, 1.6]) + np.array([2.0, -1.0, 40.0])
    r[: N // 4, 2] = 40.0 + 0.2 * rng.random(N // 4)          # a dense layer: equal-thickness slabs would be lopsided
    desc = msm_amd.plan_open(N, r.min(axis=0), r.max(axis=0), 4, 10, 1.0 / 24, 4.0 / 24, 2)
    for world in (2, 3, 8):
        e = open_slab_edges(r, desc, world)
        assert e[0] == 0 and e[-1] == desc.M[2] and all(b > a for a, b in zip(e[:-1], e[1:]))
        seen = np.zeros(N, dtype=int)
        counts = []
        halo = 4.0 / 24
        zg = (r[:, 2] - desc.origin[2]) / desc.A[2] * desc.M[2]
        for k in range(world):
            own, hal = select_open_slab(r, desc, e, k, halo)
            seen[own] += 1
            counts.append(len(own))
            assert not np.intersect1d(own, hal).size
            assert ((zg[own] >= e[k]) | (k == 0)).all() and ((zg[own] < e[k + 1]) | (k == world - 1)).all()
            hg = halo / desc.A[2] * desc.M[2]
            want = np.nonzero((seen * 0 == 0) & ~np.isin(np.arange(N), own) & (zg >= e[k] - hg) & (zg < e[k + 1] + hg))[0]
            assert np.array_equal(np.sort(hal), want)
        assert (seen == 1).all()
        assert max(counts) < 1.35 * (N / world), counts        # cuts fall on planes, so not perfectly even


class _FakeRcclApi:
    """Stand-in for the three library calls of rccl_table_or_none that fails where the test says."""

    class Table:
        closed = False

        def close(self):
            self.closed = True

    def __init__(self, rank, fail):
        self.rank, self.fail, self.calls, self.table = rank, fail, [], None

    def _maybe(self, what):
        self.calls.append(what)
        if self.fail == (what, self.rank):
            raise RuntimeError("injected failure of %s on rank %d" % (what, self.rank))

    def available(self):
        self._maybe("available")

    def unique_id(self):
        self._maybe("unique_id")
        return bytes(range(128))

    def create(self, raw, rank, world):
        assert raw == bytes(range(128))
        self._maybe("create")
        self.table = self.Table()
        return self.table


def _agree_worker(rank, world, port, cases, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)
    import json
    import torch
    import torch.distributed as dist
    from msm_amd.slab import rccl_table_or_none
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    for fail in cases:
        api = _FakeRcclApi(rank, tuple(fail) if fail else None)
        t = rccl_table_or_none(torch, dist, rank, world, "cpu", api=api)
        res.append({"got": t is not None, "calls": api.calls, "closed": bool(api.table is not None and api.table.closed)})
        dist.barrier()       # every rank left the negotiation: nobody is stuck in a collective the others skipped
    with open(os.path.join(out_dir, "agree%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


def test_rccl_negotiation_keeps_the_ranks_together(tmp_path):
    """msm_amd.slab.rccl_table_or_none with a failure injected on rank 0 or on rank 1 at each of its three library calls:
    both ranks come out with the same answer (no table), having run the same collectives, and a communicator that was
    made on one rank only is destroyed again."""
    import json
    import torch.multiprocessing as mp
    cases = [None, ("available", 0), ("available", 1), ("unique_id", 0), ("create", 0), ("create", 1)]
    world = 2
    mp.spawn(_agree_worker, args=(world, _free_port(), cases, str(tmp_path)), nprocs=world, join=True)
    res = [json.load(open(os.path.join(str(tmp_path), "agree%d.json" % k))) for k in range(world)]
    for i, fail in enumerate(cases):
        got = [res[k][i]["got"] for k in range(world)]
        assert got == [fail is None] * world, (fail, got)
        if fail and fail[0] == "available":     # nobody went on to the library's collective calls
            assert all("create" not in res[k][i]["calls"] and "unique_id" not in res[k][i]["calls"] for k in range(world))
        if fail and fail[0] == "unique_id":
            assert all("create" not in res[k][i]["calls"] for k in range(world))
        if fail and fail[0] == "create":        # the rank whose communicator came up gave it back
            other = 1 - fail[1]
            assert res[other][i]["closed"] and not res[fail[1]][i]["closed"]


class _PlanesOnlyEngine:
    """Stand-in with just what SlabRank's constructor and exchange() touch (no stages): the partition logic of an
    open-boundary plan runs without a GPU."""

    def __init__(self, d):
        import torch
        self.torch, self.d = torch, d
        self.planes = (d.zplane0, d.nzplanes)
        self.q0 = torch.zeros((d.M[2], 2, 2), dtype=torch.float64)

    def tensor_device(self):
        return "cpu"

    def set_counts(self, n_total, n_owned):
        self.counts = (n_total, n_owned)

    def set_planes(self, z0, nz):
        self.planes = (z0, nz)

    def grid_tensor(self, which, level=0):
        return self.q0

    def raw_sums_tensor(self):
        return self.torch.zeros(5, dtype=self.torch.float64)


def _open_partition_worker(rank, world, port, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)
    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd.slab import SlabRank, PlanOutgrown
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, q, desc, rn = _open_partition_case(msm_amd)
    sr = SlabRank(torch, dist, msm_amd, desc, r, q, rank, world, engine_factory=_PlanesOnlyEngine, headroom=1.5)
    sr._plan_holds = lambda lo, hi: bool(np.all(lo > np.array(desc.origin)) and np.all(hi < np.array(desc.origin) + np.array(desc.A)))
    sr.r[:sr.n_own] = torch.as_tensor(rn[sr.own_idx])
    sr.exchange()
    refused = False
    sr.r[:sr.n_own, 2] += 100.0
    try:
        sr.exchange()
    except PlanOutgrown:
        refused = True
    np.savez(os.path.join(out_dir, "op%d.npz" % rank), own=sr.own_idx, halo=sr.halo_idx, edges=np.array(sr.edges), planes=np.array(sr.eng.planes),
             counts=np.array(sr.eng.counts), refused=refused)
    dist.barrier()
    dist.destroy_process_group()


def _open_partition_case(msm_amd):
    rng = np.random.default_rng(9)
    N = 6000
    r = rng.random((N, 3)) * np.array([1.0, 0.8, 1.3]) + np.array([0.3, -0.2, 5.0])
    r[: N // 2, 2] = 5.0 + 0.3 * rng.random(N // 2)          # half of the particles in the lowest quarter: uneven slabs
    q = np.where(np.arange(N) % 2 == 0, 1.0, -1.0)
    desc = msm_amd.capi.PlanDesc()
    desc.N, desc.p, desc.mu, desc.L, desc.open_boundary = N, 6, 10, 3, 1
    for k, (m, a0, o) in enumerate(zip((64, 56, 72), (2.0, 1.75, 2.25), (-0.2, -0.7, 4.5))):
        desc.M[k], desc.A[k], desc.origin[k] = m, a0, o
    desc.a, desc.zplane0, desc.nzplanes = 0.125, 0, 72
    i = np.arange(N)
    rn = r + np.stack([0.02 * np.cos(0.7 * i), 0.03 * np.sin(1.3 * i), 0.05 + 0.2 * np.sin(0.37 * i)], axis=1)
    return r, q, desc, rn


def test_open_boundary_exchange_partition(tmp_path):
    """SlabRank.exchange() of an open-boundary plan over three gloo ranks (no GPU: a stand-in engine that records the
    planes and counts it is given): after the particles moved, the slabs are cut again from the all-reduced plane counts,
    every particle is owned exactly once, owned and halo sets are what select_open_slab gives for the new positions and
    the new cut, the engine is told its new planes, and a cloud that leaves the cell is refused on every rank."""
    import torch.multiprocessing as mp
    import msm_amd
    from msm_amd import slab
    world = 3
    mp.spawn(_open_partition_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r, q, desc, rn = _open_partition_case(msm_amd)
    edges = slab.open_slab_edges(rn, desc, world)
    assert edges != slab.open_slab_edges(r, desc, world)
    seen = np.zeros(len(q), dtype=int)
    for k in range(world):
        z = np.load(os.path.join(str(tmp_path), "op%d.npz" % k))
        assert list(z["edges"]) == edges and bool(z["refused"])
        own, hal = slab.select_open_slab(rn, desc, edges, k, slab.halo_width(desc))
        assert set(z["own"]) == set(own) and set(z["halo"]) == set(hal)
        assert tuple(z["planes"]) == (edges[k], edges[k + 1] - edges[k])
        assert tuple(z["counts"]) == (len(own) + len(hal), len(own))
        seen[z["own"]] += 1
    assert (seen == 1).all()
