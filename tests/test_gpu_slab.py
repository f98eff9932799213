"""The slab path with the real engine: two and four ranks share cuda:0 (collectives staged through a gloo
group, since RCCL refuses two ranks on one device) and must reproduce the single-GPU evaluation."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _plan(msm_amd, spec):
    """spec = ("auto", N, p, L) for the reference's parameter choices, or ("explicit", N, p, L, M, a): a cubic grid of M
    nodes per axis and cutoff a, beta and kmax from the reference's own rules (msmPeriodic.c:249-263)."""
    import math
    A = (1.0, 1.0, 1.0)
    if spec[0] == "auto":
        _, N, p, L = spec
        return N, msm_amd.plan_auto(N, A, p, 10, 4.0, L)
    _, N, p, L, M, a = spec      # ("explicit" | "blob", ...)
    lib = msm_amd.load()
    aL, hL = 2.0 ** L * a, 2.0 ** (L - 1) / M
    beta = lib.choosebetaNew(1e-12, aL, hL)
    kmax = int(math.ceil(abs(lib.choosekmax(1e-12, beta, hL))))
    return N, msm_amd.plan_explicit(N, A, (M, M, M), p, 10, L, a, beta, kmax)


def _system(N, spec):
    """Uniform random charges; for a "blob" plan three quarters of them sit in a ball of radius ~0.02 on the face between
    the slabs of ranks 0 and 1 of two: its clusters, owned, halo copies and mixed, overflow their neighbour lists and go
    to the list-free kernel, which has to book the pair energies as the list-walking kernel does for the others."""
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=43)
    if spec[0] == "blob":
        rng = np.random.default_rng(11)
        n = 3 * N // 4
        r[:n] = (0.5 + 0.01 * rng.standard_normal((n, 3))) % 1.0
    return r, q


def _worker(rank, world, port, spec, mode, c_step, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)   # a stuck rank ends with its stacks instead of hanging the run
    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd.slab import SlabRank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, desc = _plan(msm_amd, spec)
    r, q = _system(N, spec)
    sr = SlabRank(torch, dist, msm_amd, desc, r, q, rank, world, local_rank=0, perturb=False, mode=mode, c_step=c_step)
    used = sr.eng.slab_exchange_mode(world, mode) if c_step else 0
    for _ in range(2):   # twice: the second step must not see state left by the first
        en = sr.step()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), en=en, idx=sr.own_idx, acc=sr.owned_acc().cpu().numpy(), used=used,
             n_tot=sr.n_tot)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SLAB_CASES = [
    # world, plan, exchange mode asked for (0 auto, 1 all-gather, 2 halo), C step, mode expected
    (2, ("auto", 40000, 4, 2), 0, True, 1),
    (4, ("auto", 40000, 6, 3), 0, True, 1),                    # 8 planes per rank, 10 halo planes: the all-gather it is
    (4, ("auto", 40000, 6, 3), 0, False, 0),                   # the stage-by-stage Python step (what a stand-in engine gets)
    (4, ("explicit", 50000, 4, 3, 64, 0.03), 0, True, 2),      # 16 planes per rank, 5 halo planes: halo exchange
    (4, ("explicit", 50000, 4, 3, 64, 0.03), 1, True, 1),      # the same plan through the all-gather
    (2, ("explicit", 50000, 6, 2, 64, 0.03), 2, True, 2),      # two ranks: both neighbours are the same peer
    # the benchmark's grid split eight ways: 12 planes per rank, cutoff above half a slab (halo particles outnumber the
    # owned ones), 16 halo planes would not fit the neighbour's slab: all-gather
    (8, ("explicit", 60000, 6, 4, 96, 0.07), 0, True, 1),
    # a grid that is no multiple of the 8-cell tiles of the spline kernels, three ranks: the middle one visits some layers
    # of tiles only, the outer ones (their halo wraps around the odd last layer) all of them
    (3, ("explicit", 50000, 6, 3, 60, 0.05), 0, True, 2),
    # a crowded ball on the face between two slabs: list overflow on both ranks (the list-free kernel beside the list-walking one)
    (2, ("blob", 16000, 4, 2, 32, 0.1), 0, True, 1),
]


@pytest.mark.parametrize("world,spec,mode,c_step,expect", SLAB_CASES)
def test_slab_ranks_match_single_gpu(tmp_path, world, spec, mode, c_step, expect):
    import torch
    import torch.multiprocessing as mp
    import msm_amd
    mp.spawn(_worker, args=(world, _free_port(), spec, mode, c_step, str(tmp_path)), nprocs=world, join=True)
    N, desc = _plan(msm_amd, spec)
    r, q = _system(N, spec)
    eng = msm_amd.Engine(desc)
    en1, acc1 = eng.eval(r, q)
    acc1 = acc1.cpu().numpy()
    acc = np.zeros((N, 3))
    seen = np.zeros(N, dtype=int)
    scale = np.abs(en1[:6]).max()
    for k in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % k))
        acc[z["idx"]] = z["acc"]
        seen[z["idx"]] += 1
        assert int(z["used"]) == expect
        assert np.abs(z["en"] - en1).max() < 1e-12 * scale, (k, z["en"], en1)
        if world == 8:
            assert int(z["n_tot"]) > 2 * len(z["idx"])          # more halo copies than owned particles
    assert (seen == 1).all()
    err = np.sqrt(((acc - acc1) ** 2).sum() / (acc1 ** 2).sum())
    assert err < 1e-12, err


def moved(r):
    i = np.arange(len(r))
    return r + np.stack([0.03 * np.cos(0.7 * i), 0.05 * np.sin(1.3 * i), 0.11 * np.sin(0.37 * i)], axis=1)


def _exchange_worker(rank, world, port, N, p, L, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)
    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd.slab import SlabRank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=43)
    desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), p, 10, 4.0, L)
    sr = SlabRank(torch, dist, msm_amd, desc, r, q, rank, world, local_rank=0, perturb=False, headroom=1.3)
    sr.step()
    rn = moved(r)
    sr.r[:sr.n_own] = torch.as_tensor(rn[sr.own_idx], device=sr.r.device)
    sr.exchange()
    en = sr.step()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), en=en, idx=sr.own_idx, acc=sr.owned_acc().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_slab_exchange_on_gpu(tmp_path):
    """Particles move across slab boundaries; after SlabRank.exchange() four ranks (sharing cuda:0) reproduce the
    single-GPU evaluation of the new positions."""
    import torch.multiprocessing as mp
    import msm_amd
    world, N, p, L = 4, 40000, 6, 3
    mp.spawn(_exchange_worker, args=(world, _free_port(), N, p, L, str(tmp_path)), nprocs=world, join=True)
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=43)
    rn = moved(r)
    desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), p, 10, 4.0, L)
    eng = msm_amd.Engine(desc)
    en1, acc1 = eng.eval(rn, q)
    acc1 = acc1.cpu().numpy()
    acc = np.zeros((N, 3))
    seen = np.zeros(N, dtype=int)
    scale = np.abs(en1[:6]).max()
    for k in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % k))
        acc[z["idx"]] = z["acc"]
        seen[z["idx"]] += 1
        assert np.abs(z["en"] - en1).max() < 1e-12 * scale, (k, z["en"], en1)
    assert (seen == 1).all()
    assert np.sqrt(((acc - acc1) ** 2).sum() / (acc1 ** 2).sum()) < 1e-12


def _direct_worker(rank, world, port, N, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)
    import torch
    import torch.distributed as dist
    from msm_amd.slab import direct_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=47)
    u, acc, (i0, i1) = direct_sharded(torch, dist, r, q, rank, world)
    np.savez(os.path.join(out_dir, "direct%d.npz" % rank), u=u, i0=i0, i1=i1, acc=acc.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_direct_sum_rows_over_ranks(tmp_path):
    """The non-periodic direct sum with the rows of the pair matrix split over 3 ranks (sharing cuda:0): the slice
    energies add up to the one-GPU energy and the rows are the one-GPU rows, bit for bit (same kernel, same order)."""
    import torch.multiprocessing as mp
    import msm_amd
    N, world = 5001, 3
    mp.spawn(_direct_worker, args=(world, _free_port(), N, str(tmp_path)), nprocs=world, join=True)
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=47)
    u1, acc1 = msm_amd.direct(r, q)
    uo, ao = O.direct(r, q, nthreads=8)
    assert abs(u1 - uo) < 1e-12 * abs(uo)
    rows = 0
    for k in range(world):
        z = np.load(os.path.join(str(tmp_path), "direct%d.npz" % k))
        assert abs(float(z["u"]) - u1) < 1e-12 * abs(u1)
        assert np.array_equal(z["acc"], acc1[int(z["i0"]):int(z["i1"])])
        rows += int(z["i1"]) - int(z["i0"])
    assert rows == N


def test_rccl_collectives_single_rank():
    """The library's RCCL table (msm_amd_rccl_create) with a communicator of one rank: every entry runs on a stream and
    leaves the data a one-rank collective must leave (more ranks need more GPUs than this box has)."""
    import ctypes as C
    import torch
    import msm_amd
    from msm_amd import capi
    lib = msm_amd.load()
    idbuf = C.create_string_buffer(128)
    capi.check(lib.msm_amd_rccl_unique_id(idbuf), "unique_id")
    p = C.POINTER(capi.Collectives)()
    torch.cuda.set_device(0)
    capi.check(lib.msm_amd_rccl_create(idbuf.raw, 0, 1, C.byref(p)), "rccl_create")
    c = p.contents
    assert c.rank == 0 and c.world == 1
    st = torch.cuda.Stream()
    x = torch.arange(1000, dtype=torch.float64, device="cuda:0") + 0.5
    y = torch.zeros(1000, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    assert c.all_gather(c.ctx, x.data_ptr(), y.data_ptr(), 1000, st.cuda_stream) == 0
    assert c.all_reduce_sum(c.ctx, x.data_ptr(), 1000, st.cuda_stream) == 0
    up, down = torch.zeros(64, dtype=torch.float64, device="cuda:0"), torch.zeros(64, dtype=torch.float64, device="cuda:0")
    # alone in the ring a rank is its own neighbour: what it sends up it receives from below
    assert c.halo_exchange(c.ctx, x[:64].data_ptr(), x[100:164].data_ptr(), down.data_ptr(), up.data_ptr(), 64, 0, 0, st.cuda_stream) == 0
    st.synchronize()
    ref = torch.arange(1000, dtype=torch.float64) + 0.5
    assert torch.equal(y.cpu(), ref) and torch.equal(x.cpu(), ref)
    assert torch.equal(down.cpu(), ref[:64]) and torch.equal(up.cpu(), ref[100:164])
    # and a whole step through it: a handle that owns every plane
    N = 20000
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=5)
    desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), 6, 10, 4.0, 2)
    eng = msm_amd.Engine(desc)
    rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
    en1, acc1 = eng.eval(rd, qd)
    acc1 = acc1.clone()
    acc2 = torch.zeros_like(acc1)
    eng.slab_step(p, rd, qd, acc2)
    en2 = eng.slab_energies(p)
    assert np.abs(en2 - en1).max() <= 1e-12 * np.abs(en1[:6]).max()
    assert torch.equal(acc1, acc2)          # pair part + grid part: the same two numbers added either way
    eng.close()
    lib.msm_amd_rccl_destroy(p)


def test_torch_collectives_single_rank():
    """The fallback table over a torch.distributed "nccl" group of one rank: a whole step through it equals the plain
    evaluation (more ranks need more GPUs than this box has)."""
    import socket
    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd.slab import TorchCollectives
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        coll = TorchCollectives(torch, dist, 0, 1, torch.device("cuda:0"))
        c = coll.ptr.contents
        st = torch.cuda.Stream()
        x = torch.arange(1000, dtype=torch.float64, device="cuda:0") + 0.5
        y = torch.zeros(1000, dtype=torch.float64, device="cuda:0")
        up, down = torch.zeros(64, dtype=torch.float64, device="cuda:0"), torch.zeros(64, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        assert c.all_gather(c.ctx, x.data_ptr(), y.data_ptr(), 1000, st.cuda_stream) == 0
        assert c.all_reduce_sum(c.ctx, x.data_ptr(), 1000, st.cuda_stream) == 0
        assert c.halo_exchange(c.ctx, x[:64].data_ptr(), x[100:164].data_ptr(), down.data_ptr(), up.data_ptr(), 64, 0, 0, st.cuda_stream) == 0
        st.synchronize()
        torch.cuda.synchronize()
        ref = torch.arange(1000, dtype=torch.float64) + 0.5
        assert torch.equal(y.cpu(), ref) and torch.equal(x.cpu(), ref)
        assert torch.equal(down.cpu(), ref[:64]) and torch.equal(up.cpu(), ref[100:164])
        N = 20000
        r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=5)
        eng = msm_amd.Engine(msm_amd.plan_auto(N, (1.0, 1.0, 1.0), 6, 10, 4.0, 2))
        rd, qd = torch.as_tensor(r, device="cuda:0"), torch.as_tensor(q, device="cuda:0")
        en1, acc1 = eng.eval(rd, qd)
        acc1 = acc1.clone()
        acc2 = torch.zeros_like(acc1)
        eng.slab_step(coll.ptr, rd, qd, acc2)
        en2 = eng.slab_energies(coll.ptr)
        assert np.abs(en2 - en1).max() <= 1e-12 * np.abs(en1[:6]).max()
        assert torch.equal(acc1, acc2)
        eng.close()
    finally:
        dist.destroy_process_group()


def _open_worker(rank, world, port, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)
    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd.slab import SlabRank
    from test_gpu_parity import open_system
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, p, mu, h, a, L = OPEN_SLAB
    r, q = open_system(N, 5, (1.0, 0.8, 1.3), (0.3, -0.2, 5.0))
    desc = msm_amd.plan_open(N, r.min(axis=0), r.max(axis=0), p, mu, h, a, L)
    sr = SlabRank(torch, dist, msm_amd, desc, r, q, rank, world, local_rank=0)
    for _ in range(2):
        en = sr.step()
    np.savez(os.path.join(out_dir, "open%d.npz" % rank), en=en, idx=sr.own_idx, acc=sr.owned_acc().cpu().numpy(),
             z0=sr.z0, nz=sr.nz, used=sr.eng.slab_exchange_mode(world))
    dist.barrier()
    dist.destroy_process_group()


OPEN_SLAB = (30000, 6, 10, 1.0 / 32, 0.125, 3)


def test_open_boundary_slabs_match_single_gpu(tmp_path):
    """The non-periodic MSM (msm3d, msmLibrary.h:216) over four ranks sharing cuda:0: slabs cut by particle count (the
    padding planes of the cell hold none), level-0 charges summed over the ranks, everything else as in the periodic
    slab step; energies and accelerations equal the one-GPU evaluation of the same plan."""
    import torch.multiprocessing as mp
    import msm_amd
    from test_gpu_parity import open_system
    world = 4
    mp.spawn(_open_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    N, p, mu, h, a, L = OPEN_SLAB
    r, q = open_system(N, 5, (1.0, 0.8, 1.3), (0.3, -0.2, 5.0))
    desc = msm_amd.plan_open(N, r.min(axis=0), r.max(axis=0), p, mu, h, a, L)
    eng = msm_amd.Engine(desc)
    en1, acc1 = eng.eval(r, q)
    acc1 = acc1.cpu().numpy()
    acc = np.zeros((N, 3))
    seen = np.zeros(N, dtype=int)
    scale = np.abs(en1[:6]).max()
    planes, counts = 0, []
    for k in range(world):
        z = np.load(os.path.join(str(tmp_path), "open%d.npz" % k))
        acc[z["idx"]] = z["acc"]
        seen[z["idx"]] += 1
        assert int(z["used"]) == 3 and int(z["z0"]) == planes
        planes += int(z["nz"])
        counts.append(len(z["idx"]))
        assert np.abs(z["en"] - en1).max() < 1e-12 * scale, (k, z["en"], en1)
    assert planes == desc.M[2] and (seen == 1).all()
    assert max(counts) < 1.2 * min(counts), counts            # balanced by particle count, not by planes
    assert np.sqrt(((acc - acc1) ** 2).sum() / (acc1 ** 2).sum()) < 1e-12


class _NoDist:
    """torch.distributed stand-in for a rank that runs alone (the collectives are a table of no-ops)."""

    def get_backend(self):
        return "nccl"


def _one_rank_steps(rank, world, options, steps=3, kind="halo", own_stream=False):
    """`steps` evaluations of one rank of `world` with collectives that move nothing: the planes of the other ranks keep
    what they held, which depends only on the rank's own history, so two runs of the same history give the same bits
    whichever stream or launch shape the engine's options pick."""
    import torch
    import msm_amd
    from msm_amd.slab import SlabRank, NullCollectives
    if kind == "open":      # open boundaries: slabs cut by particle count, level-0 charges summed over the ranks
        from test_gpu_parity import open_system
        N, p, mu, hh, a, L = OPEN_SLAB
        r, q = open_system(N, 5, (1.0, 0.8, 1.3), (0.3, -0.2, 5.0))
        desc = msm_amd.plan_open(N, r.min(axis=0), r.max(axis=0), p, mu, hh, a, L)
    else:                   # a = 0.07 on the 96^3 grid: 16 halo planes do not fit a slab of 12, the all-gather it is
        N, desc = _plan(msm_amd, ("explicit", 100000, 6, 4, 96, {"halo": 0.04, "wide": 5.0 / 96, "allgather": 0.07}[kind]))
        r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=47)
    # own_stream: the rank on a stream of the caller's (the engine created on it, the stream current around every call)
    # instead of the legacy default stream
    side = torch.cuda.Stream() if own_stream else None
    with torch.cuda.stream(side):
        s = SlabRank(torch, _NoDist(), msm_amd, desc, r, q, rank, world, 0, perturb=False, collectives=NullCollectives(rank, world),
                     stream=side)
        assert s.eng.slab_exchange_mode(world, 0) == {"halo": 2, "wide": 2, "allgather": 1, "open": 3}[kind]
        for k, v in options.items():
            s.eng.set_option(k, v)
        for _ in range(steps):
            s.step(want_energies=False)
    torch.cuda.synchronize()
    out = s.acc[:s.n_own].clone(), s.sums.clone()
    s.eng.close()      # (and with it the handle's two streams: the cases of a test do not pile them up)
    return out


@pytest.mark.parametrize("rank", [0, 2])
def test_slab_step_options_keep_the_bits(rank):
    """The slab step with the exchange alone on the second stream (slab_overlap 1), with the coarse-level chain behind it
    (2), the whole grid stage (3), the anterpolation too (4), and on one stream (0); the slab convolution in tiles of 6 to 12 planes; the pair kernel in workgroups
    of 1, 2 and 4 waves: the same kernels on the same data, so accelerations and raw sums agree bit for bit.  Rank 0 owns
    planes whose halo wraps around the period (one plane range for the convolution), rank 2 does not."""
    import torch
    world = 4
    ref_acc, ref_sums = _one_rank_steps(rank, world, {})
    assert torch.isfinite(ref_acc).all() and float(ref_acc.abs().max()) > 0
    cases = [{"slab_overlap": 0}, {"slab_overlap": 1}, {"slab_overlap": 2}, {"slab_overlap": 3}, {"slab_overlap": 4},
             {"slab_overlap": 5},
             {"conv_tile_planes": 6}, {"conv_tile_planes": 8}, {"conv_tile_planes": 9},
             {"conv_tile_planes": 10}, {"conv_tile_planes": 12, "slab_overlap": 1}, {"pair_wpb": 2}, {"pair_wpb": 4},
             {"list_wpb": 1}, {"pair_lds_pad": 17408}, {"acc_slots": 0}, {"energy_fused": 1},
             {"conv_leads": 0}, {"pair_chunks": 1}, {"pair_chunks": 3}, {"energy_on_b": 0}, {"tail_events": 0},
             {"slab_overlap": 2, "pair_chunks_coarse": 4}, {"slab_overlap": 2, "pair_chunks_coarse": 1}]
    # (face_cut 0 changes the clusters, hence the order of the sums: compared to rounding below)
    for opts in cases:
        acc, sums = _one_rank_steps(rank, world, opts)
        assert torch.equal(acc, ref_acc), opts
        assert torch.equal(sums, ref_sums), opts
    # the rank on a stream of the caller's instead of the default stream (the step then stays on the stream it is given)
    for opts in ({}, {"slab_overlap": 2}, {"slab_overlap": 5}):
        acc, sums = _one_rank_steps(rank, world, opts, own_stream=True)
        assert torch.equal(acc, ref_acc), opts
        assert torch.equal(sums, ref_sums), opts
    # clusters cut at the faces of the slab (default) and every 8 particles whatever they are: other clusters, other
    # order of the sums, the same result to rounding
    acc, sums = _one_rank_steps(rank, world, {"face_cut": 0})
    assert float((acc - ref_acc).abs().max()) <= 1e-12 * float(ref_acc.abs().max())
    assert float(((sums - ref_sums).abs() / ref_sums.abs().clamp_min(1e-300)).max()) <= 1e-11
    # the taps of a level-0 tile shared out over 3 and over 5 workgroups (the most the partial-sum buffer holds): the
    # partial sums are added in another order; more shares than that are refused
    for split in (3, 5):
        acc, sums = _one_rank_steps(rank, world, {"conv_split": split})
        assert float((acc - ref_acc).abs().max()) <= 1e-12 * float(ref_acc.abs().max()), split
        assert float(((sums - ref_sums).abs() / ref_sums.abs().clamp_min(1e-300)).max()) <= 1e-11, split
    import msm_amd
    with pytest.raises(msm_amd.MsmAmdError):
        _one_rank_steps(rank, world, {"conv_split": 6})


def test_pair_trace_hook():
    """msm_amd_set_pair_trace: every wave of the pair kernel leaves {start, end, hardware id, kind}; the evaluation's
    results do not change, and the hook is off again after NULL."""
    import ctypes as C
    import torch
    import msm_amd
    N = 60000
    desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), 6, 10, 4.0, 3)
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=5)
    eng = msm_amd.Engine(desc)
    en0, acc0 = eng.eval(r, q)
    lib = msm_amd.load()
    slots = lib.msm_amd_pair_trace_slots(eng._h)
    assert slots >= N // 8
    buf = torch.zeros((slots, 4), dtype=torch.int64, device="cuda")
    assert lib.msm_amd_set_pair_trace(eng._h, C.c_void_p(buf.data_ptr())) == 0
    en1, acc1 = eng.eval(r, q)
    assert lib.msm_amd_set_pair_trace(eng._h, None) == 0
    t = buf.cpu().numpy()
    kinds = t[:, 3]
    assert set(np.unique(kinds)) <= {0, 1, 3}                     # every particle is owned: no wave of halo copies only
    done = kinds == 3
    assert N / 8 <= done.sum() <= slots
    assert (t[done, 1] >= t[done, 0]).all() and (t[done, 0] > 0).all()
    assert len(np.unique((t[done, 2] >> 32) & 0xF)) == 8        # waves on all eight XCDs
    assert np.array_equal(en0, en1) and torch.equal(acc0, acc1)
    buf.zero_()
    eng.eval(r, q)
    torch.cuda.synchronize()
    assert int(buf.abs().sum()) == 0


@pytest.mark.parametrize("kind,world,rank", [("allgather", 8, 0), ("open", 4, 1)])
def test_slab_step_streams_keep_the_bits_other_exchanges(kind, world, rank):
    """The same for the two other exchanges of the level-0 charges: the all-gather of the whole grid (8 ranks of the
    benchmark's grid with a cutoff whose halo does not fit a slab) and the sum over the ranks of an open-boundary plan."""
    import torch
    ref_acc, ref_sums = _one_rank_steps(rank, world, {}, kind=kind)
    assert torch.isfinite(ref_acc).all() and float(ref_acc.abs().max()) > 0
    for opts in ({"slab_overlap": 0}, {"slab_overlap": 1}, {"slab_overlap": 2}, {"slab_overlap": 3}, {"slab_overlap": 4}, {"slab_overlap": 5},
                 {"conv_tile_planes": 8}, {"pair_wpb": 4}):
        acc, sums = _one_rank_steps(rank, world, opts, kind=kind)
        assert torch.equal(acc, ref_acc), (kind, opts)
        assert torch.equal(sums, ref_sums), (kind, opts)


def test_slab_convolution_of_19_taps_in_tall_tiles():
    """Stencils of 19 taps (a = 5 h) on a slab: the split kernel in tiles of 8 planes (what a grid of this size gets) and
    the whole-grid kernel with 16 to 19 outputs per thread in tiles of 32 to 38 planes (what a thick slab of a large grid
    gets; forced here) add their taps in different orders: equal to rounding."""
    import torch
    ref_acc, ref_sums = _one_rank_steps(1, 4, {"conv_tile_planes": 8}, kind="wide")
    scale = float(ref_acc.abs().max())
    assert scale > 0
    for planes in (0, 16, 17, 18, 19):
        acc, sums = _one_rank_steps(1, 4, {"conv_tile_planes": planes}, kind="wide")
        assert float((acc - ref_acc).abs().max()) <= 1e-12 * scale, planes
        assert float(((sums - ref_sums).abs() / ref_sums.abs().clamp_min(1e-300)).max()) <= 1e-11, planes
        if planes == 0:
            assert torch.equal(acc, ref_acc)      # a grid of 96 x 96 nodes has too few tiles for the tall ones


def test_bench_gpus_flag_starts_that_many_ranks():
    """python bench.py --gpus 2 with no launcher around it: the parent starts two ranks (here sharing the one GPU, with the
    collectives staged through the host: a rehearsal of the launch path, labelled as such, not a measurement) and
    forwards rank 0's line, which says two ranks ran, what each of them held, and carries the per-rank roofline."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MSM_AMD_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--particles", "40000", "--steps", "2",
                        "--warmup", "1", "--spinup-steps", "2", "--no-cpu-baseline", "--no-accuracy"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    mg = out["multi_gpu"]
    assert len(mg["per_rank"]) == 2 and sum(x["n_owned"] for x in mg["per_rank"]) == 40000
    assert all(x["n_total"] > x["n_owned"] for x in mg["per_rank"])
    assert "rehearsal" in out["config"] and out["roofline"]["frac"] > 0
    assert np.isfinite(list(out["energies_last_timed_step"].values())).all()


def test_rank_of_eight_at_ten_million_tall_tiles():
    """One rank of 8 of the 10M-particle system (grid 256^3, 32 owned planes, 19-tap stencils, halo exchange) with
    collectives that move nothing: the level-0 convolution of its 38 planes by the whole-grid kernel in one layer of tall
    tiles (what the step chooses at this size) and by the slab kernel in tiles of 8 planes add the taps in different
    orders - equal to rounding - and each is bitwise repeatable."""
    import torch
    import msm_amd
    from msm_amd.slab import SlabRank, NullCollectives
    N, world, rank = 10000000, 8, 3
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=20261111 + 5)
    desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), 6, 10, 4.0, 0)
    assert list(desc.M) == [256, 256, 256]
    res = {}
    for planes in (0, 8, 0):
        s = SlabRank(torch, _NoDist(), msm_amd, desc, r, q, rank, world, 0, perturb=False, collectives=NullCollectives(rank, world))
        assert s.eng.slab_exchange_mode(world, 0) == 2
        s.eng.set_option("conv_tile_planes", planes)
        for _ in range(2):
            s.step(want_energies=False)
        torch.cuda.synchronize()
        acc, sums = s.acc[:s.n_own].clone(), s.sums.clone()
        s.eng.close()
        assert bool(torch.isfinite(acc).all()) and 1.2e6 < s.n_own < 1.3e6
        if planes in res:
            assert torch.equal(acc, res[planes][0]) and torch.equal(sums, res[planes][1])
        res[planes] = (acc, sums)
    scale = float(res[0][0].abs().max())
    assert float((res[0][0] - res[8][0]).abs().max()) <= 1e-12 * scale
    assert float(((res[0][1] - res[8][1]).abs() / res[0][1].abs().clamp_min(1e-300)).max()) <= 1e-11


OPEN_PAD = 0.2


def _moved_open(r):
    """Displacements that carry particles across the slab cuts and shift the cloud as a whole, inside the plan's padding."""
    i = np.arange(len(r))
    return r + np.stack([0.02 * np.cos(0.7 * i), 0.03 * np.sin(1.3 * i), 0.04 + 0.1 * np.sin(0.37 * i)], axis=1)


def _open_exchange_worker(rank, world, port, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)
    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd.slab import SlabRank, PlanOutgrown
    from test_gpu_parity import open_system
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, p, mu, h, a, L = OPEN_SLAB
    r, q = open_system(N, 5, (1.0, 0.8, 1.3), (0.3, -0.2, 5.0))
    desc = msm_amd.plan_open(N, r.min(axis=0) - OPEN_PAD, r.max(axis=0) + OPEN_PAD, p, mu, h, a, L)   # room for the particles to move
    sr = SlabRank(torch, dist, msm_amd, desc, r, q, rank, world, local_rank=0, headroom=1.4)
    sr.step()
    before = (sr.z0, sr.nz)
    rn = _moved_open(r)
    sr.r[:sr.n_own] = torch.as_tensor(rn[sr.own_idx], device=sr.r.device)
    sr.exchange()
    en = sr.step()
    # and a move that leaves the padded cell is refused on every rank alike
    far = sr.r.clone()
    far[:sr.n_own, 2] += 3.0 * (sr.gid[:sr.n_own] % 2).to(torch.float64)
    keep = sr.r
    sr.r = far
    refused = False
    try:
        sr.exchange()
    except PlanOutgrown:
        refused = True
    sr.r = keep
    np.savez(os.path.join(out_dir, "oex%d.npz" % rank), en=en, idx=sr.own_idx, acc=sr.owned_acc().cpu().numpy(), z0=sr.z0, nz=sr.nz,
             z0_before=before[0], nz_before=before[1], refused=refused)
    dist.barrier()
    dist.destroy_process_group()


def test_open_boundary_exchange_on_gpu(tmp_path):
    """SlabRank.exchange() of an open-boundary plan (msm3d, msm/msmLibrary.h:216): the particles move inside the plan's
    padded cell, the slabs are cut again where they now are (the owned planes of the ranks change), particles migrate
    and the halos are rebuilt; four ranks sharing cuda:0 then reproduce the one-GPU evaluation of the new positions with
    the same plan.  Particles that leave the cell make every rank raise PlanOutgrown."""
    import torch.multiprocessing as mp
    import msm_amd
    from test_gpu_parity import open_system
    world = 4
    mp.spawn(_open_exchange_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    N, p, mu, h, a, L = OPEN_SLAB
    r, q = open_system(N, 5, (1.0, 0.8, 1.3), (0.3, -0.2, 5.0))
    desc = msm_amd.plan_open(N, r.min(axis=0) - OPEN_PAD, r.max(axis=0) + OPEN_PAD, p, mu, h, a, L)
    rn = _moved_open(r)
    assert msm_amd.plan_open_holds(desc, rn.min(axis=0), rn.max(axis=0))
    eng = msm_amd.Engine(desc)
    en1, acc1 = eng.eval(rn, q)
    acc1 = acc1.cpu().numpy()
    acc = np.zeros((N, 3))
    seen = np.zeros(N, dtype=int)
    scale = np.abs(en1[:6]).max()
    planes, recut = 0, False
    for k in range(world):
        z = np.load(os.path.join(str(tmp_path), "oex%d.npz" % k))
        acc[z["idx"]] = z["acc"]
        seen[z["idx"]] += 1
        assert int(z["z0"]) == planes and bool(z["refused"])
        planes += int(z["nz"])
        recut = recut or (int(z["z0"]), int(z["nz"])) != (int(z["z0_before"]), int(z["nz_before"]))
        assert np.abs(z["en"] - en1).max() < 1e-12 * scale, (k, z["en"], en1)
    assert planes == desc.M[2] and (seen == 1).all() and recut
    assert np.sqrt(((acc - acc1) ** 2).sum() / (acc1 ** 2).sum()) < 1e-12
