"""The slab path with the real engine: two and four ranks share cuda:0 (collectives staged through a gloo
group, since RCCL refuses two ranks on one device) and must reproduce the single-GPU evaluation."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, N, p, L, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)   # a stuck rank ends with its stacks instead of hanging the run
    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd.slab import SlabRank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=43)
    desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), p, 10, 4.0, L)
    sr = SlabRank(torch, dist, msm_amd, desc, r, q, rank, world, local_rank=0, perturb=False)
    for _ in range(2):   # twice: the second step must not see state left by the first
        en = sr.step()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), en=en, idx=sr.own_idx, acc=sr.owned_acc().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,N,p,L", [(2, 40000, 4, 2), (4, 40000, 6, 3)])
def test_slab_ranks_match_single_gpu(tmp_path, world, N, p, L):
    import torch
    import torch.multiprocessing as mp
    import msm_amd
    mp.spawn(_worker, args=(world, _free_port(), N, p, L, str(tmp_path)), nprocs=world, join=True)
    r, q = O.neutral_system(N, (1.0, 1.0, 1.0), seed=43)
    desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), p, 10, 4.0, L)
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
        assert np.abs(z["en"] - en1).max() < 1e-12 * scale, (k, z["en"], en1)
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
