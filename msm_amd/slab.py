"""z-slab decomposition of one periodic MSM evaluation over the GPUs of a node (one process per GPU).

Rank k owns the level-0 grid planes [k*Mz/W, (k+1)*Mz/W) and the particles whose wrapped z lies in that slab;
it also holds read-only halo copies of the particles within max(a, p/2*hz) of the slab, so the pair sum and the
anterpolation gather of the owned planes need no communication.  Per evaluation there are exactly two
collectives (RCCL over xGMI when the backend is "nccl"):

  1. all-gather of the owned planes of the level-0 charge grid (each rank then holds the whole grid; the
     coarser levels are restricted and solved redundantly because they are too small to shard), issued
     asynchronously so that it overlaps the pair kernel, which does not depend on it;
  2. all-reduce of five scalars (pair energy, q.e over owned planes, Fourier term, sum q, sum q^2).

When the particles have moved (a time-stepping loop writes new positions of the owned particles into SlabRank.r),
SlabRank.exchange() migrates particles to their new owners and rebuilds every rank's halo with two all-to-all
collectives (counts, then records of x, y, z, q and the global index) - the particle halo exchange.

The level-0 convolution — the only grid stage with real work — is evaluated by each rank on its own planes
plus p/2 planes either side (the interpolation support), so potentials need no halo exchange either.
"""
import copy
import ctypes as C

import numpy as np

from . import capi


# ---------------------------------------------------------------------------
# collectives for msm_amd_slab_step (struct msm_amd_collectives)
# ---------------------------------------------------------------------------
class RcclCollectives:
    """The library's own RCCL table (msm_amd_rccl_create): collectives queued on the engine's streams.  Made by
    rccl_table_or_none below, which gets the 128-byte unique id from rank 0 over the torch.distributed group."""

    def __init__(self, ptr):
        self._p = ptr
        self.ptr = ptr

    def comm_info(self):
        """(ranks, rank) as the communicator reports them (ncclCommCount, ncclCommUserRank)."""
        n, r = C.c_int(0), C.c_int(0)
        capi.check(capi.load().msm_amd_rccl_comm_info(self._p, C.byref(n), C.byref(r)), "msm_amd_rccl_comm_info")
        return n.value, r.value

    def close(self):
        if self._p:
            capi.load().msm_amd_rccl_destroy(self._p)
            self._p = C.POINTER(capi.Collectives)()


class _LibRcclApi:
    """The three library calls rccl_table_or_none makes (tests substitute an object that fails where they want)."""

    def __init__(self, torch, device):
        self.torch, self.device, self.lib = torch, device, capi.load()

    def available(self):
        capi.check(self.lib.msm_amd_rccl_available(), "msm_amd_rccl_available")

    def unique_id(self):
        idbuf = C.create_string_buffer(128)
        capi.check(self.lib.msm_amd_rccl_unique_id(idbuf), "msm_amd_rccl_unique_id")
        return idbuf.raw

    def create(self, raw, rank, world):
        p = C.POINTER(capi.Collectives)()
        with self.torch.cuda.device(self.device):
            capi.check(self.lib.msm_amd_rccl_create(raw, rank, world, C.byref(p)), "msm_amd_rccl_create")
        return RcclCollectives(p)


def rccl_table_or_none(torch, dist, rank, world, device, api=None):
    """The library's RCCL table on every rank, or None on every rank.

    Whatever fails on whichever rank, all ranks run the same sequence of collectives on the torch.distributed group, so
    that no rank ever waits in one collective while another has moved on to the next:

      1. all-reduce (min) of "librccl.so loads in this process" - before anyone calls into the library;
      2. broadcast from rank 0 of 129 bytes: a status byte and the unique id (rank 0 always takes part: an id it could not
         make travels as status 0);
      3. msm_amd_rccl_create on every rank - itself a collective (ncclCommInitRank): if it fails on some ranks only, the
         others leave it through RCCL's own timeout;
      4. all-reduce (min) of "my communicator is up"; a rank that has one and learns that another has not destroys it.
    """
    import sys
    api = api or _LibRcclApi(torch, device)
    cdev = device if dist.get_backend() == "nccl" else "cpu"

    def all_ok(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def note(what, e):
        print("msm_amd.slab: rank %d: %s: %s; collectives through torch.distributed instead" % (rank, what, e), file=sys.stderr)

    ok = True
    try:
        api.available()
    except Exception as e:  # noqa: BLE001 - whatever went wrong, the ranks only need to agree
        note("librccl", e)
        ok = False
    if not all_ok(ok):
        return None
    msg = torch.zeros(129, dtype=torch.uint8)
    if rank == 0:
        try:
            raw = api.unique_id()
            msg[0] = 1
            msg[1:] = torch.tensor(list(raw), dtype=torch.uint8)
        except Exception as e:  # noqa: BLE001
            note("unique id", e)
    msg = msg.to(cdev)
    dist.broadcast(msg, 0)
    msg = msg.cpu()
    if int(msg[0]) != 1:
        return None
    table = None
    try:
        table = api.create(bytes(msg[1:].tolist()), rank, world)
    except Exception as e:  # noqa: BLE001
        note("communicator", e)
    if not all_ok(table is not None):
        if table is not None:
            table.close()
        return None
    return table


class _PythonCollectives:
    """A struct msm_amd_collectives whose functions are Python callables (kept alive by this object)."""

    def __init__(self, rank, world, all_gather, all_reduce_sum, halo_exchange):
        self._fns = (capi.ALL_GATHER_FN(all_gather), capi.ALL_REDUCE_FN(all_reduce_sum), capi.HALO_EXCHANGE_FN(halo_exchange))
        self._c = capi.Collectives(rank, world, None, *self._fns)
        self.ptr = C.pointer(self._c)

    def close(self):
        pass


class NullCollectives(_PythonCollectives):
    """Collectives that move nothing: the work of ONE rank of a `world`-rank run timed on one GPU (tools/rank_sim.py)."""

    def __init__(self, rank, world):
        super().__init__(rank, world, lambda *a: 0, lambda *a: 0, lambda *a: 0)


class HostStagedCollectives(_PythonCollectives):
    """The table over a torch.distributed group that cannot move device memory (gloo): every call synchronises the
    device, stages through the host and copies the result back.  For functional tests with several ranks on one GPU."""

    def __init__(self, torch, dist, rank, world, device):
        def view(ptr, count):
            class _W:
                pass
            w = _W()
            w.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(w, device=device)

        def guarded(fn):
            def call(*a):
                try:
                    fn(*a)
                    return 0
                except Exception:        # an exception must not unwind through the C caller
                    import traceback
                    traceback.print_exc()
                    return 1
            return call

        def all_gather(ctx, send, recv, count, stream):
            torch.cuda.synchronize(device)
            mine = view(send, count).cpu()
            parts = [torch.empty(count, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, mine)
            view(recv, count * world).copy_(torch.cat(parts))
            torch.cuda.synchronize(device)

        def all_reduce_sum(ctx, buf, count, stream):
            torch.cuda.synchronize(device)
            t = view(buf, count).cpu()
            dist.all_reduce(t)
            view(buf, count).copy_(t)
            torch.cuda.synchronize(device)

        def halo_exchange(ctx, send_up, send_down, recv_from_down, recv_from_up, count, up, down, stream):
            torch.cuda.synchronize(device)
            su, sd = view(send_up, count).cpu(), view(send_down, count).cpu()
            rd, ru = torch.empty(count, dtype=torch.float64), torch.empty(count, dtype=torch.float64)
            # tags tell the two messages apart when up == down (two ranks)
            ops = [dist.P2POp(dist.isend, su, up, tag=1), dist.P2POp(dist.irecv, rd, down, tag=1),
                   dist.P2POp(dist.isend, sd, down, tag=2), dist.P2POp(dist.irecv, ru, up, tag=2)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            view(recv_from_down, count).copy_(rd)
            view(recv_from_up, count).copy_(ru)
            torch.cuda.synchronize(device)

        super().__init__(rank, world, guarded(all_gather), guarded(all_reduce_sum), guarded(halo_exchange))


class TorchCollectives(_PythonCollectives):
    """The table over a torch.distributed group that moves device memory (backend "nccl", i.e. RCCL): the fallback when
    the library's own communicator cannot be made.  A call runs with the stream it is handed as torch's current stream,
    so the process group orders its work after what that stream holds and the stream waits for the collective; the
    host does not wait."""

    def __init__(self, torch, dist, rank, world, device):
        def view(ptr, count):
            class _W:
                pass
            w = _W()
            w.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(w, device=device)

        def on_stream(fn):
            def call(*a):
                try:
                    ext = torch.cuda.ExternalStream(int(a[-1] or 0), device=device)
                    with torch.cuda.stream(ext):
                        fn(*a[:-1])
                    return 0
                except Exception:        # an exception must not unwind through the C caller
                    import traceback
                    traceback.print_exc()
                    return 1
            return call

        def all_gather(ctx, send, recv, count):
            # (in place: `send` is this rank's share of `recv`, which the process group allows)
            dist.all_gather_into_tensor(view(recv, count * world), view(send, count))

        def all_reduce_sum(ctx, buf, count):
            dist.all_reduce(view(buf, count))

        def halo_exchange(ctx, send_up, send_down, recv_from_down, recv_from_up, count, up, down):
            ops = [dist.P2POp(dist.isend, view(send_up, count), up), dist.P2POp(dist.irecv, view(recv_from_down, count), down),
                   dist.P2POp(dist.isend, view(send_down, count), down), dist.P2POp(dist.irecv, view(recv_from_up, count), up)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()

        super().__init__(rank, world, on_stream(all_gather), on_stream(all_reduce_sum), on_stream(halo_exchange))


def slab_planes(Mz, world, rank):
    """First owned level-0 plane and the number of owned planes of `rank`."""
    if Mz % world != 0:
        raise ValueError("the level-0 grid has %d z planes, not divisible by %d ranks" % (Mz, world))
    nz = Mz // world
    return rank * nz, nz


def select_slab(r, Az, Mz, world, rank, halo):
    """Indices of the particles `rank` owns and of its halo copies.

    Ownership is decided from the wrapped fractional z alone, with the same expression on every rank, so every
    particle is owned exactly once.  halo is a physical distance (periodic in z)."""
    z0, nz = slab_planes(Mz, world, rank)
    zf = np.asarray(r)[:, 2] / Az
    zf = zf - np.floor(zf)
    lo, hi = z0 / Mz, (z0 + nz) / Mz
    owned = (zf >= lo) & (zf < hi)
    hf = halo / Az
    below = np.mod(lo - zf, 1.0)   # periodic distance below the slab
    above = np.mod(zf - hi, 1.0)   # periodic distance above the slab
    halo_mask = (~owned) & ((below <= hf) | (above < hf))
    return np.nonzero(owned)[0], np.nonzero(halo_mask)[0]


def open_plane_counts(zg, Mz):
    """Particles per level-0 plane (numpy): plane = floor of the grid coordinate, clamped to the grid."""
    plane = np.clip(np.floor(np.asarray(zg)).astype(np.int64), 0, Mz - 1)
    return np.bincount(plane, minlength=Mz).astype(np.int64)


def open_edges_from_counts(counts, world):
    """Plane boundaries e[0] = 0 <= e[1] <= ... <= e[world] = Mz from the particle counts per plane: e[k] is the first
    plane boundary with at least k / world of the particles below it, then made strictly increasing (at least one plane
    each).  A function of the counts alone: ranks that hold the all-reduced counts compute the same cut."""
    counts = np.asarray(counts, dtype=np.int64)
    Mz, total = len(counts), int(counts.sum())
    below = np.concatenate([[0], np.cumsum(counts)])          # below[j]: particles in planes < j
    e = [0] + [int(np.searchsorted(below, -(-k * total // world), side="left")) for k in range(1, world)] + [Mz]
    for k in range(1, world):
        e[k] = min(max(e[k], e[k - 1] + 1), Mz - (world - k))
    return e


def open_slab_edges(r, desc, world):
    """Plane boundaries of an open-boundary plan, cut so that every rank owns about the same number of particles (the
    cell is padded: planes at either end hold none).  Deterministic in the positions, so every rank computes the same
    cut; SlabRank.exchange() recomputes it from all-reduced plane counts by the same rule."""
    Mz = desc.M[2]
    zg = (np.asarray(r)[:, 2] - desc.origin[2]) / desc.A[2] * Mz      # grid units
    return open_edges_from_counts(open_plane_counts(zg, Mz), world)


def select_open_slab(r, desc, edges, rank, halo):
    """Owned and halo particle indices of `rank` for an open-boundary plan: no periodic images along z."""
    Mz = desc.M[2]
    zg = (np.asarray(r)[:, 2] - desc.origin[2]) / desc.A[2] * Mz
    lo, hi = edges[rank], edges[rank + 1]
    owner = np.clip(np.searchsorted(np.asarray(edges[1:-1], dtype=np.float64), zg, side="right"), 0, len(edges) - 2)
    owned = owner == rank
    hg = halo / desc.A[2] * Mz
    halo_mask = (~owned) & (zg >= lo - hg) & (zg < hi + hg)
    return np.nonzero(owned)[0], np.nonzero(halo_mask)[0]


def halo_width(desc, margin=0.0):
    hz = desc.A[2] / desc.M[2]
    return max(desc.a, (desc.p / 2) * hz) + margin


class PlanOutgrown(RuntimeError):
    """SlabRank.exchange(): the particles have left the padded cell of an open-boundary plan."""


class SlabRank:
    """One rank of a slab-decomposed evaluation.  `engine_factory(desc)` returns an object with the Engine
    interface (tests inject a CPU stand-in to exercise the partition and collective logic without a GPU).

    perturb=True is a benchmarking device only (bench.py, tools/rank_sim.py): every step() then evaluates the next of
    a few pre-made copies of the held positions, each displaced by up to margin/2, so that no step repeats the previous
    one, and the halo is widened by `margin` to keep every partner of a displaced particle.  The default leaves the
    caller's positions alone."""

    def __init__(self, torch, dist, msm_amd, desc, r_all, q_all, rank, world, local_rank=0, margin=2e-4,
                 engine_factory=None, perturb=False, headroom=1.02, mode=capi.SLAB_AUTO, collectives=None, c_step=True,
                 stream=None):
        """stream: the torch.cuda.Stream the caller queues this rank's work on (its integrator's stream; the caller keeps
        it current around every call).  The engine then runs on it directly.  None: the engine is created on the legacy
        default stream, whose implicit synchronisation with every other stream of the process the C step keeps out of
        its two-stream schedule by moving to streams of its own - two hand-overs between streams per evaluation, about
        20 us on an MI355X."""
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.stream = stream
        self.desc = desc
        self.open = bool(getattr(desc, "open_boundary", 0))
        if self.open:
            # slabs cut by particle count; the level-0 charges are summed over the ranks (msm_amd_slab_step)
            self.edges = open_slab_edges(r_all, desc, world)
            self.z0, self.nz = self.edges[rank], self.edges[rank + 1] - self.edges[rank]
            own, hal = select_open_slab(r_all, desc, self.edges, rank, halo_width(desc, margin if perturb else 0.0))
        else:
            self.z0, self.nz = slab_planes(desc.M[2], world, rank)
            own, hal = select_slab(r_all, desc.A[2], desc.M[2], world, rank, halo_width(desc, margin if perturb else 0.0))
        self.own_idx, self.halo_idx = own, hal
        n_own, n_tot = len(own), len(own) + len(hal)
        d = copy.copy(desc)
        d.N = int(n_tot * headroom) + 64   # capacity; the counts of a step are set with set_counts
        d.zplane0, d.nzplanes, d.n_owned = self.z0, self.nz, 0
        self.eng = engine_factory(d) if engine_factory else msm_amd.Engine(
            d, device=local_rank, stream=stream.cuda_stream if stream is not None else None)
        self.eng.set_counts(n_tot, n_own)
        idx = np.concatenate([own, hal])
        dev = self.eng.tensor_device()
        self.r = torch.as_tensor(np.ascontiguousarray(np.asarray(r_all)[idx]), device=dev)
        self.q = torch.as_tensor(np.ascontiguousarray(np.asarray(q_all)[idx]), device=dev)
        self.acc = torch.zeros((n_tot, 3), dtype=torch.float64, device=dev)
        self.n_own, self.n_tot, self.capacity = n_own, n_tot, d.N
        self.gid = torch.as_tensor(idx.astype(np.int64), device=dev)   # global index of every held particle
        self.Az, self.halo, self.margin = desc.A[2], halo_width(desc, margin if perturb else 0.0), margin
        self.q0 = self.eng.grid_tensor(capi.GRID_Q, 0)          # (Mz, My, Mx) view of the engine's level-0 charges
        self.send = torch.empty_like(self.q0[self.z0:self.z0 + self.nz])
        self.sums = self.eng.raw_sums_tensor()
        self.perturb = perturb
        self._n_all = len(q_all)
        if perturb:
            self._make_position_sets()
        self._pending = self._pending_work = self._energies = None
        # The real engine runs the whole step in C (msm_amd_slab_step: the exchange on a second stream beside the pair sum,
        # collectives from a table); a stand-in engine (CPU tests of the partition logic) is driven stage by stage below.
        self.mode = mode
        self.coll = None
        if hasattr(self.eng, "slab_step") and world > 1 and c_step:
            if collectives is not None:
                self.coll = collectives
            elif dist.get_backend() == "nccl":
                # The library's own communicator, or - when any rank cannot make one - the table over the process group on
                # every rank (rccl_table_or_none keeps the ranks on one sequence of collectives while they find out).
                self.coll = rccl_table_or_none(torch, dist, rank, world, dev)
                if self.coll is None:
                    self.coll = TorchCollectives(torch, dist, rank, world, dev)
            else:
                self.coll = HostStagedCollectives(torch, dist, rank, world, dev)
                self.eng.set_option("slab_overlap", 0)   # ranks sharing a GPU: one stream per process

    POSITION_SETS = 16

    def _make_position_sets(self):
        """Benchmarking device: POSITION_SETS displaced copies of the held positions, made before any timed step, so
        that consecutive steps evaluate different inputs without a displacement kernel between them.  The displacement
        of a particle is a function of its global index (the same table on every rank), so halo copies move with
        their owners; it is at most margin / 2."""
        t = self.torch
        g = t.Generator(device="cpu")
        g.manual_seed(7)
        table = (t.rand((self._n_all, 3), generator=g, dtype=t.float64) - 0.5) * self.margin
        delta = table[self.gid.cpu()].to(self.r.device)
        K = self.POSITION_SETS
        self.r0 = self.r
        self.r_sets = t.stack([self.r0 + delta * (((k + 1) / K) * (1.0 if k % 2 == 0 else -1.0)) for k in range(K)])
        self._set = 0

    def exchange(self):
        """Particle migration and halo refresh after the owned particles have moved (self.r[:n_own] rewritten by the
        caller; halo rows are ignored).  Ownership and halo membership follow select_slab's expressions on the new
        positions; every rank sends each of its owned particles to the new owner and to every rank whose halo it
        falls in, owned records before halo records, so a receiver holds [owned from rank 0, 1, ... | halo from rank
        0, 1, ...].  Two collectives: all-to-all of the counts, all-to-all of the records.  An open-boundary plan adds two
        all-reduces in front (bounding box, particles per plane) and cuts its slabs again; it raises PlanOutgrown when the
        particles no longer fit the plan's padded cell."""
        t, W = self.torch, self.world
        n = self.n_own
        r, q, gid = self.r[:n], self.q[:n], self.gid[:n]
        rec = t.cat([r, q[:, None], gid.to(t.float64)[:, None]], dim=1)   # global indices below 2^53 are exact
        own_parts, halo_parts, counts = [], [], t.zeros((W, 2), dtype=t.int64)
        if self.open:
            # Open boundaries: the padded cell stays while the particles' bounding box stays inside it
            # (msm_amd_plan_open_holds), and the slabs are cut again where the particles now are - by the rule of
            # open_slab_edges, from plane counts added over the ranks - so the owned planes of a rank move too.
            d = self.desc
            Mz = d.M[2]
            big = 1e300
            lo3 = r.min(dim=0).values if n else t.full((3,), big, dtype=t.float64, device=r.device)
            hi3 = r.max(dim=0).values if n else t.full((3,), -big, dtype=t.float64, device=r.device)
            zg = (r[:, 2] - d.origin[2]) / d.A[2] * Mz
            pc = t.bincount(t.clamp(t.floor(zg), 0, Mz - 1).to(t.int64), minlength=Mz)
            cdev = "cpu" if self._host_staged() or self.dist.get_backend() == "gloo" else r.device
            box = t.cat([-lo3, hi3]).to(cdev)
            pc = pc.to(cdev)
            self.dist.all_reduce(box, op=self.dist.ReduceOp.MAX)
            self.dist.all_reduce(pc)
            box = box.cpu().numpy()
            lo_all, hi_all = -box[:3], box[3:]
            if not self._plan_holds(lo_all, hi_all):
                raise PlanOutgrown("the particles' bounding box [%s, %s] has left the padded cell of the open-boundary plan: make a "
                                   "new plan around them (msm_amd_plan_open) and a new SlabRank" % (lo_all, hi_all))
            self.edges = open_edges_from_counts(pc.cpu().numpy(), W)
            inner = t.as_tensor(np.asarray(self.edges[1:-1], dtype=np.float64), device=r.device)
            owner = t.clamp(t.bucketize(zg, inner, right=True), 0, W - 1)     # = searchsorted(..., side="right") of select_open_slab
            hg = self.halo / d.A[2] * Mz
            for k in range(W):
                mine = owner == k
                halo = (~mine) & (zg >= self.edges[k] - hg) & (zg < self.edges[k + 1] + hg)
                own_parts.append(rec[mine])
                halo_parts.append(rec[halo])
                counts[k, 0], counts[k, 1] = int(mine.sum()), int(halo.sum())
            z0, nz = self.edges[self.rank], self.edges[self.rank + 1] - self.edges[self.rank]
            if (z0, nz) != (self.z0, self.nz):
                self.eng.set_planes(z0, nz)
                self.z0, self.nz = z0, nz
                self.send = t.empty_like(self.q0[self.z0:self.z0 + self.nz])
        else:
            zf = r[:, 2] / self.Az
            zf = zf - t.floor(zf)
            edges = t.arange(W + 1, dtype=t.float64, device=r.device) / W   # k / W: the bounds select_slab compares with
            owner = t.clamp(t.bucketize(zf, edges, right=True) - 1, 0, W - 1)
            hf = self.halo / self.Az
            for k in range(W):
                lo, hi = edges[k], edges[k + 1]
                mine = owner == k
                below = t.remainder(lo - zf, 1.0)
                above = t.remainder(zf - hi, 1.0)
                halo = (~mine) & ((below <= hf) | (above < hf))
                own_parts.append(rec[mine])
                halo_parts.append(rec[halo])
                counts[k, 0], counts[k, 1] = int(mine.sum()), int(halo.sum())
        send = t.cat([x for k in range(W) for x in (own_parts[k], halo_parts[k])], dim=0)
        staged = self._host_staged()
        cdev = "cpu" if staged or self.dist.get_backend() == "gloo" else r.device
        cs, cr = counts.to(cdev), t.zeros((W, 2), dtype=t.int64, device=cdev)
        self.dist.all_to_all_single(cr, cs)
        cr = cr.cpu()
        in_rows = [int(cr[k, 0] + cr[k, 1]) for k in range(W)]
        out_rows = [int(counts[k, 0] + counts[k, 1]) for k in range(W)]
        send_b = send.cpu() if staged else send
        recv = t.empty((sum(in_rows), 5), dtype=t.float64, device=send_b.device)
        self.dist.all_to_all_single(recv, send_b, output_split_sizes=in_rows, input_split_sizes=out_rows)
        recv = recv.to(r.device)
        owned, halo, at = [], [], 0
        for k in range(W):
            no, nh = int(cr[k, 0]), int(cr[k, 1])
            owned.append(recv[at:at + no])
            halo.append(recv[at + no:at + no + nh])
            at += no + nh
        new = t.cat(owned + halo, dim=0)
        n_own, n_tot = sum(int(cr[k, 0]) for k in range(W)), new.shape[0]
        if n_tot > self.capacity:
            raise RuntimeError("slab rank %d: %d particles (owned + halo) exceed the engine's capacity %d; create the "
                               "SlabRank with a larger headroom" % (self.rank, n_tot, self.capacity))
        self.r = new[:, 0:3].contiguous()
        self.q = new[:, 3].contiguous()
        self.gid = new[:, 4].to(t.int64)
        self.acc = t.zeros((n_tot, 3), dtype=t.float64, device=r.device)
        self.n_own, self.n_tot = n_own, n_tot
        ids = self.gid.cpu().numpy()
        self.own_idx, self.halo_idx = ids[:n_own], ids[n_own:]
        self.eng.set_counts(n_tot, n_own)
        if self.perturb:
            self._make_position_sets()

    def _plan_holds(self, lo, hi):
        from . import engine
        return engine.plan_open_holds(self.desc, lo, hi)

    def _host_staged(self):
        # a gloo group cannot move device memory: stage through the host (functional testing on one GPU)
        return self.dist.get_backend() == "gloo" and self.q0.is_cuda

    def _all_gather_planes(self, async_op):
        self.send.copy_(self.q0[self.z0:self.z0 + self.nz])
        if self._host_staged():
            parts = [self.torch.empty(self.send.shape, dtype=self.torch.float64) for _ in range(self.world)]
            self.dist.all_gather(parts, self.send.cpu())
            for k in range(self.world):
                self.q0[k * self.nz:(k + 1) * self.nz].copy_(parts[k])
            return None
        try:
            return self.dist.all_gather_into_tensor(self.q0.view(-1), self.send.view(-1), async_op=async_op)
        except (RuntimeError, NotImplementedError):
            parts = [self.q0[k * self.nz:(k + 1) * self.nz] for k in range(self.world)]
            return self.dist.all_gather(parts, self.send, async_op=async_op)

    def step(self, marks=None, want_energies=True):
        """One evaluation.  marks: optional callable(name) invoked between stages (bench.py records events).

        With want_energies=False nothing waits for the GPU: the reduced sums stay on the device and `last_energies`
        combines them when it is read, so a time-stepping loop queues its evaluations back to back."""
        eng = self.eng
        mark = marks if marks is not None else (lambda name: None)
        if self.perturb:
            self.r = self.r_sets[self._set % self.POSITION_SETS]
            self._set += 1
        if self.open and self.coll is None:
            raise ValueError("open-boundary slabs need the C step (msm_amd_slab_step)")
        if self.coll is not None:
            mark("step")
            eng.slab_step(self.coll.ptr, self.r, self.q, self.acc, self.mode)
            mark("end")
            self._pending, self._pending_work, self._energies = "c", None, None
            return self.last_energies if want_energies else None
        mark("bin")
        eng.stage_bin(self.r, self.q)
        mark("anterpolation")
        eng.stage_anterpolate()
        mark("allgather_issue")
        work = self._all_gather_planes(async_op=True)
        mark("pairs")
        eng.stage_pairs(self.acc)          # independent of the gather: runs while the planes travel
        mark("allgather_wait")
        if work is not None:
            work.wait()
        mark("grid")
        eng.stage_grid()
        mark("interpolation")
        eng.stage_interpolate(self.acc)
        mark("reductions")
        eng.stage_raw_sums()
        mark("end")
        s = self.sums.clone()
        if self._host_staged():
            s = s.cpu()
        # asynchronous: the kernels of the next evaluation do not wait for this collective, only a reader of the energies does
        work = self.dist.all_reduce(s, async_op=True)
        self._pending, self._pending_work, self._energies = s, work, None
        return self.last_energies if want_energies else None

    @property
    def last_energies(self):
        if self._energies is None and isinstance(self._pending, str):
            self._energies = self.eng.slab_energies(self.coll.ptr)
        if self._energies is None and self._pending is not None:
            if self._pending_work is not None:
                self._pending_work.wait()
                self._pending_work = None
            sums = self._pending.cpu().numpy().copy()
            sums[2] /= self.world          # the Fourier term is replicated on every rank: the all-reduce added it W times
            self._energies = self.eng.combine_energies(sums)
        return self._energies

    def owned_acc(self):
        return self.acc[:self.n_own]


def direct_rows(world, rank, N):
    """Rows [i0, i1) of the open-boundary direct sum that `rank` computes: equal shares, the remainder to the first ranks."""
    base, rem = divmod(N, world)
    i0 = rank * base + min(rank, rem)
    return i0, i0 + base + (1 if rank < rem else 0)


def direct_sharded(torch, dist, r_all, q_all, rank, world, local_rank=0, host_staged=None):
    """Open-boundary direct sum (direct() of msm/msmLibrary.h:212, loop of msm/nonperiodic.c:12-22) over the GPUs of a node.

    Every rank holds all positions and computes the rows [i0, i1) of the pair matrix (msm_amd_direct_device); the
    path has no exchange step: the slice energies are added with one all-reduce of a scalar, and each rank
    keeps the accelerations of its own rows.  Returns (u_total, acc_rows (i1 - i0, 3) CUDA tensor, (i0, i1))."""
    import ctypes as C
    lib = capi.load()
    dev = torch.device("cuda:%d" % local_rank)
    rd = torch.as_tensor(np.ascontiguousarray(r_all, dtype=np.float64), device=dev)
    qd = torch.as_tensor(np.ascontiguousarray(q_all, dtype=np.float64), device=dev)
    N = qd.shape[0]
    i0, i1 = direct_rows(world, rank, N)
    acc = torch.empty((max(i1 - i0, 1), 3), dtype=torch.float64, device=dev)
    work = torch.empty(lib.msm_amd_direct_work_doubles(N), dtype=torch.float64, device=dev)
    u = C.c_double(0.0)
    if i1 > i0:
        with torch.cuda.device(dev):
            capi.check(lib.msm_amd_direct_device(N, rd.data_ptr(), qd.data_ptr(), i0, i1, capi.ACC_GRADIENT, acc.data_ptr(),
                                                 work.data_ptr(), torch.cuda.current_stream().cuda_stream, C.byref(u)),
                       "direct_device")
    if host_staged is None:
        host_staged = dist.get_backend() != "nccl"
    t = torch.tensor([u.value], dtype=torch.float64, device="cpu" if host_staged else dev)
    dist.all_reduce(t)
    return float(t.item()), acc[:i1 - i0], (i0, i1)
