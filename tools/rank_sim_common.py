"""A stand-in for torch.distributed in the one-rank simulations (tools/rank_sim.py, tools/pair_timeline.py)."""


class FakeWork:
    def wait(self): pass


class FakeDist:
    def get_backend(self): return "nccl"
    def all_gather_into_tensor(self, out, inp, async_op=False): return FakeWork()
    def all_reduce(self, t, async_op=False): return None


def delay_collectives(torch, rank, world, us_halo, us_gather, us_reduce):
    """Collectives that move nothing but hold the stream they are queued on for a given time (a one-wave kernel that
    spins on the clock): the work of one rank with the latency of its exchanges, on one GPU.  The scalar all-reduce at
    the end of a step gets us_reduce."""
    from msm_amd.slab import _PythonCollectives
    # clock ticks of torch.cuda._sleep per microsecond, measured
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    e0.record(); torch.cuda._sleep(2000000); e1.record()
    torch.cuda.synchronize()
    per_us = 2000000 / (e0.elapsed_time(e1) * 1e3)

    def hold(us, stream):
        if us > 0:
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream or 0))):
                torch.cuda._sleep(int(us * per_us))
        return 0

    return _PythonCollectives(rank, world,
                              lambda ctx, send, recv, count, stream: hold(us_gather, stream),
                              lambda ctx, buf, count, stream: hold(us_reduce if count <= 8 else us_gather, stream),
                              lambda ctx, a, b, c, d, count, up, down, stream: hold(us_halo, stream))
