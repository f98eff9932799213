"""A stand-in for torch.distributed in the one-rank simulations (tools/rank_sim.py, tools/pair_timeline.py)."""


class FakeWork:
    def wait(self): pass


class FakeDist:
    def get_backend(self): return "nccl"
    def all_gather_into_tensor(self, out, inp, async_op=False): return FakeWork()
    def all_reduce(self, t, async_op=False): return None
