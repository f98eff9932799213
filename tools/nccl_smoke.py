"""One-rank RCCL run of the slab driver: exercises the all-gather / all-reduce calls on device memory owned by the
engine exactly as the multi-GPU bench does (a one-GPU box cannot host two RCCL ranks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch
import torch.distributed as dist
import msm_amd
from msm_amd.slab import SlabRank
from bench import make_system

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
N = 200000
r, q = make_system(N, 5)
desc = msm_amd.plan_auto(N, (1.0, 1.0, 1.0), 6, 10, 4.0, 0)
s = SlabRank(torch, dist, msm_amd, desc, r, q, 0, 1, 0, perturb=False)
en = s.step()
eng = msm_amd.Engine(desc)
en1, acc1 = eng.eval(r, q)
print("slab(nccl, 1 rank) utotal", en[6], "single", en1[6], "max |d acc|",
      float((s.owned_acc().cpu() - acc1.cpu()[torch.as_tensor(s.own_idx)]).abs().max()))
assert abs(en[6] - en1[6]) <= 1e-12 * np.abs(en1[:6]).max()
dist.destroy_process_group()
print("ok")
