"""Does RCCL accept two ranks on one device?  (It has been the only way to meet a two-rank communicator on a one-GPU box.)
Prints what happened; every outcome is an answer."""
import os, sys, subprocess, time
if "RANK" not in os.environ:
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    ps = [subprocess.Popen([sys.executable, __file__], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(2)]
    t0 = time.time()
    while time.time() - t0 < 45 and any(p.poll() is None for p in ps):
        time.sleep(0.5)
    for p in ps:
        if p.poll() is None:
            p.kill()
            print("rank still running after 45 s: killed")
    print("exit codes", [p.returncode for p in ps])
    sys.exit(0)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda:0"))
    t = torch.ones(4, device="cuda") * (int(os.environ["RANK"]) + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", os.environ["RANK"], "all_reduce ->", t.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank", os.environ["RANK"], "failed:", type(e).__name__, str(e)[:300], flush=True)
    sys.exit(3)
