"""Turn what tools/capture_profiles.sh left under gpurun_out/final/ into files under profiles/ with the given name prefix.

    python tools/process_profiles.py r01
"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_summary import summarise

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
F = os.path.join(ROOT, "gpurun_out", "final") + os.sep
P = os.path.join(ROOT, "profiles", tag + "_")
KERNEL = "k_pair_clusters_filtered"
# gpurun merges every capture into the same directories: keep the files of the newest run (one process id prefix) only
for sub in ("stats", "fetch", "write", "sq", "fetch4", "write4"):
    infos = sorted(glob.glob(F + sub + "/**/*_agent_info.csv", recursive=True), key=os.path.getmtime)
    for old in infos[:-1]:
        for f in glob.glob(old[:-len("agent_info.csv")] + "*"):
            os.remove(f)
shutil.copy(F + "bench_default.json", P + "bench_default.json")
shutil.copy(F + "bench_order4.json", P + "bench_order4.json")
shutil.copy(glob.glob(F + "stats/**/*kernel_stats.csv", recursive=True)[0], P + "bench_default_kernel_stats.csv")
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of `python3 bench.py --no-cpu-baseline "
                 "--no-accuracy --order P --steps 3 --warmup 1`, mean over the launches of %s; counters are in KB" % KERNEL,
       "correction": "gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for 16-byte-per-lane loads (the loads of this "
                     "kernel are global_load_dwordx4, dword list entries and scalar loads): bytes = 2 * FETCH_SIZE + WRITE_SIZE",
       "algorithmic_bytes_note": "one pass over the slots (41 MB), the neighbour lists (about 70 MB) and the outputs (25 MB); "
                                 "the rest is j clusters re-read after falling out of an XCD's L2"}
for key, (f, w) in {"order6": ("fetch", "write"), "order4": ("fetch4", "write4")}.items():
    fs = list(summarise(F + f, [KERNEL]).values())[0]
    ws = list(summarise(F + w, [KERNEL]).values())[0]
    out[key] = {"FETCH_SIZE_KB": round(fs["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(ws["WRITE_SIZE"], 1),
                "hbm_bytes_per_launch": int((2 * fs["FETCH_SIZE"] + ws["WRITE_SIZE"]) * 1024), "launches": fs["launches"]}
json.dump(out, open(P + "pair_traffic.json", "w"), indent=1)
sq = {k: v for k, v in summarise(F + "sq", []).items() if "msm::" in k}
res = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY "
                 "SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -- python3 bench.py --no-cpu-baseline --no-accuracy --steps 3 "
                 "--warmup 1 (p=6); mean per launch (tools/pmc_summary.py)",
       "reading": "VALU busy fraction = SQ_INSTS_VALU * 4 cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)"}
res.update(sq)
json.dump(res, open(P + "sq_counters.json", "w"), indent=1, sort_keys=True)
for name in ("host6.log", "host4.log"):
    if os.path.exists(F + name):
        shutil.copy(F + name, P + "host_rate_" + name.replace("host", "order").replace(".log", ".txt"))
d = json.load(open(F + "bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print(d["kernel_ms"])
print(out["order6"], out["order4"])
