"""Per step of a rocprofv3 --kernel-trace of tools/rank_sim.py: span of the step, and start / duration (us) of the kernels
named on the command line, to see how the two streams of a slab step fell out step by step.
python tools/trace_steps.py <dir> k_pair_clusters_filtered k_conv_split ..."""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
names = sys.argv[2:] or ["k_cl_list", "k_pair_clusters_filtered", "k_conv_split", "k_conv_small_pair", "k_interp_tiles"]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_cs_hist" in r["Kernel_Name"]]
for n, (a, b) in enumerate(zip(starts[:-1], starts[1:])):
    t0 = int(rows[a]["Start_Timestamp"])
    end = max(int(r["End_Timestamp"]) for r in rows[a:b])
    gap = (int(rows[b]["Start_Timestamp"]) - end) / 1e3
    out = ["step %2d span %6.1f gap-to-next %6.1f" % (n, (end - t0) / 1e3, gap)]
    for name in names:
        for r in rows[a:b]:
            if name in r["Kernel_Name"]:
                out.append("%s %.1f+%.1f" % (name.replace("k_", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
                break
    print("  ".join(out))
