"""Print the kernels of the last complete step found in a rocprofv3 --kernel-trace CSV with start / end times (us) relative
to the step's first kernel and the queue they ran on: shows which kernels ran side by side.
python tools/trace_overlap.py <dir> [first-kernel-substring]   (TRACE_STEP=n picks the n-th step)"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
first = sys.argv[2] if len(sys.argv) > 2 else "k_cs_hist"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
k = int(os.environ.get("TRACE_STEP", "-2"))   # which step (default: the last complete one; rank_sim's last five are profiled steps)
a, b = starts[k], starts[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print("%8.1f %8.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"),
                                     r["Kernel_Name"].split("(")[0][:80]))
