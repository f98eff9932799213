#!/bin/bash
# Per-kernel times of the rank simulation under engine options: tools/ab_trace.sh OUTDIR "W..." "options|options|..." [kernel-name filter]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
IFS='|' read -ra OPTS <<< "$3"
F=${4:-k_}
for W in $2; do
  for o in "${OPTS[@]}"; do
    tag=$(echo "W${W}_$o" | tr '=,' '__')
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $R/tools/rank_sim.py $W 6 "$o" > $O/$tag.log 2>&1 || exit 1
    echo "== W $W opts $o" >> $O/trace.log
    grep -h "ms/step" $O/$tag.log >> $O/trace.log
    python3 - $O/$tag "$F" >> $O/trace.log <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Name"] and "k_st_" not in r["Name"]:
        print("   %-64s %4s %9.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
cat $O/trace.log
