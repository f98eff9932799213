#!/bin/bash
# A/B of engine options on the rank simulation: tools/ab_rank.sh OUTDIR "W..." "options|options|..." ("-" = defaults)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
IFS='|' read -ra OPTS <<< "$3"
for W in $2; do
  for o in "${OPTS[@]}"; do
    echo "== W $W opts $o" >> $O/ab.log
    timeout -k 10 120 python3 $R/tools/rank_sim.py $W 6 "$o" >> $O/ab.log 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $O/ab.log
