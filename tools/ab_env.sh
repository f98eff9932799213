#!/bin/bash
# A/B of an environment variable on the rank simulation: tools/ab_env.sh OUTDIR "W..." VAR "value|value|..." ("-" = unset)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
IFS='|' read -ra VALS <<< "$4"
for W in $2; do
  for v in "${VALS[@]}"; do
    echo "== W $W $3=$v" >> $O/ab.log
    if [ "$v" = "-" ]; then unset $3; else export $3="$v"; fi
    timeout -k 10 120 python3 $R/tools/rank_sim.py $W 6 >> $O/ab.log 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $O/ab.log
