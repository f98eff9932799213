#!/bin/bash
# The work of one rank of 1, 2, 4, 8 at 1M and at 10M particles on one GPU (collectives as no-ops), as one table
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/rank_sim.txt
for W in 1 2 4 8; do
  timeout -k 10 150 python3 $R/tools/rank_sim.py $W 6 2>/dev/null | grep "^N " >> $O/rank_sim.txt || exit 1
done
RANK_SIM_RANK=0 timeout -k 10 150 python3 $R/tools/rank_sim.py 8 6 2>/dev/null | grep "^N " >> $O/rank_sim.txt || exit 1
for W in 1 2 4 8; do
  RANK_SIM_N=10000000 timeout -k 10 300 python3 $R/tools/rank_sim.py $W 6 2>/dev/null | grep "^N " >> $O/rank_sim.txt || exit 1
done
cat $O/rank_sim.txt
