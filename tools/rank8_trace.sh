#!/bin/bash
# Steps of a rank of 8 under rocprofv3 --kernel-trace, per engine option set: which kernels ran when, on which queue
# (one step in full, and the kernels that the two streams race for, step by step)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/rank8_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for opt in ${AB_OPTS:--}; do
  i=$((i + 1))
  rm -rf /tmp/tr$i
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$i -- python3 $R/tools/rank_sim.py ${RANKS:-8} 6 $opt > $O/run$i.log 2>&1 || exit 1
  echo "== $opt" > $O/timeline$i.txt
  TRACE_STEP=15 python3 $R/tools/trace_overlap.py /tmp/tr$i >> $O/timeline$i.txt || exit 1
  python3 $R/tools/trace_steps.py /tmp/tr$i >> $O/timeline$i.txt || exit 1
done
cat $O/timeline*.txt
