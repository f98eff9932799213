#!/bin/bash
# A/B on one rank of 8 (RANKS=n: of n): engine options given as AB_OPTS="opt,opt opt ..." ("-": none), one run each
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/rank8_ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/ab.txt
for opt in ${AB_OPTS:--}; do
  echo "== $opt" >> $O/ab.txt
  timeout -k 10 150 python3 $R/tools/rank_sim.py ${RANKS:-8} 6 $opt 2>/dev/null | grep "^N " >> $O/ab.txt || exit 1
done
cat $O/ab.txt
