#!/bin/bash
# One evaluation of the default bench.py run under rocprofv3 --kernel-trace: which kernels ran when (gaps between them show)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/bench_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/btr
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/btr -- python3 $R/bench.py --no-cpu-baseline --no-accuracy --no-list-reuse --steps 30 --warmup 3 > $O/run.log 2>&1 || exit 1
TRACE_STEP=20 python3 $R/tools/trace_overlap.py /tmp/btr > $O/timeline.txt || exit 1
python3 $R/tools/trace_steps.py /tmp/btr k_pair_clusters_filtered >> $O/timeline.txt || exit 1
cat $O/timeline.txt
