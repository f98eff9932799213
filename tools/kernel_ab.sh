#!/bin/bash
# Kernel durations (rocprofv3 kernel trace) of one rank of W under several option sets:
#   tools/kernel_ab.sh W pattern optionset [optionset ...]     (optionset: key=value,key=value or -)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for o in "$@"; do
    d=$R/gpurun_out/kab/$(echo "$o" | tr ',=' '__')
    rm -rf "$d"; mkdir -p "$d"
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 $R/tools/rank_sim.py $W 6 "$o" > "$d/log.txt" 2>&1 || exit 1
    grep "^N " "$d/log.txt" | sed "s/^/$o /"
    grep -E "$PAT" "$d"/*/*kernel_stats.csv | awk -F'","|",|,"' -v o="$o" '{n=$1; sub(/.*"/,"",n); printf "   %s  %-60.60s calls %s avg_ns %s\n", o, n, $2, $4}'
done
