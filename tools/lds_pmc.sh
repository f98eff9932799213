#!/bin/bash
# LDS counters of the kernels of a short benchmark run:  tools/lds_pmc.sh [pattern]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=${1:-interp|anterp_cols|conv_multi}
d=$R/gpurun_out/ldspmc/cur
rm -rf "$d"; mkdir -p "$d"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$d" -- python3 $R/bench.py --no-cpu-baseline --no-accuracy --steps 3 --warmup 1 > "$d.log" 2>&1 || exit 1
python3 - "$d" "$PAT" <<'PY'
import csv, collections, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    if re.search(sys.argv[2], k):
        m = {c: sum(x) / len(x) for c, x in v.items()}
        print(k, 'cycles/inst %.1f' % (m['SQ_LDS_IDX_ACTIVE'] / max(m['SQ_INSTS_LDS'], 1)),
              'conflict share %.2f' % (m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1)),
              'LDS active per CU %.0f' % (m['SQ_LDS_IDX_ACTIVE'] / 256), {c: round(x) for c, x in m.items()})
PY
