#!/bin/bash
# Profile capture on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes.
# Every step has its own time limit; the script stops at the first failure.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py > $O/stats.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --no-cpu-baseline --no-accuracy --steps 3 --warmup 1 > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --no-cpu-baseline --no-accuracy --steps 3 --warmup 1 > $O/write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $R/bench.py --no-cpu-baseline --no-accuracy --steps 3 --warmup 1 > $O/sq.log 2>&1 || exit 1
timeout -k 10 300 python3 $R/bench.py --order 4 > $O/bench_order4.json 2> $O/bench_order4.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch4 -- python3 $R/bench.py --order 4 --no-cpu-baseline --no-accuracy --steps 3 --warmup 1 > $O/fetch4.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write4 -- python3 $R/bench.py --order 4 --no-cpu-baseline --no-accuracy --steps 3 --warmup 1 > $O/write4.log 2>&1 || exit 1
echo done
