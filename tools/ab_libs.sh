#!/bin/bash
# time tools/opt_bench.py with the default library and every build under tools/ablibs/ (A/B of compile-time choices)
cd "$(dirname "$0")/.."
timeout -k 10 120 python tools/opt_bench.py "" "" 2>/dev/null | cut -c1-330 || exit 1
for lib in tools/ablibs/*.so; do
    echo "== $lib"
    MSM_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/opt_bench.py "" "" 2>/dev/null | cut -c1-330 || exit 1
done
