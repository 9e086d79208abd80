#!/bin/bash
# HBM traffic of every kernel from the L2 memory-side counters, collected as MI355X_MICROARCH.md §HBM / §rocprofv3
# PMC slots prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), counters only
# (no tracing domains besides --kernel-trace).  The profiled bench.py must not start profiled children of its own
# (--no-measure-traffic) and runs one context only (--no-preheat), so that launches and frames belong to one leg.  Usage: profiles/collect_traffic.sh <out-dir> [bench args...]
set -e
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-total --no-host-buffers --no-measure-traffic --no-preheat "$@" > "$out/fetch.json" 2> "$out/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-total --no-host-buffers --no-measure-traffic --no-preheat "$@" > "$out/write.json" 2> "$out/write.err"
python3 profiles/summarise_traffic.py "$out" 50      # 40 + 10 frames of one context
