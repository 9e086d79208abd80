#!/bin/bash
# End-of-round artifacts on the GPU box: official bench lines, rocprofv3 kernel statistics, HBM traffic.
# usage: profiles/collect_round.sh <round-tag>     (writes under gpurun_out/<tag>/; copy what is wanted into profiles/)
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench_cfg2.json 2> $out/bench_cfg2.err || exit 1
tail -c 400 $out/bench_cfg2.json; echo
python bench.py --config cfg3 --steps 200 --cpu-seconds 10 > $out/bench_cfg3.json 2> $out/bench_cfg3.err || exit 1
python bench.py --config cfg4 --steps 200 --cpu-seconds 10 > $out/bench_cfg4.json 2> $out/bench_cfg4.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 120 --warmup 20 --no-cpu-baseline --no-total --no-host-buffers > $out/stats.json 2> $out/stats.err || exit 1
python3 profiles/prof_summary.py $out/stats > $out/kernel_stats.txt
cat $out/kernel_stats.txt
timeout -k 10 400 bash profiles/collect_traffic.sh $out/traffic || exit 1
