#!/bin/bash
# End-of-round artifacts on the GPU box: official bench lines, rocprofv3 kernel statistics, HBM traffic.
# usage: profiles/collect_round.sh <round-tag>     (writes under gpurun_out/<tag>/; copy what is wanted into profiles/)
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_cfg2_driver_shape.json 2> $out/bench_cfg2_driver_shape.err || exit 1
python bench.py > $out/bench_cfg2.json 2> $out/bench_cfg2.err || exit 1
tail -c 600 $out/bench_cfg2.json; echo
python bench.py --config cfg3 --steps 200 --cpu-seconds 10 --extra-configs none > $out/bench_cfg3.json 2> $out/bench_cfg3.err || exit 1
python bench.py --config cfg4 --steps 200 --cpu-seconds 10 --extra-configs none > $out/bench_cfg4.json 2> $out/bench_cfg4.err || exit 1
for cfg in cfg2 cfg3 cfg4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$cfg -- python3 bench.py --config $cfg --steps 120 --warmup 20 --no-cpu-baseline --no-total --no-host-buffers --extra-configs none --no-measure-traffic > $out/stats_$cfg.json 2> $out/stats_$cfg.err || exit 1
  python3 profiles/prof_summary.py $out/stats_$cfg > $out/kernel_stats_$cfg.txt
  cp "$(ls $out/stats_$cfg/*/*_kernel_stats.csv | head -1)" $out/kernel_stats_$cfg.csv
  head -6 $out/kernel_stats_$cfg.txt
  timeout -k 10 400 bash profiles/collect_traffic.sh $out/traffic_$cfg --config $cfg --extra-configs none > $out/traffic_$cfg.txt 2>&1 || exit 1
  cp $out/traffic_$cfg/traffic.json $out/traffic_$cfg.json
done
