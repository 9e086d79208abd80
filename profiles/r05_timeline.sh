export TMPDIR=/tmp
for c in cfg2 cfg4; do
d=/tmp/tl_$c
rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --no-measure-traffic --extra-configs none --config $c --steps 120 --warmup 20 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 > /dev/null 2>&1
echo "== $c"; python3 profiles/timeline.py $d 60; python3 profiles/timeline.py $d 61
done
