python -m pytest tests/test_hip_kernels.py tests/test_hip_fullsize.py tests/test_hip_fuzz.py tests/test_hip_pipeline.py tests/test_hip_round3.py tests/test_hip_round5.py -m gpu -x -q -s -k "not 880 and not 300_frames" > gpurun_out/r05_tests_3.log 2>&1; tail -5 gpurun_out/r05_tests_3.log
for c in cfg2 cfg3 cfg4; do bash profiles/kstat.sh $c | head -1 | sed "s/^A=1 */$c /"; done > gpurun_out/r05_kstat_3.txt 2>&1; cat gpurun_out/r05_kstat_3.txt
bash profiles/r05_int_census.sh cfg3 gpurun_out/r05_integrate_census_cfg3_v3.txt
python bench.py --steps 200 --no-cpu-baseline --no-host-buffers --no-measure-traffic --extra-configs cfg3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('cfg2 value', d['value'], 'fps_total', d['fps_total'], 'cfg3', d['configs']['cfg3']['value'])
" > gpurun_out/r05_bench_quick_3.txt 2>&1; cat gpurun_out/r05_bench_quick_3.txt
KFH_RENDER_ZEROCOPY=0 python bench.py --steps 200 --no-cpu-baseline --no-host-buffers --no-measure-traffic --extra-configs none 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ZEROCOPY=0 cfg2 value', d['value'], 'fps_total', d['fps_total'])
" >> gpurun_out/r05_bench_quick_3.txt 2>&1; cat gpurun_out/r05_bench_quick_3.txt
