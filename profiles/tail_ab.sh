for envs in "KFH_ICP_TAIL_HEAD=0" "KFH_ICP_TAIL_HEAD=4" "KFH_ICP_TAIL_HEAD=3" "KFH_ICP_TAIL_HEAD=0 KFH_DEBUG_ICP_THRESHOLD=0" "KFH_ICP_TAIL_HEAD=4 KFH_DEBUG_ICP_THRESHOLD=0" "KFH_ICP_TAIL_HEAD=2 KFH_DEBUG_ICP_THRESHOLD=0"; do
  env $envs timeout -k 10 120 python bench.py --steps 300 --no-cpu-baseline --no-total --profile-every 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-60s fps %.1f  ms %.4f its %.2f ate %.9f %.9f' % ('$envs', d['value'], d['ms_per_step'], d['config']['icp_iterations_per_frame'], d['ate_m']['mean'], d['ate_m']['max']))
" || exit 1
done
