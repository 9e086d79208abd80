#!/bin/bash
# cycle stamps of one executed level-0 ICP launch (block 0 / thread 0), slot schedule against step launches
python - <<'PY'
import os, sys, subprocess
for name, env, it in (("slots (KFH_ICP_GENERIC=0), launch 10", {"KFH_ICP_GENERIC": "0"}, "10"), ("steps (KFH_ICP_STEPS=11), launch 8", {"KFH_ICP_STEPS": "11"}, "8")):
    out = "/tmp/icp_stamp_%s.bin" % it
    e = dict(os.environ, KFH_ICP_TRACE=out, KFH_ICP_TRACE_IT=it, **env)
    subprocess.run([sys.executable, "bench.py", "--steps", "260", "--warmup", "20", "--no-cpu-baseline", "--no-total", "--no-host-buffers", "--extra-configs", "none", "--profile-every", "0"],
                   env=e, capture_output=True)
    print("==", name)
    print(subprocess.run([sys.executable, "profiles/icp_trace.py", out], capture_output=True, text=True).stdout)
PY
