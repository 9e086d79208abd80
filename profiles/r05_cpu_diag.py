import os, sys, time
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
if mode == "pin":
    from slambench_amd import affinity
    print(affinity.pin_to_gpu_numa_node(0))
print("affinity cpus:", len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
os.environ["OMP_NUM_THREADS"] = sys.argv[2] if len(sys.argv) > 2 else "16"
from oracle import orc
from slambench_amd import synth
orc.build()
s = synth.make_stream(10, 640, 480)
o = orc.OracleKfusion((640, 480), (256,)*3, (4.8,)*3, s.init_pos, omp=True)
k = list(s.k)
t0 = time.perf_counter()
for f in range(10):
    t = time.perf_counter()
    o.preprocessing(s.depth_mm[f]); o.tracking(k, 1e-5, 1, f); o.integration(k, 2, 0.1, f); o.raycasting(k, 0.1, f)
    print("frame %d %.3f s" % (f, time.perf_counter() - t), flush=True)
print(mode, "total", time.perf_counter() - t0)
