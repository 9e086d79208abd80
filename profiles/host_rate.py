import sys, time, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from slambench_amd import kfusion, synth
s = synth.cached_stream(120)
d = torch.from_numpy(stream := s.depth_mm.view(np.int16)).cuda()
k = list(s.k)
kf = kfusion.Kfusion((640,480),(256,)*3,(4.8,)*3,s.init_pos)
base = d.data_ptr(); fb = 640*480*2
for f in range(20): kf.compute_frame(base+f*fb,(640,480),k,2,1,1e-5,0.1,f)
kf.synchronise()
t0=time.perf_counter()
for f in range(20,120): kf.compute_frame(base+f*fb,(640,480),k,2,1,1e-5,0.1,f)
t1=time.perf_counter(); kf.synchronise(); t2=time.perf_counter()
print("host enqueue per frame %.1f us, total per frame %.1f us" % ((t1-t0)/100*1e6,(t2-t0)/100*1e6))

