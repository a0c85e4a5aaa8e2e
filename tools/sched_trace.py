import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import mfsgd_amd as mf
name, scale, gen = sys.argv[1], float(sys.argv[2]), sys.argv[3]
w = mf.synth.workload(name, scale, generator=gen)
print(name, scale, w["nnz"], flush=True)
os.environ["MFSGD_SCHED_TRACE"] = "1"
for rep in range(2):
    with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 3, host_threads=16) as m:
        t = time.time(); m.set_ratings(w["u"], w["i"], w["r"]); print(f"== set_ratings rep {rep}: {time.time()-t:.3f} s", m.schedule_info()["split_cells"], flush=True)
