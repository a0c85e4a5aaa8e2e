mkdir -p gpurun_out/r03f
timeout -k 10 120 python - <<'PY' > gpurun_out/r03f/gen_small.log 2>&1
import time, numpy as np, mfsgd_amd as mf
t=time.time(); w=mf.synth.workload("cfg4_powerlaw", 0.02, generator="device"); print("device gen", time.time()-t, w["nnz"], w["U"], w["I"])
du=np.bincount(w["u"],minlength=w["U"]); di=np.bincount(w["i"],minlength=w["I"]); print("max deg", du.max(), di.max(), "dups", w["nnz"]-np.unique(w["u"].astype(np.int64)*w["I"]+w["i"]).size)
t=time.time(); w2=mf.synth.workload("cfg4_powerlaw", 0.02); print("host gen", time.time()-t)
du=np.bincount(w2["u"],minlength=w2["U"]); di=np.bincount(w2["i"],minlength=w2["I"]); print("max deg host", du.max(), di.max(), float(w["r"].mean()), float(w2["r"].mean()))
PY
cat gpurun_out/r03f/gen_small.log
timeout -k 10 1000 python tests/gpu_large_extra.py full > gpurun_out/r03f/large_full.log 2>&1; echo "rc=$?" >> gpurun_out/r03f/large_full.log
grep -v amdgpu.ids gpurun_out/r03f/large_full.log | tail -25
