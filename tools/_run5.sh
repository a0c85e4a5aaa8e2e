mkdir -p gpurun_out/r03e
for lib in asm gencpp; do
  if [ $lib = gencpp ]; then export MFSGD_LIBRARY=$PWD/matrixfactorizationsgd.java_amd/lib/libmfsgd_gencpp.so; fi
  timeout -k 10 200 python tools/phase_profile.py cfg2_uniform 1.0 > gpurun_out/r03e/pp_uniform_$lib.log 2>&1
  timeout -k 10 200 python tools/phase_profile.py cfg2_uniform 1.0 256 2 > gpurun_out/r03e/pp_uniform_W2_$lib.log 2>&1
  MFSGD_EMU=8 timeout -k 10 200 python tools/phase_profile.py cfg2_ml20m 1.0 > gpurun_out/r03e/pp_emu8_$lib.log 2>&1
done
tail -n 14 gpurun_out/r03e/*.log
