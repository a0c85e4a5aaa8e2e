#!/bin/bash
# Timing-only ablations of the run loop, built and run on the GPU box.
cd "$GRAFT_REPO_ROOT/matrixfactorizationsgd.java_amd/csrc"
H=/opt/rocm/bin/hipcc
$H -O3 -std=c++17 -fPIC -c schedule.cpp -o schedule.o 2>/dev/null
$H -O3 -std=c++17 -fPIC -c capi.cpp -o capi.o 2>/dev/null
for e in $EXPS; do
  $H -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DMFSGD_EXP=$e -c kernels.hip -o kernels.o 2>/dev/null && $H -shared -o ../lib/libmfsgd.so kernels.o schedule.o capi.o -pthread
  echo "== EXP $e"; (cd "$GRAFT_REPO_ROOT" && timeout -k 10 200 python tools/phase_stamps.py 2>&1 | tail -3 | head -2)
done
