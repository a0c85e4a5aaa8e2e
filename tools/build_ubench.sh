#!/bin/bash
# Builds the microbenchmarks of the rating loops into tools/bin/ (git-ignored; they travel to the GPU box with gpurun).
#   tools/build_ubench.sh && gpurun -- 'for L in 16 32 64; do tools/bin/ub3_$L; done'
set -e
cd "$(dirname "$0")"
mkdir -p bin
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for L in 16 32 64; do
    $HIPCC --offload-arch=gfx950 -O2 -ffp-contract=off -DLG=$L -Wno-unused-value ubench3.hip -o bin/ub3_$L
done
# L = 64 with the two v_permlane*_swap levels instead of the row_bcast reduction (what round 1 ran)
$HIPCC --offload-arch=gfx950 -O2 -ffp-contract=off -DLG=64 -DOLD64 -Wno-unused-value ubench3.hip -o bin/ub3_64old
$HIPCC --offload-arch=gfx950 -O2 ubench.hip -o bin/ubench
$HIPCC --offload-arch=gfx950 -O2 -Wno-unused-value ubench4.hip -o bin/ub4
