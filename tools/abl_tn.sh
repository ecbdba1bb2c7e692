#!/bin/bash
# Ablation builds of the weight-gradient kernel (csrc/gemm_tn8p.h, T8_ABL bits) as separate libraries under tools/abl/; the product
# library is untouched.  usage (CPU box): tools/abl_tn.sh 1 2 4 8 ...   then on the GPU box: AP_LIB_PATH=tools/abl/lib_tn_abl8.so python tools/bench_tn.py
set -e
cd "$(dirname "$0")/../autoprog_amd/csrc"
mkdir -p ../../tools/abl
OBJS=$(ls *.o | grep -v '^gemm.o$')
for x in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -ffp-contract=fast -DT8_ABL=$x -c gemm.hip -o /tmp/gemm_abl_$x.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gemm_abl_$x.o -o ../../tools/abl/lib_tn_abl$x.so
done
