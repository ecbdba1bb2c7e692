#!/bin/bash
# Ablation builds of ONE source of the library as separate libraries under tools/abl/ (the product library is untouched).
# usage (CPU box): tools/abl.sh outlook OLK_ABL 1 2 4 ...   ->  tools/abl/lib_outlook_OLK_ABL_<n>.so
# then on the GPU box:  AP_LIB_PATH=tools/abl/lib_outlook_OLK_ABL_4.so python tools/bench_outlook.py      (results are WRONG by design)
set -e
SRC=$1; MACRO=$2; shift 2
cd "$(dirname "$0")/../autoprog_amd/csrc"
mkdir -p ../../tools/abl
OBJS=$(ls *.o | grep -v "^$SRC.o$")
for x in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -ffp-contract=fast -D$MACRO=$x -c $SRC.hip -o /tmp/${SRC}_abl_$x.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/${SRC}_abl_$x.o -o ../../tools/abl/lib_${SRC}_${MACRO}_$x.so
done
