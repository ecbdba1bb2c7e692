#!/bin/bash
# end-to-end check of the fused MLP under training (GPU box): with AP_DETERMINISTIC=1 (ordered weight-gradient sums) the loss / gradient-norm
# trajectory of tools/soak.py must read the same, to every printed digit, with the fused MLP (default) and with the launches it replaces
# (AP_FUSED_MLP=0).  (The parameter hash tools/soak.py prints is NOT compared: two runs of the SAME configuration differ in it -- the stem's and the
# normalisation layers' parameter gradients are still summed with atomics in that mode.)   tools/soak_fused_pair.sh [steps]
N=${1:-100}
export AP_DETERMINISTIC=1
for v in 0 1; do
  AP_FUSED_MLP=$v python tools/soak.py $N 2>/dev/null | grep -v amdgpu | grep "^0:" > /tmp/soak_f$v.txt
  echo "AP_FUSED_MLP=$v: $(cut -c1-230 /tmp/soak_f$v.txt | tr '\n' ' ')"
done
if cmp -s /tmp/soak_f0.txt /tmp/soak_f1.txt; then echo "IDENTICAL trajectories over $N steps"; else echo "TRAJECTORIES DIFFER"; fi
