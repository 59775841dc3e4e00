#!/bin/bash
# experiment helper (GPU box): bench the apply class with variant libraries  coxgraph_amd/lib/var_*.so
cp coxgraph_amd/lib/libcoxgraph_hip.so /tmp/orig.so
for v in "$@"; do
  cp coxgraph_amd/lib/var_$v.so coxgraph_amd/lib/libcoxgraph_hip.so
  python bench.py --cpu-frames 0 --pcie-frames 0 --reg-iters 0 --other-frames 0 --voxel ${VOX:-0.01} --steps 40 --warmup 10 > gpurun_out/exp_$v.log 2>&1 || echo "variant $v failed"
done
cp /tmp/orig.so coxgraph_amd/lib/libcoxgraph_hip.so
