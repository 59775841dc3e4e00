#!/bin/bash
# gpurun -- 'bash scripts/fine_sweep.sh "ENV=.. ENV=.." ...': fine-voxel bench lines (merged, 1 cm and 2 cm) per environment
LINE="--cpu-frames 0 --reg-iters 0 --other-frames 0 --other-config-frames 0 --pcie-frames 0"
for E in "$@"; do
  for V in 0.01 0.02; do
    env $E timeout -k 10 200 python3 bench.py --voxel $V --steps 40 --warmup 10 $LINE > gpurun_out/fine.json 2>gpurun_out/fine.err || { tail -5 gpurun_out/fine.err; exit 1; }
    python3 - "$E" $V <<'PY'
import json, sys
d = json.load(open("gpurun_out/fine.json"))
c = d["roofline"].get("class_ms_per_frame") or {}
print(sys.argv[1], "voxel", sys.argv[2], round(d["value"]), "frames/s;", {k: round(v, 3) for k, v in c.items()}, flush=True)
PY
  done
done
