#!/bin/bash
# one-frame-in-flight kernel times of the layer update's apply kernels at 1 cm, per environment
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for E in "$@"; do
  rm -rf gpurun_out/aps
  env $E timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/aps -o t -- python3 bench.py --method merged --voxel ${VOXEL:-0.01} --steps 30 --warmup 10 --serial --no-events --no-profile-pass --cpu-frames 0 --reg-iters 0 --other-frames 0 --other-config-frames 0 --pcie-frames 0 --no-ramp > gpurun_out/aps.log 2>&1 || { tail -5 gpurun_out/aps.log; exit 1; }
  python3 - "$E" <<'PY'
import csv, sys, glob
f = glob.glob("gpurun_out/aps/**/t_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = {}
for r in rows:
    n = r["Name"]
    for key in ("k_apply_block", "k_apply_wave", "k_big_classify", "k_big_tiles", "k_touch_pieces", "k_piece_expand"):
        if key in n:
            out[key] = round(float(r["AverageNs"]) / 1000, 1)
print(sys.argv[1], out, "apply total", round(sum(v for k, v in out.items() if k.startswith("k_apply") or k.startswith("k_big")), 1), flush=True)
PY
done
rm -rf gpurun_out/aps
