#!/bin/bash
# Runs on the GPU box (gpurun): the bench line + rocprofv3 kernel traces and PMC passes that profiles/ is made from.
#   gpurun --timeout 1200 -- 'bash scripts/collect_profiles.sh r2prof'
# then here: python scripts/make_profiles.py gpurun_out/r2prof r02
# Every rocprofv3 invocation is its own run; --pmc passes carry --kernel-trace only (never combined with other trace domains).
set -e
OUT="$PWD/gpurun_out/${1:-r2prof}"
mkdir -p "$OUT"
export TMPDIR=/tmp
COMMON="--cpu-frames 0 --reg-iters 8 --other-frames 0 --pcie-frames 0 --no-profile-pass --no-ramp"
python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
echo "bench line done"
for M in merged fast; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${M}_async" -o t -- python3 bench.py --method $M $COMMON > "$OUT/${M}_async.log" 2>&1
  echo "$M async trace done"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${M}_serial" -o t -- python3 bench.py --method $M $COMMON --serial > "$OUT/${M}_serial.log" 2>&1
  echo "$M serial trace done"
  PMC="--method $M --serial --steps 60 --warmup 20 --cpu-frames 0 --reg-iters 0 --other-frames 0 --pcie-frames 0 --no-profile-pass --no-events --no-ramp"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${M}_pmc_fetch" -o p -- python3 bench.py $PMC > "$OUT/${M}_pmc_fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${M}_pmc_write" -o p -- python3 bench.py $PMC > "$OUT/${M}_pmc_write.log" 2>&1
  echo "$M pmc passes done"
done
# keep what comes back small: the per-dispatch traces of the async runs are only needed for the concurrency timeline of merged
find "$OUT" -name "*_agent_info.csv" -delete
ls -la "$OUT" "$OUT"/*/ | head -60
