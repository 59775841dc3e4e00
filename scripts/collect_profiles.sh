#!/bin/bash
# Runs on the GPU box (gpurun): the bench line + rocprofv3 kernel traces and PMC passes that profiles/ is made from.
#   gpurun --timeout 1200 -- 'bash scripts/collect_profiles.sh r3prof'
# then here: python scripts/make_profiles.py gpurun_out/r3prof r03
# Every rocprofv3 invocation is its own run; --pmc passes carry --kernel-trace only (never combined with other trace domains).
set -e
OUT="$PWD/gpurun_out/${1:-r2prof}"
mkdir -p "$OUT"
export TMPDIR=/tmp
COMMON="--cpu-frames 0 --reg-iters 0 --other-frames 0 --other-config-frames 0 --pcie-frames 0 --no-profile-pass --no-ramp"
python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
echo "bench line done"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_line_driver.json" 2> "$OUT/bench_line_driver.err"
echo "bench line at the driver's settings done"
for M in merged fast; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${M}_async" -o t -- python3 bench.py --method $M $COMMON > "$OUT/${M}_async.log" 2>&1
  echo "$M async trace done"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${M}_serial" -o t -- python3 bench.py --method $M $COMMON --serial > "$OUT/${M}_serial.log" 2>&1
  echo "$M serial trace done"
  PMC="--method $M --serial --steps 60 --warmup 20 --cpu-frames 0 --reg-iters 0 --other-frames 0 --other-config-frames 0 --pcie-frames 0 --no-profile-pass --no-events --no-ramp"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${M}_pmc_fetch" -o p -- python3 bench.py $PMC > "$OUT/${M}_pmc_fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${M}_pmc_write" -o p -- python3 bench.py $PMC > "$OUT/${M}_pmc_write.log" 2>&1
  echo "$M pmc passes done"
done
# fine voxels: bench line + one-frame-in-flight kernel statistics
FINE="--cpu-frames 0 --reg-iters 0 --other-frames 0 --other-config-frames 0 --pcie-frames 0 --no-ramp"
LINE="--cpu-frames 0 --reg-iters 0 --other-frames 40 --pcie-frames 0"
python3 bench.py --voxel 0.10 --steps 100 --warmup 10 $LINE > "$OUT/bench_line_10cm.json" 2> "$OUT/bench_line_10cm.err"
python3 bench.py --voxel 0.02 --steps 60 --warmup 10 $LINE > "$OUT/bench_line_2cm.json" 2> "$OUT/bench_line_2cm.err"
python3 bench.py --voxel 0.01 --steps 40 --warmup 10 $LINE > "$OUT/bench_line_1cm.json" 2> "$OUT/bench_line_1cm.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/merged_serial_2cm" -o t -- python3 bench.py --method merged --voxel 0.02 --steps 40 --warmup 10 --serial --no-events --no-profile-pass $FINE > "$OUT/merged_serial_2cm.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/merged_serial_1cm" -o t -- python3 bench.py --method merged --voxel 0.01 --steps 30 --warmup 10 --serial --no-events --no-profile-pass $FINE > "$OUT/merged_serial_1cm.log" 2>&1
rm -f "$OUT"/merged_serial_?cm/*kernel_trace.csv
echo "fine voxel runs done"
# where the kernel classes of neighbouring frames sit in time, without a profiler attached (COX_TIMELINE: HIP events)
COX_TIMELINE="$OUT/stage_timeline_5cm.txt" python3 bench.py --steps 60 --warmup 20 --cpu-frames 0 --reg-iters 0 --other-frames 0 --other-config-frames 0 --pcie-frames 0 --no-ramp > "$OUT/stage_timeline_5cm.log" 2>&1
# keep what comes back small: the per-dispatch traces of the async runs are only needed for the concurrency timeline of merged
find "$OUT" -name "*_agent_info.csv" -delete
ls -la "$OUT" "$OUT"/*/ | head -60
