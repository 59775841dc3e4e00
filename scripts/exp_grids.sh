#!/bin/bash
# experiment helper (GPU box): frames/s at 5 cm for a few grid sizes of the grid-stride kernels
B="python bench.py --cpu-frames 0 --pcie-frames 0 --reg-iters 0 --other-frames 0 --no-profile-pass --no-events"
export COX_APPLY=block
run() { echo -n "$* : "; env "$@" $B 2>/dev/null | grep -o "\"value\": [0-9.]*"; }
run X=1
run COX_GRID_APPLY=1024
run COX_GRID_APPLY=2048
run COX_GRID_APPLY=768 COX_GRID_MERGE=2048 COX_GRID_TOUCH=1024
run COX_GRID_APPLY=2048 COX_GRID_MERGE=2048 COX_GRID_TOUCH=1024
run COX_STREAMS=4 COX_GRID_APPLY=1024 COX_GRID_MERGE=2048 COX_GRID_TOUCH=1024
run COX_STREAMS=4
