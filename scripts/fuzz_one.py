"""Re-run one seed of tests/test_gpu_fuzz.py and print the per-frame statistics of both engines."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.zeros(1, device="cuda")
import coxgraph_amd
from coxgraph_amd.capi import Engine, Layer, Integrator
import test_gpu_fuzz as F
hip = coxgraph_amd.load_engine(); ora = Engine(os.path.join(ROOT, "oracle", "libcoxoracle.so"), "coxo_")
seed = int(sys.argv[1])
orig = F.compare_stats
def show(sa, sb, keys=None):
    for i, (a, b) in enumerate(zip(sa, sb)):
        print("frame", i, {k: (a[k], b[k]) for k in ("n_points", "n_valid", "n_rays", "n_updates", "n_new_blocks")})
    return orig(sa, sb, keys) if keys else orig(sa, sb)
F.compare_stats = show
try:
    print(F.run_case(seed, hip, ora))
except AssertionError as e:
    print("MISMATCH", str(e)[:600])
