import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import coxgraph_amd
from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, Integrator
eng = coxgraph_amd.load_engine()
for voxel, chunk in ((0.01, 4096), (0.01, 1024), (0.02, 4096)):
    os.environ["COX_BIG_CHUNK"] = str(chunk)
    cfg = eng.default_config(**synth.integrator_overrides(voxel))
    I = Integrator(eng, Layer(eng, voxel, capacity_blocks=60000), cfg, "merged")
    for t in range(12):
        T, p, c, _ = synth.make_frame(t)
        I.integrate_points(T, p, c)
    st = I.last_stats(); us = I.update_stats()
    print(f"voxel {voxel} chunk {chunk}: updates {st['n_updates']}, touched blocks {st['n_touched_blocks']}, tiles split {us['split_tiles']}, chunks {us['chunks']} (~{us['chunks'] * chunk / max(st['n_updates'], 1) * 100:.0f} % of the records at most)", flush=True)
    del I
