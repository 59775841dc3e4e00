import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("count (no init):", torch.cuda.device_count(), "initialized:", torch.cuda.is_initialized())
import coxgraph_amd
from coxgraph_amd.capi import Layer
eng = coxgraph_amd.load_engine()
print("engine devices:", eng.device_count())
l = Layer(eng, 0.05, capacity_blocks=64)
print("layer ok; torch initialized:", torch.cuda.is_initialized())
try:
    x = torch.zeros(4).cuda()
    print("torch cuda ok", x.device)
except Exception as e:
    print("torch cuda FAILED:", repr(e)[:300])
    try:
        torch.cuda.init()
        print("after init:", torch.cuda.is_initialized(), torch.cuda.current_device())
    except Exception as e2:
        print("init failed", repr(e2)[:300])
