"""How much of one GPU does one client's host thread use?  K clients (own layer, own integrator, own host thread) on one GPU."""
import sys, time, threading, numpy as np
sys.path.insert(0, '.')
import torch; torch.zeros(1, device='cuda')
import coxgraph_amd
from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, Integrator
eng = coxgraph_amd.load_engine()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = 200
frames = []
for t in range(N):
    T, pts, rgba, _ = synth.make_frame(t)
    frames.append((T, torch.from_numpy(pts).cuda(), torch.from_numpy(rgba).cuda(), pts.shape[0]))
torch.cuda.synchronize()
cfg = eng.default_config(**synth.integrator_overrides(0.05))
clients = []
for k in range(K):
    layer = Layer(eng, 0.05, capacity_blocks=32768)
    clients.append((layer, Integrator(eng, layer, cfg, "merged")))
def run(integ, n0, n1):
    for i in range(n0, n1):
        T, xyz, rgba, n = frames[i]
        integ.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), n)
    integ.sync()
for _, integ in clients:
    run(integ, 0, 20)
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(integ, 20, N)) for _, integ in clients]
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
print(f"{K} clients: {K * (N - 20) / dt:.0f} frames/s aggregate, {(N - 20) / dt:.0f} per client")
