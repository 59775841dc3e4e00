import sys, time, numpy as np
sys.path.insert(0, '.')
import torch; torch.zeros(1, device='cuda')
import coxgraph_amd
from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, Integrator, RegPoints, Registration
eng = coxgraph_amd.load_engine()
cfg = eng.default_config(**synth.integrator_overrides(0.05))
layer = Layer(eng, 0.05, capacity_blocks=32768)
integ = Integrator(eng, layer, cfg, "merged")
for t in range(20):
    T, pts, rgba, _ = synth.make_frame(t)
    integ.integrate_points(T, pts, rgba)
ref = RegPoints.from_layer(eng, layer, 1.0, 0.15)
rng = np.random.default_rng(7)
sidx = rng.integers(0, ref.n, size=int(0.3 * ref.n)).astype(np.uint32)
g = Registration(eng, ref, layer)
pr, pd = np.zeros(4), np.array([0.05, -0.03, 0.02, 0.017])
def timeit(name, fn, n=300):
    fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    print(f"{name:28s} {n / (time.perf_counter() - t0):10.0f} per s")

g2 = Registration(eng, ref, layer)
g2.set_samples(sidx)
batch = [Registration(eng, ref, layer) for _ in range(8)]
for b in batch: b.set_samples(sidx)
def all8():
    for b in batch: b.normal_eq_begin(pr, pd)
    for b in batch: b.normal_eq_finish()
for rep in range(3):
    timeit("explicit idx", lambda: g.normal_eq(pr, pd, sidx))
    timeit("stored idx", lambda: g2.normal_eq(pr, pd))
    timeit("8 in flight (x8 per call)", all8, 100)
