"""Run one fuzz seed frame by frame on fresh layers and print where the two engines' layers differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.zeros(1, device="cuda")
import coxgraph_amd
from coxgraph_amd.capi import Engine, Layer, Integrator, words_to_fields
import test_gpu_fuzz as F
hip = coxgraph_amd.load_engine(); ora = Engine(os.path.join(ROOT, "oracle", "libcoxoracle.so"), "coxo_")
seed = int(sys.argv[1]); only = int(sys.argv[2]) if len(sys.argv) > 2 else None
# regenerate the case exactly as run_case does
rng = np.random.default_rng(seed)
voxel = float(rng.choice([0.03, 0.05, 0.08, 0.1, 0.2])); method = str(rng.choice(["merged", "simple", "fast"])); max_ray = float(rng.choice([1.5, 3.0, 5.0]))
ov = dict(default_truncation_distance=float(rng.choice([2, 3, 4])) * voxel, min_ray_length_m=float(rng.choice([0.05, 0.2, 0.5])),
          max_ray_length_m=max_ray, use_const_weight=int(rng.integers(0, 2)), allow_clear=int(rng.integers(0, 2)),
          voxel_carving_enabled=int(rng.integers(0, 2)), use_weight_dropoff=int(rng.integers(0, 2)),
          use_sparsity_compensation_factor=int(rng.integers(0, 2)), sparsity_compensation_factor=float(rng.choice([1.0, 10.0])),
          max_weight=float(rng.choice([50.0, 10000.0])), enable_anti_grazing=int(rng.integers(0, 2)), integrator_threads=1,
          max_consecutive_ray_collisions=int(rng.integers(0, 5)), clear_checks_every_n_frames=int(rng.integers(1, 4)),
          start_voxel_subsampling_factor=float(rng.choice([1.0, 2.0, 3.0])))
frames = []
for _ in range(int(rng.integers(2, 6))):
    n = int(rng.choice([1, 63, 64, 65, 700, 1024, 1025, 5000, 20000]))
    frames.append((F.random_pose(rng), *F.random_cloud(rng, n, max_ray), bool(rng.random() < 0.15)))
print(method, voxel, ov)
for fi, (T, p, c, fs) in enumerate(frames):
    if only is not None and fi != only:
        continue
    res = []
    for eng in (hip, ora):
        layer = Layer(eng, voxel, capacity_blocks=60000)
        integ = Integrator(eng, layer, eng.default_config(**ov), method)
        integ.integrate_points(T, p, c if seed % 5 else None, freespace=fs)
        res.append((layer.download(), integ.last_stats()))
    ((ia, va), sa), ((ib, vb), sb) = res
    print("frame", fi, "alone: stats", {k: (sa[k], sb[k]) for k in ("n_valid", "n_rays", "n_updates", "n_new_blocks", "max_bundle_points", "max_voxel_updates")})
    sa_ = {tuple(x) for x in ia}; sb_ = {tuple(x) for x in ib}
    print("  blocks only hip:", sorted(sa_ - sb_)[:5], "only oracle:", sorted(sb_ - sa_)[:5])
    common = sorted(sa_ & sb_)
    da = {tuple(k): v for k, v in zip(ia, va)}; db = {tuple(k): v for k, v in zip(ib, vb)}
    nd = 0
    for k in common:
        wa = da[k][:, 1].view(np.float32); wb = db[k][:, 1].view(np.float32)
        diff = np.nonzero(wa != wb)[0]
        if len(diff) and nd < 6:
            i = diff[0]
            print("  block", k, "voxels differing", len(diff), "first lin", i, (i % 16, (i // 16) % 16, i // 256), "w hip/oracle", wa[i], wb[i],
                  "d", da[k][i, 0].view(np.float32), db[k][i, 0].view(np.float32))
            nd += 1
