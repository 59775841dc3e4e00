#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_golden.npz from the CPU oracle.

The reference (mfkiwl/coxgraph) holds NO golden vectors for this path (it has no tests at all and its hot-path
arithmetic lives in un-vendored forks), so these fixtures are NOT reference outputs: they freeze the behaviour of
this repository's own oracle (parity unpinned, see oracle/cox_oracle.hpp) so that (a) an accidental change of the
oracle shows up in the CPU suite and (b) the GPU suite has a second, oracle-independent-at-run-time target.
Inputs are small and synthetic; every array is data (inputs + expected outputs), no code.

    python tests/golden/make_golden.py
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from coxgraph_amd import synth  # noqa: E402
from coxgraph_amd.capi import Engine, Layer, Integrator, RegPoints, Registration  # noqa: E402


def raycast(eng, origin, point, clearing, carving, max_len, inv, trunc, from_origin):
    cap = 4096
    out = (C.c_int64 * (3 * cap))()
    n = C.c_uint64()
    eng.fn("raycast")((C.c_float * 3)(*origin), (C.c_float * 3)(*point), clearing, carving, C.c_float(max_len), C.c_float(inv), C.c_float(trunc),
                      from_origin, out, C.c_uint64(cap), C.byref(n))
    return np.array(out[:3 * n.value], np.int64).reshape(-1, 3)


def build(eng):
    g = {}
    # 1. ray paths (generic, clearing, no carving, backwards, axis-aligned quirk)
    rng = np.random.default_rng(42)
    rays = []
    for i in range(12):
        o = rng.uniform(-1, 1, 3).astype(np.float32)
        p = (o + rng.uniform(-3, 3, 3)).astype(np.float32)
        rays.append((o, p, int(i % 4 == 1), int(i % 3 != 2), 5.0, 10.0 if i % 2 else 20.0, 0.3 if i % 2 else 0.15, int(i % 5 != 4)))
    rays.append((np.array([0.05, 0.05, 0.05], np.float32), np.array([1.05, 0.05, 0.05], np.float32), 0, 1, 5.0, 10.0, 0.3, 1))
    g["ray_params"] = np.array([[*o, *p, c, cv, ml, inv, tr, fo] for (o, p, c, cv, ml, inv, tr, fo) in rays], np.float64)
    paths = [raycast(eng, *r) for r in rays]
    g["ray_path_lengths"] = np.array([len(p) for p in paths], np.int64)
    g["ray_paths"] = np.concatenate(paths, axis=0)
    # 2. small integration: every 16th pixel of frames 0 and 7, 10 cm, merged and simple
    for method in ("merged", "simple", "fast"):
        cfg = eng.default_config(integrator_threads=1, **synth.integrator_overrides(0.10))  # fast: only the single-threaded result is defined
        layer = Layer(eng, 0.10, capacity_blocks=4096)
        integ = Integrator(eng, layer, cfg, method)
        stats = []
        for t in (0, 7):
            T, pts, rgba, _ = synth.make_frame(t)
            integ.integrate_points(T, pts[::16], rgba[::16])
            s = integ.last_stats()
            stats.append([s[k] for k in ("n_valid", "n_rays", "n_updates", "n_touched_voxels", "n_new_blocks")])
        idx, vox = layer.download()
        g[f"{method}_block_idx"] = idx
        g[f"{method}_stats"] = np.array(stats, np.int64)
        g[f"{method}_words_sha256"] = np.frombuffer(hashlib.sha256(vox.tobytes()).digest(), np.uint8)
        # a sparse sample of voxels in full (block, linear index, 3 words)
        nz = np.argwhere(vox[..., 1] != 0)
        pick = nz[:: max(1, len(nz) // 400)]
        g[f"{method}_sample_where"] = pick.astype(np.int32)
        g[f"{method}_sample_words"] = vox[pick[:, 0], pick[:, 1]]
        if method == "merged":
            reading = (idx, vox)
    # 3. registration against the merged layer
    idx, vox = reading
    from test_gpu_registration import registration_points  # noqa: E402
    layer = Layer(eng, 0.10)
    layer.upload(idx, vox)
    pts = registration_points(layer, 0.3, min_weight=0.5)[::5]
    g["reg_points"] = pts
    reg = Registration(eng, RegPoints(eng, pts), layer)
    pr, pd = np.array([0.01, -0.02, 0.0, 0.01]), np.array([0.04, -0.03, 0.02, np.radians(1.0)])
    r, jf, jr = reg.evaluate(pr, pd)
    H, b, cost, nc = reg.normal_eq(pr, pd)
    g["reg_pose_ref"], g["reg_pose_read"] = pr, pd
    g["reg_residuals"], g["reg_jac_ref"], g["reg_jac_read"] = r, jf, jr
    g["reg_H"], g["reg_b"], g["reg_cost_ncorr"] = H, b, np.array([cost, nc])
    # 4. recover mode: mesh with history -> per-pose clouds -> layer (synthetic wall mesh, seed 11)
    from coxgraph_amd.capi import MeshMsg, MeshConverter  # noqa: E402
    m = synth.make_wall_mesh(seed=11, n_frames=8)
    msg = MeshMsg(m["block_edge_length"], m["blocks"], m["trajectory"])
    conv = MeshConverter(eng, 0.07)
    conv.set_mesh(msg)
    ok, rec, rgb = conv.convert()
    clouds = conv.pose_clouds()
    g["mesh_recovered_sha256"] = np.frombuffer(hashlib.sha256(rec.tobytes() + rgb.tobytes()).digest(), np.uint8)
    g["mesh_cloud_sizes"] = np.array([len(c[1]) for c in clouds], np.int64)
    g["mesh_clouds_sha256"] = np.frombuffer(hashlib.sha256(b"".join(c[1].tobytes() + c[2].tobytes() for c in clouds)).digest(), np.uint8)
    g["mesh_cloud3_head"] = clouds[3][1][:16].copy()
    cfg = eng.default_config(**synth.integrator_overrides(0.05))
    layer = Layer(eng, 0.05, capacity_blocks=4096)
    integ = Integrator(eng, layer, cfg, "merged")
    n_rec, n_int = MeshConverter(eng, 0.07).process_mesh(integ, msg)
    idx, vox = layer.download()
    g["mesh_process_counts"] = np.array([n_rec, n_int, len(idx)], np.int64)
    g["mesh_layer_sha256"] = np.frombuffer(hashlib.sha256(idx.tobytes() + vox.tobytes()).digest(), np.uint8)
    return g


if __name__ == "__main__":
    eng = Engine(os.path.join(ROOT, "oracle", "libcoxoracle.so"), "coxo_")
    out = os.path.join(ROOT, "tests", "golden", "oracle_golden.npz")
    np.savez_compressed(out, **build(eng))
    print("wrote", out, os.path.getsize(out), "bytes")
