"""Submap hand-over between GPUs + inter-robot registration (BASELINE configs[2] / configs[4]).

The reference's server pulls whole submaps from its clients as ROS messages (coxgraph/src/server/client_handler.cpp:82-104,
coxgraph_server.cpp:253-258) and registers client a's submap against client b's (coxgraph_server.cpp:396-476).  Here the
hand-over is a GPU-to-GPU copy (`cox_layer_clone_to_device`, `cox_regpoints_clone_to_device`; on a one-GPU box the peer copy
degenerates to a device copy, so the path is testable here) or, between processes, device-resident wire arrays
(`cox_layer_export_dev` -> all-gather -> `cox_layer_upload_dev`).  Every constraint (a, b) below uses CLIENT a's registration
points against CLIENT b's layer -- never a map against itself -- and is checked against the oracle.
"""
import numpy as np
import pytest

from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, Integrator, RegPoints, Registration
from util import run_frames, compare_layers

pytestmark = pytest.mark.gpu

VOXEL = 0.05
N_CLIENTS = 3
RING = 12  # clients sit 30 degrees apart on the camera circle, so neighbouring clients see overlapping parts of the room


@pytest.fixture(scope="module")
def client_submaps(hip, oracle):
    """One submap per client (frames 0..70 step 10 of its own trajectory, every 3rd pixel, 5 cm), fused on the GPU by the
    HIP engine and on the CPU by the oracle."""
    out = {}
    for name, eng in (("hip", hip), ("oracle", oracle)):
        layers = []
        for c in range(N_CLIENTS):
            layer, _, _ = run_frames(eng, method="merged", voxel=VOXEL, frames=range(0, 80, 10), subsample=3, client=c, n_clients=RING,
                                     capacity_blocks=4096)
            layers.append(layer)
        out[name] = layers
    return out


def test_clone_is_the_same_submap(hip, client_submaps):
    src = client_submaps["hip"][1]
    dst = src.clone_to_device(0)
    rep = compare_layers(dst, src)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0 and rep["blocks"] > 50
    # the clone is a layer of its own: it keeps working after the source is cleared, and can be integrated into
    cfg = hip.default_config(**synth.integrator_overrides(VOXEL))
    n0 = dst.stats()[0]
    T, pts, rgba, _ = synth.make_frame(300, client=1, n_clients=RING)
    Integrator(hip, dst, cfg, "merged").integrate_points(T, pts[::5], rgba[::5])
    assert dst.stats()[0] >= n0
    assert src.stats()[0] == n0


def test_all_pairs_inter_robot_registration_after_hand_over(hip, oracle, client_submaps):
    """configs[4]'s per-GPU workload: all pairs (a, b), a != b; client a's registration points and client b's layer both
    arrive by clone on the evaluating device; residuals / Jacobians / normal equations equal the oracle's."""
    trunc = 3 * VOXEL
    lh, lo = client_submaps["hip"], client_submaps["oracle"]
    ref_h = [RegPoints.from_layer(hip, l, 1.0, trunc) for l in lh]
    ref_o = [RegPoints.from_layer(oracle, l, 1.0, trunc) for l in lo]
    rng = np.random.default_rng(3)
    n_checked = 0
    for a in range(N_CLIENTS):
        pts_a = ref_h[a].clone_to_device(0)
        assert pts_a.n == ref_o[a].n > 3000
        assert np.array_equal(pts_a.download(), ref_o[a].download())
        for b in range(N_CLIENTS):
            if a == b:
                continue
            layer_b = lh[b].clone_to_device(0)
            gh = Registration(hip, pts_a, layer_b, 0.0)
            go = Registration(oracle, ref_o[a], lo[b], 0.0)
            n_res = int(0.3 * pts_a.n)
            idx = rng.integers(0, pts_a.n, size=n_res).astype(np.uint32)
            pr = np.array([0.0, 0.0, 0.0, 0.0])
            pd = np.array([0.05, -0.03, 0.02, np.radians(1.0)])
            rh, jfh, jrh = gh.evaluate(pr, pd, idx)
            ro, jfo, jro = go.evaluate(pr, pd, idx)
            assert np.count_nonzero(ro) > 100, "clients' submaps must overlap (same room, outward-looking circle)"
            assert np.max(np.abs(rh - ro)) <= 1e-4
            assert np.array_equal(rh, ro) and np.array_equal(jfh, jfo) and np.array_equal(jrh, jro)
            Hh, bh, ch, nh = gh.normal_eq(pr, pd, idx)
            Ho, bo, co, no = go.normal_eq(pr, pd, idx)
            assert nh == no
            assert np.max(np.abs(Hh - Ho)) <= 1e-6 * max(1.0, float(np.max(np.abs(Ho))))
            assert np.max(np.abs(bh - bo)) <= 1e-6 * max(1.0, float(np.max(np.abs(bo))))
            assert abs(ch - co) <= 1e-6 * max(1.0, co)
            n_checked += 1
    assert n_checked == N_CLIENTS * (N_CLIENTS - 1)


def test_device_resident_wire_hand_over(hip, client_submaps):
    """The multi-process flavour: export to device buffers (what a rank gives the all-gather), upload from device buffers."""
    import torch
    src = client_submaps["hip"][2]
    nb = src.n_blocks()
    idx = torch.empty((nb, 3), dtype=torch.int32, device="cuda")
    vox = torch.empty((nb, 4096, 3), dtype=torch.int32, device="cuda")
    got = src.export_dev(idx.data_ptr(), vox.data_ptr(), nb)
    assert got == nb
    hidx, hvox = src.download()
    assert np.array_equal(idx.cpu().numpy(), hidx)
    assert np.array_equal(vox.cpu().numpy().view(np.uint32), hvox)
    dst = Layer(hip, VOXEL, capacity_blocks=16)  # smaller than the message: the upload grows the pool like Layer does
    dst.upload_dev(idx.data_ptr(), vox.data_ptr(), nb)
    rep = compare_layers(dst, src)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0
    assert dst.capacity() >= nb
    # the point set travels the same way
    trunc = 3 * VOXEL
    ref = RegPoints.from_layer(hip, src, 1.0, trunc)
    ptr, n = ref.data_ptr()
    assert n == ref.n
    assert ptr != 0
    t = torch.from_numpy(ref.download()).cuda()
    ref2 = RegPoints.from_device(hip, t.data_ptr(), n)
    assert np.array_equal(ref2.download(), ref.download())


def test_caller_stream_ordering(hip):
    """inputs produced on the caller's stream right before the call, dropped right after it -- with the
    producer stream registered the result equals the synchronous path."""
    import torch
    cfg = hip.default_config(**synth.integrator_overrides(0.10))
    frames = [synth.make_frame(t) for t in range(0, 48, 4)]
    want = Layer(hip, 0.10, capacity_blocks=4096)
    iw = Integrator(hip, want, cfg, "merged")
    for T, p, c, _ in frames:
        iw.integrate_points(T, p[::4], c[::4])
    got = Layer(hip, 0.10, capacity_blocks=4096)
    ig = Integrator(hip, got, cfg, "merged")
    side = torch.cuda.Stream()
    ig.set_input_stream(side.cuda_stream)
    pinned = [(torch.from_numpy(np.ascontiguousarray(p[::4])).pin_memory(), torch.from_numpy(np.ascontiguousarray(c[::4])).pin_memory()) for _, p, c, _ in frames]
    with torch.cuda.stream(side):
        for (T, _, _, _), (hp, hc) in zip(frames, pinned):
            xyz = hp.to("cuda", non_blocking=True)   # async H2D on the side stream, no synchronisation before the call
            rgba = hc.to("cuda", non_blocking=True)
            xyz = xyz * 1.0                          # a kernel of the producer, still in flight when the engine is called
            ig.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), xyz.shape[0])
            del xyz, rgba                            # the caching allocator may hand the memory to the next iteration
    ig.sync()
    rep = compare_layers(got, want)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep


def test_registration_waits_for_frames_in_flight(hip, oracle):
    """cox_reg_* reads the layer on its own stream: it must see every frame enqueued before it."""
    import torch
    cfg = hip.default_config(**synth.integrator_overrides(0.10))
    lo, _, _ = run_frames(oracle, method="merged", voxel=0.10, frames=range(0, 40, 4), subsample=4)
    layer = Layer(hip, 0.10, capacity_blocks=4096)
    integ = Integrator(hip, layer, cfg, "merged")
    dev = []
    for t in range(0, 40, 4):
        T, p, c, _ = synth.make_frame(t)
        dev.append((T, torch.from_numpy(np.ascontiguousarray(p[::4])).cuda(), torch.from_numpy(np.ascontiguousarray(c[::4])).cuda()))
    torch.cuda.synchronize()
    pts = RegPoints.from_layer(oracle, lo, 1.0, 0.3).download()
    ref = RegPoints(hip, pts)
    g = Registration(hip, ref, layer, 0.0)
    for T, x, c in dev:
        integ.integrate_points_dev(T, x.data_ptr(), c.data_ptr(), x.shape[0])
    rh, _, _ = g.evaluate(np.zeros(4), np.array([0.02, 0.01, 0.0, 0.01]))   # no integ.sync() in between
    integ.sync()
    go = Registration(oracle, RegPoints(oracle, pts), lo, 0.0)
    ro, _, _ = go.evaluate(np.zeros(4), np.array([0.02, 0.01, 0.0, 0.01]))
    assert np.array_equal(rh, ro)


def test_rccl_paths_on_one_rank():
    """The RCCL branches of the multi-GPU server path (bench.py --gpus N over backend "nccl": all-gather of device-resident wire
    arrays, cox_layer_upload_dev / cox_regpoints_create_dev on the receiving side, posegraph.py's packed all-reduce on the
    device) executed on real hardware -- with one rank, which is what a one-GPU box allows; tests/nccl_single_rank.py."""
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(here, "nccl_single_rank.py"), str(port)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("OK"), (r.stdout[-2000:], r.stderr[-4000:])  # (RCCL prints a banner first)


def test_batched_evaluation_of_all_constraints_in_one_launch(hip, oracle, client_submaps):
    """cox_reg_normal_eq_batch (round 3): the six constraints (a, b), a != b, of one pose-graph evaluation in ONE launch -- each
    with its own poses, its own drawn samples (different sizes) -- are, bit for bit, what six calls of cox_reg_normal_eq give,
    equal the oracle's to 1e-6, and PoseGraph.build takes the batched path and returns the same cost / gradient / matrix."""
    from coxgraph_amd.posegraph import PoseGraph, RegistrationConstraint
    trunc = 3 * VOXEL
    lh, lo = client_submaps["hip"], client_submaps["oracle"]
    ref_h = [RegPoints.from_layer(hip, l, 1.0, trunc) for l in lh]
    ref_o = [RegPoints.from_layer(oracle, l, 1.0, trunc) for l in lo]
    pairs = [(a, b) for a in range(N_CLIENTS) for b in range(N_CLIENTS) if a != b]
    regs_h, regs_o, poses_ref, poses_read = [], [], [], []
    rng = np.random.default_rng(11)
    for k, (a, b) in enumerate(pairs):
        gh, go = Registration(hip, ref_h[a], lh[b], 0.0), Registration(oracle, ref_o[a], lo[b], 0.0)
        if k == 2:
            pass  # no stored samples: every point once
        else:
            gh.draw_samples(int((0.1 + 0.05 * k) * ref_h[a].n), 500 + k)
            go.set_samples(gh.get_samples())
        regs_h.append(gh)
        regs_o.append(go)
        poses_ref.append(rng.normal(scale=0.01, size=4))
        poses_read.append(np.array([0.05, -0.03, 0.02, np.radians(1.0)]) + rng.normal(scale=0.01, size=4))
    batch = Registration.normal_eq_batch(regs_h, poses_ref, poses_read)
    for k, (gh, go) in enumerate(zip(regs_h, regs_o)):
        Hs, bs, cs, ns = gh.normal_eq(poses_ref[k], poses_read[k])
        Hb, bb, cb, nb_ = batch[k]
        assert np.array_equal(Hs, Hb) and np.array_equal(bs, bb) and cs == cb and ns == nb_, k
        Ho, bo, co, no = go.normal_eq(poses_ref[k], poses_read[k])
        assert ns == no and ns > 100
        assert np.max(np.abs(Hb - Ho)) <= 1e-6 * max(1.0, float(np.max(np.abs(Ho))))
        assert np.max(np.abs(bb - bo)) <= 1e-6 * max(1.0, float(np.max(np.abs(bo)))) and abs(cb - co) <= 1e-6 * max(1.0, co)
    # through the pose graph: batched (sample_idx None on every constraint) == one call per constraint
    def graph(regs, force_single):
        pg = PoseGraph()
        for c in range(N_CLIENTS):
            pg.add_node(c, [0.01 * c, -0.005 * c, 0.002 * c, 0.001 * c], constant=(c == 0))
        for (a, b), g in zip(pairs, regs):
            pg.reg.append(RegistrationConstraint(a, b, g, sample_idx=(g.get_samples() if force_single and g.get_samples().size else None)))
        return pg.build({k: v.copy() for k, v in pg.poses.items()})
    cost_b, g_b, H_b, _ = graph(regs_h, False)
    cost_s, g_s, H_s, _ = graph([r for r in regs_h], True)
    assert np.allclose(H_b, H_s, rtol=1e-12, atol=0) and np.allclose(g_b, g_s, rtol=1e-12, atol=1e-12) and abs(cost_b - cost_s) <= 1e-12 * max(1.0, cost_s)
    cost_o, g_o, H_o, _ = graph(regs_o, False)
    assert np.allclose(H_b, H_o, rtol=1e-6, atol=1e-6 * np.max(np.abs(H_o))) and abs(cost_b - cost_o) <= 1e-6 * max(1.0, cost_o)
