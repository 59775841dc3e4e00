"""Recover mode front end (voxblox::MeshConverter, coxgraph/include/coxgraph/map_comm/mesh_converter.h) and the
TsdfRecover::processMesh loop (map_comm/tsdf_recover.h:59-99).

CPU part: known-answer tests of the oracle restatement, worked by hand from the reference source (the reference
holds no fixtures for this path: parity unpinned).  GPU part: the HIP converter against the oracle, bit for bit.
"""
import math

import numpy as np
import pytest

from coxgraph_amd import synth
from coxgraph_amd.capi import MeshMsg, MeshConverter, Layer, Integrator, CoxError
from util import compare_layers

IDENT = np.array([1, 0, 0, 0, 0, 0, 0], np.float32)
F = np.float32


def one_triangle(history, verts=((0, 0, 0), (16384, 0, 0), (0, 16384, 0)), cols=((255, 0, 0), (0, 255, 0), (0, 0, 255)), index=(1, -2, 0),
                 traj=((10, 0, IDENT),), edge=0.8):
    v = np.array(verts, np.uint16)
    c = np.array(cols, np.uint8)
    blk = dict(index=index, x=v[:, 0], y=v[:, 1], z=v[:, 2], r=c[:, 0], g=c[:, 1], b=c[:, 2], history=[list(history)] if history is not None else None)
    return MeshMsg(F(edge), [blk], list(traj))


def decode(u, idx, edge=0.8):
    return (F(u) * (F(2.0) / F(65535)) + F(idx)) * F(edge)


def test_vertex_decode_and_recovered_cloud(oracle):
    conv = MeshConverter(oracle, 0.2)
    conv.set_mesh(one_triangle([0, 0]))
    ok, xyz, rgb = conv.convert()
    assert ok and xyz.shape == (3, 3)
    # mesh_converter.h:96-112: (float(x) * (2/65535) + float(index)) * block_edge_length, in float
    exp = np.array([[decode(0, 1), decode(0, -2), decode(0, 0)], [decode(16384, 1), decode(0, -2), decode(0, 0)],
                    [decode(0, 1), decode(16384, -2), decode(0, 0)]], np.float32)
    assert np.array_equal(xyz.view(np.uint32), exp.view(np.uint32))
    assert np.array_equal(rgb, [[255, 0, 0], [0, 255, 0], [0, 0, 255]])


def test_interpolate_triangle_order_counts_and_colour_quirk(oracle):
    # legs of 0.4 m (16384 * 2/65535 * 0.8 = 0.400006), hypotenuse 0.5657 m; step 0.15
    conv = MeshConverter(oracle, 0.15)
    conv.set_mesh(one_triangle([0, 0]))
    conv.convert()
    (T, pts, col), = conv.pose_clouds()
    # triangle (3) + e01: 0.15, 0.30 (2) + centroid (1) + e02: 2 + e12: 0.15, 0.30, 0.45 (3) = 11
    assert pts.shape == (11, 3)
    p0, p1, p2 = pts[0], pts[1], pts[2]
    assert np.allclose(pts[3] - p0, [0.15, 0, 0], atol=1e-6) and np.allclose(pts[4] - p0, [0.30, 0, 0], atol=1e-6)
    assert np.array_equal(pts[5], ((p0 + p1) + p2) / F(3))                     # centroid sits between e01 and e02
    assert np.allclose(pts[6] - p0, [0, 0.15, 0], atol=1e-6) and np.allclose(pts[7] - p0, [0, 0.30, 0], atol=1e-6)
    d = (p2 - p1) / np.linalg.norm(p2 - p1)
    assert np.allclose(pts[8] - p1, d * 0.15, atol=1e-6) and np.allclose(pts[10] - p1, d * 0.45, atol=1e-6)
    # colours: vertices have a = 255
    assert np.array_equal(col[:3], [[255, 0, 0, 255], [0, 255, 0, 255], [0, 0, 255, 255]])
    # e01 at 0.15 / 0.400006: blend(c0, 1 - f, c1, f)
    f = 0.15 / 0.400006
    assert np.array_equal(col[3], [round(255 * (1 - f)), round(255 * f), 0, 255])
    # e02 blends colors[0] with colors[1] as well (mesh_converter.h:236-238), so no blue appears on that edge
    assert np.array_equal(col[6], col[3]) and col[6][2] == 0
    # e12 blends colors[1] and colors[2]
    f = 0.15 / float(np.linalg.norm(p2 - p1))
    assert np.array_equal(col[8], [0, round(255 * (1 - f)), round(255 * f), 255])
    # centroid colour: blend(c2, 1/3, blend(c0, .5, c1, .5), 2/3): blend(c0,c1) = (128,128,0) -> (85, 85, 85)
    assert np.array_equal(col[5], [85, 85, 85, 255])


def test_history_runs_uint8_keys_and_frame_ids(oracle):
    # runs [0,1] and [258,258]: frame ids 0, 1, 258 -> map keys 0, 1, 2 (uint8 key, mesh_converter.h:287);
    # pose stamps: 0 s, +0.05 s (id 1), +0.10 s (id 2), +0.26 s (id round(5.2) = 5: empty), +0.05 s again (id 1)
    t0 = (10, 950000000)
    stamps = [t0, (11, 0), (11, 50000000), (11, 210000000), (11, 0)]
    traj = [(s, ns, IDENT) for s, ns in stamps]
    conv = MeshConverter(oracle, 1.0)  # step larger than any edge: only the centroid is added
    conv.set_mesh(one_triangle([0, 1, 258, 258], traj=traj))
    conv.convert()
    clouds = conv.pose_clouds()
    assert [len(c[1]) for c in clouds] == [4, 4, 4, 0, 4]
    # overlapping runs append the triangle twice to the same frame
    conv2 = MeshConverter(oracle, 1.0)
    conv2.set_mesh(one_triangle([0, 1, 1, 1], traj=traj))
    conv2.convert()
    assert [len(c[1]) for c in conv2.pose_clouds()] == [4, 8, 0, 0, 8]
    # a reversed run contributes nothing
    conv3 = MeshConverter(oracle, 1.0)
    conv3.set_mesh(one_triangle([3, 2], traj=traj))
    conv3.convert()
    assert [len(c[1]) for c in conv3.pose_clouds()] == [0, 0, 0, 0, 0]


def test_cloud_is_moved_into_the_camera_frame(oracle):
    R, origin, T = synth.camera_pose(30)
    conv = MeshConverter(oracle, 1.0)
    conv.set_mesh(one_triangle([0, 0], traj=[(5, 0, T)]))
    _, rec, _ = conv.convert()
    (T_out, pts, _), = conv.pose_clouds()
    assert np.array_equal(T_out, T)
    exp = (rec.astype(np.float64) - origin) @ R       # R^T (p - t)
    assert np.allclose(pts[:3], exp, atol=2e-6)


def test_blocks_without_history_and_empty_inputs(oracle):
    conv = MeshConverter(oracle, 0.2)
    conv.set_mesh(one_triangle(None))
    ok, xyz, _ = conv.convert()
    assert ok and len(xyz) == 0                        # block skipped (mesh_converter.h:87), but the mesh was not empty
    assert [len(c[1]) for c in conv.pose_clouds()] == [0]
    conv = MeshConverter(oracle, 0.2)
    conv.set_mesh(MeshMsg(F(0.8), [], [(1, 0, IDENT)]))
    ok, xyz, _ = conv.convert()
    assert not ok and len(xyz) == 0                    # "if (mesh_.mesh_blocks.empty()) return false"
    conv = MeshConverter(oracle, 0.2)
    conv.set_mesh(one_triangle([0, 0], traj=[]))       # empty trajectory: setMesh ignores the message
    ok, xyz, _ = conv.convert()
    assert not ok and conv.pose_clouds() == []
    with pytest.raises(CoxError):
        c = MeshConverter(oracle, 0.2)
        c.set_mesh(one_triangle([0, 1, 2]))            # odd run list (the reference reads past the end): refused


def test_process_mesh_oracle_matches_manual_loop(oracle):
    m = synth.make_wall_mesh(seed=3, n_frames=6)
    msg = MeshMsg(m["block_edge_length"], m["blocks"], m["trajectory"])
    cfg = oracle.default_config(**synth.integrator_overrides(0.05))
    la, lb = Layer(oracle, 0.05), Layer(oracle, 0.05)
    ia, ib = Integrator(oracle, la, cfg, "merged"), Integrator(oracle, lb, cfg, "merged")
    conv = MeshConverter(oracle, 0.05)
    n_rec, n_int = conv.process_mesh(ia, msg)
    conv2 = MeshConverter(oracle, 0.05)
    conv2.set_mesh(msg)
    ok, rec, _ = conv2.convert()
    calls = 0
    for T, pts, col in conv2.pose_clouds():
        if len(pts) == 0:
            continue
        ib.integrate_points(T, pts, col)
        calls += 1
    assert n_rec == len(rec) == 3 * 32 * 14 and n_int == calls and calls > 0
    rep = compare_layers(la, lb, tol=0.0)
    assert rep["blocks"] > 0 and rep["observed"] > 1000


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("step,seed,frame_step", [(0.05, 1, 1), (0.2, 2, 1), (0.01, 3, 2)])
def test_gpu_clouds_match_oracle_bit_for_bit(hip, oracle, step, seed, frame_step):
    m = synth.make_wall_mesh(seed=seed, n_frames=10, frame_step=frame_step)
    msg = MeshMsg(m["block_edge_length"], m["blocks"], m["trajectory"])
    out = {}
    for name, eng in (("hip", hip), ("oracle", oracle)):
        conv = MeshConverter(eng, step)
        conv.set_mesh(msg)
        ok, rec, rgb = conv.convert()
        out[name] = (ok, rec, rgb, conv.pose_clouds())
    (oka, ra, ca, pa), (okb, rb, cb, pb) = out["hip"], out["oracle"]
    assert oka and okb
    assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32)) and np.array_equal(ca, cb)
    assert len(pa) == len(pb) == 10
    total = 0
    for (Ta, xa, qa), (Tb, xb, qb) in zip(pa, pb):
        assert np.array_equal(Ta, Tb)
        assert xa.shape == xb.shape
        assert np.array_equal(xa.view(np.uint32), xb.view(np.uint32))
        assert np.array_equal(qa, qb)
        total += len(xa)
    assert total > 1000


@pytest.mark.gpu
def test_gpu_kat_cases_match_oracle(hip, oracle):
    t0 = (10, 950000000)
    stamps = [t0, (11, 0), (11, 50000000), (11, 210000000), (11, 0)]
    traj = [(s, ns, synth.camera_pose(7 * k)[2]) for k, (s, ns) in enumerate(stamps)]
    for hist in ([0, 1, 258, 258], [0, 1, 1, 1], [3, 2], [], [0, 300]):
        res = []
        for eng in (hip, oracle):
            conv = MeshConverter(eng, 0.15)
            conv.set_mesh(one_triangle(hist, traj=traj))
            ok, rec, rgb = conv.convert()
            res.append((ok, rec, rgb, conv.pose_clouds()))
        (oka, ra, ca, pa), (okb, rb, cb, pb) = res
        assert oka == okb and np.array_equal(ra, rb) and np.array_equal(ca, cb)
        for (Ta, xa, qa), (Tb, xb, qb) in zip(pa, pb):
            assert xa.shape == xb.shape and np.array_equal(xa.view(np.uint32), xb.view(np.uint32)) and np.array_equal(qa, qb)
    # edge cases of the message itself
    for msg in (one_triangle(None, traj=traj), MeshMsg(F(0.8), [], traj), one_triangle([0, 0], traj=[])):
        res = []
        for eng in (hip, oracle):
            conv = MeshConverter(eng, 0.15)
            conv.set_mesh(msg)
            ok, rec, _ = conv.convert()
            res.append((ok, len(rec), [len(c[1]) for c in conv.pose_clouds()]))
        assert res[0] == res[1]
    with pytest.raises(CoxError):
        c = MeshConverter(hip, 0.15)
        c.set_mesh(one_triangle([0, 1, 2], traj=traj))
        c.convert()
    with pytest.raises(CoxError):                      # a step this small would stall the reference's float loop
        c = MeshConverter(hip, 1e-9)
        c.set_mesh(one_triangle([0, 0], traj=traj))
        c.convert()


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["merged", "simple", "fast"])  # "fast" is what coxgraph/config/tsdf_recover.yaml:6 configures
def test_gpu_process_mesh_matches_oracle(hip, oracle, method):
    m = synth.make_wall_mesh(seed=5, n_frames=12)
    msg = MeshMsg(m["block_edge_length"], m["blocks"], m["trajectory"])
    layers = {}
    for name, eng in (("hip", hip), ("oracle", oracle)):
        cfg = eng.default_config(integrator_threads=1, **synth.integrator_overrides(0.05))
        layer = Layer(eng, 0.05)
        integ = Integrator(eng, layer, cfg, method)
        # something already in the layer: processMesh starts with removeAllBlocks
        T, pts, rgba, _ = synth.make_frame(0)
        integ.integrate_points(T, pts[::16], rgba[::16])
        conv = MeshConverter(eng, 0.05)
        n_rec, n_int = conv.process_mesh(integ, msg)
        layers[name] = (layer, integ, conv, n_rec, n_int)
    assert layers["hip"][3:] == layers["oracle"][3:]
    rep = compare_layers(layers["hip"][0], layers["oracle"][0])
    assert rep["blocks"] > 0 and rep["observed"] > 1000
    assert rep["bitexact_d"] and rep["bitexact_w"]
    # the recovered cloud outlives clear(), like the caller-owned recovered_pointcloud
    xa, ca = layers["hip"][2].recovered()
    assert len(xa) == layers["hip"][3]
