"""Committed fixtures (tests/golden/oracle_golden.npz, made by tests/golden/make_golden.py).

They are NOT reference outputs -- the reference holds none for this path -- they freeze this repository's oracle.
CPU: the oracle still reproduces them.  GPU: the HIP engine reproduces them without the oracle in the loop."""
import hashlib
import os

import numpy as np
import pytest

from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, Integrator, RegPoints, Registration

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.npz"), allow_pickle=False)


def _integrate(eng, method):
    cfg = eng.default_config(integrator_threads=1, **synth.integrator_overrides(0.10))
    layer = Layer(eng, 0.10, capacity_blocks=4096)
    integ = Integrator(eng, layer, cfg, method)
    stats = []
    for t in (0, 7):
        T, pts, rgba, _ = synth.make_frame(t)
        integ.integrate_points(T, pts[::16], rgba[::16])
        s = integ.last_stats()
        stats.append([s[k] for k in ("n_valid", "n_rays", "n_updates", "n_touched_voxels", "n_new_blocks")])
    return layer, np.array(stats, np.int64)


def _check_integration(eng, method):
    layer, stats = _integrate(eng, method)
    idx, vox = layer.download()
    assert np.array_equal(idx, G[f"{method}_block_idx"])
    assert np.array_equal(stats, G[f"{method}_stats"])
    where = G[f"{method}_sample_where"]
    assert np.array_equal(vox[where[:, 0], where[:, 1]], G[f"{method}_sample_words"])
    assert np.array_equal(np.frombuffer(hashlib.sha256(vox.tobytes()).digest(), np.uint8), G[f"{method}_words_sha256"])
    return idx, vox


def _check_registration(eng, idx, vox):
    layer = Layer(eng, 0.10, capacity_blocks=4096)
    layer.upload(idx, vox)
    reg = Registration(eng, RegPoints(eng, G["reg_points"]), layer)
    r, jf, jr = reg.evaluate(G["reg_pose_ref"], G["reg_pose_read"])
    assert np.max(np.abs(r - G["reg_residuals"])) <= 1e-4
    assert np.max(np.abs(jf - G["reg_jac_ref"])) <= 1e-3 * max(1.0, np.max(np.abs(G["reg_jac_ref"])))
    assert np.max(np.abs(jr - G["reg_jac_read"])) <= 1e-3 * max(1.0, np.max(np.abs(G["reg_jac_read"])))
    H, b, cost, nc = reg.normal_eq(G["reg_pose_ref"], G["reg_pose_read"])
    assert np.allclose(H, G["reg_H"], rtol=1e-6, atol=1e-6 * np.max(np.abs(G["reg_H"])))
    assert np.allclose(b, G["reg_b"], rtol=1e-6, atol=1e-6 * np.max(np.abs(G["reg_b"])))
    assert abs(cost - G["reg_cost_ncorr"][0]) <= 1e-6 * max(1.0, G["reg_cost_ncorr"][0]) and nc == int(G["reg_cost_ncorr"][1])


def test_oracle_reproduces_golden_ray_paths(oracle):
    import ctypes as C
    off = 0
    for row, n in zip(G["ray_params"], G["ray_path_lengths"]):
        cap = 4096
        out = (C.c_int64 * (3 * cap))()
        cnt = C.c_uint64()
        oracle.fn("raycast")((C.c_float * 3)(*row[0:3]), (C.c_float * 3)(*row[3:6]), int(row[6]), int(row[7]), C.c_float(row[8]), C.c_float(row[9]),
                             C.c_float(row[10]), int(row[11]), out, C.c_uint64(cap), C.byref(cnt))
        assert cnt.value == n
        assert np.array_equal(np.array(out[:3 * n], np.int64).reshape(-1, 3), G["ray_paths"][off:off + n])
        off += n


def _check_mesh(eng):
    from coxgraph_amd.capi import MeshMsg, MeshConverter
    m = synth.make_wall_mesh(seed=11, n_frames=8)
    msg = MeshMsg(m["block_edge_length"], m["blocks"], m["trajectory"])
    conv = MeshConverter(eng, 0.07)
    conv.set_mesh(msg)
    ok, rec, rgb = conv.convert()
    clouds = conv.pose_clouds()
    sha = lambda b: np.frombuffer(hashlib.sha256(b).digest(), np.uint8)
    assert ok and np.array_equal(sha(rec.tobytes() + rgb.tobytes()), G["mesh_recovered_sha256"])
    assert np.array_equal([len(c[1]) for c in clouds], G["mesh_cloud_sizes"])
    assert np.array_equal(clouds[3][1][:16], G["mesh_cloud3_head"])
    assert np.array_equal(sha(b"".join(c[1].tobytes() + c[2].tobytes() for c in clouds)), G["mesh_clouds_sha256"])
    cfg = eng.default_config(**synth.integrator_overrides(0.05))
    layer = Layer(eng, 0.05, capacity_blocks=4096)
    integ = Integrator(eng, layer, cfg, "merged")
    n_rec, n_int = MeshConverter(eng, 0.07).process_mesh(integ, msg)
    idx, vox = layer.download()
    assert np.array_equal([n_rec, n_int, len(idx)], G["mesh_process_counts"])
    assert np.array_equal(sha(idx.tobytes() + vox.tobytes()), G["mesh_layer_sha256"])


def test_oracle_reproduces_golden_mesh_recover(oracle):
    _check_mesh(oracle)


@pytest.mark.gpu
def test_hip_reproduces_golden_mesh_recover(hip):
    _check_mesh(hip)


@pytest.mark.parametrize("method", ["merged", "simple", "fast"])
def test_oracle_reproduces_golden_layers(oracle, method):
    _check_integration(oracle, method)


def test_oracle_reproduces_golden_registration(oracle):
    idx, vox = _check_integration(oracle, "merged")
    _check_registration(oracle, idx, vox)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["merged", "simple", "fast"])
def test_hip_reproduces_golden_layers(hip, method):
    """Bit-identical voxel words (sha256 of the whole serialised layer) without the oracle in the loop."""
    _check_integration(hip, method)


@pytest.mark.gpu
def test_hip_reproduces_golden_registration(hip):
    idx, vox = _check_integration(hip, "merged")
    _check_registration(hip, idx, vox)
