"""Differential fuzzing of the paths around the integrator: recover-mode clouds, layer merge / rigid resample,
registration cost.  HIP engine vs CPU oracle, same random inputs, bit-identical results (registration: the documented
tolerances, in practice identical as well)."""
import os

import numpy as np
import pytest

from coxgraph_amd import synth
from coxgraph_amd.capi import MeshMsg, MeshConverter, Layer, Integrator, RegPoints, Registration
from util import run_frames, compare_layers

pytestmark = pytest.mark.gpu
MORE = int(os.environ.get("COX_FUZZ_SEEDS", "0"))  # COX_FUZZ_SEEDS=500 python -m pytest tests/test_gpu_fuzz_aux.py -m gpu: a campaign


def random_pose7(rng, scale=2.0):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([q, rng.uniform(-scale, scale, 3)]).astype(np.float32)


def random_mesh(rng):
    edge = float(rng.choice([0.4, 0.8, 1.6]))
    n_frames = int(rng.integers(1, 40))
    blocks = []
    for _ in range(int(rng.integers(1, 7))):
        nt = int(rng.integers(0, 30))
        hi = int(rng.choice([2000, 32767, 65535]))          # small triangles / in-block / spilling into the neighbours
        v = rng.integers(0, hi + 1, (3 * nt, 3)).astype(np.uint16)
        c = rng.integers(0, 256, (3 * nt, 3)).astype(np.uint8)
        hist = []
        for _t in range(nt):
            runs = []
            for _r in range(int(rng.integers(0, 4))):
                a = int(rng.integers(0, n_frames + 300 * (rng.random() < 0.1)))   # now and then past 255: the uint8 key aliases
                b = a + int(rng.integers(-2, 6))                                  # reversed runs contribute nothing
                runs += [a, max(b, 0)]
            hist.append(runs)
        blocks.append(dict(index=tuple(int(x) for x in rng.integers(-3, 4, 3)), x=v[:, 0], y=v[:, 1], z=v[:, 2], r=c[:, 0], g=c[:, 1], b=c[:, 2],
                           history=None if (nt == 0 or rng.random() < 0.15) else hist))
    sec, nsec = int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 10 ** 9))
    traj = []
    for k in range(n_frames):
        ns = nsec + int(rng.choice([50000000, 50000000, 49000000, 120000000])) * k
        traj.append((sec + ns // 10 ** 9, ns % 10 ** 9, random_pose7(rng)))
    return MeshMsg(np.float32(edge), blocks, traj)


@pytest.mark.parametrize("seed", range(max(16, MORE)))
def test_recover_clouds_fuzz(hip, oracle, seed):
    rng = np.random.default_rng(500 + seed)
    msg = random_mesh(rng)
    step = float(rng.choice([0.02, 0.05, 0.2, 1.0]))
    out = []
    for eng in (hip, oracle):
        conv = MeshConverter(eng, step)
        conv.set_mesh(msg)
        ok, rec, rgb = conv.convert()
        out.append((ok, rec, rgb, conv.pose_clouds()))
    (oka, ra, ca, pa), (okb, rb, cb, pb) = out
    assert oka == okb and np.array_equal(ra.view(np.uint32), rb.view(np.uint32)) and np.array_equal(ca, cb)
    assert len(pa) == len(pb)
    for (Ta, xa, qa), (Tb, xb, qb) in zip(pa, pb):
        assert np.array_equal(Ta, Tb) and xa.shape == xb.shape
        assert np.array_equal(xa.view(np.uint32), xb.view(np.uint32)) and np.array_equal(qa, qb)


@pytest.fixture(scope="module")
def two_layers(oracle):
    la, _, _ = run_frames(oracle, method="merged", voxel=0.10, frames=range(0, 40, 5), subsample=6)
    lb, _, _ = run_frames(oracle, method="merged", voxel=0.10, frames=range(20, 60, 5), subsample=6)
    return la.download(), lb.download()


@pytest.mark.parametrize("seed", range(max(8, MORE // 4)))
def test_layer_merge_fuzz(hip, oracle, two_layers, seed):
    rng = np.random.default_rng(700 + seed)
    (ia, va), (ib, vb) = two_layers
    T = None
    if seed:  # seed 0: same-grid merge
        yaw, tilt = rng.uniform(-np.pi, np.pi), rng.uniform(-0.2, 0.2)
        q = np.array([np.cos(yaw / 2) * np.cos(tilt / 2), np.sin(tilt / 2), 0.0, np.sin(yaw / 2) * np.cos(tilt / 2)])
        T = np.concatenate([q / np.linalg.norm(q), rng.uniform(-1.0, 1.0, 3)]).astype(np.float32)
    res = []
    for eng in (hip, oracle):
        A, B = Layer(eng, 0.10, capacity_blocks=8192), Layer(eng, 0.10, capacity_blocks=8192)
        A.upload(ia, va)
        B.upload(ib, vb)
        B.merge_from(A, T)
        res.append(B)
    rep = compare_layers(res[0], res[1])
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep


@pytest.mark.parametrize("seed", range(max(8, MORE // 2)))
def test_registration_fuzz(hip, oracle, two_layers, seed):
    (ia, va), (ib, vb) = two_layers
    rng = np.random.default_rng(900 + seed)
    mw = float(rng.choice([0.5, 1.0]))
    pr = np.concatenate([rng.uniform(-0.2, 0.2, 3), [rng.uniform(-0.1, 0.1)]])
    pd = pr + np.concatenate([rng.uniform(-0.08, 0.08, 3), [rng.uniform(-0.03, 0.03)]])
    ncc = float(rng.choice([0.0, 0.05]))
    vals = []
    for eng in (hip, oracle):
        A, B = Layer(eng, 0.10, capacity_blocks=8192), Layer(eng, 0.10, capacity_blocks=8192)
        A.upload(ia, va)
        B.upload(ib, vb)
        pts = A.registration_points(mw, 0.3)
        idx = rng.integers(0, len(pts), size=max(1, len(pts) // 3)).astype(np.uint32) if eng is hip else idx
        g = Registration(eng, RegPoints(eng, pts), B, ncc)
        vals.append((pts, g.evaluate(pr, pd, idx), g.normal_eq(pr, pd, idx)))
    (pa, (ra, jfa, jra), (Ha, ba, ca, na)), (pb, (rb, jfb, jrb), (Hb, bb, cb, nb)) = vals
    assert np.array_equal(pa, pb)
    assert np.max(np.abs(ra - rb)) <= 1e-4 and na == nb
    assert np.max(np.abs(jfa - jfb)) <= 1e-3 * max(1.0, np.max(np.abs(jfb))) and np.max(np.abs(jra - jrb)) <= 1e-3 * max(1.0, np.max(np.abs(jrb)))
    assert np.allclose(Ha, Hb, rtol=1e-6, atol=1e-6 * max(1.0, np.max(np.abs(Hb)))) and np.allclose(ba, bb, rtol=1e-6, atol=1e-6 * max(1.0, np.max(np.abs(bb))))
    assert abs(ca - cb) <= 1e-6 * max(1.0, cb)


@pytest.mark.parametrize("seed", range(max(8, MORE // 8)))
def test_recover_process_mesh_fuzz(hip, oracle, seed):
    """The whole recover-mode call: random mesh -> clouds -> one integratePointCloud per pose -> layer."""
    rng = np.random.default_rng(1100 + seed)
    msg = random_mesh(rng)
    step = float(rng.choice([0.05, 0.2]))
    method = str(rng.choice(["merged", "fast", "simple"]))
    voxel = float(rng.choice([0.05, 0.1]))
    res = []
    for eng in (hip, oracle):
        cfg = eng.default_config(integrator_threads=1, default_truncation_distance=3 * voxel, max_ray_length_m=8.0, min_ray_length_m=0.05, use_const_weight=1)
        layer = Layer(eng, voxel, capacity_blocks=60000)
        integ = Integrator(eng, layer, cfg, method)
        res.append((layer, MeshConverter(eng, step).process_mesh(integ, msg)))
    assert res[0][1] == res[1][1]
    rep = compare_layers(res[0][0], res[1][0])
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep
