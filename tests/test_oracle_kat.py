"""Known-answer tests for the CPU oracle, derivable by hand.

The reference ships no tests or golden vectors for this path (parity unpinned); these analytic
cases are the only independent check of the oracle.
"""
import ctypes as C
import math

import numpy as np
import pytest

from coxgraph_amd.capi import Layer, Integrator, RegPoints, Registration, words_to_fields


def f3(*v):
    return (C.c_float * 3)(*v)


def test_grid_index_epsilon_and_negative_floor(oracle):
    out = (C.c_int64 * 3)()
    oracle.fn("grid_index")(f3(0.1, -0.1, 0.0), C.c_float(10.0), out)
    # 0.1f*10 = 1.0000000149 -> 1 ; -0.1f*10 + 1e-6 = -0.999999 -> floor = -1 ; 0 + 1e-6 -> 0
    assert list(out) == [1, -1, 0]
    oracle.fn("grid_index")(f3(-0.25, 0.9999999, 15.99), C.c_float(1.0), out)
    assert list(out) == [-1, 1, 15]  # 0.9999999f + 1e-6f rounds to >= 1.0


def test_block_local_linear(oracle):
    g = (C.c_int64 * 3)(-1, 0, 16)
    blk, loc, lin = (C.c_int32 * 3)(), (C.c_int32 * 3)(), C.c_int32()
    oracle.fn("block_local")(g, 16, blk, loc, C.byref(lin))
    assert list(blk) == [-1, 0, 1] and list(loc) == [15, 0, 0] and lin.value == 15
    g = (C.c_int64 * 3)(17, -17, 5)
    oracle.fn("block_local")(g, 16, blk, loc, C.byref(lin))
    assert list(blk) == [1, -2, 0] and list(loc) == [1, 15, 5] and lin.value == 1 + 16 * (15 + 16 * 5)


def test_mixed_order_is_a_bijection(oracle):
    f = oracle.fn("mixed_index", C.c_uint64)
    for n in (0, 5, 1024, 3000, 307200):
        idx = [f(C.c_uint64(s), C.c_uint64(n)) for s in range(n)] if n <= 3000 else None
        if idx is not None:
            assert sorted(idx) == list(range(n))
    # N = 307200: 300 groups of 1024; consecutive sequence numbers hop between groups
    assert f(C.c_uint64(0), C.c_uint64(307200)) == 0
    assert f(C.c_uint64(1), C.c_uint64(307200)) == 1024
    assert f(C.c_uint64(300), C.c_uint64(307200)) == 1
    # tail beyond groups*1024 maps to itself
    assert f(C.c_uint64(2999), C.c_uint64(3000)) == 2999


def _raycast(oracle, origin, point, clearing=0, carving=1, max_len=5.0, inv=10.0, trunc=0.3, from_origin=1):
    cap = 4096
    out = (C.c_int64 * (3 * cap))()
    n = C.c_uint64()
    oracle.fn("raycast")(f3(*origin), f3(*point), clearing, carving, C.c_float(max_len), C.c_float(inv), C.c_float(trunc),
                         from_origin, out, C.c_uint64(cap), C.byref(n))
    return np.array(out[:3 * n.value], np.int64).reshape(-1, 3)


def test_raycast_generic_ray_visits_face_connected_inclusive_path(oracle):
    idx = _raycast(oracle, (0.05, 0.07, 0.03), (1.03, 0.41, 0.22))
    # inclusive of start and end voxel, each step moves exactly one axis by one
    steps = np.abs(np.diff(idx, axis=0)).sum(axis=1)
    assert np.all(steps == 1)
    assert tuple(idx[0]) == (0, 0, 0)
    # end = point + unit*0.3 ; unit = d/|d|
    d = np.array([0.98, 0.34, 0.19])
    end = np.array([1.03, 0.41, 0.22]) + d / np.linalg.norm(d) * 0.3
    assert tuple(idx[-1]) == tuple(np.floor(end * 10 + 1e-6).astype(int))
    assert len(idx) == np.abs(idx[-1] - idx[0]).sum() + 1


def test_raycast_axis_aligned_ray_keeps_upstream_zero_division_quirk(oracle):
    """Appendix D.3 geometry.  ray.y == ray.z == 0 exactly, so t_to_next = -0.5/0 = -inf on y and z
    and t_step = 0/0 = NaN: the y and z axes each win one argmin (index unchanged), turn NaN, and only
    then does x advance.  ray_length_in_steps = 13 -> 14 indices are still emitted: the start voxel three
    times, then x = 1..11.  (A DDA without the quirk would give x = 0..13.)"""
    idx = _raycast(oracle, (0.05, 0.05, 0.05), (1.05, 0.05, 0.05))
    assert len(idx) == 14
    assert np.all(idx[:, 1:] == 0)
    assert list(idx[:, 0]) == [0, 0, 0] + list(range(1, 12))


def test_raycast_clearing_and_no_carving(oracle):
    # clearing ray: ends at min(max(len - trunc, 0), max_len) along the ray
    idx = _raycast(oracle, (0.05, 0.07, 0.03), (8.0, 0.3, 0.2), clearing=1, max_len=2.0)
    d = np.array([7.95, 0.23, 0.17])
    end = np.array([0.05, 0.07, 0.03]) + d / np.linalg.norm(d) * 2.0
    assert tuple(idx[-1]) == tuple(np.floor(end * 10 + 1e-6).astype(int))
    # carving off: starts trunc in front of the surface
    idx2 = _raycast(oracle, (0.05, 0.07, 0.03), (1.03, 0.41, 0.22), carving=0)
    d = np.array([0.98, 0.34, 0.19])
    start = np.array([1.03, 0.41, 0.22]) - d / np.linalg.norm(d) * 0.3
    assert tuple(idx2[0]) == tuple(np.floor(start * 10 + 1e-6).astype(int))
    # cast_from_origin = false walks the same voxels backwards when no tie-breaking is involved
    fwd = _raycast(oracle, (0.05, 0.07, 0.03), (1.03, 0.41, 0.22))
    bwd = _raycast(oracle, (0.05, 0.07, 0.03), (1.03, 0.41, 0.22), from_origin=0)
    assert tuple(bwd[0]) == tuple(fwd[-1]) and tuple(bwd[-1]) == tuple(fwd[0]) and len(fwd) == len(bwd)


IDENT = np.array([1, 0, 0, 0, 0, 0, 0], np.float32)


def _layer_dict(layer):
    idx, vox = layer.download()
    d, w, rgba = words_to_fields(vox)
    return {tuple(b): (d[i], w[i], rgba[i]) for i, b in enumerate(idx)}


def _voxel(blocks, g):
    b = tuple(int(math.floor(x / 16)) for x in g)
    lin = (g[0] % 16) + 16 * ((g[1] % 16) + 16 * (g[2] % 16))
    d, w, c = blocks[b]
    return float(d[lin]), float(w[lin]), c[lin]


def test_single_generic_ray_simple_integrator_values(oracle):
    """One ray, voxel 0.1, trunc 0.3, const weight, weight drop-off: check sdf, clamping, drop-off and
    the 'block allocated even when the update early-returns' rule against a float64 re-derivation."""
    cfg = oracle.default_config(default_truncation_distance=0.3, use_const_weight=1, min_ray_length_m=0.1, max_ray_length_m=5.0)
    layer = Layer(oracle, 0.1)
    integ = Integrator(oracle, layer, cfg, "simple")
    T = IDENT.copy()
    T[4:] = (0.05, 0.07, 0.03)
    p_c = np.array([[0.98, 0.34, 0.19]], np.float32)
    integ.integrate_points(T, p_c, np.array([[10, 20, 30, 255]], np.uint8))
    st = integ.last_stats()
    blocks = _layer_dict(layer)
    origin = np.array([0.05, 0.07, 0.03])
    point = origin + np.array([0.98, 0.34, 0.19])
    path = _raycast(oracle, origin, point)
    assert st["n_rays"] == 1 and st["n_updates"] == len(path) and st["n_touched_voxels"] == len(path)
    dvec = point - origin
    dist = np.linalg.norm(dvec)
    n_nonzero = 0
    for g in path:
        c = (g + 0.5) * 0.1
        sdf = dist - np.dot(c - origin, dvec) / dist
        uw = 1.0
        if sdf < -0.1:
            uw = max((0.3 + sdf) / (0.3 - 0.1), 0.0)
        d, w, col = _voxel(blocks, tuple(int(x) for x in g))
        if uw < 1e-6:
            assert w == 0.0 and d == 0.0  # early return, voxel untouched (its block exists)
            continue
        n_nonzero += 1
        assert abs(w - uw) < 1e-5
        assert abs(d - min(0.3, max(-0.3, sdf))) < 1e-5
        if abs(sdf) < 0.3:
            assert tuple(col) == (10, 20, 30, 255)
        else:
            assert tuple(col) == (0, 0, 0, 0)
    assert n_nonzero >= len(path) - 2


def test_merged_three_coincident_points_weight_three(oracle):
    cfg = oracle.default_config(default_truncation_distance=0.3, use_const_weight=1, min_ray_length_m=0.1, max_ray_length_m=5.0)
    T = IDENT.copy()
    T[4:] = (0.05, 0.07, 0.03)
    p = np.array([[0.98, 0.34, 0.19]], np.float32)
    l1, l3 = Layer(oracle, 0.1), Layer(oracle, 0.1)
    Integrator(oracle, l1, cfg, "simple").integrate_points(T, p, None)
    m = Integrator(oracle, l3, cfg, "merged")
    m.integrate_points(T, np.repeat(p, 3, axis=0), None)
    assert m.last_stats()["n_rays"] == 1
    i1, v1 = l1.download()
    i3, v3 = l3.download()
    assert np.array_equal(i1, i3)
    d1, w1, _ = words_to_fields(v1)
    d3, w3, _ = words_to_fields(v3)
    assert np.allclose(d1, d3, atol=1e-6) and np.allclose(3 * w1, w3, atol=1e-5)


def test_fast_dedups_start_voxel_and_terminates_on_seen_voxels(oracle):
    cfg = oracle.default_config(default_truncation_distance=0.3, use_const_weight=1, min_ray_length_m=0.1, max_ray_length_m=5.0,
                                max_consecutive_ray_collisions=2)
    T = IDENT.copy()
    T[4:] = (0.05, 0.07, 0.03)
    layer = Layer(oracle, 0.1)
    f = Integrator(oracle, layer, cfg, "fast")
    p = np.array([[0.98, 0.34, 0.19]], np.float32)
    f.integrate_points(T, np.repeat(p, 2, axis=0), None)
    st = f.last_stats()
    assert st["n_valid"] == 2 and st["n_rays"] == 1  # second point starts in an already-used half-voxel
    full = st["n_updates"]
    # a second, slightly offset ray ends in another sub-voxel but re-enters already observed voxels:
    # it stops after max_consecutive_ray_collisions + 1 = 3 seen voxels in a row
    layer2 = Layer(oracle, 0.1)
    f2 = Integrator(oracle, layer2, cfg, "fast")
    p2 = np.array([[0.98, 0.34, 0.19], [0.98, 0.40, 0.19]], np.float32)
    f2.integrate_points(T, p2, None)
    st2 = f2.last_stats()
    assert st2["n_rays"] == 2 and full < st2["n_updates"] < 2 * full


def _plane_layer(oracle, voxel=0.1, a=0.05, bx=0.3, by=-0.2, bz=0.1, blocks=((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0))):
    """Upload a layer whose voxel distances sample the linear field d = a + b . c on voxel centres."""
    idx = np.array(blocks, np.int32)
    vox = np.zeros((len(blocks), 4096, 3), np.uint32)
    lin = np.arange(4096)
    lx, ly, lz = lin % 16, (lin // 16) % 16, lin // 256
    for i, b in enumerate(blocks):
        cx = (b[0] * 16 + lx + 0.5) * voxel
        cy = (b[1] * 16 + ly + 0.5) * voxel
        cz = (b[2] * 16 + lz + 0.5) * voxel
        d = (a + bx * cx + by * cy + bz * cz).astype(np.float32)
        vox[i, :, 0] = d.view(np.uint32)
        vox[i, :, 1] = np.float32(2.0).view(np.uint32)
    layer = Layer(oracle, voxel)
    layer.upload(idx, vox, action=0)
    return layer


def test_trilinear_reproduces_linear_field_and_fails_near_missing_block(oracle):
    layer = _plane_layer(oracle)
    val, grad = C.c_float(), (C.c_float * 3)()
    probe = oracle.fn("interp")
    for pos in [(0.77, 0.33, 0.41), (1.58, 1.61, 0.8), (1.6001, 0.2, 0.9), (0.05, 0.05, 0.05), (2.9, 3.1, 1.5)]:
        ok = probe(layer.h, f3(*pos), C.byref(val), grad)
        assert ok == 1, pos
        expect = 0.05 + 0.3 * pos[0] - 0.2 * pos[1] + 0.1 * pos[2]
        assert abs(val.value - expect) < 2e-6
        assert np.allclose(list(grad), [0.3, -0.2, 0.1], atol=2e-5)
    # value at a voxel centre equals that voxel's distance
    ok = probe(layer.h, f3(0.25, 0.35, 0.45), C.byref(val), grad)
    assert ok == 1 and abs(val.value - (0.05 + 0.3 * 0.25 - 0.2 * 0.35 + 0.1 * 0.45)) < 1e-6
    # within half a voxel of an unallocated neighbour block the 8-neighbourhood is incomplete
    assert probe(layer.h, f3(3.17, 0.5, 0.5), C.byref(val), grad) == 0   # +x neighbour block (2,0,0) missing
    assert probe(layer.h, f3(0.5, 0.5, 1.58), C.byref(val), grad) == 0   # +z neighbour block missing
    assert probe(layer.h, f3(0.02, 0.5, 0.5), C.byref(val), grad) == 0   # lower corner falls into block (-1,0,0)
    assert probe(layer.h, f3(5.0, 5.0, 5.0), C.byref(val), grad) == 0    # no block at all


def test_registration_identity_and_translated_wall(oracle):
    """Appendix D.7: planar wall normal to x; translating the reading pose by delta along x gives
    residual = -delta * w * scale and J_ref[:,0] = -w * scale (wall distance grows with x)."""
    voxel = 0.1
    layer = _plane_layer(oracle, voxel=voxel, a=-1.0, bx=1.0, by=0.0, bz=0.0)  # d = x - 1: wall at x = 1
    rng = np.random.default_rng(3)
    n = 200
    xyz = np.stack([rng.uniform(0.7, 1.3, n), rng.uniform(0.4, 2.6, n), rng.uniform(0.4, 1.2, n)], axis=1)
    w = rng.uniform(0.5, 2.0, n)
    pts = np.concatenate([xyz, (xyz[:, :1] - 1.0), w[:, None]], axis=1).astype(np.float32)
    reg = Registration(oracle, RegPoints(oracle, pts), layer)
    zero = np.zeros(4)
    r, jf, jr = reg.evaluate(zero, zero)
    assert np.max(np.abs(r)) < 1e-5
    delta = 0.03
    scale = n / pts[:, 4].astype(np.float64).sum()
    r, jf, jr = reg.evaluate(zero, np.array([delta, 0, 0, 0.0]))
    # reading frame shifted by +delta -> points appear at x - delta -> reading distance smaller by delta
    assert np.allclose(r, delta * pts[:, 4] * scale, atol=2e-5)
    assert np.allclose(jf[:, 0], -pts[:, 4] * scale, atol=2e-4)
    assert np.allclose(jr[:, 0], pts[:, 4] * scale, atol=2e-4)
    assert np.allclose(jf[:, 1:3], 0, atol=2e-4)
    H, b, cost, nc = reg.normal_eq(zero, np.array([delta, 0, 0, 0.0]))
    J = np.concatenate([jf, jr], axis=1)
    assert nc == n
    assert np.allclose(H, J.T @ J, rtol=1e-9, atol=1e-9)
    assert np.allclose(b, J.T @ r, rtol=1e-9, atol=1e-9)
    assert abs(cost - 0.5 * float(r @ r)) < 1e-12


def test_registration_jacobian_matches_central_differences(oracle):
    layer = _plane_layer(oracle, voxel=0.1, a=-0.4, bx=0.35, by=0.2, bz=-0.15)
    rng = np.random.default_rng(5)
    n = 64
    xyz = np.stack([rng.uniform(0.9, 2.2, n), rng.uniform(0.9, 2.2, n), rng.uniform(0.5, 1.0, n)], axis=1)
    pts = np.concatenate([xyz, rng.uniform(-0.1, 0.1, (n, 1)), rng.uniform(0.5, 2.0, (n, 1))], axis=1).astype(np.float32)
    reg = Registration(oracle, RegPoints(oracle, pts), layer)
    p_ref = np.array([0.05, -0.02, 0.01, 0.02])
    p_read = np.array([-0.03, 0.04, -0.02, -0.015])
    r0, jf, jr = reg.evaluate(p_ref, p_read)
    h = 1e-3
    for k in range(4):
        e = np.zeros(4)
        e[k] = h
        num_f = (reg.evaluate(p_ref + e, p_read, jacobians=False)[0] - reg.evaluate(p_ref - e, p_read, jacobians=False)[0]) / (2 * h)
        num_r = (reg.evaluate(p_ref, p_read + e, jacobians=False)[0] - reg.evaluate(p_ref, p_read - e, jacobians=False)[0]) / (2 * h)
        # the field is linear so central differences are exact up to float32 interpolation noise
        assert np.allclose(jf[:, k], num_f, atol=3e-3), k
        assert np.allclose(jr[:, k], num_r, atol=3e-3), k


def test_wire_format_words(oracle):
    """Appendix D.9: voxel (d=1.0, w=2.0, rgba=(1,2,3,4)) -> 0x3F800000, 0x40000000, 0x01020304."""
    layer = Layer(oracle, 0.1)
    vox = np.zeros((1, 4096, 3), np.uint32)
    vox[0, 7] = (0x3F800000, 0x40000000, 0x01020304)
    layer.upload(np.array([[2, -3, 4]], np.int32), vox)
    idx, out = layer.download()
    assert np.array_equal(idx, [[2, -3, 4]]) and np.array_equal(out, vox)
    d, w, rgba = words_to_fields(out)
    assert d[0, 7] == 1.0 and w[0, 7] == 2.0 and tuple(rgba[0, 7]) == (1, 2, 3, 4)
    assert layer.stats() == (1, 49152)
    # merge action: mergeVoxelAIntoVoxelB weighted mean
    vox2 = np.zeros_like(vox)
    vox2[0, 7] = (np.float32(0.0).view(np.uint32), np.float32(2.0).view(np.uint32), 0x03040506)
    layer.upload(np.array([[2, -3, 4]], np.int32), vox2, action=1)
    d, w, rgba = words_to_fields(layer.download()[1])
    assert d[0, 7] == 0.5 and w[0, 7] == 4.0 and tuple(rgba[0, 7]) == (2, 3, 4, 5)
    layer.upload(np.zeros((0, 3), np.int32), np.zeros((0, 4096, 3), np.uint32), action=2)
    assert layer.stats()[0] == 0


def test_registration_points_of_a_layer(oracle):
    """findRelevantVoxelIndices: weight > min and |d| < max, voxel-centre positions, (z,y,x) block / linear voxel order."""
    layer = _plane_layer(oracle, voxel=0.1, a=-1.0, bx=1.0, by=0.0, bz=0.0, blocks=((0, 0, 0), (1, 0, 0), (0, 0, 1)))
    pts = layer.registration_points(min_voxel_weight=1.0, max_voxel_distance=0.3)
    # d = x - 1 on voxel centres x = 0.05 + 0.1 k: |d| < 0.3 <=> k in 7..12, all inside blocks with x index 0 (0..1.6 m);
    # 256 (y,z) columns per k in block (0,0,0) and again in block (0,0,1); block (1,0,0) contributes nothing
    assert pts.shape == (6 * 256 * 2, 5)
    assert np.allclose(pts[:, 3], pts[:, 0] - 1.0, atol=1e-6) and np.all(pts[:, 4] == 2.0)
    assert np.all(np.abs(pts[:, 3]) < 0.3)
    # order: block (0,0,0), then (1,0,0), then (0,0,1); inside a block x fastest
    assert np.allclose(pts[0, :3], [0.75, 0.05, 0.05]) and np.allclose(pts[1, :3], [0.85, 0.05, 0.05])
    assert np.all(pts[:6 * 256, 2] < 1.6) and np.all(pts[6 * 256:, 2] > 1.6)
    assert layer.registration_points(min_voxel_weight=2.0).shape[0] == 0  # strict >
    rp = RegPoints.from_layer(oracle, layer, 1.0, 0.3)
    assert rp.n == len(pts)


def test_layer_merge_same_grid_and_resampled(oracle):
    """mergeLayerAintoLayerB: same grid -> weights add, distances are the weighted mean; with a transform the source is
    resampled first -- a linear field shifted by one voxel along x lands one voxel over, exactly."""
    a = _plane_layer(oracle, voxel=0.1, a=0.05, bx=0.3, by=-0.2, bz=0.1, blocks=((0, 0, 0), (1, 0, 0)))
    b = _plane_layer(oracle, voxel=0.1, a=0.25, bx=0.3, by=-0.2, bz=0.1, blocks=((0, 0, 0),))
    b.merge_from(a)
    idx, vox = b.download()
    assert [tuple(i) for i in idx] == [(0, 0, 0), (1, 0, 0)]
    d, w, _ = words_to_fields(vox)
    da = words_to_fields(a.download()[1])[0]
    assert np.all(w[0] == 4.0) and np.all(w[1] == 2.0)
    assert np.allclose(d[0], da[0] + 0.1, atol=1e-6) and np.allclose(d[1], da[1], atol=1e-6)
    # resample: T_B_A = translation by +0.1 m in x; B's voxel at x sees A's field at x - 0.1
    c = Layer(oracle, 0.1)
    T = np.array([1, 0, 0, 0, 0.1, 0, 0], np.float32)
    c.merge_from(a, T)
    idx, vox = c.download()
    d, w, _ = words_to_fields(vox)
    blocks = {tuple(i): k for k, i in enumerate(idx)}
    assert (0, 0, 0) in blocks and (1, 0, 0) in blocks
    lin = np.arange(4096)
    for bi, k in blocks.items():
        cx = (bi[0] * 16 + lin % 16 + 0.5) * 0.1 - 0.1
        cy = (bi[1] * 16 + (lin // 16) % 16 + 0.5) * 0.1
        cz = (bi[2] * 16 + lin // 256 + 0.5) * 0.1
        seen = w[k] > 0
        expect = 0.05 + 0.3 * cx - 0.2 * cy + 0.1 * cz
        assert np.allclose(d[k][seen], expect[seen], atol=2e-6)
        assert np.all(w[k][seen] == 2.0)
    # the source covers x in [0, 3.2): after the shift the first voxel column of block (0,0,0) has no source -> unobserved
    k0 = blocks[(0, 0, 0)]
    assert np.all(w[k0][lin % 16 == 0] == 0) and np.all(w[k0][lin % 16 == 1] == 2.0)
