"""Known-answer tests of the oracle's finishSubmap() pieces (isosurface vertices, ESDF, surface box, weighted sampler).

The reference holds no fixtures for these (parity unpinned), so the anchors are geometric facts that
do not depend on any restatement: marching cubes puts its vertices ON an analytic plane / sphere, a quasi-Euclidean ESDF
equals the true distance along the grid axes, a weighted sampler reproduces the weights.
"""
import numpy as np
import pytest

from coxgraph_amd.capi import Layer, RegPoints, Registration, words_to_fields

VOXEL = 0.05
TRUNC = 0.15


def analytic_layer(eng, sdf, lo, hi, voxel=VOXEL, trunc=TRUNC, weight=5.0, band=None):
    """Blocks lo..hi (inclusive block indices) filled with clamp(sdf(centre), +-trunc); voxels farther than `band` from the
    surface stay unobserved when band is given."""
    idx, vox = [], []
    lin = np.arange(4096)
    loc = np.stack([lin % 16, (lin // 16) % 16, lin // 256], axis=1)
    for bz in range(lo[2], hi[2] + 1):
        for by in range(lo[1], hi[1] + 1):
            for bx in range(lo[0], hi[0] + 1):
                g = np.array([bx, by, bz]) * 16 + loc
                c = ((g.astype(np.float32) + np.float32(0.5)) * np.float32(voxel)).astype(np.float32)
                d = sdf(c.astype(np.float64))
                w = np.full(4096, weight, np.float32)
                if band is not None:
                    w[np.abs(d) > band] = 0.0
                dd = np.clip(d, -trunc, trunc).astype(np.float32)
                dd[w == 0] = 0.0
                words = np.zeros((4096, 3), np.uint32)
                words[:, 0] = dd.view(np.uint32)
                words[:, 1] = w.view(np.uint32)
                idx.append([bx, by, bz])
                vox.append(words)
    layer = Layer(eng, voxel, capacity_blocks=max(64, len(idx)))
    layer.upload(np.array(idx, np.int32), np.array(vox, np.uint32))
    return layer


def test_isosurface_vertices_lie_on_an_analytic_plane(oracle):
    x0 = 0.237
    layer = analytic_layer(oracle, lambda c: c[:, 0] - x0, (-1, -1, -1), (0, 0, 0))
    pts = RegPoints.from_isosurface(oracle, layer, min_weight=1.0)
    p = pts.download()
    # the plane cuts exactly one x edge per (y, z) voxel-centre line; cubes exist between the 32 centres per axis -> 32 x 32 lattice
    # points, each shared by up to 4 cubes and merged by the connected mesh
    assert pts.n_mesh_vertices > 4 * 31 * 31 and pts.n_connected_vertices == 32 * 32
    assert np.max(np.abs(p[:, 0] - x0)) < 2e-6
    yz = np.unique(np.round((p[:, 1:3] / VOXEL - 0.5)).astype(int), axis=0)
    assert len(yz) == len(p)  # one vertex per lattice point
    # vertices on the outer ring have no 8 observed neighbours on one side only if the layer ends there: all interior ones interpolate
    assert 30 * 30 <= len(p) <= 32 * 32
    assert np.max(np.abs(p[:, 3])) < 2e-6 and np.allclose(p[:, 4], 5.0)


def test_isosurface_vertices_lie_on_an_analytic_sphere(oracle):
    c0, r = np.array([0.11, -0.07, 0.05]), 0.6
    layer = analytic_layer(oracle, lambda c: np.linalg.norm(c - c0, axis=1) - r, (-1, -1, -1), (0, 0, 0))
    p = RegPoints.from_isosurface(oracle, layer, min_weight=1.0).download()
    rr = np.linalg.norm(p[:, :3].astype(np.float64) - c0, axis=1)
    assert len(p) > 2000
    assert np.max(np.abs(rr - r)) < VOXEL ** 2 / (2 * r) + 1e-5      # chord error of a linear interpolant on a sphere
    assert np.max(np.abs(p[:, 3])) < VOXEL ** 2 / r                     # trilinear TSDF at the vertex is ~0
    # merged at half a voxel: no two survivors share a cell of the proximity grid
    cells = np.round(p[:, :3].astype(np.float64) / (0.5 * VOXEL)).astype(np.int64)
    assert len(np.unique(cells, axis=0)) == len(p)


def test_unobserved_corners_invalidate_a_cube(oracle):
    """getSdfIfValid: a corner with weight <= min_weight kills its cubes; nothing is meshed in an unobserved slab."""
    x0 = 0.237
    layer = analytic_layer(oracle, lambda c: c[:, 0] - x0, (-1, -1, -1), (0, 0, 0), weight=1.0)
    assert RegPoints.from_isosurface(oracle, layer, min_weight=1.0).n == 0      # weight == min_weight is not enough
    assert RegPoints.from_isosurface(oracle, layer, min_weight=0.5).n > 900


def test_surface_obb_of_a_plane_slab(oracle):
    x0 = 0.237
    layer = analytic_layer(oracle, lambda c: c[:, 0] - x0, (-1, -1, -1), (0, 0, 0))
    mn, mx, n = layer.surface_obb()
    # voxels with |d| <= one voxel: centres 0.225 (d = -0.012) and 0.275 (d = 0.038), 2 per line; box grown by half a voxel
    assert n == 2 * 32 * 32
    assert np.allclose(mn, [0.225 - 0.025, -0.8, -0.8], atol=1e-6) and np.allclose(mx, [0.275 + 0.025, 0.8, 0.8], atol=1e-6)
    empty = Layer(oracle, VOXEL)
    mn, mx, n = empty.surface_obb()
    assert n == 0 and np.all(np.isinf(mn)) and np.all(np.isinf(mx))


def test_esdf_of_a_plane_is_the_distance_along_the_axis(oracle):
    x0 = 0.237
    layer = analytic_layer(oracle, lambda c: c[:, 0] - x0, (-2, -1, -1), (1, 0, 0))
    esdf = layer.esdf(max_distance_m=1.0, min_distance_m=0.1)
    idx, vox = esdf.download()
    d, w, rgba = words_to_fields(vox)
    _, tv = layer.download()
    td, tw, _ = words_to_fields(tv)
    lin = np.arange(4096)
    loc = np.stack([lin % 16, (lin // 16) % 16, lin // 256], axis=1)
    centres_x = ((idx[:, None, 0] * 16 + loc[None, :, 0]).astype(np.float64) + 0.5) * VOXEL
    true = centres_x - x0
    fixed = rgba[..., 3] == 1
    assert np.array_equal(fixed, np.abs(td) < 0.1)
    assert np.array_equal(d[fixed], td[fixed])                       # the fixed band is the TSDF itself
    assert np.all(w == 1.0)
    inside = np.abs(true) < 1.0 - VOXEL
    assert np.max(np.abs(d[inside] - true[inside])) < 1e-5           # straight propagation: exact up to float accumulation
    assert np.all(np.abs(d) <= 1.0 + 1e-6) and np.all(np.sign(d) == np.sign(true))
    far = np.abs(true) > 1.0 + VOXEL
    assert np.all(np.abs(d[far]) == 1.0)                              # beyond the maximum: +- default distance


def test_esdf_is_quasi_euclidean_around_a_sphere(oracle):
    c0, r = np.array([0.0, 0.0, 0.0]), 0.4
    layer = analytic_layer(oracle, lambda c: np.linalg.norm(c - c0, axis=1) - r, (-2, -2, -2), (1, 1, 1))
    esdf = layer.esdf(max_distance_m=1.5, min_distance_m=0.1)
    idx, vox = esdf.download()
    d, w, _ = words_to_fields(vox)
    lin = np.arange(4096)
    loc = np.stack([lin % 16, (lin // 16) % 16, lin // 256], axis=1)
    c = ((idx[:, None, :] * 16 + loc[None, :, :]).astype(np.float64) + 0.5) * VOXEL
    true = np.linalg.norm(c - c0, axis=2) - r
    m = (true > 0.1) & (true < 1.0)
    err = d[m] - true[m]
    # 26-neighbourhood path lengths over-estimate the Euclidean distance by at most ~6.6 % (+ the discretisation of the seed band)
    assert err.min() > -0.5 * VOXEL and np.max(err / true[m]) < 0.12
    assert np.all(d[true < -0.1] < 0)


def test_weighted_sampler_reproduces_the_weights_and_is_deterministic(oracle):
    rng = np.random.default_rng(5)
    n = 200
    pts = np.zeros((n, 5), np.float32)
    pts[:, :3] = rng.normal(size=(n, 3))
    pts[:, 4] = rng.uniform(0.0, 10.0, n).astype(np.float32)
    pts[7, 4] = 0.0                                                   # never drawn
    layer = Layer(oracle, 0.1)
    g = Registration(oracle, RegPoints(oracle, pts), layer)
    g.draw_samples(400000, seed=1)
    s1 = g.get_samples()
    g.draw_samples(400000, seed=1)
    assert np.array_equal(s1, g.get_samples())
    g.draw_samples(400000, seed=2)
    assert not np.array_equal(s1, g.get_samples())
    assert s1.max() < n and 7 not in s1
    freq = np.bincount(s1, minlength=n) / len(s1)
    want = pts[:, 4].astype(np.float64) / pts[:, 4].astype(np.float64).sum()
    assert np.max(np.abs(freq - want)) < 4 * np.sqrt(want.max() / len(s1))
    # the stored draws are what an evaluation without explicit indices uses
    r1, _, _ = g.evaluate(np.zeros(4), np.zeros(4), jacobians=False)
    assert len(r1) == 400000
