"""finishSubmap() on the GPU (a12 / f-2): isosurface registration points, ESDF, surface box, weighted sampler -- HIP engine
against the CPU oracle through the C ABI, plus two checks that do not rest on any restatement: the fused TSDF's zero
crossing lies on the analytic scene of coxgraph_amd/synth.py, and registering a submap against a rigidly displaced copy of
itself recovers the displacement.

Bar: bit-exact (vertex coordinates, distances, weights, ESDF words, box corners, sample indices).  The ESDF relaxation and
the min / max / mesh order are order-free by construction, so there is nothing to tolerate.
"""
import numpy as np
import pytest

from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, RegPoints, Registration, words_to_fields
from util import run_frames, compare_layers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0.10, 0.05])
def submap(request, hip, oracle):
    """A finished submap (frames 0..140 step 10 of the benchmark stream), fused by the HIP engine and by the oracle."""
    voxel = request.param
    kw = dict(method="merged", voxel=voxel, frames=range(0, 150, 10), subsample=2, capacity_blocks=8192)
    lh, _, _ = run_frames(hip, **kw)
    lo, _, _ = run_frames(oracle, **kw)
    rep = compare_layers(lh, lo)
    assert rep["bitexact_d"] and rep["bitexact_w"]
    return voxel, lh, lo


def test_isosurface_points_match_oracle(hip, oracle, submap):
    voxel, lh, lo = submap
    for min_w in (1.0, 1e-4):
        ph = RegPoints.from_isosurface(hip, lh, min_weight=min_w)
        po = RegPoints.from_isosurface(oracle, lo, min_weight=min_w)
        assert (ph.n_mesh_vertices, ph.n_connected_vertices, ph.n) == (po.n_mesh_vertices, po.n_connected_vertices, po.n)
        assert po.n > 1000
        a, b = ph.download(), po.download()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # a different merge threshold changes the set, identically on both sides
    ph = RegPoints.from_isosurface(hip, lh, min_weight=1.0, vertex_proximity_threshold=1e-3)
    po = RegPoints.from_isosurface(oracle, lo, min_weight=1.0, vertex_proximity_threshold=1e-3)
    assert ph.n == po.n > 0 and np.array_equal(ph.download().view(np.uint32), po.download().view(np.uint32))


def test_zero_crossing_lies_on_the_analytic_scene(hip, submap):
    """Physics anchor (not a restatement): every isosurface vertex of the fused map is within half a voxel of the room's
    walls / floor / ceiling or of the sphere that the synthetic depth images were rendered from."""
    voxel, lh, _ = submap
    p = RegPoints.from_isosurface(hip, lh, min_weight=1.0).download()[:, :3].astype(np.float64)
    d_planes = np.minimum(np.abs(p - synth.ROOM_MIN), np.abs(p - synth.ROOM_MAX)).min(axis=1)
    d_sphere = np.abs(np.linalg.norm(p - synth.SPHERE_C, axis=1) - synth.SPHERE_R)
    err = np.minimum(d_planes, d_sphere)
    assert len(p) > 1000
    # room corners and occlusion boundaries (the sphere's silhouette against the wall) smear the TSDF along the rays: 5 % of
    # the vertices may sit further out, none further than a voxel
    q = np.quantile(err, [0.5, 0.9, 0.95, 1.0])
    print("isosurface vertex distance to the analytic scene, quantiles 50/90/95/100 % [voxels]:", q / voxel)
    assert q[2] < 0.5 * voxel and q[3] < 1.0 * voxel, q
    assert q[0] < 0.1 * voxel


@pytest.mark.parametrize("max_d,min_d", [(2.0, None), (4.0, 0.1), (2.0, 0.2)])   # half the truncation; coxgraph_client.yaml:68-69; voxblox defaults
def test_esdf_matches_oracle(hip, oracle, submap, max_d, min_d):
    voxel, lh, lo = submap
    if min_d is None:
        min_d = 1.5 * voxel
    eh = lh.esdf(max_distance_m=max_d, min_distance_m=min_d)
    eo = lo.esdf(max_distance_m=max_d, min_distance_m=min_d)
    rep = compare_layers(eh, eo, tol=0.0)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep
    idx, vox = eh.download()
    d, w, rgba = words_to_fields(vox)
    _, tv = lh.download()
    td, tw, _ = words_to_fields(tv)
    assert np.array_equal(w > 0, tw >= 1e-6)
    fixed = rgba[..., 3] == 1
    assert np.array_equal(d[fixed], td[fixed]) and np.all(np.abs(d) <= max_d)
    free = (w > 0) & ~fixed & (np.abs(d) < max_d)
    if min_d < 3 * voxel:   # else every observed voxel lies inside the fixed band (the TSDF is clamped at 3 voxels): ESDF == TSDF
        assert free.sum() > 1000
        assert np.all(np.sign(d[free]) == np.sign(td[free]))
    else:
        assert free.sum() == 0 and np.array_equal(d[w > 0], td[w > 0])


def test_surface_obb_matches_oracle(hip, oracle, submap):
    _, lh, lo = submap
    a, b = lh.surface_obb(), lo.surface_obb()
    assert a[2] == b[2] > 1000
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.all(a[0] >= synth.ROOM_MIN - 0.2) and np.all(a[1] <= synth.ROOM_MAX + 0.2)


def test_weighted_sampler_matches_oracle(hip, oracle, submap):
    _, lh, lo = submap
    ph = RegPoints.from_isosurface(hip, lh, min_weight=1.0)
    po = RegPoints.from_isosurface(oracle, lo, min_weight=1.0)
    gh, go = Registration(hip, ph, lh), Registration(oracle, po, lo)
    n_res = int(0.3 * po.n)   # sampling_ratio 0.3 (coxgraph/config/server.yaml:30)
    for seed in (0, 1, 12345678901234567):
        gh.draw_samples(n_res, seed)
        go.draw_samples(n_res, seed)
        sh, so = gh.get_samples(), go.get_samples()
        assert len(so) == n_res and np.array_equal(sh, so)
    w = po.download()[:, 4].astype(np.float64)
    gh.draw_samples(2000000, 7)
    freq = np.bincount(gh.get_samples(), minlength=po.n) / 2000000
    assert np.max(np.abs(freq - w / w.sum())) < 5 * np.sqrt((w / w.sum()).max() / 2000000)


def test_explicit_to_implicit_registration_matches_oracle(hip, oracle, submap):
    """The server's configured constraint (server.yaml:28-31): isosurface vertices of the reference submap against the ESDF
    of the reading submap, points drawn by the weighted sampler."""
    voxel, lh, lo = submap
    kw = dict(method="merged", voxel=voxel, frames=range(70, 220, 10), subsample=2, capacity_blocks=8192)
    rh, _, _ = run_frames(hip, **kw)
    ro, _, _ = run_frames(oracle, **kw)
    ph, po = RegPoints.from_isosurface(hip, lh, 1.0), RegPoints.from_isosurface(oracle, lo, 1.0)
    eh, eo = rh.esdf(max_distance_m=2.0, min_distance_m=1.5 * voxel), ro.esdf(max_distance_m=2.0, min_distance_m=1.5 * voxel)
    gh, go = Registration(hip, ph, eh), Registration(oracle, po, eo)
    n_res = int(0.3 * po.n)
    gh.draw_samples(n_res, 42)
    go.draw_samples(n_res, 42)
    pr, pd = np.zeros(4), np.array([0.05, -0.03, 0.02, np.radians(1.0)])
    a = gh.evaluate(pr, pd)
    b = go.evaluate(pr, pd)
    assert np.count_nonzero(b[0]) > 0.2 * n_res
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    Hh, bh, ch, nh = gh.normal_eq(pr, pd)
    Ho, bo, co, no = go.normal_eq(pr, pd)
    assert nh == no and np.allclose(Hh, Ho, rtol=1e-9, atol=1e-9 * np.abs(Ho).max()) and abs(ch - co) <= 1e-9 * max(1.0, co)
    # far from alignment the truncated TSDF loses its correspondences' gradient (everything reads +-trunc); the ESDF keeps it
    far = np.array([0.45, -0.3, 0.1, np.radians(4.0)])
    gt = Registration(hip, ph, rh)
    gt.draw_samples(n_res, 42)
    _, b_esdf, _, _ = gh.normal_eq(pr, far)
    _, b_tsdf, _, _ = gt.normal_eq(pr, far)
    assert np.linalg.norm(b_esdf) > np.linalg.norm(b_tsdf)


def test_registration_recovers_a_rigid_displacement(hip, submap):
    """Physics anchor: a submap registered against a copy of itself whose pose is off by (5 cm, -3 cm, 2 cm, 1 deg) is pulled
    back onto itself by the solver the server runs (two-stage optimise, only the registration constraint active)."""
    from coxgraph_amd.posegraph import PoseGraphInterface
    voxel, lh, _ = submap
    pts = RegPoints.from_isosurface(hip, lh, 1.0)
    esdf = lh.esdf(max_distance_m=2.0, min_distance_m=1.5 * voxel)
    g = Registration(hip, pts, esdf)
    g.draw_samples(int(0.3 * pts.n), 3)
    pg = PoseGraphInterface()
    pg.addSubmap(0, [0, 0, 0, 0])
    pg.addSubmap(1, [0.05, -0.03, 0.02, np.radians(1.0)])
    pg.addForceRegistrationConstraint(0, 1, g)
    pg.optimize(enable_registration=True)
    err = pg.getPoseMap()[1]
    assert np.all(np.abs(err[:3]) < 0.1 * voxel) and abs(err[3]) < np.radians(0.05), err
