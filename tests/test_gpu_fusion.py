"""Parity tests proper: the HIP TSDF engine against the CPU oracle, both through the C ABI.

Bar (north_star): voxel / block indices bit-exact, distances and weights within 1e-4.  The engine applies
updates to every voxel in the oracle's single-threaded order, so the tests also report (and for the
configurations below require) bit-identical distance and weight words.
"""
import numpy as np
import pytest

from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, Integrator, CoxError, words_to_fields
from util import run_frames, compare_layers, compare_stats, TOL

pytestmark = pytest.mark.gpu

IDENT = np.array([1, 0, 0, 0, 0, 0, 0], np.float32)


def _both(hip, oracle, **kw):
    la, ia, sa = run_frames(hip, **kw)
    lb, ib, sb = run_frames(oracle, **kw)
    return (la, ia, sa), (lb, ib, sb)


def test_single_ray_known_answer(hip, oracle):
    """Same hand-checkable ray as tests/test_oracle_kat.py::test_single_generic_ray_simple_integrator_values."""
    out = []
    for eng in (hip, oracle):
        cfg = eng.default_config(default_truncation_distance=0.3, use_const_weight=1, min_ray_length_m=0.1, max_ray_length_m=5.0)
        layer = Layer(eng, 0.1, capacity_blocks=256)
        integ = Integrator(eng, layer, cfg, "simple")
        T = IDENT.copy()
        T[4:] = (0.05, 0.07, 0.03)
        integ.integrate_points(T, np.array([[0.98, 0.34, 0.19]], np.float32), np.array([[10, 20, 30, 255]], np.uint8))
        out.append((layer, integ.last_stats()))
    rep = compare_layers(out[0][0], out[1][0])
    compare_stats([out[0][1]], [out[1][1]])
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0


def test_axis_aligned_ray_quirk_matches(hip, oracle):
    """ray.y == ray.z == 0: the -inf / NaN DDA quirk of the reference must be reproduced by the kernel."""
    out = []
    for eng in (hip, oracle):
        cfg = eng.default_config(default_truncation_distance=0.3, use_const_weight=1, min_ray_length_m=0.1, max_ray_length_m=5.0)
        layer = Layer(eng, 0.1, capacity_blocks=256)
        integ = Integrator(eng, layer, cfg, "simple")
        T = IDENT.copy()
        T[4:] = (0.05, 0.05, 0.05)
        integ.integrate_points(T, np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, -1.0, 0.0]], np.float32), None)
        out.append((layer, integ.last_stats()))
    rep = compare_layers(out[0][0], out[1][0])
    compare_stats([out[0][1]], [out[1][1]])
    assert rep["bitexact_d"] and rep["bitexact_w"]


@pytest.mark.parametrize("method", ["merged", "simple"])
@pytest.mark.parametrize("voxel", [0.10, 0.05])
def test_subsampled_frames_parity(hip, oracle, method, voxel):
    (la, _, sa), (lb, _, sb) = _both(hip, oracle, method=method, voxel=voxel, frames=[0, 1, 2, 40], subsample=7, capacity_blocks=4096)
    compare_stats(sa, sb)
    rep = compare_layers(la, lb)
    print(method, voxel, rep)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep


@pytest.mark.parametrize("method,voxel", [("merged", 0.05), ("merged", 0.10), ("simple", 0.10)])
def test_full_frames_parity(hip, oracle, method, voxel):
    """BASELINE configs[0]/[1] input shape: full 640x480 frames (307 200 points each)."""
    (la, _, sa), (lb, _, sb) = _both(hip, oracle, method=method, voxel=voxel, frames=[0, 1, 2], capacity_blocks=8192)
    compare_stats(sa, sb)
    rep = compare_layers(la, lb)
    print(method, voxel, rep, sa[-1])
    assert rep["n_diff_w"] == 0 and rep["err_d"] <= TOL


def test_simple_full_frame_5cm_parity(hip, oracle):
    """Heaviest semantics at the headline resolution: ~2.5e7 (ray, voxel) updates in one frame."""
    (la, _, sa), (lb, _, sb) = _both(hip, oracle, method="simple", voxel=0.05, frames=[3], capacity_blocks=8192)
    compare_stats(sa, sb)
    rep = compare_layers(la, lb)
    print(rep, sa)
    assert rep["err_d"] <= TOL and rep["n_diff_w"] == 0


@pytest.mark.parametrize("select", ["COX_APPLY=pieces", "COX_APPLY=records", "COX_PARTITION=records", "COX_PARTITION=records,COX_BUCKETS=1",
                                    "COX_PARTITION=pieces", "COX_BUCKETS=0",
                                    # round 3, piece partition: large tiles classified chunk by chunk (k_big_tiles / k_big_classify; chunks of
                                    # 1024 records so that test-sized clouds have tiles of more than two chunks), a wave per small tile
                                    # (k_apply_wave<256> / <512>), and each of the two switched off
                                    "COX_PARTITION=pieces,COX_BIG_CHUNK=1024,COX_WAVE_TILE_MAX=256", "COX_PARTITION=pieces,COX_BIG_CHUNK=1024",
                                    "COX_PARTITION=pieces,COX_SPLIT_TILES=0", "COX_PARTITION=pieces,COX_APPLY_WAVE=0"])
@pytest.mark.parametrize("voxel,sub", [(0.10, 1), (0.05, 1), (0.02, 3)])
def test_alternative_layer_update_paths_are_bit_identical(hip, oracle, monkeypatch, select, voxel, sub):
    """Environment switches (read when an integrator is created) select the other implementations of the layer-update half of
    a merged frame: COX_APPLY=pieces (k_touch_pieces / k_apply_pieces: (ray, tile) runs instead of records), COX_APPLY=records
    (full record sort + per-record kernels, round 1), COX_PARTITION=records / pieces (the tile apply fed by partitioned records or
    by sorted and expanded pieces, whichever is not the default at that voxel size), COX_BUCKETS=1 / 0 (records partitioned in one
    pass into 4096 buckets -- at 2 cm ~1 500 touched blocks, six tiles per bucket -- or by the whole tile id in two).  Same
    oracle, same bar."""
    env = dict(kv.split("=") for kv in select.split(","))
    for name, value in env.items():
        monkeypatch.setenv(name, value)
    la, _, sa = run_frames(hip, method="merged", voxel=voxel, frames=[0, 1, 2, 40], subsample=sub, capacity_blocks=40000 if voxel < 0.05 else 8192)
    for name in env:
        monkeypatch.delenv(name)
    lb, _, sb = run_frames(oracle, method="merged", voxel=voxel, frames=[0, 1, 2, 40], subsample=sub, capacity_blocks=40000 if voxel < 0.05 else 8192)
    compare_stats(sa, sb)
    rep = compare_layers(la, lb)
    print(select, voxel, rep)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep


def test_large_tiles_are_split_and_still_bit_identical(hip, oracle, monkeypatch):
    """Every ray of a frame leaves from the sensor, so the tiles around it hold a large part of the frame's records; with the piece
    partition their classification is cut into chunks for the whole chip (k_big_tiles / k_big_classify) and their workgroup starts
    from the chunks' sums.  The path must actually be taken here (cox_integrator_update_stats), with and without hard voxels in such
    tiles (second pass over the same frames: the tiles around the sensor then hold observed voxels), and equal the oracle bit for bit."""
    monkeypatch.setenv("COX_PARTITION", "pieces")
    monkeypatch.setenv("COX_BIG_CHUNK", "1024")
    frames = [0, 1, 2, 0, 1, 2]
    la, ia, sa = run_frames(hip, method="merged", voxel=0.02, frames=frames, subsample=3, capacity_blocks=40000)
    split = ia.update_stats()
    monkeypatch.delenv("COX_PARTITION")
    monkeypatch.delenv("COX_BIG_CHUNK")
    assert split["split_tiles"] > 0 and split["chunks"] > 2 * split["split_tiles"], split
    lb, _, sb = run_frames(oracle, method="merged", voxel=0.02, frames=frames, subsample=3, capacity_blocks=40000)
    compare_stats(sa, sb)
    rep = compare_layers(la, lb)
    print(split, rep)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep


def test_fine_voxels_2cm(hip, oracle):
    (la, _, sa), (lb, _, sb) = _both(hip, oracle, method="merged", voxel=0.02, frames=[0, 5], subsample=3, capacity_blocks=40000)
    compare_stats(sa, sb)
    print(compare_layers(la, lb))


@pytest.mark.parametrize("overrides", [
    dict(use_const_weight=0),                                        # 1/z^2 weights
    dict(voxel_carving_enabled=0),                                   # rays start trunc in front of the surface
    dict(use_weight_dropoff=0),
    dict(use_sparsity_compensation_factor=1, sparsity_compensation_factor=20.0, max_weight=1000.0),  # coxgraph_client.yaml:58,64-66
    dict(max_ray_length_m=2.5),                                      # many clearing rays
    dict(max_ray_length_m=2.5, allow_clear=0),                       # ... which are then dropped instead
    dict(enable_anti_grazing=1),
    dict(max_weight=3.0),                                            # weight saturation inside the sequence
])
def test_config_variants(hip, oracle, overrides):
    for method in ("merged", "simple"):
        if method == "simple" and "enable_anti_grazing" in overrides:
            continue
        (la, _, sa), (lb, _, sb) = _both(hip, oracle, method=method, voxel=0.10, frames=[0, 1, 2, 3], subsample=5, capacity_blocks=4096,
                                          cfg_overrides=overrides)
        compare_stats(sa, sb)
        rep = compare_layers(la, lb)
        print(method, overrides, rep)


def test_nan_and_noise_inputs(hip, oracle):
    (la, _, sa), (lb, _, sb) = _both(hip, oracle, method="merged", voxel=0.05, frames=[0, 1], subsample=3, capacity_blocks=8192,
                                      nan_fraction=0.02, noise=True)
    compare_stats(sa, sb)
    compare_layers(la, lb)


def test_freespace_points_and_ragged_inputs(hip, oracle):
    rng = np.random.default_rng(11)
    for n in (0, 1, 63, 64, 65, 1023, 1024, 1025, 5000):
        pts = (rng.uniform(-1, 1, (n, 3)) * np.array([3.0, 2.0, 3.0]) + np.array([0, 0, 3.5])).astype(np.float32)
        rgba = rng.integers(0, 256, (n, 4)).astype(np.uint8)
        out = []
        for eng in (hip, oracle):
            cfg = eng.default_config(**synth.integrator_overrides(0.10))
            layer = Layer(eng, 0.10, capacity_blocks=4096)
            integ = Integrator(eng, layer, cfg, "merged")
            T = np.array([0.9238795, 0.0, 0.3826834, 0.0, 0.3, -0.2, 1.0], np.float32)
            integ.integrate_points(T, pts, rgba, freespace=(n % 2 == 1))
            integ.integrate_points(T, pts[: n // 2], None)
            out.append((layer, integ.last_stats()))
        compare_layers(out[0][0], out[1][0])
        compare_stats([out[0][1]], [out[1][1]])


def test_all_points_invalid_allocates_nothing(hip):
    cfg = hip.default_config(**synth.integrator_overrides(0.05))
    layer = Layer(hip, 0.05, capacity_blocks=64)
    integ = Integrator(hip, layer, cfg, "merged")
    pts = np.full((1000, 3), 0.01, np.float32)  # closer than min_ray_length_m
    integ.integrate_points(IDENT, pts, None)
    st = integ.last_stats()
    assert st["n_valid"] == 0 and st["n_rays"] == 0 and st["n_updates"] == 0 and layer.stats()[0] == 0


def test_pool_exhaustion_is_reported(hip):
    """A pinned pool that runs out is an error in the frame that fills it AND in every later frame that meets one of the
    blocks it could not store (never a partial map behind COX_OK); the block counter never passes the capacity."""
    cfg = hip.default_config(**synth.integrator_overrides(0.05))
    layer = Layer(hip, 0.05, capacity_blocks=8)
    layer.set_auto_grow(False)
    integ = Integrator(hip, layer, cfg, "merged")
    T, pts, rgba, _ = synth.make_frame(0)
    for _ in range(3):
        with pytest.raises(CoxError) as e:
            integ.integrate_points(T, pts[::4], rgba[::4])
        assert e.value.status == -4
        n = C_uint64_blocks(layer)
        assert n <= 8
    # the same through a device upload: keys left without storage stay an error
    idx = np.array([[100 + k, 0, 0] for k in range(4)], np.int32)
    with pytest.raises(CoxError) as e:
        layer.upload(idx, np.zeros((4, 4096, 3), np.uint32))
    assert e.value.status == -4
    # growing the pool heals the layer: the rebuilt table drops the keys without storage
    layer.reserve(4096)
    assert layer.capacity() == 4096
    integ.integrate_points(T, pts[::4], rgba[::4])
    integ.integrate_points(T, pts[::4], rgba[::4])


def C_uint64_blocks(layer):
    import ctypes as C
    n, b = C.c_uint64(), C.c_uint64()
    layer.eng.fn("layer_stats")(layer.h, C.byref(n), C.byref(b))  # status ignored: the layer is in its error state
    return int(n.value)


@pytest.mark.parametrize("method", ["merged", "fast"])
def test_layer_grows_like_the_reference_map(hip, oracle, method):
    """voxblox's Layer grows without bound: starting from a pool of 8 blocks the engine doubles it as it fills (first frame
    sized by hand, as a frame cannot be replayed) and ends with the oracle's map, bit for bit."""
    kw = dict(method=method, voxel=0.05, frames=[0, 10, 20, 30, 40, 50, 60, 70], subsample=3)
    lb, _, sb = run_frames(oracle, **kw)
    first = sb[0]["n_new_blocks"]
    la, _, sa = run_frames(hip, capacity_blocks=first + 1, **kw)
    compare_stats(sa, sb)
    rep = compare_layers(la, lb)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep
    assert la.capacity() > first + 1 and la.stats()[0] == lb.stats()[0]


def test_unsupported_configurations_are_refused_loudly(hip):
    layer = Layer(hip, 0.05, capacity_blocks=64)
    with pytest.raises(CoxError) as e:   # a wall-clock budget cannot be reproduced
        Integrator(hip, layer, hip.default_config(max_integration_time_s=0.01), "fast")
    assert e.value.status == -6
    with pytest.raises(CoxError) as e:   # the exact-set variant exists in the oracle only
        Integrator(hip, layer, hip.default_config(fast_exact_sets=1), "fast")
    assert e.value.status == -6
    with pytest.raises(CoxError):
        Integrator(hip, layer, hip.default_config(integration_order_mode=1), "merged")


FAST1 = dict(integrator_threads=1)  # the reference's fast integrator is only reproducible single-threaded


@pytest.mark.parametrize("voxel", [0.10, 0.05])
def test_fast_subsampled_frames_parity(hip, oracle, voxel):
    """`method: fast` (what coxgraph's tsdf_server_*.yaml configure): start-voxel dedup and early termination through
    the two lossy 2^20-slot sets, reproduced bit for bit."""
    (la, _, sa), (lb, _, sb) = _both(hip, oracle, method="fast", voxel=voxel, frames=[0, 1, 2, 40, 41], subsample=5, capacity_blocks=4096, cfg_overrides=FAST1)
    compare_stats(sa, sb)
    rep = compare_layers(la, lb)
    print(voxel, rep, sa[-1])
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep
    assert sa[0]["n_rays"] < sa[0]["n_valid"]            # the start set skipped points
    assert sa[0]["n_updates"] < 40 * sa[0]["n_rays"]     # and rays stopped early


def test_fast_full_frames_5cm_parity(hip, oracle):
    (la, _, sa), (lb, _, sb) = _both(hip, oracle, method="fast", voxel=0.05, frames=[0, 1, 2, 3], capacity_blocks=8192, cfg_overrides=FAST1)
    compare_stats(sa, sb)
    rep = compare_layers(la, lb)
    print(rep, sa)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep


@pytest.mark.parametrize("overrides", [
    dict(clear_checks_every_n_frames=3),                 # the sets survive from frame to frame
    dict(max_consecutive_ray_collisions=0),
    dict(max_consecutive_ray_collisions=6),
    dict(start_voxel_subsampling_factor=1.0),
    dict(start_voxel_subsampling_factor=4.0, voxel_carving_enabled=0),
    dict(use_const_weight=0, use_weight_dropoff=0, allow_clear=0),
])
def test_fast_config_variants(hip, oracle, overrides):
    ov = dict(FAST1)
    ov.update(overrides)
    (la, _, sa), (lb, _, sb) = _both(hip, oracle, method="fast", voxel=0.05, frames=[0, 1, 2, 3, 4], subsample=3, capacity_blocks=8192, cfg_overrides=ov)
    compare_stats(sa, sb)
    rep = compare_layers(la, lb)
    print(overrides, rep, sa[-1])
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep


def test_fast_known_answers(hip, oracle):
    """two identical points -> the second is skipped; a slightly offset ray stops after
    max_consecutive_ray_collisions + 1 already-seen voxels."""
    cfg_kw = dict(default_truncation_distance=0.3, use_const_weight=1, min_ray_length_m=0.1, max_ray_length_m=5.0, integrator_threads=1)
    pts = np.array([[1.05, 0.05, 0.05], [1.05, 0.05, 0.05], [1.05, 0.06, 0.05]], np.float32)
    T = IDENT.copy()
    T[4:] = [0.05, 0.05, 0.05]
    res = []
    for eng in (hip, oracle):
        layer = Layer(eng, 0.1, capacity_blocks=64)
        integ = Integrator(eng, layer, eng.default_config(**cfg_kw), "fast")
        integ.integrate_points(T, pts - T[4:], None)
        res.append((layer, integ.last_stats()))
    (la, sa), (lb, sb) = res
    assert {k: sa[k] for k in ("n_valid", "n_rays", "n_updates")} == {k: sb[k] for k in ("n_valid", "n_rays", "n_updates")}
    assert sa["n_valid"] == 3 and sa["n_rays"] < 3      # the repeated point does not start a second ray
    rep = compare_layers(la, lb)
    assert rep["bitexact_d"] and rep["bitexact_w"]


def test_layer_wire_roundtrip_merge_clear(hip, oracle):
    la, _, _ = run_frames(hip, method="merged", voxel=0.10, frames=[0], subsample=9, capacity_blocks=2048)
    idx, vox = la.download()
    assert la.stats() == (len(idx), len(idx) * 49152)
    # oracle deserialises what the engine serialised, and vice versa, identically
    lo = Layer(oracle, 0.10)
    lo.upload(idx, vox)
    i2, v2 = lo.download()
    assert np.array_equal(idx, i2) and np.array_equal(vox, v2)
    lh = Layer(hip, 0.10, capacity_blocks=2048)
    lh.upload(idx[::-1].copy(), vox[::-1].copy())  # arbitrary block order in the message
    i3, v3 = lh.download()
    assert np.array_equal(idx, i3) and np.array_equal(vox, v3)
    # merge action == mergeVoxelAIntoVoxelB on both sides
    lh.upload(idx, vox, action=1)
    lo.upload(idx, vox, action=1)
    rep = compare_layers(lh, lo)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0
    # reset action and removeAllBlocks
    lh.upload(idx[:3], vox[:3], action=2)
    assert lh.stats()[0] == 3
    lh.clear()
    assert lh.stats() == (0, 0)
    # a cleared layer integrates like a new one
    cfg = hip.default_config(**synth.integrator_overrides(0.10))
    T, pts, rgba, _ = synth.make_frame(0)
    Integrator(hip, lh, cfg, "merged").integrate_points(T, pts[::9], rgba[::9])
    compare_layers(lh, la)


@pytest.mark.parametrize("method", ["merged", "fast"])
def test_depth_front_end_matches_point_path(hip, method):
    import torch
    T, pts, rgba, depth = synth.make_frame(2, nan_fraction=0.02)
    cfg = hip.default_config(**synth.integrator_overrides(0.05))
    l1 = Layer(hip, 0.05, capacity_blocks=8192)
    l2 = Layer(hip, 0.05, capacity_blocks=8192)
    Integrator(hip, l1, cfg, method).integrate_points(T, pts, rgba)
    d = torch.from_numpy(depth).cuda()
    c = torch.from_numpy(synth.frame_colors()).cuda()
    i2 = Integrator(hip, l2, cfg, method)
    i2.integrate_depth_dev(T, d.data_ptr(), c.data_ptr(), 640, 480, synth.INTRINSICS[(640, 480)])
    i2.sync()
    rep = compare_layers(l2, l1)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0


@pytest.mark.parametrize("method", ["merged", "fast"])
def test_results_are_reproducible_run_to_run(hip, method):
    a, _, _ = run_frames(hip, method=method, voxel=0.05, frames=[0, 1, 2], capacity_blocks=8192)
    b, _, _ = run_frames(hip, method=method, voxel=0.05, frames=[0, 1, 2], capacity_blocks=8192)
    ia, va = a.download()
    ib, vb = b.download()
    assert np.array_equal(ia, ib) and np.array_equal(va, vb)


def test_config4_shape_1280x720_2cm(hip, oracle):
    """BASELINE configs[3] input shape: 1280x720 depth (921 600 points), 2 cm voxels, rays 0.1-3 m (tsdf_server_rs.yaml:12-17)."""
    (la, _, sa), (lb, _, sb) = _both(hip, oracle, method="merged", voxel=0.02, frames=[0, 30], capacity_blocks=65536, wh=(1280, 720))
    compare_stats(sa, sb)
    rep = compare_layers(la, lb)
    print(rep, sa[-1])
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0


def test_one_centimetre_voxels(hip, oracle):
    """BASELINE configs[4] voxel size (extrapolated ray limits) at the full 640x480: ~1e4 blocks, 3e6 distinct voxels and 2.5e7
    (ray, voxel) updates in the frame."""
    (la, _, sa), (lb, _, sb) = _both(hip, oracle, method="merged", voxel=0.01, frames=[0], subsample=1, capacity_blocks=131072)
    compare_stats(sa, sb)
    rep = compare_layers(la, lb)
    print(rep, sa[-1])
    assert rep["bitexact_d"] and rep["bitexact_w"]


def test_hoisted_reciprocal_division_is_correctly_rounded(hip):
    """k_bundle_merge's division (reciprocal refined off the dependent chain) must equal IEEE '/' bit for bit."""
    import ctypes as C
    bad = C.c_uint64(123)
    for seed in (1, 2, 3):
        hip.check(hip.fn("selftest_division")(C.c_int(0), C.c_uint64(1 << 28), C.c_uint64(seed), C.byref(bad)), "selftest_division")
        assert bad.value == 0, bad.value


def test_layer_merge_matches_oracle(hip, oracle):
    """mergeLayerAintoLayerB (map_server.cpp:67-69, submap_collection.cpp:31-33): same grid, and resampled through a rigid
    transform (trilinear / nearest fallback), on real fused submaps."""
    out = {}
    for name, eng in (("hip", hip), ("oracle", oracle)):
        kw = dict(capacity_blocks=8192) if eng is hip else {}
        a, _, _ = run_frames(eng, method="merged", voxel=0.10, frames=[0, 10, 20], subsample=5, **kw)
        b, _, _ = run_frames(eng, method="merged", voxel=0.10, frames=[15, 25], subsample=5, **kw)
        c, _, _ = run_frames(eng, method="merged", voxel=0.10, frames=[40], subsample=5, **kw)
        b.merge_from(a)                                   # same grid
        yaw = np.radians(10.0)
        T = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2), 0.13, -0.07, 0.02], np.float32)
        c.merge_from(a, T)                                # resampled
        out[name] = (b, c)
    for k, what in ((0, "same grid"), (1, "resampled")):
        rep = compare_layers(out["hip"][k], out["oracle"][k], tol=1e-6)
        print(what, rep)
        assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, (what, rep)


@pytest.mark.parametrize("method,sub", [("merged", 1), ("merged", 5), ("fast", 1), ("simple", 11)])
def test_async_device_path_equals_the_synchronous_one(hip, method, sub):
    """bench.py's path: frames resident on the GPU, enqueued back to back without waiting (several frames in flight on
    the engine's streams, buffers rotating) -- must give the layer the frame-by-frame synchronous host path gives."""
    import torch
    n_frames = 48
    frames = [synth.make_frame(t) for t in range(n_frames)]
    cfg = hip.default_config(integrator_threads=1, **synth.integrator_overrides(0.05))
    a, b = Layer(hip, 0.05, capacity_blocks=16384), Layer(hip, 0.05, capacity_blocks=16384)
    ia, ib = Integrator(hip, a, cfg, method), Integrator(hip, b, cfg, method)
    dev = [(T, torch.from_numpy(np.ascontiguousarray(p[::sub])).cuda(), torch.from_numpy(np.ascontiguousarray(c[::sub])).cuda()) for T, p, c, _ in frames]
    torch.cuda.synchronize()
    for T, xyz, rgba in dev:                      # no sync between frames
        ia.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), xyz.shape[0])
    ia.sync()
    for T, p, c, _ in frames:                     # host buffers, one frame at a time
        ib.integrate_points(T, p[::sub], c[::sub])
    assert ia.last_stats() == ib.last_stats()
    ja, va = a.download()
    jb, vb = b.download()
    assert np.array_equal(ja, jb) and np.array_equal(va, vb)


@pytest.mark.parametrize("env", [{"COX_STREAMS": "2"}, {"COX_STREAMS": "4s"}, {"COX_STREAMS": "3p"}, {"COX_STREAMS": "6"}, {"COX_SUBMIT_THREAD": "0"}, {"COX_STREAMS": "6", "COX_SUBMIT_THREAD": "0"},
                                 {"COX_GRAPH": "1"}, {"COX_STREAM_MAP": "001234"}, {"COX_TILE": "9"}, {"COX_TILE": "9", "COX_PARTITION": "pieces"},
                                 {"COX_PARTITION": "pieces"}])
@pytest.mark.parametrize("method", ["merged", "simple"])
def test_pipeline_configurations_give_the_same_layer(hip, monkeypatch, env, method):
    """The six stages of a frame on 2 / 4 / 6 streams (or any other map), with or without the submission thread, replayed as
    HIP graphs: 40 frames enqueued back to back (up to six in flight, every buffer set reused several times) must give,
    bit for bit, the layer the frame-by-frame synchronous path of the default configuration gives."""
    import torch
    sub = 3 if method == "merged" else 17
    frames = [synth.make_frame(t) for t in range(40)]
    cfg = hip.default_config(integrator_threads=1, **synth.integrator_overrides(0.05))
    a, b = Layer(hip, 0.05, capacity_blocks=16384), Layer(hip, 0.05, capacity_blocks=16384)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ia = Integrator(hip, a, cfg, method)          # the environment is read when an integrator is created
    for k in env:
        monkeypatch.delenv(k)
    ib = Integrator(hip, b, cfg, method)
    dev = [(T, torch.from_numpy(np.ascontiguousarray(p[::sub])).cuda(), torch.from_numpy(np.ascontiguousarray(c[::sub])).cuda()) for T, p, c, _ in frames]
    torch.cuda.synchronize()
    for T, xyz, rgba in dev:
        ia.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), xyz.shape[0])
    ia.sync()
    for T, p, c, _ in frames:
        ib.integrate_points(T, p[::sub], c[::sub])
    assert ia.last_stats() == ib.last_stats()
    ja, va = a.download()
    jb, vb = b.download()
    assert np.array_equal(ja, jb) and np.array_equal(va, vb)


@pytest.mark.parametrize("method,voxel,sub", [("merged", 0.05, 2), ("merged", 0.02, 4), ("fast", 0.05, 2)])
def test_long_stream_async_equals_sync(hip, method, voxel, sub):
    """Soak: 600 frames of the trajectory (216 degrees of the camera circle: the map and its pool keep growing) enqueued back to
    back -- every buffer set, hand-over event and submission-thread job reused a hundred times -- against the same frames one
    at a time through the synchronous host path."""
    import torch
    n_frames = 600
    cfg = hip.default_config(integrator_threads=1, **synth.integrator_overrides(voxel))
    cap0 = 128 if voxel >= 0.05 else 4096   # small pools (the first frame alone has to fit): several doublings on the way
    a, b = Layer(hip, voxel, capacity_blocks=cap0), Layer(hip, voxel, capacity_blocks=cap0)
    ia, ib = Integrator(hip, a, cfg, method), Integrator(hip, b, cfg, method)
    chunk = 100
    for c0 in range(0, n_frames, chunk):
        frames = [synth.make_frame(t) for t in range(c0, c0 + chunk)]
        dev = [(T, torch.from_numpy(np.ascontiguousarray(p[::sub])).cuda(), torch.from_numpy(np.ascontiguousarray(c[::sub])).cuda()) for T, p, c, _ in frames]
        torch.cuda.synchronize()
        for T, xyz, rgba in dev:
            ia.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), xyz.shape[0])
        ia.sync()           # (the device tensors of the chunk go away after this)
        for T, p, c, _ in frames:
            ib.integrate_points(T, p[::sub], c[::sub])
    assert ia.last_stats() == ib.last_stats()
    ja, va = a.download()
    jb, vb = b.download()
    assert ja.shape[0] > 200 and np.array_equal(ja, jb) and np.array_equal(va, vb)


def test_async_stream_with_changing_sizes_and_a_capacity_growth(hip, oracle):
    """Frames of very different sizes back to back, one of them larger than the integrator's initial capacity (buffers are
    reallocated in the middle of the stream), an empty one in between; checked against the oracle."""
    import torch
    rng = np.random.default_rng(5)
    clouds = []
    for t in range(14):
        T, p, c, _ = synth.make_frame(3 * t)
        if t == 6:      # 614 400 points > 640 x 480
            _, p2, c2, _ = synth.make_frame(3 * t + 1)
            p, c = np.concatenate([p, p2]), np.concatenate([c, c2])
        elif t == 9:
            p, c = p[:0], c[:0]
        else:
            keep = rng.random(len(p)) < rng.choice([0.002, 0.05, 0.3, 1.0])
            p, c = p[keep], c[keep]
        clouds.append((T, np.ascontiguousarray(p), np.ascontiguousarray(c)))
    cfg_kw = dict(integrator_threads=1, **synth.integrator_overrides(0.10))
    a = Layer(hip, 0.10, capacity_blocks=8192)
    ia = Integrator(hip, a, hip.default_config(**cfg_kw), "merged")
    dev = [(T, torch.from_numpy(p).cuda(), torch.from_numpy(c).cuda()) for T, p, c in clouds]
    torch.cuda.synchronize()
    for T, xyz, rgba in dev:
        ia.integrate_points_dev(T, xyz.data_ptr() if xyz.shape[0] else 0, rgba.data_ptr() if xyz.shape[0] else 0, xyz.shape[0])
    ia.sync()
    b = Layer(oracle, 0.10)
    ib = Integrator(oracle, b, oracle.default_config(**cfg_kw), "merged")
    for T, p, c in clouds:
        ib.integrate_points(T, p, c)
    rep = compare_layers(a, b)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep


def test_two_integrators_share_one_layer(hip, oracle):
    """A client may drive one layer with more than one integrator (e.g. a merged one for the live stream and a fast one for
    recover mode): frame ids and per-frame block ordinals live in the layer, so alternating between them stays exact."""
    res = []
    for eng in (hip, oracle):
        cfg = eng.default_config(integrator_threads=1, **synth.integrator_overrides(0.10))
        layer = Layer(eng, 0.10, capacity_blocks=8192)
        ints = [Integrator(eng, layer, cfg, m) for m in ("merged", "fast", "simple")]
        for t in range(9):
            T, p, c, _ = synth.make_frame(4 * t)
            ints[t % 3].integrate_points(T, p[::9], c[::9])
        res.append(layer)
    rep = compare_layers(res[0], res[1])
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep


def test_pool_exactly_full_and_one_block_short(hip, oracle):
    """A layer created with exactly as many blocks as the frames need works and equals the oracle; with one block less
    the engine reports COX_ERR_POOL_EXHAUSTED (never a silent partial map)."""
    kw = dict(method="merged", voxel=0.05, frames=[0, 20, 40], subsample=3)
    lb, _, _ = run_frames(oracle, **kw)
    need = lb.stats()[0]
    assert need > 50
    la, _, _ = run_frames(hip, capacity_blocks=need, auto_grow=False, **kw)
    rep = compare_layers(la, lb)
    assert rep["bitexact_d"] and rep["bitexact_w"] and la.stats()[0] == need
    with pytest.raises(CoxError) as e:
        run_frames(hip, capacity_blocks=need - 1, auto_grow=False, **kw)
    assert e.value.status == -4


def test_independent_handles_from_several_host_threads(hip):
    """Handles are not thread-safe, but different handles may be driven from different threads (several clients on one
    GPU): three clients fused concurrently give the layers they give one after the other."""
    import threading
    import torch
    cfg = hip.default_config(integrator_threads=1, **synth.integrator_overrides(0.10))
    n_clients, n_frames = 3, 24
    streams = []
    for k in range(n_clients):
        fr = []
        for t in range(n_frames):
            T, p, c, _ = synth.make_frame(2 * t, client=k, n_clients=n_clients)
            fr.append((T, torch.from_numpy(np.ascontiguousarray(p[::3])).cuda(), torch.from_numpy(np.ascontiguousarray(c[::3])).cuda()))
        streams.append(fr)
    torch.cuda.synchronize()

    def fuse(frames, method):
        layer = Layer(hip, 0.10, capacity_blocks=8192)
        integ = Integrator(hip, layer, cfg, method)
        for T, xyz, rgba in frames:
            integ.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), xyz.shape[0])
        integ.sync()
        return layer.download()

    methods = ["merged", "fast", "merged"]
    alone = [fuse(streams[k], methods[k]) for k in range(n_clients)]
    together = [None] * n_clients
    errors = []

    def work(k):
        try:
            together[k] = fuse(streams[k], methods[k])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(n_clients)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for (ia, va), (ib, vb) in zip(alone, together):
        assert np.array_equal(ia, ib) and np.array_equal(va, vb)


# ---- round 3: fast without host round trips, host buffers without waiting, depth frames without a host sync -----------------
@pytest.mark.parametrize("env", [{}, {"COX_FAST_SEQUENTIAL": "1"}, {"COX_FAST_CAP": "8,8"}, {"COX_FAST_CAP": "16,32"}, {"COX_FAST_CAP": "32,32"}, {"COX_FAST_STREAMS": "1"},
                                 {"COX_SUBMIT_THREAD": "0"}])
def test_fast_relaxation_on_the_device_and_its_sequential_fallback(hip, oracle, monkeypatch, env):
    """`fast` decides convergence, list growth and overflow on the device (one persistent relaxation launch per round, no host round
    trip), and a frame the two rounds do not finish is redone by ONE lane running the reference's loop (k_fast_sequential;
    COX_FAST_SEQUENTIAL=1 sends every frame there).  Either way the layer is the oracle's, bit for bit."""
    import torch
    sub = 3
    frames = [synth.make_frame(t) for t in (0, 1, 2, 3, 40, 41)]
    cfg_kw = dict(integrator_threads=1, **synth.integrator_overrides(0.05))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    a = Layer(hip, 0.05, capacity_blocks=8192)
    ia = Integrator(hip, a, hip.default_config(**cfg_kw), "fast")
    for k in env:
        monkeypatch.delenv(k)
    b = Layer(oracle, 0.05, capacity_blocks=8192)
    ib = Integrator(oracle, b, oracle.default_config(**cfg_kw), "fast")
    dev = [(T, torch.from_numpy(np.ascontiguousarray(p[::sub])).cuda(), torch.from_numpy(np.ascontiguousarray(c[::sub])).cuda()) for T, p, c, _ in frames]
    torch.cuda.synchronize()
    for T, xyz, rgba in dev:  # enqueued back to back: the three chains of consecutive frames overlap
        ia.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), xyz.shape[0])
    ia.sync()
    for T, p, c, _ in frames:
        ib.integrate_points(T, p[::sub], c[::sub])
    st = ia.fast_stats()
    print(env, st)
    assert st["frames"] == len(frames)
    assert st["sequential_frames_barrier_gave_up"] == 0, st
    if not env:
        assert st["round1_frames"] >= 1, st       # some ray gets through its 8 candidate steps in most frames of this scene
        assert st["sequential_frames"] <= 2, st   # (a ray that outgrows its round-1 list sends its frame to the sequential kernel: rare)
    if env.get("COX_FAST_SEQUENTIAL"):
        assert st["sequential_frames"] == len(frames), st
    compare_stats([ia.last_stats()], [ib.last_stats()])
    rep = compare_layers(a, b)
    assert rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep


@pytest.mark.parametrize("method", ["merged", "fast", "simple"])
@pytest.mark.parametrize("pinned,select", [(False, ""), (True, ""), (True, "COX_INPUT_STREAM=0"), (True, "COX_H2D=kernel"), (False, "COX_COPY_THREADS=0")])
def test_async_host_entry_equals_the_synchronous_one(hip, monkeypatch, method, pinned, select):
    """cox_integrate_points_async: host buffers, frames not waited for (H2D of frame t+1 on the input stream beside the kernels of
    frame t, six staging sets reused many times), from pageable memory (bounce buffer, copied by the caller and three helper threads)
    and from pinned memory (copied from directly); and the switches: copies on the frame's own stream, a copy kernel instead of the copy
    engine, the bounce copy by the caller alone."""
    import torch
    sub = 2 if method != "simple" else 13
    n_frames = 30
    frames = [synth.make_frame(t) for t in range(n_frames)]
    cfg = hip.default_config(integrator_threads=1, **synth.integrator_overrides(0.05))
    a, b = Layer(hip, 0.05, capacity_blocks=16384), Layer(hip, 0.05, capacity_blocks=16384)
    env = dict(kv.split("=") for kv in select.split(",")) if select else {}
    for name, value in env.items():
        monkeypatch.setenv(name, value)   # (read when the integrator is created / at its first host frame)
    ia = Integrator(hip, a, cfg, method)
    ib = Integrator(hip, b, cfg, method)
    host = []
    for T, p, c, _ in frames:
        xyz, rgba = torch.from_numpy(np.ascontiguousarray(p[::sub])), torch.from_numpy(np.ascontiguousarray(c[::sub]))
        if pinned:
            xyz, rgba = xyz.pin_memory(), rgba.pin_memory()
        host.append((T, xyz, rgba))
    for T, xyz, rgba in host:
        ia.integrate_points_async(T, xyz.data_ptr(), rgba.data_ptr(), xyz.shape[0])
    ia.wait_inputs()
    ia.sync()
    for T, p, c, _ in frames:
        ib.integrate_points(T, p[::sub], c[::sub])
    assert ia.last_stats() == ib.last_stats()
    ja, va = a.download()
    jb, vb = b.download()
    assert np.array_equal(ja, jb) and np.array_equal(va, vb)


@pytest.mark.parametrize("method", ["merged", "fast", "simple"])
def test_depth_stream_without_host_sync(hip, method):
    """Depth frames enqueued back to back: the point count of every frame stays on the device (the mixed visiting order is
    computed from it there).  Frames with 2 % invalid pixels, one frame without a single valid pixel in the middle of the stream;
    against the point path fed with the same points one frame at a time."""
    import torch
    n_frames = 12
    K = synth.INTRINSICS[(640, 480)]
    frames = [synth.make_frame(t, nan_fraction=0.02) for t in range(n_frames)]
    cfg = hip.default_config(integrator_threads=1, **synth.integrator_overrides(0.05))
    a, b = Layer(hip, 0.05, capacity_blocks=16384), Layer(hip, 0.05, capacity_blocks=16384)
    ia, ib = Integrator(hip, a, cfg, method), Integrator(hip, b, cfg, method)
    colors = torch.from_numpy(synth.frame_colors()).cuda()
    depths = [torch.from_numpy(d).cuda() for _, _, _, d in frames]
    empty = torch.zeros((480, 640), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    host_colors = synth.frame_colors()
    for t, ((T, _, _, dh), d) in enumerate(zip(frames, depths)):
        if t % 3 == 2:  # every third frame as HOST images (cox_integrate_depth_async: staged, converted, count on the device)
            ia.integrate_depth_async(T, dh.ctypes.data, host_colors.ctypes.data, 640, 480, K)
        else:
            ia.integrate_depth_dev(T, d.data_ptr(), colors.data_ptr(), 640, 480, K)
        if t == 5:
            ia.integrate_depth_dev(T, empty.data_ptr(), colors.data_ptr(), 640, 480, K)
    ia.sync()
    assert ia.last_stats()["n_points"] == frames[-1][1].shape[0]  # fetched from the device
    for t, (T, p, c, _) in enumerate(frames):
        ib.integrate_points(T, p, c)
        if t == 5:  # (an empty cloud still resets the fast integrator's sets)
            ib.integrate_points(T, np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8))
    sa, sb = ia.last_stats(), ib.last_stats()
    assert sa == sb, (sa, sb)
    ja, va = a.download()
    jb, vb = b.download()
    assert np.array_equal(ja, jb) and np.array_equal(va, vb)


def test_integrators_sharing_a_layer_asynchronously_with_pool_growth(hip, oracle):
    """two or three integrators alternate on ONE layer through the asynchronous device path (frames in flight on each
    integrator's own streams, half of every frame waiting in its submission thread), with a pool so small that it is
    reallocated several times on the way -- by whichever integrator happens to notice.  Frames of different integrators must
    reach the layer in call order and nobody may touch a freed pool: the layer equals the oracle's, bit for bit."""
    import torch
    n_frames, sub = 36, 4
    frames = [synth.make_frame(3 * t) for t in range(n_frames)]
    methods = ("merged", "fast", "merged", "simple")
    dev = [(T, torch.from_numpy(np.ascontiguousarray(p[::sub])).cuda(), torch.from_numpy(np.ascontiguousarray(c[::sub])).cuda()) for T, p, c, _ in frames]
    torch.cuda.synchronize()
    res = []
    for eng in (hip, oracle):
        cfg = eng.default_config(integrator_threads=1, **synth.integrator_overrides(0.05))
        layer = Layer(eng, 0.05, capacity_blocks=64)  # the first frame alone nearly fills it
        ints = [Integrator(eng, layer, cfg, m) for m in methods]
        for t, (T, p, c, _) in enumerate(frames):
            k = (t // 2) % len(ints)  # two frames in a row per integrator, then the next one
            if eng is hip:
                ints[k].integrate_points_dev(T, dev[t][1].data_ptr(), dev[t][2].data_ptr(), dev[t][1].shape[0])
            else:
                ints[k].integrate_points(T, p[::sub], c[::sub])
        for i in ints:
            i.sync()
        res.append(layer)
        if eng is hip:
            assert layer.capacity() > 64  # it grew
    rep = compare_layers(res[0], res[1])
    assert rep["blocks"] > 64 and rep["bitexact_d"] and rep["bitexact_w"] and rep["n_diff_color"] == 0, rep
