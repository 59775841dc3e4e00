"""Shared helpers for the parity tests (HIP engine vs. CPU oracle through the same C ABI)."""
import numpy as np

from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, Integrator, words_to_fields

TOL = 1e-4  # north_star: distances / weights within 1e-4, voxel / block indices bit-exact


def run_frames(eng, method, voxel, frames, capacity_blocks=0, subsample=1, cfg_overrides=None, client=0, n_clients=1,
               nan_fraction=0.0, noise=False, wh=(640, 480), auto_grow=True):
    """Integrate synthetic frames; returns (layer, integrator, [stats per frame])."""
    ov = synth.integrator_overrides(voxel)
    if cfg_overrides:
        ov.update(cfg_overrides)
    cfg = eng.default_config(**ov)
    layer = Layer(eng, voxel, capacity_blocks=capacity_blocks)
    if not auto_grow:
        layer.set_auto_grow(False)
    integ = Integrator(eng, layer, cfg, method)
    stats = []
    for t in frames:
        T, pts, rgba, _ = synth.make_frame(t, client, n_clients, w=wh[0], h=wh[1], noise=noise, nan_fraction=nan_fraction)
        integ.integrate_points(T, pts[::subsample], rgba[::subsample])
        stats.append(integ.last_stats())
    return layer, integ, stats


def compare_layers(la, lb, tol=TOL, check_color=True):
    """Bit-exact block set and observed-voxel set; distances/weights within tol. Returns a report dict."""
    ia, va = la.download()
    ib, vb = lb.download()
    assert ia.shape == ib.shape and np.array_equal(ia, ib), f"block index sets differ: {len(ia)} vs {len(ib)}"
    da, wa, ca = words_to_fields(va)
    db, wb, cb = words_to_fields(vb)
    assert np.array_equal(wa > 0, wb > 0), "observed voxel sets differ"
    err_d = float(np.max(np.abs(da - db))) if da.size else 0.0
    err_w = float(np.max(np.abs(wa - wb))) if wa.size else 0.0
    assert err_d <= tol, f"distance error {err_d}"
    assert err_w <= tol * max(1.0, float(np.max(wb)) if wb.size else 1.0), f"weight error {err_w}"
    rep = dict(blocks=len(ia), observed=int((wb > 0).sum()), err_d=err_d, err_w=err_w,
               bitexact_d=bool(np.array_equal(va[..., 0], vb[..., 0])), bitexact_w=bool(np.array_equal(va[..., 1], vb[..., 1])),
               n_diff_d=int((va[..., 0] != vb[..., 0]).sum()), n_diff_w=int((va[..., 1] != vb[..., 1]).sum()),
               n_diff_color=int((va[..., 2] != vb[..., 2]).sum()))
    if check_color:
        dc = np.abs(ca.astype(np.int32) - cb.astype(np.int32))
        rep["err_color"] = int(dc.max()) if dc.size else 0
    return rep


def compare_stats(sa, sb, keys=("n_points", "n_valid", "n_rays", "n_updates", "n_touched_voxels", "n_new_blocks", "max_bundle_points", "max_voxel_updates")):
    for a, b in zip(sa, sb):
        for k in keys:
            assert a[k] == b[k], (k, a[k], b[k])
