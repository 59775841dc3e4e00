#!/usr/bin/env python3
"""Headline benchmark: 640x480 depth frames/s fused into a 5 cm TSDF submap per MI355X (BASELINE.json configs[1]) +
submap registrations/s, one coxgraph client per GPU.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic depth frame: integratePointCloud(T_G_C, points_C, colors) of
307 200 points into the client's submap layer, inputs already resident in HBM.  The reference configures two integrators on
this path: `merged` (coxgraph_sim/launch/experiments/mav_3dplanning_2d3dhouse_two.launch:10) and `fast`
(coxgraph/config/tsdf_server_euroc.yaml:6).  `value` is `merged` (as in rounds 1 and 2); `fast` is a first-class block of its
own under `other_methods` (own roofline, same frames) and both are repeated side by side under `headline`; each method's
GPU / CPU ratio is taken against the CPU restatement of THE SAME method at the reference's 8 integrator threads
(`cpu_baseline` = fast, `cpu_baseline.same_method` = merged).

The timed stream carries NO instrumentation: kernel-class times come from two passes of their own over the same frames (HIP
events around the classes inside a stream that keeps frames in flight, and one frame in flight).  `other_configs` holds short,
bounded runs of the other two shapes BASELINE.json names -- configs[3] (1280x720, 2 cm) and configs[4]'s 1 cm voxels -- each
with a roofline of its own, so that the fine-voxel numbers are driver-run too.  `registration` registers the two halves of the
stream that was run (50 % overlap: SURVEY.md section 8d's frames 0-149 / 75-224 at the default 300 steps).

For N > 1 the driver launches one rank per GPU (torch.distributed over RCCL); clients are independent (weak scaling, no
data-path collective -- SURVEY.md section 8e).  After the fusion timing the ranks run the server's inter-robot leg
(configs[2] / [4]): every client finishes its submap on its own GPU (ESDF + isosurface points), the submaps are exchanged
GPU to GPU (one all-gather of the wire arrays), every constraint (a, b) registers CLIENT a's points against CLIENT b's
distance field, and the packed normal equations are summed with ONE all-reduce per pose-graph evaluation.

One JSON line on rank 0.  `roofline` prices the longest kernel class of a frame against HBM peak with the ALGORITHMIC bytes
of SURVEY.md section 8d (16 B per valid point + 24 B per touched voxel); `cpu_baseline` times the CPU restatement of the
reference's configured integrator (fast, 8 threads) on a bounded sample of the same frames on this box's host cores.
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# An integrator runs its four stages on four HIP streams; the ROCm runtime multiplexes a process's
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with torch's own streams), and two stages that land on
# one queue run back to back: 6.1 k frames/s with 4 queues, 7.5 k with 8.  Read when the runtime initialises, so set it first.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
KERNELS_OF_CLASS = {
    "merge": "k_bundle_merge", "apply": "k_block_starts+k_apply_block", "bundle_hash": "fillBuffer+k_bundle_insert+k_bundle_keys",
    "point_sort": "k_rs_hist/offsets/scatter<11> (points)", "touch_emit": "k_scan_small+k_touch*+k_emit*",
    "record_sort": "k_rs_hist/offsets/scatter<12> (records)", "fast_start": "k_fast_points..k_fast_rays (+ point sort)",
    "fast_visits": "k_fast_visits+visit sort+k_fast_inverse (round 0: 8 candidate steps per ray)", "fast_sweeps": "k_fast_sweep (round 0's relaxation launches)",
    "fast_round1": "k_fast_grow+scan+k_fast_visits+visit sort+k_fast_inverse+k_fast_sweep (round 1: whole walks of the rays that got through)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)  # the 300-frame 30 Hz stream of SURVEY.md section 8d
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--method", default="merged", choices=["merged", "fast", "simple"])
    ap.add_argument("--no-events", action="store_true", help="skip the instrumented pass (HIP-event kernel-class timing with frames in flight); the timed stream itself never carries events")
    ap.add_argument("--other-frames", type=int, default=300, help="frames of the same stream also run through the other method (0 = skip)")
    ap.add_argument("--voxel", type=float, default=0.05)
    ap.add_argument("--width", type=int, default=640, help="depth image width (640 or 1280: the two intrinsics of coxgraph_amd/synth.py)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--other-config-frames", type=int, default=24, help="timed frames of each `other_configs` block (0 = skip)")
    ap.add_argument("--cpu-frames", type=int, default=320, help="frames of the CPU baseline sample (0 = skip); ~10 s of CPU work at the default")
    ap.add_argument("--reg-iters", type=int, default=50)
    ap.add_argument("--pcie-frames", type=int, default=-1, help="timed frames of the PCIe-inclusive legs (default: as many as --steps; 0 = skip)")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--serial", action="store_true", help="sync after every frame (profiling aid: kernel times without cross-frame overlap)")
    ap.add_argument("--no-ramp", action="store_true", help="no untimed clock-ramp frames (PMC passes: every dispatch of the run then belongs to warmup + steps frames)")
    return ap.parse_args()


def respawn_per_gpu(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU (the same command the driver uses) before
    anything touches the GPU, and leave with its exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def cpu_baseline(frames, voxel, n_frames, threads):
    """Oracle 'fast' integrator (the reference's configured method, tsdf_server_euroc.yaml:6,10) on host cores."""
    from coxgraph_amd import synth
    from coxgraph_amd.capi import Engine, Layer, Integrator
    import ctypes as C
    lib = os.path.join(ROOT, "oracle", "libcoxoracle.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build_oracle()
    eng = Engine(lib, "coxo_")
    out = {}
    for method, thr in (("fast", threads), ("merged", threads)):
        cfg = eng.default_config(integrator_threads=thr, **synth.integrator_overrides(voxel))
        layer = Layer(eng, voxel)
        integ = Integrator(eng, layer, cfg, method)
        eng.fn("integrator_set_count_touched")(integ.h, C.c_int(0))
        nf = n_frames if method == "fast" else max(2, n_frames // 2)
        t0 = time.perf_counter()
        for (T, pts, rgba) in frames[:nf]:
            integ.integrate_points(T, pts, rgba)
        dt = time.perf_counter() - t0
        out[method] = (nf / dt, nf, dt)
    return out


def pmc_traffic(method):
    """Whole-frame HBM traffic from the committed PMC passes of the same command (rocprofv3 cannot run inside this
    process): sum over the kernels of (FETCH_SIZE + WRITE_SIZE per launch) x (launches per frame)."""
    path = os.path.join(ROOT, "profiles", f"r03_pmc_traffic_{method}.json")
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", f"r02_pmc_traffic_{method}.json")
    try:
        pm = json.load(open(path))
        return float(pm["bytes_per_frame"]), {"source": os.path.relpath(path, ROOT), "commit": pm.get("commit"), "frames": pm.get("frames"),
                                              "note": pm.get("note")}
    except Exception:
        return None, {"source": None, "note": "no PMC summary committed for this method yet"}


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        respawn_per_gpu(args)
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={env_world}")
    import torch
    import torch.distributed as dist
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)  # one rank per GPU; a rehearsal on a smaller box shares devices
    torch.cuda.set_device(dev_index)
    # "gloo" to rehearse N > 1 on a single-GPU box; "rccl-capi": rendezvous over gloo, the two collectives of the server leg through the
    # engine's own cox_comm_* (RCCL behind the C ABI: what the C++ host uses) instead of torch.distributed
    backend = os.environ.get("COX_DIST_BACKEND", "nccl")
    capi_comm = backend == "rccl-capi"
    if capi_comm:
        backend = "gloo"
    if world > 1:
        import datetime
        tmo = datetime.timedelta(seconds=180)  # a rank that dies must not leave the others waiting for the default 10 min
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index), timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    local_rank = dev_index
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    import coxgraph_amd
    from coxgraph_amd import synth
    from coxgraph_amd.capi import Layer, Integrator, RegPoints, Registration
    eng = coxgraph_amd.load_engine()

    # ---- synthetic stream of this rank's client, resident in HBM before anything is timed ----------
    W, H = args.width, args.height
    if (W, H) not in synth.INTRINSICS:
        raise SystemExit(f"bench.py: no intrinsics for {W}x{H} (have {sorted(synth.INTRINSICS)})")
    n_frames = args.warmup + args.steps
    keep_host = max(args.cpu_frames, (args.warmup + args.steps) if args.pcie_frames < 0 else (args.warmup + args.pcie_frames if args.pcie_frames else 0), 1)

    def make_frames(n, w, h, keep):
        """-> (host frames (the first `keep`), device-resident frames)"""
        host, dev = [], []
        for t in range(n):
            # one client per GPU; with several clients they stand 30 degrees apart on the camera circle, so that neighbouring
            # clients' submaps overlap and the inter-robot constraints have correspondences
            T, pts, rgba, depth = synth.make_frame(t, client=rank, n_clients=(12 if world > 1 else 1), w=w, h=h)
            if t < keep:
                host.append((T, pts, rgba, depth))
            dev.append((T, torch.from_numpy(pts).cuda(), torch.from_numpy(rgba).cuda(), pts.shape[0]))
        torch.cuda.synchronize()
        return host, dev

    host_frames, dev_frames = make_frames(n_frames, W, H, keep_host)
    # The interpreter's cyclic collector stays off from here on and is run by hand between the legs: a full collection that happens to
    # fall into a timed region is a 40 ms hole in 3-30 ms of frames (seen at 10 cm: 4 950 frames/s read as 1 700; which leg it hits
    # depends on how many objects the clock ramp before it allocated, i.e. on how fast the GPU is).
    gc.disable()

    def config_for(voxel):
        return eng.default_config(**synth.integrator_overrides(voxel))
    cfg = config_for(args.voxel)

    def clock_ramp(method, dev, cfg_, voxel, seconds=0.3):
        """Untimed extra frames on a scratch layer so that the timed region starts at running clocks (the reported `warmup`
        frames still go through the measured layer)."""
        if args.no_ramp:
            return
        scratch = Layer(eng, voxel, device=local_rank, capacity_blocks=32768)
        si = Integrator(eng, scratch, cfg_, method)
        t0 = time.perf_counter()
        i = 0
        while time.perf_counter() - t0 < seconds:
            T, xyz, rgba, n = dev[i % len(dev)]
            si.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), n)
            i += 1
            if i % 16 == 0:
                si.sync()
        si.sync()

    def run_stream(method, steps, warmup, dev, cfg_, voxel, events=0, collective=True):
        """warm-up + `steps` timed frames on a fresh layer; barrier + device sync on both sides, MAX over ranks.  events = n > 0:
        HIP-event markers around the kernel classes of every n-th frame (a pass of its own: never the run `value` comes from).
        collective=False: a pass that only rank 0 makes (no barrier, no reduction)."""
        layer = Layer(eng, voxel, device=local_rank, capacity_blocks=32768)
        integ = Integrator(eng, layer, cfg_, method)
        if events:
            integ.set_profiling(events)
        clock_ramp(method, dev, cfg_, voxel)
        for i in range(warmup):
            T, xyz, rgba, n = dev[i]
            integ.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), n)
        integ.sync()
        integ.class_times(reset=True)
        integ.host_time(reset=True)
        torch.cuda.synchronize()
        gc.collect()   # (the collector is off for the whole run -- main() -- and runs here, between the legs)
        if world > 1 and collective:
            dist.barrier()
        t0 = time.perf_counter()
        marks = []
        for i in range(warmup, warmup + steps):
            T, xyz, rgba, n = dev[i]
            integ.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), n)
            if args.serial:
                integ.sync()
            if os.environ.get("BENCH_DEBUG") and (i - warmup) % 10 == 9:
                marks.append(round((time.perf_counter() - t0) * 1e3, 2))
        if marks:
            print("[bench debug] ms at every 10th enqueue:", marks, file=sys.stderr, flush=True)
        integ.sync()
        torch.cuda.synchronize()
        if world > 1 and collective:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1 and collective:
            tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        live = integ.class_times() if events else None
        hms, hn = integ.host_time()
        run_stream.host_ms_per_frame = hms / max(hn, 1)   # caller's thread inside the integrate calls
        return layer, integ, dt, live

    def roofline_of(method, steps, warmup, dev, cfg_, voxel, live, live_every):
        """Algorithmic bytes need per-frame counters, which only a synchronous pass can read: the same frames once more, one in
        flight, on a fresh layer; that pass also gives the kernel classes' undisturbed durations."""
        stats_sum = dict(n_valid=0, n_touched_voxels=0, n_updates=0, n_rays=0)
        stats_max = dict(max_bundle_points=0, max_voxel_updates=0)
        layer2 = Layer(eng, voxel, device=local_rank, capacity_blocks=32768)
        integ2 = Integrator(eng, layer2, cfg_, method)
        integ2.set_profiling(True)
        for i in range(warmup + steps):
            T, xyz, rgba, n = dev[i]
            integ2.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), n)
            integ2.sync()
            if i >= warmup:
                st = integ2.last_stats()
                for k in stats_sum:
                    stats_sum[k] += st[k]
                for k in stats_max:
                    stats_max[k] = max(stats_max[k], st[k])
            elif i == warmup - 1:
                integ2.class_times(reset=True)
        serial = integ2.class_times()
        fast_stats = integ2.fast_stats() if method == "fast" else None
        st = live if live is not None else serial
        # a class may have several regions per frame: price it per FRAME
        frames_timed = max(st["apply"][1], 1)
        per_frame_ms = {k: v[0] / frames_timed for k, v in st.items() if v[1]}
        # B_frame = 16 B per valid point + 24 B per touched voxel (12-B TsdfVoxel read + written)
        alg_bytes = (16.0 * stats_sum["n_valid"] + 24.0 * stats_sum["n_touched_voxels"]) / max(steps, 1)
        dom = max(per_frame_ms, key=per_frame_ms.get)
        ms = per_frame_ms[dom]
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        traffic, src = pmc_traffic(method) if (abs(voxel - 0.05) < 1e-9 and len(dev) and dev[0][3] <= 640 * 480) else (None, {"source": None, "note": "PMC passes are collected for the headline configuration only"})
        roof = {"bound": "hbm", "kernel": KERNELS_OF_CLASS[dom], "kernel_class": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_ratio": (traffic / alg_bytes) if traffic else None,
                "traffic_scope": "whole frame: sum over all kernels of (FETCH_SIZE + WRITE_SIZE) per launch x launches per frame", "traffic_source": src,
                "avg_launch_ms": ms, "algorithmic_bytes_per_launch": alg_bytes, "launches_timed": frames_timed,
                "launch_unit": "one frame's launches of this kernel class (HIP events around the class on its own stream)",
                "timing": (f"HIP events around the classes of every {live_every}. frame in a pass of its own with frames in flight (the timed stream carries none)"
                           if live is not None else "HIP events, one frame in flight"),
                "class_ms_per_frame": per_frame_ms, "class_ms_per_frame_one_in_flight": {k: v[0] / max(serial["apply"][1], 1) for k, v in serial.items() if v[1]},
                "whole_frame_algorithmic_GBps": None,
                "note": "the path is bound by dependent in-order chains and launch latency, not by HBM, at 5 cm (DESIGN.md section 6)"}
        fstats = {k: v / max(steps, 1) for k, v in stats_sum.items()}
        return roof, fstats, stats_max, alg_bytes, fast_stats

    def measure(method, steps, warmup, dev, cfg_, voxel, profile=True, events=True):
        """clean timed run (no instrumentation) -> frames/s; then, unless disabled, the instrumented pass and the one-in-flight pass"""
        layer, integ, dt, _ = run_stream(method, steps, warmup, dev, cfg_, voxel, events=0)
        host_ms = run_stream.host_ms_per_frame
        out = {"value": world * steps / dt, "unit": "frames/s", "frames": steps, "ms_per_step": dt / steps * 1e3, "host_submit_ms_per_frame": host_ms}
        if profile and rank == 0:
            live, every = None, 0
            if events:
                every = 2 if steps < 64 else 8
                _, _, _, live = run_stream(method, steps, warmup, dev, cfg_, voxel, events=every, collective=False)  # (rank 0 only)
            roof, fstats, crit, ab, fst = roofline_of(method, steps, warmup, dev, cfg_, voxel, live, every)
            roof["whole_frame_algorithmic_GBps"] = ab / (dt / steps) / 1e9
            out.update({"roofline": roof, "frame_stats_mean": fstats, "critical_path": crit})
            if fst:
                out["relaxation"] = fst
        return layer, integ, dt, out

    # ---- headline ----------------------------------------------------------------------------------------------------
    layer, integ, dt, head = measure(args.method, args.steps, args.warmup, dev_frames, cfg, args.voxel, profile=not args.no_profile_pass, events=not args.no_events)
    fps = head["value"]
    host_ms = head["host_submit_ms_per_frame"]
    roofline, frame_stats, crit = head.get("roofline"), head.get("frame_stats_mean"), head.get("critical_path")

    # ---- the other method on the same stream, as a block of its own -------------------------------------------------------
    other = None
    if rank == 0 and world == 1 and args.other_frames > 0 and args.method in ("fast", "merged"):
        om = "merged" if args.method == "fast" else "fast"
        nf = min(args.other_frames, args.steps)
        _, _, _, blk = measure(om, nf, args.warmup, dev_frames, cfg, args.voxel, profile=not args.no_profile_pass, events=not args.no_events)
        blk["note"] = ("MergedTsdfIntegrator semantics" if om == "merged" else "FastTsdfIntegrator semantics at integrator_threads=1") + \
                      ", bit-exact vs the CPU oracle (DESIGN.md section 5)"
        other = {om: blk}

        # the fourth integrator the reference configures (method: "projective", tsdf_server_default.yaml:6, tsdf_server_carla.yaml:6):
        # a gather, not a ray caster; same frames through the yaml files' sensor model (1280 x 960 over 360 degrees)
        try:
            nfp = min(nf, 100)
            pcfg = eng.default_config(sensor_horizontal_resolution=1280, sensor_vertical_resolution=960, sensor_vertical_field_of_view_degrees=360.0,
                                      **{k: v for k, v in synth.integrator_overrides(args.voxel).items()
                                         if k in ("default_truncation_distance", "min_ray_length_m", "max_ray_length_m", "use_const_weight", "max_weight")})
            lp_ = Layer(eng, args.voxel, device=local_rank, capacity_blocks=32768)
            ip_ = Integrator(eng, lp_, pcfg, "projective")
            for i in range(min(10, args.warmup)):
                T, xyz, rgba, n = dev_frames[i]
                ip_.integrate_points_dev(T, xyz.data_ptr(), 0, n)
            ip_.sync()
            t0p = time.perf_counter()
            for i in range(args.warmup, args.warmup + nfp):
                T, xyz, rgba, n = dev_frames[i]
                ip_.integrate_points_dev(T, xyz.data_ptr(), 0, n)
            ip_.sync()
            dtp = time.perf_counter() - t0p
            stp = ip_.last_stats()
            other["projective"] = {"value": nfp / dtp, "unit": "frames/s", "frames": nfp, "ms_per_step": dtp / nfp * 1e3,
                                   "last_frame": {"n_valid": stp["n_valid"], "n_updates": stp["n_updates"], "n_touched_blocks": stp["n_touched_blocks"]},
                                   "update_kernel_bytes_last_frame": stp["n_touched_blocks"] * 2 * 49152,
                                   "note": "ProjectiveTsdfIntegrator semantics (range image + per-voxel gather), frames queued on one stream, no colours; bit-exact vs "
                                           "the CPU oracle (tests/test_gpu_projective.py); k_proj_update streams 96 KB per marked block (DESIGN.md section 5d)"}
            del ip_, lp_
        except Exception as e:  # never let the extra block cost the line
            other["projective"] = {"error": str(e)}

    # ---- the other shapes BASELINE.json names, short and bounded, each with a roofline of its own ---------------------------------
    other_configs = None
    if rank == 0 and world == 1 and args.other_config_frames > 0 and abs(args.voxel - 0.05) < 1e-9 and (W, H) == (640, 480):
        other_configs = {}
        ocf, ocw = args.other_config_frames, 6
        for name, (w2, h2, vox2) in (("configs[3]: 1280x720, 2 cm (coxgraph/config/tsdf_server_rs.yaml:12-17)", (1280, 720, 0.02)),
                                     ("configs[4] voxel size: 640x480, 1 cm (rays to 3 m, extrapolated: the reference has no 1 cm config)", (640, 480, 0.01))):
            try:
                if (w2, h2) == (W, H):
                    dev2 = dev_frames          # the headline's frames (as many of them as the run has)
                else:
                    _, dev2 = make_frames(ocw + ocf, w2, h2, 0)
                nf2 = min(ocf, len(dev2) - ocw)
                if nf2 < 4:
                    raise RuntimeError(f"only {len(dev2)} frames available")
                cfg2 = config_for(vox2)
                blocks = {}
                for m2 in ("merged", "fast"):
                    nfm = nf2 if m2 == "merged" else max(4, nf2 // 2)
                    _, _, _, blk = measure(m2, nfm, ocw, dev2, cfg2, vox2, profile=(m2 == "merged" and not args.no_profile_pass), events=False)
                    blocks[m2] = blk
                other_configs[name] = {"width": w2, "height": h2, "voxel_size_m": vox2, "points_per_frame": w2 * h2, "warmup": ocw, **blocks}
                if dev2 is not dev_frames:
                    del dev2
                torch.cuda.empty_cache()
            except Exception as e:
                other_configs[name] = {"error": str(e)}

    # ---- PCIe-inclusive: what the boundary costs when the caller hands over host buffers -----------------------------------
    # Every leg repeats the resident run with another entry point: the same frames (warm-up, then `steps` timed), a fresh layer, the
    # clock ramp; `--pcie-frames` bounds the timed frames (default: all of them).
    pcie = None
    if rank == 0 and world == 1 and args.pcie_frames != 0:
        nw = min(args.warmup, len(host_frames))
        nf = min(args.steps if args.pcie_frames < 0 else min(args.pcie_frames, args.steps), len(host_frames) - nw)
        pcie = {"frames": nf, "warmup": nw, "bytes_per_frame_points": int(host_frames[0][1].nbytes + host_frames[0][2].nbytes),
                "bytes_per_frame_depth": int(host_frames[0][3].nbytes + W * H * 4),
                "note": "never `value`.  The same frames as the resident run through the host entry points, frames in flight: cox_integrate_points_async (the "
                        "reference's boundary, tsdf_recover.h:71-77: host point clouds) from pinned and from pageable memory (pageable: one CPU copy into a "
                        "pinned bounce buffer), copied by the copy engine on the integrator's input stream; cox_integrate_depth_async (pinned depth + colour "
                        "images, converted on the device, the point count stays there); device images the caller copies on a stream of its own "
                        "(cox_integrate_depth_dev); and the synchronous cox_integrate_points, one frame in flight"}
        K = np.array(synth.INTRINSICS[(W, H)], np.float32)
        rgba_img = torch.from_numpy(synth.frame_colors(W, H)).pin_memory()

        def timed_leg(method, call, frames, sync_each=False):
            """every buffer once untimed on a scratch layer (a pinned buffer's first DMA pays for its mappings; a sensor driver reuses a ring of
            them), then warm-up + timed frames on a fresh layer like run_stream"""
            scratch = Integrator(eng, Layer(eng, args.voxel, device=local_rank, capacity_blocks=32768), cfg, method)
            for f in frames[:nw + nf]:
                call(scratch, f)
            scratch.sync()
            del scratch
            integ = Integrator(eng, Layer(eng, args.voxel, device=local_rank, capacity_blocks=32768), cfg, method)
            clock_ramp(method, dev_frames, cfg, args.voxel)
            for f in frames[:nw]:
                call(integ, f)
            integ.sync()
            torch.cuda.synchronize()
            gc.collect()
            t0 = time.perf_counter()
            for f in frames[nw:nw + nf]:
                call(integ, f)
                if sync_each:
                    integ.sync()
            integ.sync()
            torch.cuda.synchronize()
            return nf / (time.perf_counter() - t0)

        for m in (["merged", "fast"] if args.method in ("merged", "fast") else [args.method]):
            res = {}
            pinned = [(T, torch.from_numpy(p).pin_memory(), torch.from_numpy(c).pin_memory()) for T, p, c, _ in host_frames[:nw + nf]]
            pageable = [(T, torch.from_numpy(p), torch.from_numpy(c)) for T, p, c, _ in host_frames[:nw + nf]]
            by_address = lambda I, f: I.integrate_points_async(f[0], f[1].data_ptr(), f[2].data_ptr(), f[1].shape[0])
            res["host_points_pinned_frames_per_s"] = timed_leg(m, by_address, pinned)
            res["host_points_pageable_frames_per_s"] = timed_leg(m, by_address, pageable)
            del pinned, pageable
            ns = min(nf, 30)
            sync_frames = host_frames[:nw + nf]
            lp = Integrator(eng, Layer(eng, args.voxel, device=local_rank, capacity_blocks=32768), cfg, m)
            for T, p, c, _ in sync_frames[:3]:
                lp.integrate_points(T, p, c)
            t0 = time.perf_counter()
            for T, p, c, _ in sync_frames[3:3 + ns]:
                lp.integrate_points(T, p, c)  # cox_integrate_points: H2D + frame + sync, one frame in flight
            res["host_points_synchronous_frames_per_s"] = ns / (time.perf_counter() - t0)
            del lp
            # depth images (1.2 MB + 1.2 MB colour) from pinned host memory through cox_integrate_depth_async ...
            pinned_d = [(T, torch.from_numpy(d).pin_memory()) for T, _, _, d in host_frames[:nw + nf]]
            res["depth_image_frames_per_s"] = timed_leg(m, lambda I, f: I.integrate_depth_async(f[0], f[1].data_ptr(), rgba_img.data_ptr(), W, H, K), pinned_d)
            # ... and as device images the caller copies on a stream of its own, ordered against the engine (cox_integrator_set_input_stream).
            # (ONE stream made with hipStreamCreateWithFlags, as a host program has: the first torch.cuda.Stream() creates a pool of 64, and
            # hardware queues shared between streams are what DESIGN.md section 5 is about)
            import ctypes
            raw = ctypes.c_void_p()
            if ctypes.CDLL("libamdhip64.so").hipStreamCreateWithFlags(ctypes.byref(raw), 1) != 0:
                raise SystemExit("bench.py: hipStreamCreateWithFlags failed")
            side = torch.cuda.ExternalStream(raw.value)
            ring = [(torch.empty(W * H, dtype=torch.float32, device="cuda"), torch.empty(W * H * 4, dtype=torch.uint8, device="cuda")) for _ in range(8)]
            turn = [0]

            def on_caller_stream(I, f):
                dd, cc = ring[turn[0] % len(ring)]  # (the engine orders the caller's stream behind its last read of the images)
                turn[0] += 1
                I.set_input_stream(side.cuda_stream)
                with torch.cuda.stream(side):
                    dd.copy_(f[1].view(-1), non_blocking=True)
                    cc.copy_(rgba_img.view(-1), non_blocking=True)
                I.integrate_depth_dev(f[0], dd.data_ptr(), cc.data_ptr(), W, H, K)
            res["depth_image_on_caller_stream_frames_per_s"] = timed_leg(m, on_caller_stream, pinned_d)
            torch.cuda.synchronize()
            del pinned_d, ring  # (the stream is left to the process: torch keeps a reference to it)
            resident = fps if m == args.method else ((other or {}).get(m, {}).get("value"))
            if resident:
                res["resident_frames_per_s"] = resident
                res["fraction_of_resident"] = {k.replace("_frames_per_s", ""): v / resident for k, v in res.items() if k.endswith("_frames_per_s") and k != "resident_frames_per_s"}
            pcie[m] = res

    # ---- registrations/s: the server's configured constraint (config/server.yaml:28-31): isosurface vertices of the reference
    # submap against the ESDF of the reading submap, sampling_ratio 0.3 drawn by the weighted sampler on the GPU.  The pair is
    # SURVEY.md section 8d's: two submaps of the stream that was run with 50 % overlap -- frames [0, 2m) and [m, 3m), m = a third
    # of the frames (0-149 / 75-224 of the full stream's first 225) -- the reading one perturbed by (5 cm, -3 cm, 2 cm, 1 degree) ---
    reg = None
    trunc = cfg.default_truncation_distance
    esdf_cfg = dict(max_distance_m=2.0, min_distance_m=0.5 * trunc)
    if rank == 0 and world == 1 and args.reg_iters > 0:  # (N > 1: the inter-robot leg below is the registration measurement)
        from coxgraph_amd.posegraph import PoseGraphInterface
        m3 = max(1, len(dev_frames) // 3)
        subs = []
        for lo, hi in ((0, 2 * m3), (m3, min(3 * m3, len(dev_frames)))):
            ls = Layer(eng, args.voxel, device=local_rank, capacity_blocks=32768)
            isub = Integrator(eng, ls, cfg, "merged")
            for T, xyz, rgba, n in dev_frames[lo:hi]:
                isub.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), n)
            isub.sync()
            subs.append((ls, isub, (lo, hi)))
        t0 = time.perf_counter()
        ref = RegPoints.from_isosurface(eng, subs[0][0], 1.0)   # finishSubmap() of the reference submap ...
        esdf = subs[1][0].esdf(**esdf_cfg)                       # ... and of the reading one
        finish_ms = (time.perf_counter() - t0) * 1e3
        n_res = int(0.3 * ref.n)
        g = Registration(eng, ref, esdf)
        g.draw_samples(n_res, 7)
        pr, pd = np.zeros(4), np.array([0.05, -0.03, 0.02, np.radians(1.0)])

        def rate(fn, calls, per_call=1, reps=3):
            """median calls/s over `reps` timed repetitions (the first call of a loop pays clock ramp-up)"""
            fn()
            out = []
            for _ in range(reps):
                t1 = time.perf_counter()
                for _ in range(calls):
                    fn()
                out.append(per_call * calls / (time.perf_counter() - t1))
            return sorted(out)[len(out) // 2]
        n_corr = g.normal_eq(pr, pd)[3]
        g.kernel_time(reset=True)
        one_at_a_time = rate(lambda: g.normal_eq(pr, pd), args.reg_iters)
        kms, kl = g.kernel_time()
        batch = [Registration(eng, ref, esdf) for _ in range(8)]
        for k, b in enumerate(batch):
            b.draw_samples(n_res, 100 + k)

        def evaluate_all():
            for b in batch:
                b.normal_eq_begin(pr, pd)
            return [b.normal_eq_finish() for b in batch]
        in_flight = rate(evaluate_all, max(1, args.reg_iters // 8), per_call=8)
        # configs[4]'s pose-graph evaluation: 28 constraints (all pairs of 8 clients) in ONE launch (cox_reg_normal_eq_batch)
        batch28 = [Registration(eng, ref, esdf) for _ in range(28)]
        for k, b in enumerate(batch28):
            b.draw_samples(n_res, 200 + k)
        prs, pds = [pr] * 28, [pd] * 28
        Registration.normal_eq_batch(batch28, prs, pds)
        batch28[0].kernel_time(reset=True)
        batched = rate(lambda: Registration.normal_eq_batch(batch28, prs, pds), max(2, args.reg_iters // 8), per_call=28)
        bms, bl = batch28[0].kernel_time()
        pg = PoseGraphInterface()
        pg.addSubmap(0, [0, 0, 0, 0])
        pg.addSubmap(1, pd)
        pg.addLoopClosureMeasurement(0, 1, [0.02, 0.0, -0.01, np.radians(0.5)])
        pg.addForceRegistrationConstraint(0, 1, g)
        t2 = time.perf_counter()
        _, second = pg.optimize(enable_registration=True)
        solve_ms = (time.perf_counter() - t2) * 1e3
        reg = {"registrations_per_s": batched, "registrations_per_s_8_handles_in_flight": in_flight, "registrations_per_s_one_at_a_time": one_at_a_time,
               "batch": {"constraints_per_launch": 28, "kernel_ms_per_launch": bms / max(bl, 1), "kernel_GBps_algorithmic": 28 * n_res * (20 + 8 * 12) / (bms / max(bl, 1) * 1e-3) / 1e9,
                         "note": "one pose-graph evaluation of configs[4] (28 constraints) = one launch of k_reg_normal_eq_batch + one 57 KB read-back"}, "residuals_per_registration": n_res, "registration_points": int(ref.n),
               "correspondences": int(n_corr),
               "pair": f"reference submap = frames {subs[0][2][0]}..{subs[0][2][1] - 1}, reading submap = frames {subs[1][2][0]}..{subs[1][2][1] - 1} of this run's stream (50 % overlap), "
                       "reading pose off by (0.05, -0.03, 0.02) m and 1 degree",
               "point_set": "isosurface vertices (explicit_to_implicit), ESDF reading", "finish_submap_ms": finish_ms,
               "kernel_ms": kms / max(kl, 1), "kernel_GBps_algorithmic": n_res * (20 + 8 * 12) / (kms / max(kl, 1) * 1e-3) / 1e9,
               "roofline_frac": n_res * (20 + 8 * 12) / (kms / max(kl, 1) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
               "two_stage_solve_ms": solve_ms, "solve_evaluations": second["evaluations"], "solved_pose_error": [float(x) for x in pg.getPoseMap()[1]]}
        del subs

    # ---- N > 1: submap exchange + inter-robot registration ------------------------------------------------------------------
    dist_reg = None
    if world > 1 and args.reg_iters > 0:
        from coxgraph_amd.posegraph import PoseGraph, RegistrationConstraint
        # finishSubmap() on the client's own GPU
        ref = RegPoints.from_isosurface(eng, layer, 1.0)
        esdf = layer.esdf(**esdf_cfg)
        nb, npts = esdf.n_blocks(), ref.n
        sizes = torch.tensor([nb, npts], dtype=torch.int64, device=coll_dev)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes)
        all_sizes = [tuple(int(v) for v in t.tolist()) for t in all_sizes]
        max_nb, max_np = max(s[0] for s in all_sizes), max(s[1] for s in all_sizes)
        idx_t = torch.zeros((max_nb, 3), dtype=torch.int32, device="cuda")
        vox_t = torch.zeros((max_nb, 4096, 3), dtype=torch.int32, device="cuda")
        pts_t = torch.zeros((max_np, 5), dtype=torch.float32, device="cuda")
        esdf.export_dev(idx_t.data_ptr(), vox_t.data_ptr(), max_nb)
        pts_t[:npts].copy_(torch.from_numpy(ref.download()).cuda())  # preparation, not part of the timed exchange
        torch.cuda.synchronize()
        dist.barrier()
        comm = None
        if capi_comm:
            from coxgraph_amd.capi import Comm
            ids = [Comm.unique_id(eng) if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            comm = Comm(eng, local_rank, rank, world, ids[0])
            dist.barrier()
        # the exchange: one all-gather of (block indices, voxel words, point sets); device to device over xGMI with RCCL
        t0 = time.perf_counter()
        if comm is not None:
            g_idx = torch.empty((world, max_nb, 3), dtype=torch.int32, device="cuda")
            g_vox = torch.empty((world, max_nb, 4096, 3), dtype=torch.int32, device="cuda")
            g_pts = torch.empty((world, max_np, 5), dtype=torch.float32, device="cuda")
            cur = torch.cuda.current_stream().cuda_stream
            comm.allgather_dev(idx_t.data_ptr(), g_idx.data_ptr(), idx_t.numel() * 4, cur)
            comm.allgather_dev(vox_t.data_ptr(), g_vox.data_ptr(), vox_t.numel() * 4, cur)
            comm.allgather_dev(pts_t.data_ptr(), g_pts.data_ptr(), pts_t.numel() * 4, cur)
        elif backend == "nccl":
            g_idx = torch.empty((world, max_nb, 3), dtype=torch.int32, device="cuda")
            g_vox = torch.empty((world, max_nb, 4096, 3), dtype=torch.int32, device="cuda")
            g_pts = torch.empty((world, max_np, 5), dtype=torch.float32, device="cuda")
            dist.all_gather_into_tensor(g_idx, idx_t)
            dist.all_gather_into_tensor(g_vox, vox_t)
            dist.all_gather_into_tensor(g_pts, pts_t)
        else:  # rehearsal over gloo: through the host
            def gather(t):
                lst = [torch.empty_like(t, device="cpu") for _ in range(world)]
                dist.all_gather(lst, t.cpu())
                return torch.stack(lst).cuda()
            g_idx, g_vox, g_pts = gather(idx_t), gather(vox_t), gather(pts_t)
        torch.cuda.synchronize()
        dt_x = time.perf_counter() - t0
        tt = torch.tensor([dt_x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_x = float(tt.item())
        recv_bytes = sum((s[0] * (12 + 49152) + s[1] * 20) for k, s in enumerate(all_sizes) if k != rank)
        # constraints (a, b), a < b: CLIENT a's isosurface points against CLIENT b's ESDF, dealt over the ranks
        pairs = [(a, b) for a in range(world) for b in range(a + 1, world)]
        pg = PoseGraph()
        for k in range(world):
            pg.add_node(k, [0.01 * k, -0.005 * k, 0.002 * k, 0.001 * k], constant=(k == 0))
        layers, points = {}, {}
        n_corr_total = 0
        for k, (a, b) in enumerate(pairs):
            if k % world != rank:
                pg.reg.append(None)  # another rank's constraint: never touched here
                continue
            if b not in layers:
                lb = Layer(eng, args.voxel, device=local_rank, capacity_blocks=max(64, all_sizes[b][0]))
                lb.upload_dev(g_idx[b].data_ptr(), g_vox[b].data_ptr(), all_sizes[b][0])
                layers[b] = lb
            if a not in points:
                points[a] = RegPoints.from_device(eng, g_pts[a].data_ptr(), all_sizes[a][1], device=local_rank)
            gab = Registration(eng, points[a], layers[b])
            gab.draw_samples(int(0.3 * points[a].n), 1000 + k)
            n_corr_total += gab.normal_eq(pg.poses[a], pg.poses[b])[3]
            pg.reg.append(RegistrationConstraint(a, b, gab))
        poses = {k: v.copy() for k, v in pg.poses.items()}
        red = dict(comm=comm) if comm is not None else dict(group=dist.group.WORLD)
        pg.build(poses, **red)
        dist.barrier()
        n_eval = max(4, args.reg_iters // 10)
        t4 = time.perf_counter()
        for _ in range(n_eval):
            cost, _, _, _ = pg.build(poses, **red)
        torch.cuda.synchronize()
        dist.barrier()
        dt4 = time.perf_counter() - t4
        tt = torch.tensor([dt4, float(n_corr_total)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(tt[1:], op=dist.ReduceOp.SUM)
        dt4 = float(tt[0].item())
        dist_reg = {"pose_graph_evaluations_per_s": n_eval / dt4, "registrations_per_s": n_eval * len(pairs) / dt4, "constraints": len(pairs),
                    "constraint": "client a's isosurface points against client b's ESDF (a < b), after the exchange",
                    "correspondences_over_all_constraints": int(tt[1].item()),
                    "submap_exchange_ms": dt_x * 1e3, "submap_exchange_GBps": recv_bytes / dt_x / 1e9, "submap_exchange_bytes_received_per_rank": recv_bytes,
                    "submap_blocks": [s[0] for s in all_sizes], "isosurface_points": [s[1] for s in all_sizes],
                    "all_reduce_doubles": (4 * (world - 1)) ** 2 + 4 * (world - 1) + 2,
                    "backend": "rccl through cox_comm_* (C ABI)" if comm is not None else ("rccl through torch.distributed" if backend == "nccl" else backend), "cost": cost}

    # ---- CPU baseline on rank 0, N = 1 only ---------------------------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        threads = min(8, os.cpu_count() or 1)
        res = cpu_baseline([f[:3] for f in host_frames], args.voxel, min(args.cpu_frames, len(host_frames)), threads)
        cpu = {"value": res["fast"][0], "unit": "frames/s", "cores": threads, "kind": "port",
               "sample": f"oracle FastTsdfIntegrator restatement (the integrator the reference's tsdf servers configure), {threads} threads (reference "
                         f"integrator_threads: 8), first {res['fast'][1]} frames of the same stream, {res['fast'][2]:.1f} s",
               "same_method": {"method": "merged", "value": res["merged"][0], "unit": "frames/s", "cores": threads,
                               "sample": f"oracle MergedTsdfIntegrator restatement, {threads} threads, first {res['merged'][1]} frames, {res['merged'][2]:.1f} s"}}

    if rank == 0:
        line = {
            "metric": "640x480 depth frames/sec fused (TSDF, 5 cm voxels) per node; submap registrations/sec reported alongside",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: 1 client per GPU, {W}x{H} synthetic depth stream, {args.voxel * 100:.0f} cm voxels, "
                                   f"{args.method} integrator semantics (the reference's configured method is fast; bit-exact vs CPU oracle), points resident in HBM",
                       "points_per_frame": W * H, "method": args.method, "voxel_size_m": args.voxel, "clients": world},
            # both integrators the reference configures on this path, side by side (value = the `--method` one)
            "headline": {args.method: fps, **({k: v.get("value") for k, v in (other or {}).items() if k in ("merged", "fast")})},
            "relaxation": head.get("relaxation"),
            "frame_stats_mean": frame_stats,
            "host_submit_ms_per_frame": host_ms,   # caller's thread inside cox_integrate_points_dev (the other half of a frame is enqueued by the integrator's submission thread)
            "critical_path": dict(crit, note="longest sequential chains of any timed frame: points of the largest bundle (merged), updates of the busiest voxel") if crit else None,
            "roofline": roofline, "cpu_baseline": cpu,
            "gpu_over_cpu": ({"merged": (fps if args.method == "merged" else (other or {}).get("merged", {}).get("value", 0.0)) / cpu["same_method"]["value"],
                              "fast": (fps if args.method == "fast" else (other or {}).get("fast", {}).get("value", 0.0)) / cpu["value"]} if cpu else None),
            "other_methods": other, "other_configs": other_configs, "pcie_inclusive": pcie, "registration": reg, "distributed_registration": dist_reg,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
