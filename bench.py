#!/usr/bin/env python3
"""Headline benchmark: 640x480 depth frames/s fused into a 5 cm TSDF submap per MI355X
(BASELINE.json configs[1]) + submap registrations/s, one coxgraph client per GPU.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic depth frame: integratePointCloud(T_G_C,
points_C, colors) of 307 200 points into the client's submap layer, inputs already resident in HBM.
For N > 1 the driver launches one rank per GPU (torch.distributed over RCCL); clients are independent
(weak scaling, no data-path collective -- SURVEY.md section 8e); the only collective is the
barrier/max-reduce of the timing harness.

One JSON line on rank 0.  `roofline` prices the dominant kernel against HBM peak using the ALGORITHMIC
bytes of SURVEY.md section 8d (16 B per valid point + 24 B per touched voxel); `cpu_baseline` times
the CPU restatement of the reference's configured integrator (fast, 8 threads) on a bounded sample of
the same frames on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)  # the 300-frame 30 Hz stream of SURVEY.md section 8d
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--method", default="merged", choices=["merged", "simple", "fast"])
    ap.add_argument("--no-events", action="store_true", help="no HIP-event kernel timing inside the timed region")
    ap.add_argument("--fast-frames", type=int, default=300, help="frames of the same stream also run through method 'fast' (0 = skip)")
    ap.add_argument("--voxel", type=float, default=0.05)
    ap.add_argument("--cpu-frames", type=int, default=320, help="frames of the CPU baseline sample (0 = skip); ~10 s of CPU work at the default")
    ap.add_argument("--reg-iters", type=int, default=50)
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--serial", action="store_true", help="sync after every frame (profiling aid: kernel times without cross-frame overlap)")
    return ap.parse_args()


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
    for method, thr in (("fast", threads), ("merged", 1)):
        cfg = eng.default_config(integrator_threads=thr, **synth.integrator_overrides(voxel))
        layer = Layer(eng, voxel)
        integ = Integrator(eng, layer, cfg, method)
        eng.fn("integrator_set_count_touched")(integ.h, C.c_int(0))
        nf = n_frames if method == "fast" else max(2, n_frames // 4)  # the single-threaded merged run is context, keep it short
        t0 = time.perf_counter()
        for (T, pts, rgba) in frames[:nf]:
            integ.integrate_points(T, pts, rgba)
        dt = time.perf_counter() - t0
        out[method] = (nf / dt, nf, dt)
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    backend = os.environ.get("COX_DIST_BACKEND", "nccl")  # "gloo" to rehearse N > 1 on a single-GPU box
    if world > 1:
        import datetime
        tmo = datetime.timedelta(seconds=180)  # a rank that dies must not leave the others waiting for the default 10 min
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index), timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    local_rank = dev_index

    import coxgraph_amd
    from coxgraph_amd import synth
    from coxgraph_amd.capi import Layer, Integrator, RegPoints, Registration, words_to_fields
    eng = coxgraph_amd.load_engine()

    # ---- synthetic stream of this rank's client, resident in HBM before anything is timed ----------
    n_frames = args.warmup + args.steps
    host_frames = []
    dev_frames = []
    for t in range(n_frames):
        T, pts, rgba, _ = synth.make_frame(t, client=rank, n_clients=max(world, 1))
        if t < max(args.cpu_frames, 1):
            host_frames.append((T, pts, rgba))
        dev_frames.append((T, torch.from_numpy(pts).cuda(), torch.from_numpy(rgba).cuda(), pts.shape[0]))
    torch.cuda.synchronize()

    cfg = eng.default_config(**synth.integrator_overrides(args.voxel))
    layer = Layer(eng, args.voxel, device=local_rank, capacity_blocks=32768)
    integ = Integrator(eng, layer, cfg, args.method)

    def step(i):
        T, xyz, rgba, n = dev_frames[i]
        integ.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), n)
        if args.serial:
            integ.sync()

    # HIP events around the two candidate dominant kernels are recorded on the engine's own streams INSIDE the timed
    # region, for every 4th frame (measured: every frame costs 6.6 % of the throughput, every 4th < 2 %); --no-events
    # times the region without them
    if not args.no_events:
        integ.set_profiling(4)
    for i in range(args.warmup):
        step(i)
    integ.sync()
    integ.stage_times(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_frames):
        step(i)
    integ.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fps = world * args.steps / dt
    live_times = integ.stage_times() if not args.no_events else None

    # ---- roofline of the dominant kernel ---------------------------------------------------------------------------
    # durations: HIP events recorded inside the timed region above (frames overlap on two streams there, so a kernel's
    # duration includes what the other stream's kernels take from it -- the same view rocprofv3 --kernel-trace has of
    # this command).  Algorithmic bytes need per-frame counters, which only a synchronous pass can read: the same frames
    # are run again, one in flight, on a fresh layer; that pass also gives the kernels' undisturbed durations.
    roofline = None
    stats_sum = dict(n_valid=0, n_touched_voxels=0, n_updates=0, n_rays=0)
    stats_max = dict(max_bundle_points=0, max_voxel_updates=0)
    if rank == 0 and not args.no_profile_pass:
        layer2 = Layer(eng, args.voxel, device=local_rank, capacity_blocks=32768)
        integ2 = Integrator(eng, layer2, cfg, args.method)
        integ2.set_profiling(True)
        for i in range(n_frames):
            T, xyz, rgba, n = dev_frames[i]
            integ2.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), n)
            integ2.sync()
            if i >= args.warmup:
                st = integ2.last_stats()
                for k in stats_sum:
                    stats_sum[k] += st[k]
                for k in stats_max:
                    stats_max[k] = max(stats_max[k], st[k])
            elif i == args.warmup - 1:
                integ2.stage_times(reset=True)
        serial = integ2.stage_times()
        st = live_times if live_times is not None else serial
        # B_frame = 16 B per valid point + 24 B per touched voxel (12-B TsdfVoxel read + written)
        alg_bytes = 16.0 * stats_sum["n_valid"] + 24.0 * stats_sum["n_touched_voxels"]
        kernels = {"merge": "k_bundle_merge", "apply": "k_apply_eval+k_apply_long"}
        stage = max(st, key=lambda k: st[k][0]) if args.method == "merged" else "apply"
        ms, launches = st[stage]
        if launches:
            per_launch_bytes = alg_bytes / args.steps
            avg_ms = ms / launches
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            # HBM bytes per launch from the committed PMC passes (profiles/, same command with --serial): rocprofv3 cannot
            # run inside this process, so the figure is read back from the summary it produced
            traffic = None
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
                names = kernels[stage].split("+")
                traffic = sum(pm[k]["fetch_bytes"] + pm[k]["write_bytes"] for k in names)
            except Exception:
                traffic = None
            avg = lambda d: {kernels[k]: (v[0] / v[1] if v[1] else None) for k, v in d.items()}
            roofline = {"bound": "hbm", "kernel": kernels[stage], "achieved": achieved, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "avg_launch_ms": avg_ms,
                        "algorithmic_bytes_per_launch": per_launch_bytes, "launches_timed": launches,
                        "timing": "HIP events on the engine's streams inside the timed region, every 4th frame" if live_times is not None else "HIP events, one frame in flight",
                        "note": "one launch = one frame; the path is bound by dependent in-order update chains, not by HBM, at 5 cm (DESIGN.md section 6)",
                        "stage_avg_ms": avg(st), "stage_avg_ms_one_frame_in_flight": avg(serial),
                        "updates_per_s_in_apply": (stats_sum["n_updates"] / args.steps) / (st["apply"][0] / st["apply"][1] * 1e-3) if st["apply"][1] else None}
        del integ2, layer2

    # ---- registrations/s: one fused residual+Jacobian+normal-equation evaluation of one constraint ------
    reg = None
    if rank == 0 and args.reg_iters > 0:
        from coxgraph_amd.posegraph import PoseGraphInterface
        trunc = cfg.default_truncation_distance
        # finishSubmap()'s relevant-voxel point set, extracted and kept on the GPU; reading layer = the fused submap itself,
        # displaced by the SURVEY.md section 8d perturbation
        ref = RegPoints.from_layer(eng, layer, 1.0, trunc)
        ww = layer.registration_points(1.0, trunc)[:, 4].astype(np.float64)
        rng = np.random.default_rng(7)
        n_res = int(0.3 * ref.n)  # sampling_ratio 0.3, coxgraph/config/server.yaml:30
        sidx = rng.choice(ref.n, size=n_res, replace=True, p=ww / ww.sum()).astype(np.uint32)
        g = Registration(eng, ref, layer)
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
        g.normal_eq(pr, pd, sidx)
        g.kernel_time(reset=True)
        one_at_a_time = rate(lambda: g.normal_eq(pr, pd, sidx), args.reg_iters)
        kms, kl = g.kernel_time()
        # the server's pattern: one pose-graph evaluation begins all of its constraints, then collects them; sample
        # indices stay on the GPU (8 constraints in flight: configs[4] has 28 inter-robot pairs over 8 ranks)
        batch = [Registration(eng, ref, layer) for _ in range(8)]
        for b in batch:
            b.set_samples(sidx)
        def evaluate_all():
            for b in batch:
                b.normal_eq_begin(pr, pd)
            return [b.normal_eq_finish() for b in batch]
        in_flight = rate(evaluate_all, max(1, args.reg_iters // 8), per_call=8)
        # full two-stage solve of a 2-node graph (loop closure + forced registration constraint), pose_graph_interface.cpp:32-49
        pg = PoseGraphInterface()
        pg.addSubmap(0, [0, 0, 0, 0])
        pg.addSubmap(1, pd)
        pg.addLoopClosureMeasurement(0, 1, [0.02, 0.0, -0.01, np.radians(0.5)])
        pg.addForceRegistrationConstraint(0, 1, g, sidx)
        t2 = time.perf_counter()
        _, second = pg.optimize(enable_registration=True)
        solve_ms = (time.perf_counter() - t2) * 1e3
        reg = {"registrations_per_s": in_flight, "registrations_per_s_one_at_a_time": one_at_a_time, "residuals_per_registration": n_res, "registration_points": int(ref.n),
               "kernel_ms": kms / max(kl, 1), "kernel_GBps_algorithmic": n_res * (20 + 8 * 12) / (kms / max(kl, 1) * 1e-3) / 1e9,
               "two_stage_solve_ms": solve_ms, "solve_evaluations": second["evaluations"],
               "solved_pose_error": [float(x) for x in pg.getPoseMap()[1]]}

    # ---- N > 1: the server's inter-robot registration, constraints dealt over the ranks, one all-reduce per evaluation ----
    # (SURVEY.md section 8e / BASELINE configs[2], [4]): all pairs of 8 submaps = 28 forced registration constraints; every
    # rank evaluates its share against its own client's map (begin all, then collect) and the packed (4N)^2 + 4N + 1
    # doubles are summed with ONE all-reduce (RCCL when the backend is nccl).  Max over ranks, like the fusion timing.
    dist_reg = None
    if world > 1 and args.reg_iters > 0:
        from coxgraph_amd.posegraph import PoseGraph, RegistrationConstraint
        trunc = cfg.default_truncation_distance
        ref = RegPoints.from_layer(eng, layer, 1.0, trunc)
        rng = np.random.default_rng(7 + rank)
        n_res = int(0.3 * ref.n)
        sidx = rng.integers(0, max(ref.n, 1), size=n_res).astype(np.uint32)
        pg = PoseGraph()
        n_nodes = 8
        for k in range(n_nodes):
            pg.add_node(k, [0.01 * k, -0.005 * k, 0.002 * k, 0.001 * k], constant=(k == 0))
        pairs = [(a, b) for a in range(n_nodes) for b in range(a + 1, n_nodes)]
        for k, (a, b) in enumerate(pairs):
            if k % world == rank:
                g = Registration(eng, ref, layer)
                g.set_samples(sidx)
                pg.reg.append(RegistrationConstraint(a, b, g))
            else:
                pg.reg.append(None)  # another rank's constraint: never touched here
        poses = {k: v.copy() for k, v in pg.poses.items()}
        pg.build(poses, group=dist.group.WORLD)
        dist.barrier()
        n_eval = max(4, args.reg_iters // 10)
        t4 = time.perf_counter()
        for _ in range(n_eval):
            cost, _, _, _ = pg.build(poses, group=dist.group.WORLD)
        torch.cuda.synchronize()
        dist.barrier()
        dt4 = time.perf_counter() - t4
        tt = torch.tensor([dt4], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt4 = float(tt.item())
        dist_reg = {"pose_graph_evaluations_per_s": n_eval / dt4, "registrations_per_s": n_eval * len(pairs) / dt4, "constraints": len(pairs),
                    "residuals_per_constraint": n_res, "all_reduce_doubles": (4 * (n_nodes - 1)) ** 2 + 4 * (n_nodes - 1) + 1,
                    "backend": "rccl" if backend == "nccl" else backend, "cost": cost}

    # ---- the reference's configured method (`method: "fast"`, tsdf_server_euroc.yaml:6) on the same stream, for context ----
    other = None
    if rank == 0 and world == 1 and args.method != "fast" and args.fast_frames > 0:
        nf = min(args.fast_frames, args.steps)  # the same frames as the headline number: warm-up frames first, untimed
        layer3 = Layer(eng, args.voxel, device=local_rank, capacity_blocks=32768)
        integ3 = Integrator(eng, layer3, cfg, "fast")
        for i in range(args.warmup):
            T, xyz, rgba, n = dev_frames[i]
            integ3.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), n)
        integ3.sync()
        t3 = time.perf_counter()
        for i in range(args.warmup, args.warmup + nf):
            T, xyz, rgba, n = dev_frames[i]
            integ3.integrate_points_dev(T, xyz.data_ptr(), rgba.data_ptr(), n)
        integ3.sync()
        dt3 = time.perf_counter() - t3
        other = {"fast": {"value": nf / dt3, "unit": "frames/s", "frames": nf,
                          "note": "FastTsdfIntegrator semantics at integrator_threads=1, bit-exact vs the CPU oracle (DESIGN.md section 5c)"}}
        del integ3, layer3

    # ---- CPU baseline on rank 0, N = 1 only ---------------------------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        threads = min(8, os.cpu_count() or 1)
        res = cpu_baseline(host_frames, args.voxel, min(args.cpu_frames, len(host_frames)), threads)
        cpu = {"value": res["fast"][0], "unit": "frames/s", "cores": threads, "kind": "port",
               "sample": f"oracle FastTsdfIntegrator restatement, {threads} threads (reference integrator_threads: 8), first {res['fast'][1]} frames of the "
                         f"same stream, {res['fast'][2]:.1f} s; oracle merged 1 thread: {res['merged'][0]:.2f} frames/s over {res['merged'][1]} frames",
               "merged_1thread_frames_per_s": res["merged"][0]}

    if rank == 0:
        line = {
            "metric": "640x480 depth frames/sec fused (TSDF, 5 cm voxels) per node; submap registrations/sec reported alongside",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: 1 client per GPU, 640x480 synthetic depth stream, {args.voxel * 100:.0f} cm voxels, "
                                   f"{args.method} integrator semantics (bit-exact vs CPU oracle), points resident in HBM",
                       "points_per_frame": 307200, "method": args.method, "voxel_size_m": args.voxel, "clients": world},
            "frame_stats_mean": {k: v / max(args.steps, 1) for k, v in stats_sum.items()},
            "critical_path": dict(stats_max, note="longest sequential chains of any timed frame: points of the largest bundle (k_bundle_merge), "
                                                  "updates of the busiest voxel (k_apply_*; free-space runs fold)"),
            "roofline": roofline, "cpu_baseline": cpu, "registration": reg, "distributed_registration": dist_reg, "other_methods": other,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
