#!/usr/bin/env python3
"""Round-3 probe of the host boundary: what a pinned -> device copy costs by itself (per copy, by size, one stream / two streams),
and where the time of cox_integrate_points_async goes (time inside the calls vs the rate of the whole loop).
  python scripts/h2d_probe.py [frames] [method]          (COX_H2D=kernel|memcpy in the environment)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch

import coxgraph_amd
from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, Integrator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
method = sys.argv[2] if len(sys.argv) > 2 else "merged"
voxel = 0.05
env = {k: v for k, v in os.environ.items() if k.startswith("COX_")}
eng = coxgraph_amd.load_engine()
cfg = eng.default_config(**synth.integrator_overrides(voxel))

# ---- A: raw copies ------------------------------------------------------------------------------------------------------
if not os.environ.get("PROBE_SKIP_RAW"):
    for mb in (0.3, 1.2, 3.7, 4.9):
        nb = int(mb * 1e6) // 16 * 16
        src = [torch.empty(nb, dtype=torch.uint8).pin_memory() for _ in range(8)]
        dst = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(8)]
        for n_streams in (1, 2):
            streams = [torch.cuda.Stream() for _ in range(n_streams)]
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(200):
                    with torch.cuda.stream(streams[i % n_streams]):
                        dst[i % 8].copy_(src[i % 8], non_blocking=True)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            print(f"raw copy {mb} MB, {n_streams} stream(s): submit {(t1 - t0) / 200 * 1e6:.1f} us/copy, done {(t2 - t0) / 200 * 1e6:.1f} us/copy = {nb / ((t2 - t0) / 200) / 1e9:.1f} GB/s", flush=True)
        del src, dst

# streams created (and never used) before the engine's own: does the NUMBER of streams of the process matter, or only the active ones?
if os.environ.get("PROBE_DUMMY_STREAMS"):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    dummies = []
    for _ in range(int(os.environ["PROBE_DUMMY_STREAMS"])):
        h = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0
        dummies.append(h)

# ---- B: the engine ------------------------------------------------------------------------------------------------------
host = [synth.make_frame(t) for t in range(n)]
dev = [(T, torch.from_numpy(p).cuda(), torch.from_numpy(c).cuda(), p.shape[0]) for T, p, c, _ in host]
pinned = [(T, torch.from_numpy(p).pin_memory(), torch.from_numpy(c).pin_memory()) for T, p, c, _ in host]
torch.cuda.synchronize()


def run(label, call, frames, reps=int(os.environ.get('PROBE_REPS', '3'))):
    for rep in range(reps):
        integ = Integrator(eng, Layer(eng, voxel, capacity_blocks=32768), cfg, method)
        for f in frames[:10]:
            call(integ, f)
        integ.sync()
        inside = 0.0
        t0 = time.perf_counter()
        for f in frames:
            a = time.perf_counter()
            call(integ, f)
            inside += time.perf_counter() - a
        t1 = time.perf_counter()
        integ.sync()
        t2 = time.perf_counter()
        hm, hf = integ.host_time()
        print(f"{method} {label} env {env}: {len(frames) / (t2 - t0):.0f} frames/s; inside the calls {inside / len(frames) * 1e6:.0f} us/frame, loop {(t1 - t0) / len(frames) * 1e6:.0f} us/frame, "
              f"drain {(t2 - t1) * 1e6:.0f} us; engine host time {hm / max(hf, 1) * 1e3:.0f} us/frame", flush=True)
        del integ


run("resident", lambda I, f: I.integrate_points_dev(f[0], f[1].data_ptr(), f[2].data_ptr(), f[3]), dev)
if os.environ.get("PROBE_RESIDENT_ONLY"):
    sys.exit(0)
run("pinned xyz+rgba", lambda I, f: I.integrate_points_async(f[0], f[1].data_ptr(), f[2].data_ptr(), f[1].shape[0]), pinned)
K = np.array(synth.INTRINSICS[(640, 480)], np.float32)
rgba_img = torch.from_numpy(synth.frame_colors(640, 480)).pin_memory()
pinned_d = [(T, torch.from_numpy(d).pin_memory()) for T, _, _, d in host]
run("pinned depth+rgba images", lambda I, f: I.integrate_depth_async(f[0], f[1].data_ptr(), rgba_img.data_ptr(), 640, 480, K), pinned_d)
import ctypes
_hip = ctypes.CDLL("libamdhip64.so")
_h = ctypes.c_void_p()
assert _hip.hipStreamCreateWithFlags(ctypes.byref(_h), 1) == 0
side = torch.cuda.ExternalStream(_h.value)
dd = [torch.empty(640 * 480, dtype=torch.float32, device="cuda") for _ in range(8)]
cc = [torch.empty(640 * 480 * 4, dtype=torch.uint8, device="cuda") for _ in range(8)]
count = [0]


def depth_on_caller_stream(I, f):
    i = count[0] % 8
    count[0] += 1
    I.set_input_stream(side.cuda_stream)
    with torch.cuda.stream(side):
        dd[i].copy_(f[1].view(-1), non_blocking=True)
        cc[i].copy_(rgba_img.view(-1), non_blocking=True)
    I.integrate_depth_dev(f[0], dd[i].data_ptr(), cc[i].data_ptr(), 640, 480, K)


run("device images copied on the caller's stream", depth_on_caller_stream, pinned_d)
run("pinned xyz only", lambda I, f: I.integrate_points_async(f[0], f[1].data_ptr(), None, f[1].shape[0]), pinned)

# ---- C: resident frames with unrelated H2D traffic beside them (does a transfer by itself slow the kernels?) ------------------
if os.environ.get("PROBE_BACKGROUND"):
    import threading
    stop = False
    copied = [0]
    side = torch.cuda.Stream()
    bsrc = [torch.empty(4915200, dtype=torch.uint8).pin_memory() for _ in range(4)]
    bdst = [torch.empty(4915200, dtype=torch.uint8, device="cuda") for _ in range(4)]

    def background():
        ev = [torch.cuda.Event() for _ in range(4)]
        i = 0
        while not stop:
            with torch.cuda.stream(side):
                if i >= 4:
                    ev[i % 4].synchronize()
                bdst[i % 4].copy_(bsrc[i % 4], non_blocking=True)
                ev[i % 4].record(side)
            i += 1
            copied[0] = i
    th = threading.Thread(target=background)
    th.start()
    time.sleep(0.05)
    c0, t0 = copied[0], time.perf_counter()
    run("resident, 4.9 MB H2D copies running beside it on a stream of their own", lambda I, f: I.integrate_points_dev(f[0], f[1].data_ptr(), f[2].data_ptr(), f[3]), dev)
    print(f"  background copies: {(copied[0] - c0) / (time.perf_counter() - t0):.0f} /s", flush=True)
    stop = True
    th.join()
    torch.cuda.synchronize()
