"""Run by tests/test_gpu_exchange.py::test_rccl_paths_on_one_rank in a process of its own: the RCCL (backend "nccl") branches of
the multi-GPU server path -- device-to-device all-gather of a submap's wire arrays and point set, upload on the "receiving"
side, the packed all-reduce of the pose graph -- with world size 1 on the one GPU of the test box.  Prints OK on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import coxgraph_amd  # noqa: E402
from coxgraph_amd import synth  # noqa: E402
from coxgraph_amd.capi import Layer, Integrator, RegPoints, Registration  # noqa: E402
from coxgraph_amd.posegraph import PoseGraph, RegistrationConstraint  # noqa: E402


def main():
    port = int(sys.argv[1])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    eng = coxgraph_amd.load_engine()
    voxel = 0.10
    cfg = eng.default_config(**synth.integrator_overrides(voxel))
    subs = []
    for client in (0, 1):  # two clients' submaps on this GPU
        layer = Layer(eng, voxel)
        integ = Integrator(eng, layer, cfg, "merged")
        for t in range(0, 60, 10):
            T, pts, rgba, _ = synth.make_frame(t, client=client, n_clients=12)
            integ.integrate_points(T, pts[::6], rgba[::6])
        subs.append((layer.esdf(max_distance_m=2.0, min_distance_m=1.5 * voxel), RegPoints.from_isosurface(eng, layer, 1.0)))
    # the exchange of bench.py's N > 1 leg, through RCCL
    esdf, ref = subs[1][0], subs[0][1]
    nb, npts = esdf.n_blocks(), ref.n
    sizes = torch.tensor([nb, npts], dtype=torch.int64, device="cuda")
    all_sizes = [torch.zeros_like(sizes)]
    dist.all_gather(all_sizes, sizes)
    assert all_sizes[0].tolist() == [nb, npts]
    idx_t = torch.zeros((nb, 3), dtype=torch.int32, device="cuda")
    vox_t = torch.zeros((nb, 4096, 3), dtype=torch.int32, device="cuda")
    pts_t = torch.from_numpy(ref.download()).cuda()
    esdf.export_dev(idx_t.data_ptr(), vox_t.data_ptr(), nb)
    g_idx, g_vox, g_pts = (torch.empty((1,) + tuple(t.shape), dtype=t.dtype, device="cuda") for t in (idx_t, vox_t, pts_t))
    dist.all_gather_into_tensor(g_idx, idx_t)
    dist.all_gather_into_tensor(g_vox, vox_t)
    dist.all_gather_into_tensor(g_pts, pts_t)
    torch.cuda.synchronize()
    lb = Layer(eng, voxel, capacity_blocks=max(64, nb))
    lb.upload_dev(g_idx[0].data_ptr(), g_vox[0].data_ptr(), nb)
    pa = RegPoints.from_device(eng, g_pts[0].data_ptr(), npts)
    assert lb.stats()[0] == nb and pa.n == npts
    # the pose graph's packed all-reduce on the device (posegraph.py, backend "nccl") == the sum without a group
    results = []
    for group in (None, dist.group.WORLD):
        pg = PoseGraph()
        pg.add_node(0, [0.0, 0.0, 0.0, 0.0], constant=True)
        pg.add_node(1, [0.02, -0.01, 0.0, 0.002])
        reg = Registration(eng, pa, lb)
        reg.draw_samples(int(0.3 * pa.n), 1234)
        pg.reg.append(RegistrationConstraint(0, 1, reg))
        cost, g, H, _ = pg.build({k: v.copy() for k, v in pg.poses.items()}, group=group)
        results.append((cost, g.copy(), H.copy()))
    assert results[0][0] > 0 and np.count_nonzero(results[0][2]) > 0
    assert results[0][0] == results[1][0] and np.array_equal(results[0][1], results[1][1]) and np.array_equal(results[0][2], results[1][2])
    # the same two collectives through the engine's own communicator (cox_comm_*: RCCL behind the C ABI, what the C++ host and
    # bench.py's COX_DIST_BACKEND=rccl-capi use): the gathered bytes and the reduced normal equations are the same
    from coxgraph_amd.capi import Comm  # noqa: E402
    comm = Comm(eng, 0, 0, 1, Comm.unique_id(eng))
    c_idx, c_vox, c_pts = (torch.zeros((1,) + tuple(t.shape), dtype=t.dtype, device="cuda") for t in (idx_t, vox_t, pts_t))
    cur = torch.cuda.current_stream().cuda_stream
    for src, dst in ((idx_t, c_idx), (vox_t, c_vox), (pts_t, c_pts)):
        comm.allgather_dev(src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size(), cur)
    assert torch.equal(c_idx, g_idx) and torch.equal(c_vox, g_vox) and torch.equal(c_pts, g_pts)
    pg = PoseGraph()
    pg.add_node(0, [0.0, 0.0, 0.0, 0.0], constant=True)
    pg.add_node(1, [0.02, -0.01, 0.0, 0.002])
    reg = Registration(eng, pa, lb)
    reg.draw_samples(int(0.3 * pa.n), 1234)
    pg.reg.append(RegistrationConstraint(0, 1, reg))
    cost_c, g_c, H_c, _ = pg.build({k: v.copy() for k, v in pg.poses.items()}, comm=comm)
    assert cost_c == results[0][0] and np.array_equal(g_c, results[0][1]) and np.array_equal(H_c, results[0][2])
    comm.close()
    tt = torch.tensor([1.5, 2.0], dtype=torch.float64, device="cuda")
    dist.all_reduce(tt[:1], op=dist.ReduceOp.MAX)
    dist.all_reduce(tt[1:], op=dist.ReduceOp.SUM)
    assert tt.tolist() == [1.5, 2.0]
    dist.barrier()
    dist.destroy_process_group()
    print("OK", nb, npts, results[0][0])


if __name__ == "__main__":
    main()
