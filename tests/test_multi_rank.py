"""The N > 1 path on CPU: two ranks over gloo.

(1) fusion shards by client with no data-path collective: each rank builds its own client's stream; the only
    collectives are the harness barrier and the MAX-reduce of the timing (bench.py);
(2) registration: constraints are dealt round-robin over ranks and every LM evaluation sums ONE packed
    (4N)^2 + 4N + 1 buffer with a single all-reduce; all ranks end with identical poses, equal to the
    single-process result.  The registration cost is evaluated by the CPU ORACLE here (no GPU in this container).
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(eng):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_logic import plane_layer
    from coxgraph_amd.capi import RegPoints, Registration
    from coxgraph_amd.posegraph import PoseGraphInterface, RelativePoseConstraint
    normals = [np.array([1.0, 0.5, 0.25]), np.array([0.2, 1.0, -0.3]), np.array([-0.3, 0.2, 1.0])]
    g = PoseGraphInterface()
    g.addSubmap(0, [0, 0, 0, 0], client_id=0)
    keep = []
    rng = np.random.default_rng(9)
    for k, nrm in enumerate(normals, start=1):
        n = nrm / np.linalg.norm(nrm)
        layer = plane_layer(eng, a=-0.9, b=tuple(n))
        xyz = rng.uniform([0.0, 0.0, 0.0], [2.0, 2.0, 1.0], (300, 3))
        d = xyz @ n - 0.9
        m = np.abs(d) < 0.25
        pts = np.concatenate([xyz[m], d[m, None], rng.uniform(0.5, 2.0, (m.sum(), 1))], axis=1).astype(np.float32)
        reg = Registration(eng, RegPoints(eng, pts), layer)
        keep.append((layer, reg))
        off = 0.05 * k * n
        g.addSubmap(k, [off[0], off[1], off[2], 0.0], client_id=k % 2)
        g.addForceRegistrationConstraint(0, k, reg)
        g.pose_graph.rel.append(RelativePoseConstraint(0, k, [off[0], off[1], off[2], 0.0], np.eye(4) * 1e-3))
    return g, keep


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from coxgraph_amd.capi import Engine
    from coxgraph_amd import synth
    eng = Engine(os.path.join(ROOT, "oracle", "libcoxoracle.so"), "coxo_")
    # (1) client sharding: rank k integrates client k's stream only; timing is MAX-reduced like bench.py does
    T, pts, _, _ = synth.make_frame(3, client=rank, n_clients=world)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # (2) registration reduce
    g, keep = _problem(eng)
    g.optimize(enable_registration=True, group=dist.group.WORLD)
    poses = g.getPoseMap()
    out[rank] = dict(T=T.tolist(), n=int(pts.shape[0]), tmax=float(t.item()), poses={k: v.tolist() for k, v in poses.items()},
                     summary=g.pose_graph.last_summary)
    dist.destroy_process_group()


def test_two_ranks_gloo(oracle):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    # clients differ (different trajectories), timing is the max over ranks
    assert res[0]["T"] != res[1]["T"] and res[0]["n"] == res[1]["n"] == 307200
    assert res[0]["tmax"] == res[1]["tmax"] == pytest.approx(0.2)
    # every rank solved the same system after the all-reduce
    for k in res[0]["poses"]:
        assert res[0]["poses"][k] == res[1]["poses"][k]
    # and it equals the single-process result
    g, keep = _problem(oracle)
    g.optimize(enable_registration=True)
    single = g.getPoseMap()
    for k, v in single.items():
        assert np.allclose(v, res[0]["poses"][k], atol=1e-9), (k, v, res[0]["poses"][k])
    assert res[0]["summary"]["final_cost"] < 1e-4 * max(1.0, res[0]["summary"]["initial_cost"])


# ---- the server's inter-robot leg (bench.py --gpus N, BASELINE configs[2] / [4]) over gloo ---------------------------------
def _client_submap(eng, client, voxel=0.10):
    """One client's finished submap (oracle): TSDF -> ESDF layer + isosurface registration points."""
    from coxgraph_amd import synth
    from coxgraph_amd.capi import Layer, Integrator, RegPoints
    cfg = eng.default_config(**synth.integrator_overrides(voxel))
    layer = Layer(eng, voxel)
    integ = Integrator(eng, layer, cfg, "merged")
    for t in range(0, 60, 10):
        T, pts, rgba, _ = synth.make_frame(t, client=client, n_clients=12)   # 30 degrees apart: neighbours overlap
        integ.integrate_points(T, pts[::6], rgba[::6])
    esdf = layer.esdf(max_distance_m=2.0, min_distance_m=1.5 * voxel)
    pts = RegPoints.from_isosurface(eng, layer, 1.0)
    return esdf, pts


def _exchange_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from coxgraph_amd.capi import Engine, Layer, RegPoints, Registration
    from coxgraph_amd.posegraph import PoseGraph, RegistrationConstraint
    eng = Engine(os.path.join(ROOT, "oracle", "libcoxoracle.so"), "coxo_")
    esdf, pts = _client_submap(eng, rank)
    # exchange: wire arrays of the ESDF layer + the point set, padded to the largest rank, one all-gather each
    idx, vox = esdf.download()
    p = pts.download()
    sizes = torch.tensor([idx.shape[0], p.shape[0]], dtype=torch.int64)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = [tuple(int(v) for v in t.tolist()) for t in all_sizes]
    max_nb, max_np = max(s[0] for s in all_sizes), max(s[1] for s in all_sizes)

    def gather(a, n_max, tail, dtype):
        buf = torch.zeros((n_max,) + tail, dtype=dtype)
        buf[:a.shape[0]] = torch.from_numpy(a)
        lst = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(lst, buf)
        return [t.numpy() for t in lst]
    g_idx = gather(idx, max_nb, (3,), torch.int32)
    g_vox = gather(vox.view(np.int32), max_nb, (4096, 3), torch.int32)
    g_pts = gather(p, max_np, (5,), torch.float32)
    # constraints (a, b), a < b, dealt over the ranks: client a's points against client b's distance field
    pairs = [(a, b) for a in range(world) for b in range(a + 1, world)]
    pg = PoseGraph()
    for k in range(world):
        pg.add_node(k, [0.02 * k, -0.01 * k, 0.0, 0.002 * k], constant=(k == 0))
    keep = []
    for k, (a, b) in enumerate(pairs):
        if k % world != rank:
            pg.reg.append(None)
            continue
        lb = Layer(eng, 0.10)
        lb.upload(g_idx[b][:all_sizes[b][0]], g_vox[b][:all_sizes[b][0]].view(np.uint32))
        pa = RegPoints(eng, g_pts[a][:all_sizes[a][1]])
        reg = Registration(eng, pa, lb)
        reg.draw_samples(int(0.3 * pa.n), 1000 + k)
        keep.append((lb, pa, reg))
        pg.reg.append(RegistrationConstraint(a, b, reg))
    poses = {k: v.copy() for k, v in pg.poses.items()}
    cost, g, H, free = pg.build(poses, group=dist.group.WORLD)
    out[rank] = dict(cost=float(cost), g=g.tolist(), H=H.tolist(), sizes=all_sizes)
    dist.destroy_process_group()


def test_submap_exchange_and_inter_robot_registration_over_gloo(oracle):
    """Three ranks = three clients: every rank finishes its own submap, the submaps are all-gathered as wire arrays, each
    constraint registers CLIENT a's isosurface points against CLIENT b's ESDF on the rank that owns it, one all-reduce sums
    the packed normal equations -- every rank ends with the system a single process builds from all three submaps."""
    from coxgraph_amd.capi import Registration
    from coxgraph_amd.posegraph import PoseGraph, RegistrationConstraint
    world = 3
    port = 31500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_exchange_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    for r in range(1, world):
        assert res[r]["cost"] == res[0]["cost"] and res[r]["H"] == res[0]["H"] and res[r]["g"] == res[0]["g"]
    subs = [_client_submap(oracle, c) for c in range(world)]
    assert [tuple(s) for s in res[0]["sizes"]] == [(e.stats()[0], p.n) for e, p in subs]
    pg = PoseGraph()
    for k in range(world):
        pg.add_node(k, [0.02 * k, -0.01 * k, 0.0, 0.002 * k], constant=(k == 0))
    keep = []
    for k, (a, b) in enumerate([(a, b) for a in range(world) for b in range(a + 1, world)]):
        reg = Registration(oracle, subs[a][1], subs[b][0])
        reg.draw_samples(int(0.3 * subs[a][1].n), 1000 + k)
        keep.append(reg)
        pg.reg.append(RegistrationConstraint(a, b, reg))
    cost, g, H, _ = pg.build({k: v.copy() for k, v in pg.poses.items()})
    assert cost > 0 and np.count_nonzero(H) > 0      # neighbouring clients overlap: the constraints have correspondences
    assert np.isclose(cost, res[0]["cost"], rtol=1e-12) and np.allclose(g, res[0]["g"], rtol=1e-10, atol=1e-12) and np.allclose(H, res[0]["H"], rtol=1e-10, atol=1e-12)


def _worker_failing_rank(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    sys.path.insert(0, ROOT)
    from coxgraph_amd.capi import Engine
    eng = Engine(os.path.join(ROOT, "oracle", "libcoxoracle.so"), "coxo_")
    g, keep = _problem(eng)
    if rank == 1:  # this rank's share of the constraints cannot be evaluated (a pending begin without its finish)
        mine = [c for k, c in enumerate(g.pose_graph.reg) if k % world == rank]
        mine[0].reg.normal_eq_begin(np.zeros(4), np.zeros(4))
    try:
        g.pose_graph.build({k: v.copy() for k, v in g.pose_graph.poses.items()}, group=dist.group.WORLD)
        out[rank] = "no error"
    except RuntimeError as e:
        out[rank] = str(e)
    dist.barrier()  # both ranks are still in step: nobody is left waiting inside the all-reduce
    dist.destroy_process_group()


def test_a_failing_rank_does_not_leave_the_others_in_the_all_reduce(oracle):
    """a rank that cannot evaluate its constraints still joins the all-reduce (zero contribution + an error count), and
    EVERY rank raises -- nobody waits for ever."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_failing_rank, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert "failed on 1 rank" in res[0] and "failed on 1 rank" in res[1], res
