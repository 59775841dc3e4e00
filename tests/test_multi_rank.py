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
