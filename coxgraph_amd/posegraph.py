"""Server-side pose graph over 4-DoF (x, y, z, yaw) submap poses -- the caller of the registration kernel.

Host-side mirror of what coxgraph's server drives through voxgraph + Ceres:
  PoseGraphInterface::addSubmap / addLoopClosureMeasurement / addForceRegistrationConstraint /
  updateSubmapRPConstraints / optimize(enable_registration)     coxgraph/src/server/pose_graph_interface.cpp:10-105
  relative-pose residual  r = sqrt_information * e              coxgraph/include/coxgraph/server/backend/relative_pose_constraint.h:28-61,114-119
  node 0 constant, yaw local parameterisation                   pose_graph_interface.cpp:20-25, backend/node_collection.h:22-24
  solver budget: parameter_tolerance 3e-3                       backend/pose_graph.h:60-64

Ceres is replaced by a small dense Levenberg-Marquardt (the graphs have tens of nodes).  Registration
constraints are evaluated on the GPU in their fused form (cox_reg_normal_eq: H 8x8, b 8, cost); with more
than one rank they are dealt round-robin and every LM evaluation sums ONE packed buffer of
(4N)^2 + 4N + 1 doubles with a single all-reduce -- never one call per constraint.
All ranks then solve the same small system redundantly, so no broadcast is needed.
"""
import math

import numpy as np


def normalize_angle(a):
    """ceres-style NormalizeAngle: a - 2*pi*floor((a + pi) / (2*pi))."""
    return a - 2.0 * math.pi * math.floor((a + math.pi) / (2.0 * math.pi))


def sqrt_information(info):
    """Cholesky factor L^T of the 4x4 information matrix (relative_pose_constraint.h:28-61: LLT, LDLT root if semi-definite)."""
    info = np.asarray(info, np.float64)
    try:
        return np.linalg.cholesky(info).T
    except np.linalg.LinAlgError:
        w, v = np.linalg.eigh(info)
        return (v * np.sqrt(np.clip(w, 0.0, None))).T


class RelativePoseConstraint:
    """voxgraph RelativePoseCostFunction <4,4,4>: e = [Rz(yaw_A)^T (t_B - t_A) - t_obs ; wrap(yaw_B - yaw_A - yaw_obs)]."""

    def __init__(self, a, b, T_ab, information):
        self.a, self.b = a, b
        self.obs = np.asarray(T_ab, np.float64)  # x, y, z, yaw of B in A
        self.sqrt_info = sqrt_information(information)

    def evaluate(self, pa, pb):
        c, s = math.cos(pa[3]), math.sin(pa[3])
        d = pb[:3] - pa[:3]
        e = np.array([c * d[0] + s * d[1] - self.obs[0], -s * d[0] + c * d[1] - self.obs[1], d[2] - self.obs[2],
                      normalize_angle(pb[3] - pa[3] - self.obs[3])])
        Ja = np.array([[-c, -s, 0.0, -s * d[0] + c * d[1]],
                       [s, -c, 0.0, -c * d[0] - s * d[1]],
                       [0.0, 0.0, -1.0, 0.0],
                       [0.0, 0.0, 0.0, -1.0]])
        Jb = np.array([[c, s, 0.0, 0.0], [-s, c, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
        return self.sqrt_info @ e, self.sqrt_info @ Ja, self.sqrt_info @ Jb


class RegistrationConstraint:
    """voxgraph RegistrationConstraint: reference submap a (registration points) against reading submap b (TSDF layer).

    `registration` is a coxgraph_amd.capi.Registration (any engine); `sample_idx` are the weighted sampler's draws.
    """

    def __init__(self, a, b, registration, sample_idx=None):
        self.a, self.b, self.reg, self.sample_idx = a, b, registration, sample_idx

    def normal_eq(self, pa, pb):
        H, g, cost, _ = self.reg.normal_eq(pa, pb, self.sample_idx)
        return H, g, cost

    # split form: all constraints of one evaluation are begun before any is waited for (their kernels overlap, one round of
    # launch + PCIe latency per solver iteration instead of one per constraint)
    def begin(self, pa, pb):
        self.reg.normal_eq_begin(pa, pb, self.sample_idx)

    def finish(self):
        H, g, cost, _ = self.reg.normal_eq_finish()
        return H, g, cost


class PoseGraph:
    def __init__(self):
        self.poses = {}       # id -> np.float64[4]
        self.constant = set()
        self.rel = []         # RelativePoseConstraint
        self.submap_rel = []  # consecutive-submap relative poses (reset every update, pose_graph_interface.cpp:51-71)
        self.reg = []         # forced RegistrationConstraints (addForceRegistrationConstraint): kept
        self.overlap_reg = []  # overlap-driven ones: rebuilt by updateRegistrationConstraints()
        self.last_summary = None

    # ---- nodes / constraints ---------------------------------------------------------------------------------
    def add_node(self, node_id, pose, constant=False):
        self.poses[node_id] = np.array(pose, np.float64)
        if constant:
            self.constant.add(node_id)

    def _free_index(self):
        free = [i for i in sorted(self.poses) if i not in self.constant]
        return {nid: k for k, nid in enumerate(free)}, free

    # ---- one evaluation of the whole problem: cost, gradient, Gauss-Newton matrix ------------------------------
    def build(self, poses, exclude_registration=False, group=None):
        idx, free = self._free_index()
        n = 4 * len(free)
        H = np.zeros((n, n))
        g = np.zeros(n)
        cost = 0.0

        def scatter(nid_a, nid_b, Haa, Hab, Hbb, ga, gb):
            ia, ib = idx.get(nid_a), idx.get(nid_b)
            if ia is not None:
                H[4 * ia:4 * ia + 4, 4 * ia:4 * ia + 4] += Haa
                g[4 * ia:4 * ia + 4] += ga
            if ib is not None:
                H[4 * ib:4 * ib + 4, 4 * ib:4 * ib + 4] += Hbb
                g[4 * ib:4 * ib + 4] += gb
            if ia is not None and ib is not None:
                H[4 * ia:4 * ia + 4, 4 * ib:4 * ib + 4] += Hab
                H[4 * ib:4 * ib + 4, 4 * ia:4 * ia + 4] += Hab.T

        # registration constraints: dealt round-robin over ranks, summed with ONE all-reduce of a packed buffer
        all_reg = self.reg + self.overlap_reg
        if not exclude_registration and all_reg:
            rank, world = (0, 1)
            if group is not None:
                import torch.distributed as dist
                rank, world = dist.get_rank(group), dist.get_world_size(group)
            Hr = np.zeros((n, n))
            gr = np.zeros(n)
            cr = 0.0
            H_save, g_save = H, g
            H, g = Hr, gr
            mine = [c for k, c in enumerate(all_reg) if k % world == rank and c is not None]
            for c in mine:
                c.begin(poses[c.a], poses[c.b])
            for c in mine:
                H8, g8, ck = c.finish()
                scatter(c.a, c.b, H8[:4, :4], H8[:4, 4:], H8[4:, 4:], g8[:4], g8[4:])
                cr += ck
            H, g = H_save, g_save
            if group is not None:  # (also with one rank: the same code path a rehearsal on one GPU exercises)
                import torch
                import torch.distributed as dist
                buf = torch.from_numpy(np.concatenate([Hr.reshape(-1), gr, [cr]]))
                if dist.get_backend(group) == "nccl":
                    buf = buf.cuda()
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
                buf = buf.cpu().numpy()
                Hr, gr, cr = buf[:n * n].reshape(n, n), buf[n * n:n * n + n], float(buf[-1])
            H += Hr
            g += gr
            cost += cr
        # relative-pose constraints: tiny, evaluated redundantly on every rank (after the reduction, so they count once)
        for c in self.rel + self.submap_rel:
            r, Ja, Jb = c.evaluate(poses[c.a], poses[c.b])
            scatter(c.a, c.b, Ja.T @ Ja, Ja.T @ Jb, Jb.T @ Jb, Ja.T @ r, Jb.T @ r)
            cost += 0.5 * float(r @ r)
        return cost, g, H, free

    # ---- Levenberg-Marquardt ------------------------------------------------------------------------------------
    def optimize(self, exclude_registration=False, max_iterations=50, parameter_tolerance=3e-3, group=None):
        poses = {k: v.copy() for k, v in self.poses.items()}
        cost, g, H, free = self.build(poses, exclude_registration, group)
        lam, it, n_eval = 1e-4, 0, 1
        initial = cost
        if free:
            for it in range(1, max_iterations + 1):
                A = H + lam * np.diag(np.maximum(np.diag(H), 1e-12))
                try:
                    delta = np.linalg.solve(A, -g)
                except np.linalg.LinAlgError:
                    lam *= 10.0
                    continue
                trial = {k: v.copy() for k, v in poses.items()}
                for k, nid in enumerate(free):
                    trial[nid][:3] += delta[4 * k:4 * k + 3]
                    trial[nid][3] = normalize_angle(trial[nid][3] + delta[4 * k + 3])  # angle local parameterisation
                c2, g2, H2, _ = self.build(trial, exclude_registration, group)
                n_eval += 1
                if c2 < cost:
                    x_norm = math.sqrt(sum(float(poses[nid] @ poses[nid]) for nid in free))
                    poses, cost, g, H = trial, c2, g2, H2
                    lam = max(lam / 3.0, 1e-12)
                    if np.linalg.norm(delta) <= parameter_tolerance * (x_norm + parameter_tolerance):  # Ceres parameter_tolerance test
                        break
                else:
                    lam *= 4.0
                    if lam > 1e12:
                        break
        self.poses = poses
        self.last_summary = dict(initial_cost=initial, final_cost=cost, iterations=it, evaluations=n_eval)
        return self.last_summary


class PoseGraphInterface:
    """The facade coxgraph's server calls (coxgraph/include/coxgraph/server/pose_graph_interface.h:66-90)."""

    def __init__(self, loop_closure_information=None, submap_relative_pose_information=None):
        self.pose_graph = PoseGraph()
        # coxgraph/config/server.yaml:37-51
        self.lc_info = np.diag([100.0, 100.0, 250.0, 250.0]) if loop_closure_information is None else loop_closure_information
        self.sm_rp_info = np.diag([1000.0, 1000.0, 2500.0, 2500.0]) if submap_relative_pose_information is None else submap_relative_pose_information
        self.submaps = {}  # id -> dict(client)
        self.overlap_pairs = None        # callable(poses) -> [(a, b)]: which submaps' surface boxes overlap at these poses
        self.constraint_factory = None   # callable(a, b) -> RegistrationConstraint for an overlapping pair

    def addSubmap(self, submap_id, pose4, client_id=0):
        """pose_graph_interface.cpp:10-30: submap 0 is constant."""
        self.submaps[submap_id] = dict(client=client_id)
        self.pose_graph.add_node(submap_id, pose4, constant=(submap_id == 0))

    def addLoopClosureMeasurement(self, a, b, T_ab):
        self.pose_graph.rel.append(RelativePoseConstraint(a, b, T_ab, self.lc_info))
        return True

    def addForceRegistrationConstraint(self, a, b, registration, sample_idx=None):
        """pose_graph_interface.cpp:88-105."""
        self.pose_graph.reg.append(RegistrationConstraint(a, b, registration, sample_idx))

    def updateSubmapRPConstraints(self):
        """pose_graph_interface.cpp:51-71: consecutive submaps of one client keep their current relative pose."""
        self.pose_graph.submap_rel = []
        by_client = {}
        for sid in sorted(self.submaps):
            by_client.setdefault(self.submaps[sid]["client"], []).append(sid)
        for ids in by_client.values():
            for i, j in zip(ids[:-1], ids[1:]):
                pa, pb = self.pose_graph.poses[i], self.pose_graph.poses[j]
                c, s = math.cos(pa[3]), math.sin(pa[3])
                d = pb[:3] - pa[:3]
                T_ij = np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], d[2], normalize_angle(pb[3] - pa[3])])
                self.pose_graph.submap_rel.append(RelativePoseConstraint(i, j, T_ij, self.sm_rp_info))

    def resetSubmapRelativePoseConstrains(self):
        """(sic) pose_graph_interface.h:72-74."""
        self.pose_graph.submap_rel = []

    def updateRegistrationConstraints(self):
        """voxgraph PoseGraphInterface::updateRegistrationConstraints (called at pose_graph_interface.cpp:38): drop the
        overlap-driven constraints, constrain every pair whose surface boxes overlap at the current poses.  The overlap test
        and the constraint construction are callables set by the owner of the submaps (C++: coxgraph_hip_posegraph.hpp does
        both itself from its SubmapCollection)."""
        self.pose_graph.overlap_reg = []
        if self.overlap_pairs is None or self.constraint_factory is None:
            return
        for a, b in self.overlap_pairs(self.getPoseMap()):
            self.pose_graph.overlap_reg.append(self.constraint_factory(a, b))

    def evaluateResiduals(self, constraint_type):
        """'RelPose' (loop closures) or 'SubmapRelPose' (consecutive submaps): the stacked residual vectors at the current poses."""
        lst = self.pose_graph.rel if constraint_type == "RelPose" else self.pose_graph.submap_rel
        out = []
        for c in lst:
            out.extend(c.evaluate(self.pose_graph.poses[c.a], self.pose_graph.poses[c.b])[0])
        return np.array(out)

    def checkLoopClosureCandidates(self):
        return any(c.a in self.pose_graph.poses and c.b in self.pose_graph.poses for c in self.pose_graph.rel)

    def optimize(self, enable_registration=True, group=None):
        """pose_graph_interface.cpp:32-49: solve without registration constraints; refresh the overlap-driven ones if enabled;
        solve with ALL constraints -- forced registration constraints included, whatever the flag says."""
        first = self.pose_graph.optimize(exclude_registration=True, group=group)
        if enable_registration:
            self.updateRegistrationConstraints()
        second = self.pose_graph.optimize(exclude_registration=False, group=group)
        return first, second

    def getPoseMap(self):
        return {k: v.copy() for k, v in self.pose_graph.poses.items()}
