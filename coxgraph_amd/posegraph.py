"""Server-side pose graph over 4-DoF (x, y, z, yaw) submap poses -- the caller of the registration kernel.

Host-side mirror of what coxgraph's server drives through voxgraph + Ceres:
  PoseGraphInterface::addSubmap / addLoopClosureMeasurement / addForceRegistrationConstraint /
  updateSubmapRPConstraints / optimize(enable_registration)     coxgraph/src/server/pose_graph_interface.cpp:10-105
  relative-pose residual  r = sqrt_information * e              coxgraph/include/coxgraph/server/backend/relative_pose_constraint.h:28-61,114-119
  node 0 constant, yaw local parameterisation                   pose_graph_interface.cpp:20-25, backend/node_collection.h:22-24
  solver budget: parameter_tolerance 3e-3                       backend/pose_graph.h:60-64

Ceres itself is not in the image; its trust-region minimiser is restated here from its published description (Solver::Options
defaults + the two values coxgraph overrides) on dense normal equations (the graphs have tens of nodes): TrustRegionOptions /
trust_region_minimize below, line for line the same policy as coxgraph_amd/host/coxgraph_hip_solver.hpp.  Registration
constraints are evaluated on the GPU in their fused form (cox_reg_normal_eq: H 8x8, b 8, cost); with more
than one rank they are dealt round-robin and every LM evaluation sums ONE packed buffer of
(4N)^2 + 4N + 1 doubles with a single all-reduce -- never one call per constraint.
All ranks then solve the same small system redundantly, so no broadcast is needed.
"""
import math
import time

import numpy as np


class TrustRegionOptions:
    """ceres::Solver::Options as coxgraph's server runs it (coxgraph/include/coxgraph/server/backend/pose_graph.h:56-68 sets
    parameter_tolerance 3e-3, max_solver_time_in_seconds 4, num_threads 4 and SPARSE_SCHUR; everything else is Ceres' default)."""
    max_num_iterations = 50
    max_solver_time_in_seconds = 4.0
    function_tolerance = 1e-6
    gradient_tolerance = 1e-10
    parameter_tolerance = 3e-3
    min_relative_decrease = 1e-3
    initial_trust_region_radius = 1e4
    max_trust_region_radius = 1e16
    min_trust_region_radius = 1e-32
    min_lm_diagonal = 1e-6
    max_lm_diagonal = 1e32
    max_num_consecutive_invalid_steps = 5
    jacobi_scaling = True

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(type(self), k):
                raise AttributeError(k)
            setattr(self, k, v)


def trust_region_minimize(evaluate, x0, plus, opt=None, clock=time.perf_counter):
    """Ceres' TrustRegionMinimizer with the LEVENBERG_MARQUARDT strategy (monotonic steps, no inner iterations, no bounds) on the
    normal equations.  evaluate(x) -> (cost, g = J^T f, H = J^T J); plus(x, delta) = the (local) parameterisation's Plus.

    Per iteration, as Ceres does it:
      step      D^2 = clamp(diag(H_s), min_lm_diagonal, max_lm_diagonal) / radius;  (H_s + D^2) d_s = -g_s;  d = scale * d_s
                (_s: Jacobian columns scaled by 1 / (1 + ||column||), the norms taken once at the initial point: jacobi_scaling)
      validity  model_cost_change = -d_s.(g_s + H_s d_s / 2) must be > 0, else the step is invalid: radius /= 2, and
                max_num_consecutive_invalid_steps of them in a row fail the solve
      stop      ||d|| <= parameter_tolerance (||x|| + parameter_tolerance), then |cost change| <= function_tolerance * cost --
                both BEFORE the step is taken, so x stays where it was
      accept    rho = cost change / model_cost_change > min_relative_decrease: x <- candidate, stop if max |gradient| <=
                gradient_tolerance, radius <- min(max_radius, radius / max(1/3, 1 - (2 rho - 1)^3)), decrease factor <- 2
      reject    radius /= decrease factor, decrease factor *= 2
      before every iteration: max_num_iterations, max_solver_time_in_seconds, radius <= min_trust_region_radius
    -> (x, summary)."""
    opt = opt or TrustRegionOptions()
    t_start = clock()
    x = np.array(x0, np.float64)
    n = x.size
    cost, g, H = evaluate(x)
    summary = dict(initial_cost=cost, final_cost=cost, iterations=0, evaluations=1, successful_steps=0, unsuccessful_steps=0, termination="CONVERGENCE",
                   message="", trace=[])

    def finish(term, msg):
        summary.update(final_cost=cost, termination=term, message=msg)
        return x, summary
    if n == 0:
        return finish("CONVERGENCE", "no free parameters")
    g = np.asarray(g, np.float64)
    H = np.asarray(H, np.float64)
    scale = 1.0 / (1.0 + np.sqrt(np.maximum(np.diag(H), 0.0))) if opt.jacobi_scaling else np.ones(n)

    def gradient_max_norm(xx, gg):
        return float(np.max(np.abs(xx - plus(xx, -gg))))
    gmax = gradient_max_norm(x, g)
    summary["trace"].append(dict(iteration=0, cost=cost, cost_change=0.0, gradient_max_norm=gmax, step_norm=0.0, relative_decrease=0.0,
                                 trust_region_radius=opt.initial_trust_region_radius, step_is_valid=False, step_is_successful=False))
    if gmax <= opt.gradient_tolerance:
        return finish("CONVERGENCE", "Gradient tolerance reached.")
    radius, decrease_factor, invalid = opt.initial_trust_region_radius, 2.0, 0
    x_norm = float(np.linalg.norm(x))
    iteration = 0
    while True:
        if iteration >= opt.max_num_iterations:
            return finish("NO_CONVERGENCE", "Maximum number of iterations reached.")
        if clock() - t_start >= opt.max_solver_time_in_seconds:
            return finish("NO_CONVERGENCE", "Maximum solver time reached.")
        if radius <= opt.min_trust_region_radius:
            return finish("CONVERGENCE", "Minimum trust region radius reached.")
        iteration += 1
        summary["iterations"] = iteration
        Hs = H * scale[:, None] * scale[None, :]
        gs = g * scale
        D2 = np.clip(np.diag(Hs), opt.min_lm_diagonal, opt.max_lm_diagonal) / radius
        it = dict(iteration=iteration, cost=cost, cost_change=0.0, gradient_max_norm=gmax, step_norm=0.0, relative_decrease=0.0, trust_region_radius=radius,
                  step_is_valid=False, step_is_successful=False)
        summary["trace"].append(it)
        valid = True
        try:
            ds = np.linalg.solve(Hs + np.diag(D2), -gs)
            valid = bool(np.all(np.isfinite(ds)))
        except np.linalg.LinAlgError:
            valid = False
        model_cost_change = 0.0
        if valid:
            model_cost_change = -float(ds @ (gs + 0.5 * (Hs @ ds)))
            valid = model_cost_change > 0.0
        if not valid:
            invalid += 1
            if invalid >= opt.max_num_consecutive_invalid_steps:
                return finish("FAILURE", "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps.")
            radius *= 0.5
            it["trust_region_radius"] = radius
            continue
        invalid = 0
        it["step_is_valid"] = True
        delta = ds * scale
        cand = plus(x, delta)
        c2, g2, H2 = evaluate(cand)
        summary["evaluations"] += 1
        step_norm = float(np.linalg.norm(x - cand))
        it["step_norm"] = step_norm
        it["cost_change"] = cost - c2
        if step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance):
            return finish("CONVERGENCE", "Parameter tolerance reached.")
        if abs(cost - c2) <= opt.function_tolerance * cost:
            return finish("CONVERGENCE", "Function tolerance reached.")
        rho = (cost - c2) / model_cost_change
        it["relative_decrease"] = rho
        if rho > opt.min_relative_decrease:
            x, cost, g, H = cand, c2, np.asarray(g2, np.float64), np.asarray(H2, np.float64)
            x_norm = float(np.linalg.norm(x))
            it["step_is_successful"] = True
            it["cost"] = cost
            summary["successful_steps"] += 1
            gmax = gradient_max_norm(x, g)
            it["gradient_max_norm"] = gmax
            if gmax <= opt.gradient_tolerance:
                return finish("CONVERGENCE", "Gradient tolerance reached.")
            radius = min(opt.max_trust_region_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            decrease_factor = 2.0
        else:
            summary["unsuccessful_steps"] += 1
            radius /= decrease_factor
            decrease_factor *= 2.0
        it["trust_region_radius"] = radius


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
    def build(self, poses, exclude_registration=False, group=None, comm=None):
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
            if comm is not None:  # coxgraph_amd.capi.Comm: RCCL behind the C ABI (what the C++ host uses)
                rank, world = comm.rank, comm.world
            elif group is not None:
                import torch.distributed as dist
                rank, world = dist.get_rank(group), dist.get_world_size(group)
            Hr = np.zeros((n, n))
            gr = np.zeros(n)
            cr = 0.0
            H_save, g_save = H, g
            H, g = Hr, gr
            mine = [c for k, c in enumerate(all_reg) if k % world == rank and c is not None]
            # A rank whose evaluation fails must still take part in the all-reduce (the others would wait in it until the backend's
            # timeout): it contributes zeros and one in the buffer's extra "failed ranks" word, and every rank raises.
            local_error = None
            try:
                batchable = len(mine) >= 2 and all(c.sample_idx is None and hasattr(type(c.reg), "normal_eq_batch") for c in mine)
                if batchable:  # ONE launch for this rank's constraints (cox_reg_normal_eq_batch)
                    results = type(mine[0].reg).normal_eq_batch([c.reg for c in mine], [poses[c.a] for c in mine], [poses[c.b] for c in mine])
                else:          # begin them all, then collect: their kernels overlap
                    for c in mine:
                        c.begin(poses[c.a], poses[c.b])
                    results = [c.finish() for c in mine]
                for c, res in zip(mine, results):
                    H8, g8, ck = res[0], res[1], res[2]
                    scatter(c.a, c.b, H8[:4, :4], H8[:4, 4:], H8[4:, 4:], g8[:4], g8[4:])
                    cr += ck
            except Exception as e:  # noqa: BLE001
                if group is None and comm is None:
                    raise
                local_error = e
                Hr[:] = 0.0
                gr[:] = 0.0
                cr = 0.0
            H, g = H_save, g_save
            if comm is not None:
                buf = comm.allreduce_f64(np.concatenate([Hr.reshape(-1), gr, [cr, 0.0 if local_error is None else 1.0]]))
                if buf[-1] != 0.0:
                    raise RuntimeError(f"registration constraints: evaluation failed on {int(buf[-1])} rank(s)") from local_error
                Hr, gr, cr = buf[:n * n].reshape(n, n), buf[n * n:n * n + n], float(buf[-2])
            elif group is not None:  # (also with one rank: the same code path a rehearsal on one GPU exercises)
                import torch
                import torch.distributed as dist
                buf = torch.from_numpy(np.concatenate([Hr.reshape(-1), gr, [cr, 0.0 if local_error is None else 1.0]]))
                if dist.get_backend(group) == "nccl":
                    buf = buf.cuda()
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
                buf = buf.cpu().numpy()
                if buf[-1] != 0.0:
                    raise RuntimeError(f"registration constraints: evaluation failed on {int(buf[-1])} rank(s)") from local_error
                Hr, gr, cr = buf[:n * n].reshape(n, n), buf[n * n:n * n + n], float(buf[-2])
            H += Hr
            g += gr
            cost += cr
        # relative-pose constraints: tiny, evaluated redundantly on every rank (after the reduction, so they count once)
        for c in self.rel + self.submap_rel:
            r, Ja, Jb = c.evaluate(poses[c.a], poses[c.b])
            scatter(c.a, c.b, Ja.T @ Ja, Ja.T @ Jb, Jb.T @ Jb, Ja.T @ r, Jb.T @ r)
            cost += 0.5 * float(r @ r)
        return cost, g, H, free

    # ---- ceres::Solve as coxgraph configures it (backend/pose_graph.h:56-68) ---------------------------------------------
    def optimize(self, exclude_registration=False, options=None, group=None, comm=None):
        opt = options or TrustRegionOptions()
        idx, free = self._free_index()
        base = {k: v.copy() for k, v in self.poses.items()}

        def unpack(x):
            poses = {k: v.copy() for k, v in base.items()}
            for k, nid in enumerate(free):
                poses[nid] = np.array(x[4 * k:4 * k + 4], np.float64)
            return poses

        def evaluate(x):
            cost, g, H, _ = self.build(unpack(x), exclude_registration, group, comm)
            return cost, g, H

        def plus(x, delta):  # x, y, z plain; yaw: voxgraph's AngleLocalParameterization (normalised sum)
            y = np.array(x, np.float64) + delta
            y[3::4] = [normalize_angle(a) for a in y[3::4]]
            return y
        x0 = np.concatenate([self.poses[nid] for nid in free]) if free else np.zeros(0)
        x, summary = trust_region_minimize(evaluate, x0, plus, opt)
        self.poses = unpack(x)
        self.last_summary = summary
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
