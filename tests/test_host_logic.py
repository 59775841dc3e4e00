"""Host-side logic that needs no GPU: pose-graph costs/solver (driving the ORACLE registration cost as the
evaluator -- tests only), C++ adapters compile and link against the C ABI, bench.py's CPU-side pieces."""
import os
import subprocess
import sys

import numpy as np
import pytest

from coxgraph_amd.capi import Layer, RegPoints, Registration
from coxgraph_amd.posegraph import PoseGraph, PoseGraphInterface, RelativePoseConstraint, normalize_angle, sqrt_information

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_normalize_angle_and_sqrt_information():
    assert abs(normalize_angle(3.5 * np.pi) - (-0.5 * np.pi)) < 1e-12
    assert abs(normalize_angle(-np.pi) - (-np.pi)) < 1e-12 and abs(normalize_angle(np.pi) - (-np.pi)) < 1e-12
    info = np.diag([100.0, 100.0, 250.0, 250.0])  # coxgraph/config/server.yaml:37-43
    L = sqrt_information(info)
    assert np.allclose(L.T @ L, info)
    semi = np.diag([1.0, 0.0, 4.0, 9.0])
    L = sqrt_information(semi)
    assert np.allclose(L.T @ L, semi)


def test_relative_pose_cost_known_answers_and_jacobian():
    """Appendix D.8: A = (0,0,0,0), B = (1,0,0,pi/2), obs = same -> r = 0; yaw wraps at +-pi."""
    c = RelativePoseConstraint(0, 1, [1, 0, 0, np.pi / 2], np.eye(4))
    r, Ja, Jb = c.evaluate(np.zeros(4), np.array([1.0, 0, 0, np.pi / 2]))
    assert np.allclose(r, 0)
    c2 = RelativePoseConstraint(0, 1, [0, 0, 0, np.pi - 0.01], np.eye(4))
    r, _, _ = c2.evaluate(np.zeros(4), np.array([0, 0, 0, -np.pi + 0.01]))
    assert abs(r[3] - 0.02) < 1e-12  # shortest way round, not 2*pi - 0.02
    rng = np.random.default_rng(0)
    c3 = RelativePoseConstraint(0, 1, rng.normal(size=4), np.diag([100.0, 100.0, 250.0, 250.0]))
    pa, pb = rng.normal(size=4), rng.normal(size=4)
    r0, Ja, Jb = c3.evaluate(pa, pb)
    h = 1e-6
    for k in range(4):
        e = np.zeros(4)
        e[k] = h
        assert np.allclose((c3.evaluate(pa + e, pb)[0] - c3.evaluate(pa - e, pb)[0]) / (2 * h), Ja[:, k], atol=1e-5)
        assert np.allclose((c3.evaluate(pa, pb + e)[0] - c3.evaluate(pa, pb - e)[0]) / (2 * h), Jb[:, k], atol=1e-5)


def _linear(x):
    r = 2.0 * x[0] - 6.0
    return 0.5 * r * r, np.array([2.0 * r]), np.array([[4.0]])


def _cubic(x):
    r, J = x[0] ** 3 - 1.0, 3.0 * x[0] ** 2
    return 0.5 * r * r, np.array([J * r]), np.array([[J * J]])


def test_trust_region_trace_derived_by_hand():
    """Ceres' policy (backend/pose_graph.h:56-68 + Solver::Options defaults) on r = 2x - 6 from x = 0, by hand: with Jacobi scaling the
    LM step of a one-parameter linear problem is the Gauss-Newton step times R / (R + 1) (R = trust-region radius, 1e4 at first), the
    model is exact (rho = 1), so the radius triples; the second step, 3e-4, is below parameter_tolerance * (|x| + parameter_tolerance)
    = 3e-3 * 3.0027 and the solve ends BEFORE taking it."""
    from coxgraph_amd.posegraph import trust_region_minimize, TrustRegionOptions
    x, s = trust_region_minimize(_linear, [0.0], lambda x, d: x + d)
    x1 = 3.0 * 1e4 / (1e4 + 1.0)
    assert s["termination"] == "CONVERGENCE" and s["message"] == "Parameter tolerance reached." and s["iterations"] == 2 and s["evaluations"] == 3
    assert abs(x[0] - x1) < 1e-15
    t = s["trace"]
    assert t[0]["cost"] == 18.0 and t[0]["trust_region_radius"] == 1e4
    assert abs(t[1]["step_norm"] - x1) < 1e-15 and abs(t[1]["relative_decrease"] - 1.0) < 1e-9 and t[1]["trust_region_radius"] == 3e4 and t[1]["step_is_successful"]
    assert abs(t[1]["cost"] - 0.5 * (6.0 / 10001.0) ** 2) < 1e-18
    assert abs(t[2]["step_norm"] - (3.0 - x1) * 3e4 / (3e4 + 1.0)) < 1e-15 and not t[2]["step_is_successful"]
    # without the parameter test the residual shrinks by (R + 1) per step, R = 1e4, 3e4, 9e4: |gradient| = 2 |r| = 4.4e-13 <= 1e-10 after three
    x, s = trust_region_minimize(_linear, [0.0], lambda x, d: x + d, TrustRegionOptions(parameter_tolerance=0.0))
    assert s["message"] == "Gradient tolerance reached." and s["iterations"] == 3
    assert abs((2.0 * x[0] - 6.0) - (-6.0 / 10001.0 / 30001.0 / 90001.0)) < 2e-15  # (-2.2e-13, to the rounding of 2x - 6 at x = 3)
    # a start at the optimum ends in iteration 0
    x, s = trust_region_minimize(_linear, [3.0], lambda x, d: x + d)
    assert s["iterations"] == 0 and s["message"] == "Gradient tolerance reached."
    # budget checks come before every iteration
    x, s = trust_region_minimize(_linear, [0.0], lambda x, d: x + d, TrustRegionOptions(max_num_iterations=1, parameter_tolerance=0.0))
    assert s["termination"] == "NO_CONVERGENCE" and s["iterations"] == 1
    x, s = trust_region_minimize(_linear, [0.0], lambda x, d: x + d, TrustRegionOptions(max_solver_time_in_seconds=0.0))
    assert s["termination"] == "NO_CONVERGENCE" and s["message"] == "Maximum solver time reached." and s["iterations"] == 0


def test_trust_region_rejected_steps_shrink_the_radius_as_ceres_does():
    """r = x^3 - 1 from x = 0.1: the Jacobian is tiny there, the first steps overshoot by orders of magnitude and are rejected
    (rho < min_relative_decrease); every rejection divides the radius by a factor that doubles (2, 4, 8, ...), an accepted step
    resets the factor -- radius after k rejections in a row = 1e4 / 2^(k (k + 1) / 2), by hand."""
    from coxgraph_amd.posegraph import trust_region_minimize, TrustRegionOptions
    x, s = trust_region_minimize(_cubic, [0.1], lambda x, d: x + d, TrustRegionOptions(parameter_tolerance=1e-12))
    t = s["trace"]
    k = 0
    while not t[k + 1]["step_is_successful"]:
        k += 1
        assert t[k]["trust_region_radius"] == 1e4 / 2.0 ** (k * (k + 1) // 2), (k, t[k])
    assert k >= 5 and s["unsuccessful_steps"] >= k
    costs = [it["cost"] for it in t]
    assert all(b <= a for a, b in zip(costs, costs[1:]))  # monotonic steps only
    assert abs(x[0] - 1.0) < 1e-6 and s["termination"] == "CONVERGENCE"


def test_cpp_trust_region_minimiser_agrees_with_the_python_one(hip, tmp_path):
    from coxgraph_amd.posegraph import trust_region_minimize, TrustRegionOptions
    out = subprocess.run([_build_posegraph_smoke(hip, tmp_path), "lm"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = {l.split()[0]: l for l in out.stdout.strip().splitlines()}
    head, trace = lines["linear"].split("|")
    x, s = trust_region_minimize(_linear, [0.0], lambda x, d: x + d)
    assert float(head.split()[1]) == x[0] and int(head.split()[2]) == s["iterations"] and "Parameter tolerance reached." in head
    vals = [float(v) for v in trace.split()]
    for k, it in enumerate(s["trace"]):
        got = vals[5 * k:5 * k + 5]
        want = [it["iteration"], it["cost"], it["step_norm"], it["relative_decrease"], it["trust_region_radius"]]
        assert np.allclose(got, want, rtol=1e-14, atol=0), (k, got, want)
    assert "Gradient tolerance reached." in lines["linear_gradient"] and int(lines["linear_gradient"].split()[2]) == 3
    head, trace = lines["cubic"].split("|")
    x, s = trust_region_minimize(_cubic, [0.1], lambda x, d: x + d, TrustRegionOptions(parameter_tolerance=1e-12))
    h = head.split()
    assert abs(float(h[1]) - x[0]) < 1e-12 and (int(h[2]), int(h[3]), int(h[4])) == (s["iterations"], s["successful_steps"], s["unsuccessful_steps"])
    vals = [float(v) for v in trace.split()]
    for k, it in enumerate(s["trace"]):
        assert (int(vals[4 * k]), int(vals[4 * k + 1])) == (it["iteration"], int(it["step_is_successful"]))
        assert np.isclose(vals[4 * k + 2], it["cost"], rtol=1e-7, atol=1e-18) and np.isclose(vals[4 * k + 3], it["trust_region_radius"], rtol=1e-12)


def test_loop_closure_chain_optimises_to_consistent_poses():
    g = PoseGraphInterface()
    truth = {0: np.array([0.0, 0, 0, 0]), 1: np.array([1.0, 0.5, 0.1, 0.3]), 2: np.array([2.0, -0.5, 0.2, -0.2])}
    rng = np.random.default_rng(1)
    for k, p in truth.items():
        g.addSubmap(k, p + (0 if k == 0 else rng.normal(scale=0.2, size=4)))

    def rel(a, b):
        c, s = np.cos(truth[a][3]), np.sin(truth[a][3])
        d = truth[b][:3] - truth[a][:3]
        return [c * d[0] + s * d[1], -s * d[0] + c * d[1], d[2], truth[b][3] - truth[a][3]]

    for a, b in ((0, 1), (1, 2), (0, 2)):
        g.addLoopClosureMeasurement(a, b, rel(a, b))
    first, second = g.optimize(enable_registration=False)
    poses = g.getPoseMap()
    # coxgraph's parameter_tolerance is 3e-3 (backend/pose_graph.h:60): Ceres stops as soon as a step is that small relative to |x|,
    # WITHOUT taking it -- the poses agree with the truth to a fraction of that, not to rounding
    assert "Parameter tolerance reached." in (first["message"], second["message"])
    for k in truth:
        assert np.allclose(poses[k], truth[k], atol=1e-3), (k, poses[k])
    assert np.array_equal(poses[0], truth[0])  # submap 0 is constant (pose_graph_interface.cpp:20-25)
    assert second["final_cost"] < 1e-4
    # with Ceres' own default (1e-8) the same graph converges to rounding
    from coxgraph_amd.posegraph import TrustRegionOptions
    g.pose_graph.optimize(exclude_registration=True, options=TrustRegionOptions(parameter_tolerance=1e-8, function_tolerance=0.0))
    for k in truth:
        assert np.allclose(g.getPoseMap()[k], truth[k], atol=1e-6), (k, g.getPoseMap()[k])


def plane_layer(eng, voxel=0.1, a=-1.0, b=(1.0, 0.0, 0.0), blocks=None):
    blocks = blocks or [(x, y, z) for x in range(-1, 3) for y in range(-1, 3) for z in range(-1, 2)]
    idx = np.array(blocks, np.int32)
    vox = np.zeros((len(blocks), 4096, 3), np.uint32)
    lin = np.arange(4096)
    loc = np.stack([lin % 16, (lin // 16) % 16, lin // 256], axis=1)
    for i, blk in enumerate(blocks):
        c = (np.array(blk)[None, :] * 16 + loc + 0.5) * voxel
        vox[i, :, 0] = (a + c @ np.array(b)).astype(np.float32).view(np.uint32)
        vox[i, :, 1] = np.float32(2.0).view(np.uint32)
    layer = Layer(eng, voxel)
    layer.upload(idx, vox)
    return layer


def corner_problem(eng):
    """Reading submap = three orthogonal planes x=1, y=0.8, z=0.5 meeting in a corner is not expressible as one linear
    field, so use a smooth quadratic-free surrogate: distance to the plane x + 0.5 y + 0.25 z = 1.2 (normalised); yaw and all
    three translations are observable from points spread in space only along the normal, so anchor the rest by loop closure."""
    n = np.array([1.0, 0.5, 0.25])
    n /= np.linalg.norm(n)
    layer = plane_layer(eng, a=-1.2 / np.linalg.norm([1.0, 0.5, 0.25]), b=tuple(n))
    rng = np.random.default_rng(4)
    xyz = rng.uniform([0.0, 0.0, 0.0], [2.0, 2.0, 1.0], (400, 3))
    d = xyz @ n - 1.2 / np.linalg.norm([1.0, 0.5, 0.25])
    keep = np.abs(d) < 0.25
    pts = np.concatenate([xyz[keep], d[keep, None], rng.uniform(0.5, 2.0, (keep.sum(), 1))], axis=1).astype(np.float32)
    return layer, pts, n


def test_registration_constraint_pulls_pose_onto_the_surface(oracle):
    layer, pts, n = corner_problem(oracle)
    reg = Registration(oracle, RegPoints(oracle, pts), layer)
    g = PoseGraphInterface()
    g.addSubmap(0, [0, 0, 0, 0])
    off = 0.08 * n  # reading submap displaced along the surface normal: the registration cost sees exactly this
    g.addSubmap(1, [off[0], off[1], off[2], 0.0])
    g.addForceRegistrationConstraint(0, 1, reg)
    # weak prior keeps the unobservable directions where they are
    g.pose_graph.rel.append(RelativePoseConstraint(0, 1, [off[0], off[1], off[2], 0.0], np.eye(4) * 1e-3))
    before = reg.normal_eq(np.zeros(4), g.getPoseMap()[1])[2]
    g.optimize(enable_registration=True)
    p1 = g.getPoseMap()[1]
    after = reg.normal_eq(np.zeros(4), p1)[2]
    assert before > 1e-3 and after < 1e-6 * max(1.0, before)
    assert abs(float(p1[:3] @ n)) < 2e-3  # the displacement along the normal is gone


def test_forced_registration_constraint_is_optimised_whatever_the_flag(oracle):
    """pose_graph_interface.cpp:32-41: the second solve always runs with all constraints; enable_registration only gates
    updateRegistrationConstraints() (the overlap-driven constraints) between the two solves."""
    layer, pts, n = corner_problem(oracle)
    off = 0.08 * n
    calls = []

    def build(enable):
        reg = Registration(oracle, RegPoints(oracle, pts), layer)
        g = PoseGraphInterface()
        g.addSubmap(0, [0, 0, 0, 0])
        g.addSubmap(1, [off[0], off[1], off[2], 0.0])
        g.addForceRegistrationConstraint(0, 1, reg)
        g.pose_graph.rel.append(RelativePoseConstraint(0, 1, [off[0], off[1], off[2], 0.0], np.eye(4) * 1e-3))
        g.overlap_pairs = lambda poses: calls.append(enable) or []
        g.constraint_factory = lambda a, b: None
        g.optimize(enable_registration=enable)
        return g.getPoseMap()[1]

    p_off, p_on = build(False), build(True)
    assert abs(float(p_off[:3] @ n)) < 2e-3 and np.allclose(p_off, p_on)
    assert calls == [True]   # the overlap refresh ran only where it was enabled
    g = PoseGraphInterface()
    g.addSubmap(0, [0, 0, 0, 0])
    g.addSubmap(1, [1, 0, 0, 0], client_id=0)
    g.addSubmap(2, [2, 0, 0, 0.1], client_id=0)
    g.updateSubmapRPConstraints()
    assert len(g.evaluateResiduals("SubmapRelPose")) == 8 and np.allclose(g.evaluateResiduals("SubmapRelPose"), 0)
    g.resetSubmapRelativePoseConstrains()
    assert len(g.evaluateResiduals("SubmapRelPose")) == 0 and not g.checkLoopClosureCandidates()
    g.addLoopClosureMeasurement(0, 2, [2.1, 0, 0, 0.1])
    assert g.checkLoopClosureCandidates() and abs(g.evaluateResiduals("RelPose")[0] - (-1.0)) < 1e-9   # sqrt(100) * (2.0 - 2.1)


def test_cpp_adapters_compile_and_link_against_the_c_abi(hip, tmp_path):
    """The reference-shaped C++ wrappers (coxgraph_amd/host) build with -std=c++14 like the reference
    (coxgraph/CMakeLists.txt:4) and link against the shared library; without a GPU the program reports so."""
    exe = str(tmp_path / "adapter_smoke")
    libdir = os.path.dirname(hip.path)
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp"),
                           "-L" + libdir, "-lcoxgraph_hip", "-Wl,-rpath," + libdir])
    rc = subprocess.call([exe])
    assert rc == (0 if hip.device_count() > 0 else 77)


@pytest.mark.gpu
def test_cpp_adapters_run_on_the_gpu(hip, tmp_path):
    exe = str(tmp_path / "adapter_smoke")
    libdir = os.path.dirname(hip.path)
    subprocess.check_call(["g++", "-std=c++14", "-o", exe, os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp"), "-L" + libdir, "-lcoxgraph_hip",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def _build_posegraph_smoke(hip, tmp_path):
    exe = str(tmp_path / "posegraph_smoke")
    libdir = os.path.dirname(hip.path)
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "posegraph_smoke.cpp"),
                           "-L" + libdir, "-lcoxgraph_hip", "-Wl,-rpath," + libdir])
    return exe


def test_cpp_pose_graph_agrees_with_the_python_one(hip, tmp_path):
    """coxgraph_amd/host/coxgraph_hip_posegraph.hpp (the C++ host side of the server's pose graph) against
    coxgraph_amd/posegraph.py on a 5-submap, 2-client graph with loop closures and consecutive-submap constraints
    (no registration constraint, so no GPU is needed): same costs, same iteration counts, same poses."""
    from coxgraph_amd.posegraph import PoseGraphInterface
    out = subprocess.run([_build_posegraph_smoke(hip, tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    summary = [float(x) for x in lines[0].split()[1:]]
    cpp = {int(l.split()[1]): np.array([float(x) for x in l.split()[2:]]) for l in lines[1:] if l.startswith("pose")}
    pg = PoseGraphInterface()
    for sid, p, c in [(0, (0, 0, 0, 0), 0), (1, (1.0, 0.1, 0.0, 0.05), 0), (2, (2.1, 0.3, 0.02, 0.12), 0), (3, (0.2, 1.9, -0.03, 1.50), 1),
                      (4, (0.1, 3.1, 0.01, 1.62), 1)]:
        pg.addSubmap(sid, p, c)
    pg.updateSubmapRPConstraints()
    pg.addLoopClosureMeasurement(0, 3, [0.15, 2.05, 0.0, 1.57])
    pg.addLoopClosureMeasurement(2, 4, [-1.8, 2.9, 0.05, 1.49])
    pg.addLoopClosureMeasurement(1, 2, [1.05, -0.05, 0.01, 0.02])
    first, second = pg.optimize(False)
    assert abs(summary[0] - second["initial_cost"]) < 1e-9 and abs(summary[1] - second["final_cost"]) < 1e-9
    assert (int(summary[2]), int(summary[3])) == (first["iterations"], second["iterations"])
    for sid, pose in pg.getPoseMap().items():
        assert np.allclose(cpp[sid], pose, rtol=0, atol=1e-9), (sid, cpp[sid], pose)


@pytest.mark.gpu
def test_cpp_pose_graph_with_a_registration_constraint_on_the_gpu(hip, tmp_path):
    out = subprocess.run([_build_posegraph_smoke(hip, tmp_path), "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "registration" in out.stdout
