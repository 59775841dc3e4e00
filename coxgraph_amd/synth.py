"""Deterministic synthetic depth stream for the benchmark and parity tests.

Scene: axis-aligned room, interior [-4,4] x [-3,3] x [0,3] m (world frame G, z up) plus a sphere
r = 0.75 m at (2.5, 0.5, 1.0) (SURVEY.md proposed x = 1.5, which puts the surface 0.12 m from the camera
circle and makes a third of the frame invalid by min_ray_length; moved 1 m outward).  Depth by analytic ray/plane + ray/sphere intersection in float64,
rounded to float32.  Camera: pinhole, z-forward optical frame; the 640x480 intrinsics are the
RealSense ones the reference ships (coxgraph/config/realsense/rs_config_0.yaml:12-24).
Trajectory of client k of K: circle radius 1 m, height 1.5 m, yaw(t) = 2*pi*k/K + 0.36 deg * t looking
outward.  Points are produced in row-major pixel order (v outer, u inner) -- the order
depth_image_proc hands them to the integrator.
"""
import math
import numpy as np

INTRINSICS = {
    (640, 480): (611.16, 609.64, 323.45, 244.94),
    (1280, 720): (916.74, 916.74, 640.0, 360.0),
}
ROOM_MIN = np.array([-4.0, -3.0, 0.0])
ROOM_MAX = np.array([4.0, 3.0, 3.0])
SPHERE_C = np.array([2.5, 0.5, 1.0])
SPHERE_R = 0.75

# integrator parameters per voxel size, from the reference's yaml files
VOXEL_CONFIGS = {
    0.10: dict(default_truncation_distance=0.30, min_ray_length_m=0.2, max_ray_length_m=10.0),  # coxgraph_client.yaml:56-61
    0.05: dict(default_truncation_distance=0.15, min_ray_length_m=0.3, max_ray_length_m=6.0),   # tsdf_server_euroc.yaml:12-17
    0.02: dict(default_truncation_distance=0.06, min_ray_length_m=0.1, max_ray_length_m=3.0),   # tsdf_server_rs.yaml:12-17
    0.01: dict(default_truncation_distance=0.03, min_ray_length_m=0.1, max_ray_length_m=3.0),   # extrapolated (no 1 cm config in the reference)
}
COMMON_CONFIG = dict(use_const_weight=1, allow_clear=1, voxel_carving_enabled=1, max_weight=10000.0, use_weight_dropoff=1)


def integrator_overrides(voxel_size):
    key = min(VOXEL_CONFIGS, key=lambda v: abs(v - voxel_size))
    d = dict(COMMON_CONFIG)
    d.update(VOXEL_CONFIGS[key])
    return d


def quat_from_matrix(R):
    """Rotation matrix -> unit quaternion (w,x,y,z), float64."""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s])
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s])
    return q / np.linalg.norm(q)


def camera_pose(t, client=0, n_clients=1):
    """(R_G_C 3x3 float64, origin float64[3], T_G_C float32[7] = qw,qx,qy,qz,tx,ty,tz) of frame t."""
    yaw = 2.0 * math.pi * client / n_clients + math.radians(0.36) * t
    c, s = math.cos(yaw), math.sin(yaw)
    Rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    R_opt = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    R = Rz @ R_opt
    origin = np.array([c * 1.0, s * 1.0, 1.5])
    q = quat_from_matrix(R)
    T = np.concatenate([q, origin]).astype(np.float32)
    return R, origin, T


def render_depth(R, origin, w=640, h=480):
    """Analytic z-depth image (float32, metres) of the room+sphere from pose (R, origin)."""
    fx, fy, cx, cy = INTRINSICS[(w, h)]
    u = np.arange(w, dtype=np.float64)
    v = np.arange(h, dtype=np.float64)
    uu, vv = np.meshgrid(u, v)
    dirs_c = np.stack([(uu - cx) / fx, (vv - cy) / fy, np.ones_like(uu)], axis=-1)  # z = 1 -> parameter = z-depth
    dirs = dirs_c @ R.T
    tmin = np.full((h, w), np.inf)
    for ax in range(3):
        d = dirs[..., ax]
        with np.errstate(divide="ignore", invalid="ignore"):
            for bound in (ROOM_MIN[ax], ROOM_MAX[ax]):
                tt = (bound - origin[ax]) / d
                tt = np.where((tt > 1e-9) & np.isfinite(tt), tt, np.inf)
                tmin = np.minimum(tmin, tt)
    oc = origin - SPHERE_C
    a = np.sum(dirs * dirs, axis=-1)
    b = 2.0 * np.sum(dirs * oc, axis=-1)
    cc = float(oc @ oc) - SPHERE_R ** 2
    disc = b * b - 4 * a * cc
    with np.errstate(invalid="ignore"):
        ts = (-b - np.sqrt(np.where(disc >= 0, disc, np.nan))) / (2 * a)
    ts = np.where((disc >= 0) & (ts > 1e-9), ts, np.inf)
    return np.minimum(tmin, ts).astype(np.float32)


def depth_to_points(depth, w=640, h=480):
    """p_C = d * ((u-cx)/fx, (v-cy)/fy, 1) in float32, row-major; matching colours.

    The arithmetic (float32: (u - cx) / fx, then * d) is the same as the engine's depth front end
    so that cox_integrate_depth_dev and cox_integrate_points see identical points.
    """
    fx, fy, cx, cy = (np.float32(x) for x in INTRINSICS[(w, h)])
    u = np.arange(w, dtype=np.float32)
    v = np.arange(h, dtype=np.float32)
    xn = ((u - cx) / fx).astype(np.float32)
    yn = ((v - cy) / fy).astype(np.float32)
    d = depth.astype(np.float32)
    pts = np.empty((h, w, 3), np.float32)
    pts[..., 0] = d * xn[None, :]
    pts[..., 1] = d * yn[:, None]
    pts[..., 2] = d
    return pts.reshape(-1, 3)


def frame_colors(w=640, h=480):
    uu, vv = np.meshgrid(np.arange(w), np.arange(h))
    rgba = np.stack([uu & 255, vv & 255, np.full_like(uu, 128), np.full_like(uu, 255)], axis=-1).astype(np.uint8)
    return rgba.reshape(-1, 4)


def make_frame(t, client=0, n_clients=1, w=640, h=480, noise=False, nan_fraction=0.0):
    """-> (T_G_C float32[7], points_C float32[N,3], rgba uint8[N,4], depth float32[h,w]).

    Non-finite depths (optional 2 % NaN mask, seed 2000+client) are dropped from the point list,
    as voxblox_ros convertPointcloud does; the depth image keeps them as NaN.
    """
    R, origin, T = camera_pose(t, client, n_clients)
    depth = render_depth(R, origin, w, h)
    if noise:
        rng = np.random.default_rng(1000 + client + 7919 * t)
        depth = (depth + rng.normal(0.0, 1.0, depth.shape).astype(np.float32) * np.float32(0.001) * depth * depth).astype(np.float32)
    if nan_fraction > 0:
        rng = np.random.default_rng(2000 + client + 7919 * t)
        depth = depth.copy()
        depth[rng.random(depth.shape) < nan_fraction] = np.nan
    pts = depth_to_points(depth, w, h)
    rgba = frame_colors(w, h)
    keep = np.isfinite(depth.reshape(-1)) & (depth.reshape(-1) > 0)
    return T, np.ascontiguousarray(pts[keep]), np.ascontiguousarray(rgba[keep]), depth


def make_wall_mesh(seed=0, n_frames=12, voxel_size=0.05, quads=4, stamp0=(1600000000, 950000000), frame_step=1):
    """A voxblox_msgs/Mesh-with-history of the room's x = +3.0 m wall region, as recover mode receives it.

    -> dict(block_edge_length, blocks=[...], trajectory=[(sec, nsec, T_G_C float32[7]), ...]) in the form
    coxgraph_amd.capi.MeshMsg takes.  Blocks are 16 voxels wide; every block holds quads x quads quads (two
    triangles each) with a little depth jitter and random colours; every triangle carries a run-length observation
    history over frame ids 0..n_frames-1 (one or two inclusive runs); one block arrives without history (the
    reference skips such blocks) and one triangle has an empty history.  Poses are the benchmark trajectory's
    (client 0), stamped 0.05 s * frame_step apart across a second boundary.
    """
    rng = np.random.default_rng(seed)
    edge = 16 * voxel_size
    ix = int(math.floor(3.0 / edge))
    blocks = []
    iy_range = range(int(math.floor(-1.6 / edge)), int(math.floor(1.6 / edge)) + 1)
    iz_range = range(0, int(math.floor(2.4 / edge)) + 1)
    for bi, (iy, iz) in enumerate((a, b) for a in iy_range for b in iz_range):
        # vertex grid in block-local fixed point: value * 2 / 65535 = fraction of the block edge
        g = np.linspace(0.0, 1.0, quads + 1)
        depth = (3.0 - ix * edge) / edge + rng.uniform(-0.02, 0.02, (quads + 1, quads + 1))
        to_u16 = lambda f: np.clip(np.round(f * 65535.0 / 2.0), 0, 65535).astype(np.uint16)
        vx, vy, vz = to_u16(depth), to_u16(np.broadcast_to(g[:, None], depth.shape)), to_u16(np.broadcast_to(g[None, :], depth.shape))
        col = rng.integers(0, 256, (quads + 1, quads + 1, 3)).astype(np.uint8)
        x, y, z, r, gg, b, hist = [], [], [], [], [], [], []
        for a in range(quads):
            for c in range(quads):
                for tri in (((a, c), (a + 1, c), (a, c + 1)), ((a + 1, c), (a + 1, c + 1), (a, c + 1))):
                    for (p, q) in tri:
                        x.append(vx[p, q]), y.append(vy[p, q]), z.append(vz[p, q])
                        r.append(col[p, q, 0]), gg.append(col[p, q, 1]), b.append(col[p, q, 2])
                    f0 = int(rng.integers(0, n_frames))
                    f1 = int(rng.integers(f0, n_frames))
                    runs = [f0, f1]
                    if f1 + 2 < n_frames and rng.random() < 0.3:
                        f2 = int(rng.integers(f1 + 2, n_frames))
                        runs += [f2, int(rng.integers(f2, n_frames))]
                    hist.append(runs)
        if bi == 2:
            hist[5] = []     # a triangle nobody observed
        blocks.append(dict(index=(ix, iy, iz), x=x, y=y, z=z, r=r, g=gg, b=b, history=None if bi == 1 else hist))
    sec0, nsec0 = stamp0
    traj = []
    for k in range(n_frames):
        ns = nsec0 + 50000000 * frame_step * k
        traj.append((sec0 + ns // 1000000000, ns % 1000000000, camera_pose(10 * k)[2]))
    return dict(block_edge_length=np.float32(edge), blocks=blocks, trajectory=traj)
