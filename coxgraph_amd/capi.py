"""ctypes binding of the C ABI declared in include/coxgraph_hip.h.

The binding is generic over (shared library, symbol prefix): the product engine is
``coxgraph_amd/lib/libcoxgraph_hip.so`` with prefix ``cox_``; the test suite loads its CPU checker
(built from ``oracle/``, its own symbol prefix) through the same classes so parity tests read the
same on both sides.  Nothing in this package loads that checker.
"""
import ctypes as C
import os
import numpy as np

COX_OK = 0
STATUS = {
    0: "COX_OK", -1: "COX_ERR_INVALID_ARG", -2: "COX_ERR_NO_DEVICE", -3: "COX_ERR_OUT_OF_MEMORY",
    -4: "COX_ERR_POOL_EXHAUSTED", -5: "COX_ERR_INDEX_RANGE", -6: "COX_ERR_UNSUPPORTED",
    -7: "COX_ERR_BUFFER_TOO_SMALL", -8: "COX_ERR_INTERNAL",
}
METHODS = {"simple": 0, "merged": 1, "fast": 2, "projective": 3}
VOXELS_PER_BLOCK = 4096


class CoxError(RuntimeError):
    def __init__(self, status, what):
        super().__init__(f"{what}: {STATUS.get(status, status)}")
        self.status = status


class TsdfConfig(C.Structure):
    """cox_tsdf_config (voxblox TsdfIntegratorBase::Config)."""
    _fields_ = [
        ("default_truncation_distance", C.c_float), ("max_weight", C.c_float),
        ("voxel_carving_enabled", C.c_int32), ("min_ray_length_m", C.c_float),
        ("max_ray_length_m", C.c_float), ("use_const_weight", C.c_int32), ("allow_clear", C.c_int32),
        ("use_weight_dropoff", C.c_int32), ("use_sparsity_compensation_factor", C.c_int32),
        ("sparsity_compensation_factor", C.c_float), ("integrator_threads", C.c_int32),
        ("integration_order_mode", C.c_int32), ("enable_anti_grazing", C.c_int32),
        ("start_voxel_subsampling_factor", C.c_float), ("max_consecutive_ray_collisions", C.c_int32),
        ("clear_checks_every_n_frames", C.c_int32), ("max_integration_time_s", C.c_float),
        ("merged_bundle_order", C.c_int32), ("fast_exact_sets", C.c_int32),
        ("sensor_horizontal_resolution", C.c_int32), ("sensor_vertical_resolution", C.c_int32),
        ("sensor_vertical_field_of_view_degrees", C.c_float), ("projective_interpolation_scheme", C.c_int32),
        ("projective_adaptive_gap_m", C.c_float),
    ]


class FrameStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_points", "n_valid", "n_rays", "n_updates", "n_touched_voxels",
                                          "n_touched_blocks", "n_new_blocks", "max_bundle_points", "max_voxel_updates")]

    def asdict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class EsdfConfig(C.Structure):
    """cox_esdf_config (voxblox EsdfIntegrator::Config)."""
    _fields_ = [("max_distance_m", C.c_float), ("min_distance_m", C.c_float), ("default_distance_m", C.c_float), ("min_weight", C.c_float)]


class RegConfig(C.Structure):
    _fields_ = [("no_correspondence_cost", C.c_double)]


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One loaded implementation of the C ABI."""

    def __init__(self, lib_path, prefix):
        if not os.path.exists(lib_path):
            raise FileNotFoundError(lib_path)
        self.lib = C.CDLL(lib_path)
        self.prefix = prefix
        self.path = lib_path

    def fn(self, name, restype=C.c_int):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        return f

    def has(self, name):
        return hasattr(self.lib, self.prefix + name)

    def check(self, status, what):
        if status != COX_OK:
            raise CoxError(status, what)

    def default_config(self, **overrides):
        cfg = TsdfConfig()
        self.fn("tsdf_config_default", None)(C.byref(cfg))
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)
        return cfg

    def device_count(self):
        return int(self.fn("device_count")()) if self.has("device_count") else 0


class Layer:
    """voxblox::Layer<TsdfVoxel> (cox_layer_t)."""

    def __init__(self, eng, voxel_size, voxels_per_side=16, device=0, capacity_blocks=0):
        self.eng = eng
        self.voxel_size = float(voxel_size)
        self.h = C.c_void_p()
        eng.check(eng.fn("layer_create")(C.c_float(voxel_size), C.c_int(voxels_per_side), C.c_int(device),
                                         C.c_uint64(capacity_blocks), C.byref(self.h)), "layer_create")

    def close(self):
        if self.h:
            self.eng.fn("layer_destroy", None)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):  # removeAllBlocks
        self.eng.check(self.eng.fn("layer_clear")(self.h), "layer_clear")

    def stats(self):
        n, b = C.c_uint64(), C.c_uint64()
        self.eng.check(self.eng.fn("layer_stats")(self.h, C.byref(n), C.byref(b)), "layer_stats")
        return int(n.value), int(b.value)

    def download(self):
        """serializeLayerAsMsg: (block_idx int32[n,3], words uint32[n,4096,3]) sorted by (z,y,x)."""
        n = C.c_uint64()
        f = self.eng.fn("layer_download")
        self.eng.check(f(self.h, None, None, C.c_uint64(0), C.byref(n)), "layer_download(query)")
        nb = int(n.value)
        idx = np.zeros((nb, 3), np.int32)
        vox = np.zeros((nb, VOXELS_PER_BLOCK, 3), np.uint32)
        if nb:
            self.eng.check(f(self.h, _fp(idx), _fp(vox), C.c_uint64(nb), C.byref(n)), "layer_download")
        return idx, vox

    def merge_from(self, other, T_B_A=None):
        """mergeLayerAintoLayerB(other, [T_B_A,] self)."""
        T = None if T_B_A is None else np.ascontiguousarray(T_B_A, np.float32)
        self.eng.check(self.eng.fn("layer_merge")(other.h, _fp(T) if T is not None else None, self.h), "layer_merge")

    def registration_points(self, min_voxel_weight=1.0, max_voxel_distance=0.3):
        """finishSubmap()'s relevant voxels: float32 [n,5] = x, y, z, distance, weight."""
        n = C.c_uint64()
        f = self.eng.fn("layer_registration_points")
        self.eng.check(f(self.h, C.c_float(min_voxel_weight), C.c_float(max_voxel_distance), None, C.c_uint64(0), C.byref(n)), "layer_registration_points")
        out = np.zeros((int(n.value), 5), np.float32)
        if n.value:
            self.eng.check(f(self.h, C.c_float(min_voxel_weight), C.c_float(max_voxel_distance), _fp(out), C.c_uint64(n.value), C.byref(n)),
                           "layer_registration_points")
        return out

    def surface_obb(self):
        """getSubmapFrameSurfaceObb -> (min[3], max[3], n_surface_voxels)."""
        mn, mx, n = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_uint64()
        self.eng.check(self.eng.fn("layer_surface_obb")(self.h, _fp(mn), _fp(mx), C.byref(n)), "layer_surface_obb")
        return mn, mx, int(n.value)

    def esdf(self, max_distance_m=None, min_distance_m=None, default_distance_m=None, min_weight=None):
        """generateEsdf(): a new layer (TSDF wire layout: distance = ESDF distance, weight = observed, colour word = fixed)."""
        cfg = EsdfConfig()
        self.eng.fn("esdf_config_default", None)(C.byref(cfg))
        for k, v in (("max_distance_m", max_distance_m), ("min_distance_m", min_distance_m), ("default_distance_m", default_distance_m), ("min_weight", min_weight)):
            if v is not None:
                setattr(cfg, k, v)
        if max_distance_m is not None and default_distance_m is None:
            cfg.default_distance_m = max_distance_m
        out = Layer.__new__(Layer)
        out.eng, out.voxel_size, out.h = self.eng, self.voxel_size, C.c_void_p()
        self.eng.check(self.eng.fn("esdf_from_tsdf")(self.h, C.byref(cfg), C.byref(out.h)), "esdf_from_tsdf")
        return out

    def reserve(self, capacity_blocks):
        self.eng.check(self.eng.fn("layer_reserve")(self.h, C.c_uint64(capacity_blocks)), "layer_reserve")

    def capacity(self):
        n = C.c_uint64()
        self.eng.check(self.eng.fn("layer_capacity")(self.h, C.byref(n)), "layer_capacity")
        return int(n.value)

    def set_auto_grow(self, on):
        self.eng.check(self.eng.fn("layer_set_auto_grow")(self.h, C.c_int(int(on))), "layer_set_auto_grow")

    def clone_to_device(self, device, capacity_blocks=0):
        """Submap hand-over inside one process: block array + keys copied GPU to GPU, hash rebuilt on `device`."""
        out = Layer.__new__(Layer)
        out.eng, out.voxel_size, out.h = self.eng, self.voxel_size, C.c_void_p()
        self.eng.check(self.eng.fn("layer_clone_to_device")(self.h, C.c_int(device), C.c_uint64(capacity_blocks), C.byref(out.h)), "layer_clone_to_device")
        return out

    def n_blocks(self):
        return self.stats()[0]

    def export_dev(self, idx_ptr, vox_ptr, cap_blocks):
        """serializeLayerAsMsg into device buffers (pointers as ints); returns the block count."""
        n = C.c_uint64()
        self.eng.check(self.eng.fn("layer_export_dev")(self.h, C.c_void_p(idx_ptr), C.c_void_p(vox_ptr), C.c_uint64(cap_blocks), C.byref(n)), "layer_export_dev")
        return int(n.value)

    def upload_dev(self, idx_ptr, vox_ptr, n_blocks, action=0):
        """deserializeMsgToLayer from device buffers (pointers as ints)."""
        self.eng.check(self.eng.fn("layer_upload_dev")(self.h, C.c_void_p(idx_ptr), C.c_void_p(vox_ptr), C.c_uint64(n_blocks), C.c_int(action)), "layer_upload_dev")

    def upload(self, idx, vox, action=0):
        idx = np.ascontiguousarray(idx, np.int32)
        vox = np.ascontiguousarray(vox, np.uint32)
        assert vox.shape == (idx.shape[0], VOXELS_PER_BLOCK, 3)
        self.eng.check(self.eng.fn("layer_upload")(self.h, _fp(idx), _fp(vox), C.c_uint64(idx.shape[0]), C.c_int(action)),
                       "layer_upload")


class Integrator:
    """voxblox::TsdfIntegratorBase (cox_integrator_t)."""

    def __init__(self, eng, layer, cfg, method):
        self.eng, self.layer = eng, layer
        self.h = C.c_void_p()
        m = METHODS[method] if isinstance(method, str) else int(method)
        eng.check(eng.fn("integrator_create")(layer.h, C.byref(cfg), C.c_int(m), C.byref(self.h)), "integrator_create")

    def close(self):
        if self.h:
            self.eng.fn("integrator_destroy", None)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def integrate_points(self, T_G_C, xyz, rgba=None, freespace=False):
        """integratePointCloud(T_G_C, points_C, colors, freespace_points) with host buffers."""
        T = np.ascontiguousarray(T_G_C, np.float32)
        xyz = np.ascontiguousarray(xyz, np.float32)
        assert T.shape == (7,) and xyz.ndim == 2 and xyz.shape[1] == 3
        n = xyz.shape[0]
        if rgba is not None:
            rgba = np.ascontiguousarray(rgba, np.uint8)
            assert rgba.shape == (n, 4)
        self.eng.check(self.eng.fn("integrate_points")(self.h, _fp(T), _fp(xyz), _fp(rgba) if rgba is not None else None,
                                                       C.c_uint64(n), C.c_int(int(freespace))), "integrate_points")

    def deintegrate_points(self, T_G_C, xyz):
        """integratePointCloud(..., deintegrate = true): projective integrator only."""
        T = np.ascontiguousarray(T_G_C, np.float32)
        xyz = np.ascontiguousarray(xyz, np.float32)
        self.eng.check(self.eng.fn("integrate_points_ex")(self.h, _fp(T), _fp(xyz), None, C.c_uint64(xyz.shape[0]), C.c_int(0), C.c_int(1)), "integrate_points_ex")

    def integrate_points_dev(self, T_G_C, xyz_ptr, rgba_ptr, n, freespace=False):
        T = np.ascontiguousarray(T_G_C, np.float32)
        self.eng.check(self.eng.fn("integrate_points_dev")(self.h, _fp(T), C.c_void_p(xyz_ptr), C.c_void_p(rgba_ptr or 0),
                                                           C.c_uint64(n), C.c_int(int(freespace))), "integrate_points_dev")

    def integrate_points_async(self, T_G_C, xyz_ptr, rgba_ptr, n, freespace=False):
        """Host buffers (addresses), frame not waited for; pinned buffers must stay untouched until wait_inputs() / sync()."""
        T = np.ascontiguousarray(T_G_C, np.float32)
        self.eng.check(self.eng.fn("integrate_points_async")(self.h, _fp(T), C.c_void_p(xyz_ptr), C.c_void_p(rgba_ptr or 0),
                                                             C.c_uint64(n), C.c_int(int(freespace))), "integrate_points_async")

    def wait_inputs(self):
        self.eng.check(self.eng.fn("integrator_wait_inputs")(self.h), "integrator_wait_inputs")

    def integrate_depth_dev(self, T_G_C, depth_ptr, rgba_ptr, w, h, K):
        T = np.ascontiguousarray(T_G_C, np.float32)
        K = np.ascontiguousarray(K, np.float32)
        self.eng.check(self.eng.fn("integrate_depth_dev")(self.h, _fp(T), C.c_void_p(depth_ptr), C.c_void_p(rgba_ptr or 0),
                                                          C.c_int(w), C.c_int(h), _fp(K)), "integrate_depth_dev")

    def integrate_depth_async(self, T_G_C, depth_ptr, rgba_ptr, w, h, K):
        """Host depth / colour images (addresses), frame not waited for."""
        T = np.ascontiguousarray(T_G_C, np.float32)
        K = np.ascontiguousarray(K, np.float32)
        self.eng.check(self.eng.fn("integrate_depth_async")(self.h, _fp(T), C.c_void_p(depth_ptr), C.c_void_p(rgba_ptr or 0),
                                                            C.c_int(w), C.c_int(h), _fp(K)), "integrate_depth_async")

    def set_input_stream(self, stream_ptr, enable=True):
        """Order *_dev calls against the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream)."""
        self.eng.check(self.eng.fn("integrator_set_input_stream")(self.h, C.c_void_p(stream_ptr or 0), C.c_int(int(enable))), "integrator_set_input_stream")

    def sync(self):
        self.eng.check(self.eng.fn("integrator_sync")(self.h), "integrator_sync")

    def last_stats(self):
        s = FrameStats()
        self.eng.check(self.eng.fn("integrator_last_stats")(self.h, C.byref(s)), "integrator_last_stats")
        return s.asdict()

    def set_profiling(self, on=True):
        """0 / False = off, n >= 1 = HIP-event timing of the merge and apply kernels of every n-th frame."""
        self.eng.check(self.eng.fn("integrator_set_profiling")(self.h, C.c_int(int(on))), "integrator_set_profiling")

    def stage_times(self, reset=False):
        """{'merge': (ms, launches), 'apply': (ms, launches)} measured with HIP events on the kernels' own streams."""
        ms, n = (C.c_double * 2)(), (C.c_uint64 * 2)()
        self.eng.check(self.eng.fn("integrator_stage_times")(self.h, ms, n, C.c_int(int(reset))), "integrator_stage_times")
        return {"merge": (float(ms[0]), int(n[0])), "apply": (float(ms[1]), int(n[1]))}

    KERNEL_CLASSES = ("merge", "apply", "bundle_hash", "point_sort", "touch_emit", "record_sort", "fast_start", "fast_visits", "fast_sweeps", "fast_round1")

    def class_times(self, reset=False):
        """{class: (ms, regions)} for every kernel class of a frame (cox_kernel_class), HIP events on the kernels' own streams."""
        k = len(self.KERNEL_CLASSES)
        ms, n = (C.c_double * k)(), (C.c_uint64 * k)()
        self.eng.check(self.eng.fn("integrator_class_times")(self.h, ms, n, C.c_int(int(reset))), "integrator_class_times")
        return {name: (float(ms[i]), int(n[i])) for i, name in enumerate(self.KERNEL_CLASSES)}

    def fast_stats(self):
        """method 'fast': run totals of the observed-set relaxation (cox_integrator_fast_stats)."""
        out = (C.c_uint64 * 10)()
        self.eng.check(self.eng.fn("integrator_fast_stats")(self.h, out), "integrator_fast_stats")
        return dict(sequential_frames=int(out[0]), round1_frames=int(out[1]), passes_round0=int(out[2]), passes_later_rounds=int(out[3]),
                    frames=int(out[4]), sequential_frames_list_outgrown=int(out[5]), sequential_frames_barrier_gave_up=int(out[6]), frames_with_three_rounds_or_more=int(out[7]), relax_work_ns=int(out[8]), relax_barrier_wait_ns=int(out[9]))

    def update_stats(self):
        """last frame: tiles whose classification was split over the chip, and their chunks (cox_integrator_update_stats)."""
        out = (C.c_uint64 * 2)()
        self.eng.check(self.eng.fn("integrator_update_stats")(self.h, out), "integrator_update_stats")
        return dict(split_tiles=int(out[0]), chunks=int(out[1]))

    def host_time(self, reset=False):
        """(ms, frames): host time spent inside the integrate calls (enqueueing; cox_integrator_host_time)."""
        ms, n = C.c_double(), C.c_uint64()
        self.eng.check(self.eng.fn("integrator_host_time")(self.h, C.byref(ms), C.byref(n), C.c_int(int(reset))), "integrator_host_time")
        return float(ms.value), int(n.value)

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        self.eng.check(self.eng.fn("integrator_kernel_time")(self.h, C.byref(ms), C.byref(n), C.c_int(int(reset))),
                       "integrator_kernel_time")
        return float(ms.value), int(n.value)


class RegPoints:
    def __init__(self, eng, pts, device=0):
        self.eng = eng
        pts = np.ascontiguousarray(pts, np.float32)
        assert pts.ndim == 2 and pts.shape[1] == 5
        self.n = pts.shape[0]
        self.h = C.c_void_p()
        eng.check(eng.fn("regpoints_create")(C.c_int(device), _fp(pts), C.c_uint64(self.n), C.byref(self.h)), "regpoints_create")

    @classmethod
    def from_layer(cls, eng, layer, min_voxel_weight=1.0, max_voxel_distance=0.3):
        """finishSubmap()'s relevant-voxel set, built and kept on the engine's side (no host round trip)."""
        self = cls.__new__(cls)
        self.eng = eng
        self.h = C.c_void_p()
        eng.check(eng.fn("regpoints_from_layer")(layer.h, C.c_float(min_voxel_weight), C.c_float(max_voxel_distance), C.byref(self.h)),
                  "regpoints_from_layer")
        n = C.c_uint64()
        eng.check(eng.fn("regpoints_size")(self.h, C.byref(n)), "regpoints_size")
        self.n = int(n.value)
        return self

    @classmethod
    def _wrap(cls, eng, h):
        self = cls.__new__(cls)
        self.eng, self.h = eng, h
        n = C.c_uint64()
        eng.check(eng.fn("regpoints_size")(self.h, C.byref(n)), "regpoints_size")
        self.n = int(n.value)
        return self

    @classmethod
    def from_device(cls, eng, ptr, n, device=0):
        """A set whose n*5 floats already sit in HBM (pointer as int)."""
        h = C.c_void_p()
        eng.check(eng.fn("regpoints_create_dev")(C.c_int(device), C.c_void_p(ptr), C.c_uint64(n), C.byref(h)), "regpoints_create_dev")
        return cls._wrap(eng, h)

    @classmethod
    def from_isosurface(cls, eng, layer, min_weight=1.0, vertex_proximity_threshold=None):
        """finishSubmap()'s isosurface vertices (the "explicit" set), built and kept on the engine's side."""
        thr = 0.5 * layer.voxel_size if vertex_proximity_threshold is None else vertex_proximity_threshold
        h, nm, nc = C.c_void_p(), C.c_uint64(), C.c_uint64()
        eng.check(eng.fn("regpoints_from_isosurface")(layer.h, C.c_float(min_weight), C.c_float(thr), C.byref(h), C.byref(nm), C.byref(nc)), "regpoints_from_isosurface")
        self = cls._wrap(eng, h)
        self.n_mesh_vertices, self.n_connected_vertices = int(nm.value), int(nc.value)
        return self

    def clone_to_device(self, device):
        h = C.c_void_p()
        self.eng.check(self.eng.fn("regpoints_clone_to_device")(self.h, C.c_int(device), C.byref(h)), "regpoints_clone_to_device")
        return RegPoints._wrap(self.eng, h)

    def data_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self.eng.check(self.eng.fn("regpoints_data_dev")(self.h, C.byref(p), C.byref(n)), "regpoints_data_dev")
        return int(p.value or 0), int(n.value)

    def download(self):
        out = np.zeros((self.n, 5), np.float32)
        n = C.c_uint64()
        self.eng.check(self.eng.fn("regpoints_download")(self.h, _fp(out), C.c_uint64(self.n), C.byref(n)), "regpoints_download")
        return out

    def close(self):
        if self.h:
            self.eng.fn("regpoints_destroy", None)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Registration:
    """voxgraph::RegistrationCostFunction for one (reference points, reading layer) pair."""

    def __init__(self, eng, ref_points, reading_layer, no_correspondence_cost=0.0):
        self.eng, self.ref, self.reading = eng, ref_points, reading_layer
        cfg = RegConfig(no_correspondence_cost)
        self.h = C.c_void_p()
        eng.check(eng.fn("reg_create")(ref_points.h, reading_layer.h, C.byref(cfg), C.byref(self.h)), "reg_create")

    def close(self):
        if self.h:
            self.eng.fn("reg_destroy", None)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _args(self, pose_ref, pose_read, sample_idx):
        pr = np.ascontiguousarray(pose_ref, np.float64)
        pd = np.ascontiguousarray(pose_read, np.float64)
        assert pr.shape == (4,) and pd.shape == (4,)
        if sample_idx is None:
            return pr, pd, None, (getattr(self, "stored_n", None) or self.ref.n)
        si = np.ascontiguousarray(sample_idx, np.uint32)
        return pr, pd, si, si.shape[0]

    def evaluate(self, pose_ref, pose_read, sample_idx=None, jacobians=True):
        """CostFunction::Evaluate -> (residuals[n], J_ref[n,4], J_read[n,4])."""
        pr, pd, si, n = self._args(pose_ref, pose_read, sample_idx)
        r = np.zeros(n, np.float64)
        jf = np.zeros((n, 4), np.float64) if jacobians else None
        jr = np.zeros((n, 4), np.float64) if jacobians else None
        self.eng.check(self.eng.fn("reg_evaluate")(self.h, _fp(pr), _fp(pd), _fp(si) if si is not None else None, C.c_uint64(n),
                                                   _fp(r), _fp(jf) if jacobians else None, _fp(jr) if jacobians else None),
                       "reg_evaluate")
        return r, jf, jr

    def normal_eq(self, pose_ref, pose_read, sample_idx=None):
        pr, pd, si, n = self._args(pose_ref, pose_read, sample_idx)
        H = np.zeros((8, 8), np.float64)
        b = np.zeros(8, np.float64)
        cost, nc = C.c_double(), C.c_uint64()
        self.eng.check(self.eng.fn("reg_normal_eq")(self.h, _fp(pr), _fp(pd), _fp(si) if si is not None else None, C.c_uint64(n),
                                                    _fp(H), _fp(b), C.byref(cost), C.byref(nc)), "reg_normal_eq")
        return H, b, float(cost.value), int(nc.value)

    def set_samples(self, sample_idx):
        """Keep the sample indices on the engine's side; later calls pass sample_idx=None and use them."""
        if sample_idx is None:
            self.eng.check(self.eng.fn("reg_set_samples")(self.h, None, C.c_uint64(0)), "reg_set_samples")
            self.stored_n = None
            return
        si = np.ascontiguousarray(sample_idx, np.uint32)
        self.eng.check(self.eng.fn("reg_set_samples")(self.h, _fp(si), C.c_uint64(si.shape[0])), "reg_set_samples")
        self.stored_n = int(si.shape[0])

    def draw_samples(self, n_res, seed):
        """The weighted sampler's draws for one evaluation, made on the engine's side; they become the stored set."""
        self.eng.check(self.eng.fn("reg_draw_samples")(self.h, C.c_uint64(n_res), C.c_uint64(seed)), "reg_draw_samples")
        self.stored_n = int(n_res)

    def get_samples(self):
        n = C.c_uint64()
        f = self.eng.fn("reg_get_samples")
        self.eng.check(f(self.h, None, C.c_uint64(0), C.byref(n)), "reg_get_samples")
        out = np.zeros(int(n.value), np.uint32)
        if n.value:
            self.eng.check(f(self.h, _fp(out), C.c_uint64(n.value), C.byref(n)), "reg_get_samples")
        return out

    def normal_eq_begin(self, pose_ref, pose_read, sample_idx=None):
        pr, pd, si, n = self._args(pose_ref, pose_read, sample_idx)
        self.eng.check(self.eng.fn("reg_normal_eq_begin")(self.h, _fp(pr), _fp(pd), _fp(si) if si is not None else None, C.c_uint64(n)), "reg_normal_eq_begin")

    def normal_eq_finish(self):
        H = np.zeros((8, 8), np.float64)
        b = np.zeros(8, np.float64)
        cost, nc = C.c_double(), C.c_uint64()
        self.eng.check(self.eng.fn("reg_normal_eq_finish")(self.h, _fp(H), _fp(b), C.byref(cost), C.byref(nc)), "reg_normal_eq_finish")
        return H, b, float(cost.value), int(nc.value)

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        self.eng.check(self.eng.fn("reg_kernel_time")(self.h, C.byref(ms), C.byref(n), C.c_int(int(reset))), "reg_kernel_time")
        return float(ms.value), int(n.value)

    @staticmethod
    def normal_eq_batch(regs, poses_ref, poses_read):
        """cox_reg_normal_eq_batch: all the constraints of one pose-graph evaluation in ONE launch (every handle with its stored
        samples, or all its points).  -> [(H 8x8, b 8, cost, n_corr)] in the order of `regs`."""
        n = len(regs)
        eng = regs[0].eng
        handles = (C.c_void_p * n)(*[r.h for r in regs])
        pr = np.ascontiguousarray(poses_ref, np.float64).reshape(n, 4)
        pd = np.ascontiguousarray(poses_read, np.float64).reshape(n, 4)
        H = np.zeros((n, 8, 8), np.float64)
        b = np.zeros((n, 8), np.float64)
        cost = np.zeros(n, np.float64)
        nc = np.zeros(n, np.uint64)
        eng.check(eng.fn("reg_normal_eq_batch")(handles, C.c_uint64(n), _fp(pr), _fp(pd), _fp(H), _fp(b), _fp(cost), _fp(nc)), "reg_normal_eq_batch")
        return [(H[c], b[c], float(cost[c]), int(nc[c])) for c in range(n)]


class Comm:
    """cox_comm_t: RCCL behind the C ABI (one rank = one process = one GPU) -- the collectives the C++ host uses."""
    ID_BYTES = 128

    @staticmethod
    def unique_id(eng):
        buf = (C.c_uint8 * Comm.ID_BYTES)()
        eng.check(eng.fn("comm_unique_id")(buf), "comm_unique_id")
        return bytes(buf)

    def __init__(self, eng, device, rank, world, unique_id):
        self.eng = eng
        self.h = C.c_void_p()
        buf = (C.c_uint8 * Comm.ID_BYTES).from_buffer_copy(unique_id)
        eng.check(eng.fn("comm_init_rank")(C.c_int(device), C.c_int(rank), C.c_int(world), buf, C.byref(self.h)), "comm_init_rank")
        self.rank, self.world = rank, world

    def close(self):
        if self.h:
            self.eng.fn("comm_destroy", None)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allreduce_f64(self, buf):
        """in-place sum over the ranks of a float64 numpy array (host in, host out)"""
        a = np.ascontiguousarray(buf, np.float64)
        self.eng.check(self.eng.fn("comm_allreduce_f64")(self.h, _fp(a), C.c_uint64(a.size)), "comm_allreduce_f64")
        return a

    def allgather_dev(self, send_ptr, recv_ptr, bytes_per_rank, producer_stream=0):
        self.eng.check(self.eng.fn("comm_allgather_dev_on")(self.h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), C.c_uint64(bytes_per_rank),
                                                            C.c_void_p(producer_stream or 0)), "comm_allgather_dev_on")


class MeshMsgStruct(C.Structure):
    """cox_mesh_msg."""
    _fields_ = [
        ("block_edge_length", C.c_float), ("n_blocks", C.c_uint64), ("block_index", C.c_void_p), ("vertex_begin", C.c_void_p),
        ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("r", C.c_void_p), ("g", C.c_void_p), ("b", C.c_void_p),
        ("block_has_history", C.c_void_p), ("history_begin", C.c_void_p), ("history", C.c_void_p),
        ("n_poses", C.c_uint64), ("stamp_sec", C.c_void_p), ("stamp_nsec", C.c_void_p), ("T_G_C", C.c_void_p),
    ]


class MeshMsg:
    """A voxblox_msgs/Mesh with history + trajectory, flattened into the arrays cox_mesh_msg points to.

    blocks: list of dict(index=(ix,iy,iz), x,y,z=uint16[nv], r,g,b=uint8[nv], history=None | list (one per triangle) of
    run-length lists [first0, last0, first1, last1, ...]); trajectory: list of (sec, nsec, T_G_C[7] float32).
    """

    def __init__(self, block_edge_length, blocks, trajectory):
        self.block_edge_length = float(block_edge_length)
        nb = len(blocks)
        self.block_index = np.array([b["index"] for b in blocks], np.int64).reshape(nb, 3)
        counts = [len(b["x"]) for b in blocks]
        self.vertex_begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(b[k], dt) for b in blocks]) if nb else np.zeros(0, dt), dt)
        self.x, self.y, self.z = cat("x", np.uint16), cat("y", np.uint16), cat("z", np.uint16)
        self.r, self.g, self.b = cat("r", np.uint8), cat("g", np.uint8), cat("b", np.uint8)
        self.block_has_history = np.array([1 if b.get("history") else 0 for b in blocks], np.uint8)
        hb, hist = [0], []
        for b, cnt in zip(blocks, counts):
            runs = b.get("history") or [[] for _ in range(cnt // 3)]
            assert len(runs) == cnt // 3, "one ObsHistory per triangle"
            for r in runs:
                hist.extend(int(v) for v in r)
                hb.append(len(hist))
        self.history_begin = np.array(hb, np.uint64)
        self.history = np.array(hist, np.uint32)
        self.stamp_sec = np.array([p[0] for p in trajectory], np.uint32)
        self.stamp_nsec = np.array([p[1] for p in trajectory], np.uint32)
        self.T_G_C = np.ascontiguousarray(np.array([p[2] for p in trajectory], np.float32).reshape(len(trajectory), 7))
        self.n_blocks, self.n_poses = nb, len(trajectory)

    def struct(self):
        m = MeshMsgStruct()
        m.block_edge_length, m.n_blocks, m.n_poses = self.block_edge_length, self.n_blocks, self.n_poses
        for k in ("block_index", "vertex_begin", "x", "y", "z", "r", "g", "b", "block_has_history", "history_begin", "history", "stamp_sec",
                  "stamp_nsec", "T_G_C"):
            setattr(m, k, getattr(self, k).ctypes.data)
        return m


class MeshConverter:
    """voxblox::MeshConverter (cox_meshconv_t): recover mode's mesh -> per-pose point clouds."""

    def __init__(self, eng, interpolate_voxel_size=0.2, device=0):
        self.eng = eng
        self.h = C.c_void_p()
        eng.check(eng.fn("meshconv_create")(C.c_int(device), C.c_float(interpolate_voxel_size), C.byref(self.h)), "meshconv_create")

    def close(self):
        if self.h:
            self.eng.fn("meshconv_destroy", None)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mesh(self, mesh):
        m = mesh.struct()
        self.eng.check(self.eng.fn("meshconv_set_mesh")(self.h, C.byref(m)), "meshconv_set_mesh")

    def convert(self):
        """convertToPointCloud -> (converted?, recovered xyz float32[n,3], recovered rgb uint8[n,3])."""
        n, ok = C.c_uint64(), C.c_int()
        self.eng.check(self.eng.fn("meshconv_convert")(self.h, C.byref(n), C.byref(ok)), "meshconv_convert")
        return bool(ok.value), *self.recovered()

    def recovered(self):
        n = C.c_uint64()
        f = self.eng.fn("meshconv_recovered")
        self.eng.check(f(self.h, None, None, C.c_uint64(0), C.byref(n)), "meshconv_recovered(query)")
        xyz = np.zeros((int(n.value), 3), np.float32)
        rgb = np.zeros((int(n.value), 3), np.uint8)
        if n.value:
            self.eng.check(f(self.h, _fp(xyz), _fp(rgb), C.c_uint64(n.value), C.byref(n)), "meshconv_recovered")
        return xyz, rgb

    def pose_clouds(self):
        """The sequence getNextPointcloud yields: list of (T_G_C float32[7], points_C float32[n,3], colors uint8[n,4])."""
        out = []
        i = C.c_int32(0)
        T = np.zeros(7, np.float32)
        n, more = C.c_uint64(), C.c_int()
        nxt, dl = self.eng.fn("meshconv_next"), self.eng.fn("meshconv_download")
        while True:
            k = int(i.value)
            px, pc = C.c_void_p(), C.c_void_p()
            self.eng.check(nxt(self.h, C.byref(i), _fp(T), C.byref(px), C.byref(pc), C.byref(n), C.byref(more)), "meshconv_next")
            if not more.value:
                break
            xyz = np.zeros((int(n.value), 3), np.float32)
            rgba = np.zeros((int(n.value), 4), np.uint8)
            if n.value:
                self.eng.check(dl(self.h, C.c_int32(k), _fp(xyz), _fp(rgba), C.c_uint64(n.value), C.byref(n)), "meshconv_download")
            out.append((T.copy(), xyz, rgba))
        return out

    def clear(self):
        self.eng.check(self.eng.fn("meshconv_clear")(self.h), "meshconv_clear")

    def process_mesh(self, integrator, mesh):
        """TsdfRecover::processMesh up to the serialisation: -> (n_recovered_points, n_integratePointCloud_calls)."""
        m = mesh.struct()
        nr, ni = C.c_uint64(), C.c_uint64()
        self.eng.check(self.eng.fn("recover_process_mesh")(self.h, integrator.h, C.byref(m), C.byref(nr), C.byref(ni)), "recover_process_mesh")
        return int(nr.value), int(ni.value)


# ---- wire-format helpers (voxblox_msgs/Block data words) -----------------------------------------
def words_to_fields(vox):
    """uint32[...,3] wire words -> (distance f32, weight f32, rgba u8[...,4])."""
    vox = np.ascontiguousarray(vox, np.uint32)
    d = vox[..., 0].copy().view(np.float32)
    w = vox[..., 1].copy().view(np.float32)
    c = vox[..., 2]
    rgba = np.stack([(c >> 24) & 255, (c >> 16) & 255, (c >> 8) & 255, c & 255], axis=-1).astype(np.uint8)
    return d, w, rgba
