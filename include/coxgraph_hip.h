/*
 * coxgraph_hip.h -- C ABI of the MI355X-native TSDF fusion + submap registration engine.
 *
 * This is the drop-in boundary for the ONE data-parallel hot path of mfkiwl/coxgraph
 * (SURVEY.md section 8).  The reference has no FFI layer of its own: the seam is C++ virtual
 * dispatch into the un-vendored voxblox / voxgraph forks.  Every entry point below names the
 * reference interface (file:line under /root/reference) whose work it replaces; the thin C++
 * adapter classes that give these entry points the reference's own signatures live in
 * coxgraph_amd/host/ (see INTEGRATION.md for the binding a coxgraph maintainer would add).
 *
 * Conventions
 *  - every function returns COX_OK (0) or a negative cox_status; no exception crosses the ABI
 *  - handles are opaque; one handle = one GPU + one HIP stream; handles are NOT thread-safe
 *    (mirror: one ROS callback thread drives an integrator, tsdf_recover.h:71-77), but distinct
 *    handles may be used from distinct threads (mirror: Ceres calls Evaluate on different
 *    residual blocks from up to 4 threads, backend/pose_graph.h:63)
 *  - input buffers are HOST pointers borrowed for the duration of the call unless the name
 *    says _dev (then they are device pointers valid on the handle's GPU, e.g. a torch tensor's
 *    data_ptr()); the engine owns all device memory it allocates
 *  - poses T_G_C are 7 floats: unit quaternion (w,x,y,z) then translation (x,y,z)  [minkindr
 *    QuatTransformationTemplate<float>, used as voxblox::Transformation]
 *  - 4-DoF poses are 4 doubles (x, y, z, yaw)  [voxgraph Pose4D; pose graph is 4-DoF, see
 *    coxgraph/include/coxgraph/server/backend/node_collection.h:22-24]
 */
#ifndef COXGRAPH_HIP_H_
#define COXGRAPH_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cox_status {
  COX_OK = 0,
  COX_ERR_INVALID_ARG = -1,
  COX_ERR_NO_DEVICE = -2,       /* no HIP device / HIP runtime call failed */
  COX_ERR_OUT_OF_MEMORY = -3,   /* device allocation failed */
  COX_ERR_POOL_EXHAUSTED = -4,  /* block pool or hash table full: layer was created too small */
  COX_ERR_INDEX_RANGE = -5,     /* a voxel index left the +-2^20 range the packed keys support */
  COX_ERR_UNSUPPORTED = -6,     /* valid in the reference, not implemented by this engine (yet) */
  COX_ERR_BUFFER_TOO_SMALL = -7,
  COX_ERR_INTERNAL = -8,
  COX_ERR_COMM = -9             /* an RCCL call failed (communicator set-up or a collective); RCCL's own message goes to stderr */
} cox_status;

typedef struct cox_layer cox_layer_t;           /* voxblox::Layer<TsdfVoxel> on one GPU */
typedef struct cox_integrator cox_integrator_t; /* voxblox::TsdfIntegratorBase */
typedef struct cox_regpoints cox_regpoints_t;   /* voxgraph registration point set of a submap */
typedef struct cox_reg cox_reg_t;               /* voxgraph::RegistrationCostFunction */

/* integrator factory key: method in {"simple","merged","fast","projective"}; "fast" reproduces the reference at integrator_threads = 1;
 * "projective" (config/tsdf_server_default.yaml:6-9, tsdf_server_carla.yaml:6-9) is voxblox's range-image gather integrator
 * (coxgraph/config/tsdf_server_euroc.yaml:6, tsdf_server_default.yaml:6,
 *  coxgraph_sim/launch/experiments/mav_3dplanning_2d3dhouse_two.launch:10) */
typedef enum cox_method { COX_METHOD_SIMPLE = 0, COX_METHOD_MERGED = 1, COX_METHOD_FAST = 2, COX_METHOD_PROJECTIVE = 3 } cox_method;

/* voxblox::TsdfIntegratorBase::Config as read by getTsdfIntegratorConfigFromRosParam
 * (used by coxgraph/include/coxgraph/map_comm/tsdf_recover.h:48-50; keys in
 *  coxgraph/config/tsdf_server_euroc.yaml:10-24).  cox_tsdf_config_default() fills the voxblox
 * defaults. */
typedef struct cox_tsdf_config {
  float default_truncation_distance; /* truncation_distance */
  float max_weight;
  int32_t voxel_carving_enabled;
  float min_ray_length_m;
  float max_ray_length_m;
  int32_t use_const_weight;
  int32_t allow_clear;
  int32_t use_weight_dropoff;
  int32_t use_sparsity_compensation_factor;
  float sparsity_compensation_factor;
  int32_t integrator_threads;     /* CPU oracle only; the GPU engine ignores it */
  int32_t integration_order_mode; /* 0 = "mixed" (voxblox default); others unsupported */
  int32_t enable_anti_grazing;
  float start_voxel_subsampling_factor; /* fast */
  int32_t max_consecutive_ray_collisions; /* fast */
  int32_t clear_checks_every_n_frames;    /* fast */
  float max_integration_time_s;           /* fast; a finite budget is refused by the HIP engine (wall-clock dependent) */
  int32_t merged_bundle_order;            /* CPU oracle only: 0 canonical, 1 libstdc++ map order */
  int32_t fast_exact_sets;                /* CPU oracle only: 0 ApproxHashSet, 1 exact sets */
  /* projective only (sensor_horizontal_resolution / sensor_vertical_resolution / sensor_vertical_field_of_view_degrees,
   * config/tsdf_server_default.yaml:7-9); the resolutions must be set, voxblox CHECKs them > 0 */
  int32_t sensor_horizontal_resolution;
  int32_t sensor_vertical_resolution;
  float sensor_vertical_field_of_view_degrees;
  int32_t projective_interpolation_scheme; /* 0 nearest, 1 min neighbour, 2 bilinear, 3 adaptive (what TsdfIntegratorFactory instantiates) */
  float projective_adaptive_gap_m;         /* adaptive: a 2 x 2 range neighbourhood wider than this is a depth discontinuity */
} cox_tsdf_config;

/* per-frame counters of the last cox_integrate_* call (used for the roofline accounting of
 * B_frame = 16*n_valid + 24*n_touched_voxels) */
typedef struct cox_frame_stats {
  uint64_t n_points;         /* points handed in */
  uint64_t n_valid;          /* points that passed isPointValid */
  uint64_t n_rays;           /* rays cast (simple: n_valid, merged: bundles) */
  uint64_t n_updates;        /* (ray, voxel) updates = updateTsdfVoxel calls */
  uint64_t n_touched_voxels; /* distinct voxels updated */
  uint64_t n_touched_blocks; /* distinct blocks visited */
  uint64_t n_new_blocks;     /* blocks allocated by this frame */
  uint64_t max_bundle_points;  /* merged: points of the largest bundle (length of the longest sequential mean); else 0 */
  uint64_t max_voxel_updates;  /* most updates any one voxel received (length of the longest in-order replay) */
} cox_frame_stats;

void cox_tsdf_config_default(cox_tsdf_config* cfg);
const char* cox_status_string(int status);
/* number of HIP devices visible (0 without a GPU); never fails */
int cox_device_count(void);

/* ---- Layer<TsdfVoxel>  ------------------------------------------------------------------- */
/* voxblox::Layer<TsdfVoxel>(voxel_size, voxels_per_side) as owned by TsdfMap
 * (tsdf_recover.h:48 getTsdfMapConfigFromRosParam).  voxels_per_side must be 16.
 * capacity_blocks = size of the device block pool (49,152 B each); 0 picks a default. */
int cox_layer_create(float voxel_size, int voxels_per_side, int device, uint64_t capacity_blocks, cox_layer_t** out);
void cox_layer_destroy(cox_layer_t* layer);
/* Layer::removeAllBlocks()  (tsdf_recover.h:62, src/client/map_server.cpp:65) */
int cox_layer_clear(cox_layer_t* layer);
/* Layer::getNumberOfAllocatedBlocks() / getMemorySize()  (map_server.h:142, tsdf_recover.h:92) */
int cox_layer_stats(cox_layer_t* layer, uint64_t* n_blocks, uint64_t* memory_bytes);
/* serializeLayerAsMsg<TsdfVoxel>(layer, only_updated=false, &msg)  (utils/msg_converter.h:49,
 * map_server.cpp:88, tsdf_recover.h:95): block_idx_xyz gets 3 int32 per block, voxels_3u32 gets
 * 4096*3 uint32 per block in voxblox_msgs/Block wire layout (distance bits, weight bits,
 * a|b<<8|g<<16|r<<24).  Blocks are returned sorted by (z,y,x) so output is deterministic.
 * Pass cap_blocks = 0 and NULL buffers to query n_blocks only. */
int cox_layer_download(cox_layer_t* layer, int32_t* block_idx_xyz, uint32_t* voxels_3u32, uint64_t cap_blocks, uint64_t* n_blocks);
/* deserializeMsgToLayer(msg, layer)  (utils/msg_converter.h:107): action 0 = update (overwrite
 * blocks), 1 = merge (mergeVoxelAIntoVoxelB per voxel), 2 = reset (clear first, then update) */
int cox_layer_upload(cox_layer_t* layer, const int32_t* block_idx_xyz, const uint32_t* voxels_3u32, uint64_t n_blocks, int action);

/* Layer<TsdfVoxel>::allocateBlockPtrByIndex never fails in voxblox: the map grows without bound.  Here the pool is
 * doubled automatically (integrators: once it is half full, before the next frame is enqueued; uploads / merges: before
 * they insert) until hipMalloc refuses; cox_layer_reserve does it explicitly.  Contents are preserved, nothing may be in
 * flight on the layer from OTHER threads.  cox_layer_set_auto_grow(layer, 0) pins the capacity: a frame that runs out then
 * reports COX_ERR_POOL_EXHAUSTED at sync, and so does every later frame that meets a block left without storage. */
int cox_layer_reserve(cox_layer_t* layer, uint64_t capacity_blocks);
int cox_layer_capacity(cox_layer_t* layer, uint64_t* capacity_blocks);
int cox_layer_set_auto_grow(cox_layer_t* layer, int on);
/* deserializeMsgToLayer (utils/msg_converter.h:107) from a message that already sits in HBM on the layer's GPU
 * (submap hand-over between GPUs: the wire arrays arrive by RCCL all-gather / peer copy) */
int cox_layer_upload_dev(cox_layer_t* layer, const int32_t* block_idx_xyz_dev, const uint32_t* voxels_3u32_dev, uint64_t n_blocks, int action);
/* serializeLayerAsMsg (utils/msg_converter.h:49) into DEVICE buffers, same (z,y,x) block order as cox_layer_download:
 * what a rank hands to the all-gather of the submap exchange.  cap_blocks = 0 and NULL buffers query n_blocks. */
int cox_layer_export_dev(cox_layer_t* layer, int32_t* block_idx_xyz_dev, uint32_t* voxels_3u32_dev, uint64_t cap_blocks, uint64_t* n_blocks);
/* Submap hand-over inside one process: what the server obtains with ClientHandler::requestSubmapByTime -> get_client_submap
 * (src/server/client_handler.cpp:82-104, src/server/coxgraph_server.cpp:253-258) as a ROS message becomes a GPU-to-GPU copy
 * of the block array + block keys (hipMemcpyPeer: xGMI between the GPUs of a node) and a rebuild of the hash table on
 * dst_device.  capacity_blocks = 0: as many blocks as the source holds.  Same device: a device-to-device copy. */
int cox_layer_clone_to_device(const cox_layer_t* src, int dst_device, uint64_t capacity_blocks, cox_layer_t** out);

/* mergeLayerAintoLayerB(layer_A, [T_B_A,] layer_B)  (src/client/map_server.cpp:67-69, src/server/submap_collection.cpp:31-33):
 * T_B_A = NULL merges on the same grid (mergeVoxelAIntoVoxelB per voxel); otherwise A is first resampled onto B's grid
 * (transformLayer: trilinear, nearest voxel where that fails, blocks without data dropped).  Both layers on one GPU. */
int cox_layer_merge(const cox_layer_t* layer_A, const float T_B_A[7], cox_layer_t* layer_B);

/* ---- TsdfIntegratorBase -------------------------------------------------------------------- */
/* TsdfIntegratorFactory::create(method, config, layer) */
int cox_integrator_create(cox_layer_t* layer, const cox_tsdf_config* cfg, int method, cox_integrator_t** out);
void cox_integrator_destroy(cox_integrator_t* integ);
/* TsdfIntegratorBase::integratePointCloud(T_G_C, points_C, colors, freespace_points)
 * -- the call at coxgraph/include/coxgraph/map_comm/tsdf_recover.h:75.
 * xyz: n*3 floats (camera frame), rgba: n*4 bytes or NULL. Host pointers. Synchronous. */
int cox_integrate_points(cox_integrator_t* integ, const float T_G_C[7], const float* xyz, const uint8_t* rgba, uint64_t n, int freespace);
/* TsdfIntegratorBase::integratePointCloud(T_G_C, points_C, colors, freespace_points, deintegrate): the projective integrator
 * can take a cloud back out again (voxblox's TsdfServer does so for clouds that leave its queue of
 * pointcloud_deintegration_queue_length, config/tsdf_server_default.yaml:28).  COX_ERR_UNSUPPORTED for the other methods
 * with deintegrate != 0. */
int cox_integrate_points_ex(cox_integrator_t* integ, const float T_G_C[7], const float* xyz, const uint8_t* rgba, uint64_t n, int freespace, int deintegrate);
/* same, inputs already resident on the handle's GPU; asynchronous.
 * ORDERING AND LIFETIME of *_dev inputs.  A frame is enqueued on the engine's own non-blocking streams (ray generation and
 * layer update of consecutive frames overlap), which do not order against any stream of the caller.  Either
 *  (a) the caller synchronises the stream that produced xyz_dev / rgba_dev / depth_dev before the call and keeps the
 *      buffers alive and unmodified until cox_integrator_sync, or
 *  (b) the caller registers its producer stream once with cox_integrator_set_input_stream: every later *_dev call then
 *      makes the engine wait for what that stream has enqueued so far, and makes that stream wait until the engine has read
 *      the inputs -- the call behaves as if the read happened on the caller's stream, so a stream-ordered allocator
 *      (e.g. PyTorch's, for tensors allocated on that stream) may recycle the buffers as soon as the caller drops them.
 * Readers of the layer (cox_reg_*, cox_regpoints_from_layer, downloads, clones) wait for the frames in flight by themselves. */
int cox_integrate_points_dev(cox_integrator_t* integ, const float T_G_C[7], const float* xyz_dev, const uint8_t* rgba_dev, uint64_t n, int freespace);
/* TsdfIntegratorBase::integratePointCloud(T_G_C, points_C, colors, freespace_points) as the reference calls it -- with HOST
 * buffers (coxgraph/include/coxgraph/map_comm/tsdf_recover.h:71-77) -- without waiting for the frame: the buffers are copied to
 * one of six staging sets on the integrator's input stream (the copy of frame t+1 runs beside the kernels of frame t; the call waits
 * on the host only when all six sets are still being read, i.e. when the caller is six frames ahead of the device) and the frame
 * is enqueued behind the copy.  Pageable buffers are free again when the call returns (they go through a pinned bounce
 * buffer: one CPU copy, shared with three helper threads of the integrator); buffers in pinned memory (hipHostMalloc / hipHostRegister, e.g. a torch tensor after pin_memory()) are
 * copied from directly and must stay unmodified until cox_integrator_wait_inputs or cox_integrator_sync has returned.
 * Errors of the frame are reported by the next cox_integrator_sync, as for the *_dev entry points. */
int cox_integrate_points_async(cox_integrator_t* integ, const float T_G_C[7], const float* xyz, const uint8_t* rgba, uint64_t n, int freespace);
/* wait until every host buffer handed to cox_integrate_points_async so far has been copied (not for the frames themselves) */
int cox_integrator_wait_inputs(cox_integrator_t* integ);
/* depth image front end (what depth_image_proc/point_cloud_xyzrgb does ahead of the tsdf_server,
 * coxgraph/launch/cvg/tsdf_client0_cvg.launch:24-30): p_C = d*((u-cx)/fx,(v-cy)/fy,1), row-major
 * point order, non-finite or <=0 depths dropped.  depth_dev: w*h floats in metres on the GPU;
 * rgba_dev: w*h*4 bytes or NULL.  K = {fx, fy, cx, cy}.  Asynchronous: the point count never visits the host (the frame's first
 * kernel reads it from device memory), so frames stay in flight exactly as with cox_integrate_points_dev. */
int cox_integrate_depth_dev(cox_integrator_t* integ, const float T_G_C[7], const float* depth_dev, const uint8_t* rgba_dev, int w, int h,
                            const float K[4]);
/* the same from HOST images (what a depth camera driver hands over), without waiting for the frame: depth (w*h floats, metres) and
 * rgba (w*h*4 bytes or NULL) are copied like the buffers of cox_integrate_points_async (pageable: free again on return; pinned:
 * unmodified until cox_integrator_wait_inputs), converted on the GPU and integrated. */
int cox_integrate_depth_async(cox_integrator_t* integ, const float T_G_C[7], const float* depth, const uint8_t* rgba, int w, int h, const float K[4]);
/* hip_stream: the hipStream_t (as void*) the caller produces *_dev inputs on; NULL = the legacy default stream.
 * enable = 0 returns to contract (a) above. */
int cox_integrator_set_input_stream(cox_integrator_t* integ, void* hip_stream, int enable);
/* wait for the handle's streams; returns any deferred device-side error (pool exhausted, ...) */
int cox_integrator_sync(cox_integrator_t* integ);
int cox_integrator_last_stats(cox_integrator_t* integ, cox_frame_stats* stats);
/* HIP-event timing of the bundle-merge and TSDF-update ("apply") kernels on the streams they run on: 0 = off (default),
 * n >= 1 = time the kernels of every n-th frame */
int cox_integrator_set_profiling(cox_integrator_t* integ, int on);
/* HIP-event time of the dominant kernel over the calls since the last reset (bench.py roofline):
 * accumulated milliseconds and launch count of the TSDF update ("apply") stage */
int cox_integrator_kernel_time(cox_integrator_t* integ, double* apply_ms, uint64_t* apply_launches, int reset);
/* same for two kernels at once: [0] = k_bundle_merge (merged integrator's ray generation, the longest kernel of a
 * frame), [1] = the TSDF update stage (k_apply_eval + k_apply_long) */
int cox_integrator_stage_times(cox_integrator_t* integ, double ms[2], uint64_t launches[2], int reset);

/*
 * Host time the caller's thread spends enqueueing inside the cox_integrate_* calls of this integrator (waits for a free frame slot excluded):
 * at 5 cm the stream is bound by the host's launch rate, not by the GPU, and the second half of every
 * frame is enqueued by a submission thread of the integrator (COX_SUBMIT_THREAD=0 turns it off).  No reference counterpart
 * (measurement only).
 */
int cox_integrator_host_time(cox_integrator_t* integ, double* ms_total, uint64_t* frames, int reset);

/* the same for every kernel class of a frame: ms[k] = accumulated HIP-event time of the regions of class k since the last
 * reset, regions[k] = how many regions that is (one region = the consecutive launches of that class in one frame; the
 * sweeps of the fast integrator are one region per round) */
typedef enum cox_kernel_class {
  COX_KC_MERGE = 0,       /* k_bundle_merge (merged: the bundles' sequential means) */
  COX_KC_APPLY = 1,       /* k_apply_eval + k_apply_long (the TSDF update) */
  COX_KC_BUNDLE_HASH = 2, /* frame hash memset + k_bundle_insert + k_bundle_keys (merged) */
  COX_KC_POINT_SORT = 3,  /* radix sort of the points (merged: bundling) */
  COX_KC_TOUCH_EMIT = 4,  /* record offsets + touch (block allocation) + emit */
  COX_KC_RECORD_SORT = 5, /* radix sort of the (voxel, ray) records */
  COX_KC_FAST_START = 6,  /* fast: start-set sort + flags + ray list */
  COX_KC_FAST_VISITS = 7, /* fast: candidate visits + their sort + inverse */
  COX_KC_FAST_SWEEPS = 8, /* fast: the relaxation launches of round 0 (capped candidate lists) */
  COX_KC_FAST_ROUND1 = 9, /* fast: round 1 (whole walks for the rays that got through their capped lists: lists, sort, relaxation) */
  COX_KERNEL_CLASSES = 10
} cox_kernel_class;
int cox_integrator_class_times(cox_integrator_t* integ, double ms[COX_KERNEL_CLASSES], uint64_t regions[COX_KERNEL_CLASSES], int reset);
/* method "fast" only (COX_ERR_UNSUPPORTED otherwise): run totals of the observed-set relaxation since the integrator was created --
 * out[0] frames the relaxation did not settle and that were redone by the sequential kernel (exact either way), out[1] frames that
 * needed a second round (some ray got through its capped candidate list), out[2] / out[3] passes of the relaxation in round 0 /
 * the later rounds, out[4] frames, out[5] the part of out[0] in which a ray outgrew its list in the last round, out[6] the part in
 * which a grid barrier gave up, out[7] frames that needed a third round or more, out[8] / out[9] nanoseconds workgroup 0 of the
 * relaxation spent in the passes' own work / waiting at their barriers (where the slowest workgroup's work shows).  No reference
 * counterpart (measurement only).  Waits for the frames in flight. */
int cox_integrator_fast_stats(cox_integrator_t* integ, uint64_t out[10]);
/* layer update of the last frame (measurement only, no reference counterpart): out[0] tiles whose classification was split over the
 * chip (more records than two chunks; piece partition = fine voxels only, else 0), out[1] the chunks they were cut into.  Waits for
 * the frames in flight. */
int cox_integrator_update_stats(cox_integrator_t* integ, uint64_t out[2]);

/* self-test: the merged integrator evaluates its sequential mean with an IEEE division whose divisor-only part is
 * hoisted out of the dependent chain; this compares it bit for bit with the compiler's '/' on n pseudo-random operand
 * pairs in the range the kernel accepts and returns the number of differing results (must be 0) */
int cox_selftest_division(int device, uint64_t n, uint64_t seed, uint64_t* mismatches);

/* ---- registration (voxgraph RegistrationCostFunction) ------------------------------------- */
typedef struct cox_reg_config {
  double no_correspondence_cost; /* voxgraph registration.no_correspondence_cost, default 0 */
} cox_reg_config;

/* Registration points of the reference submap (VoxgraphSubmap::finishSubmap() builds them,
 * called at utils/msg_converter.h:113): n * {x, y, z, distance, weight} floats. */
int cox_regpoints_create(int device, const float* xyz_dist_weight, uint64_t n, cox_regpoints_t** out);
void cox_regpoints_destroy(cox_regpoints_t* pts);
/* the same from / to buffers in HBM, and the set on another GPU of the node (the reference ships the submap -- and with it
 * the inputs of finishSubmap() -- as a ROS message, utils/msg_converter.h:46-118; here the finished point set travels) */
int cox_regpoints_create_dev(int device, const float* xyz_dist_weight_dev, uint64_t n, cox_regpoints_t** out);
int cox_regpoints_data_dev(const cox_regpoints_t* pts, const float** xyz_dist_weight_dev, uint64_t* n);
int cox_regpoints_download(const cox_regpoints_t* pts, float* xyz_dist_weight, uint64_t cap, uint64_t* n);
int cox_regpoints_clone_to_device(const cox_regpoints_t* src, int dst_device, cox_regpoints_t** out);
/* VoxgraphSubmap::finishSubmap() -> findRelevantVoxelIndices, as triggered for every received submap at
 * utils/msg_converter.h:113: the "voxels" (implicit_to_implicit) registration point set = every voxel with
 * weight > min_voxel_weight and |distance| < max_voxel_distance, position = voxel centre, in (z,y,x) block order and
 * linear voxel order.  out may be NULL to query n; cox_regpoints_from_layer keeps the set on the GPU. */
int cox_layer_registration_points(cox_layer_t* layer, float min_voxel_weight, float max_voxel_distance, float* out_xyz_dist_weight, uint64_t cap,
                                  uint64_t* n);
int cox_regpoints_from_layer(cox_layer_t* layer, float min_voxel_weight, float max_voxel_distance, cox_regpoints_t** out);
/* VoxgraphSubmap::finishSubmap() -> findIsosurfaceVertices (utils/msg_converter.h:113; the set the server's configured
 * registration_method "explicit_to_implicit" uses, config/server.yaml:28-31): marching cubes over every block
 * (voxblox MeshIntegrator, corners with weight <= min_weight invalidate a cube), vertices closer than
 * vertex_proximity_threshold merged (createConnectedMesh; voxgraph passes half a voxel), TSDF distance and weight
 * interpolated at every vertex (vertices without 8 observed neighbours dropped).  Order: blocks by (z,y,x), upstream's cube
 * order inside a block.  n_mesh_vertices / n_connected_vertices (optional) report the raw and the merged vertex counts. */
int cox_regpoints_from_isosurface(cox_layer_t* layer, float min_weight, float vertex_proximity_threshold, cox_regpoints_t** out, uint64_t* n_mesh_vertices,
                                  uint64_t* n_connected_vertices);
/* VoxgraphSubmap::getSubmapFrameSurfaceObb: box (submap frame) of the observed voxels (weight > 1e-6) within one voxel of the
 * surface, grown by half a voxel; +-inf when there is none.  overlapsWith() -- the test behind
 * updateRegistrationConstraints(), src/server/pose_graph_interface.cpp:38 -- compares the boxes' world-frame AABBs. */
int cox_layer_surface_obb(cox_layer_t* layer, float min_xyz[3], float max_xyz[3], uint64_t* n_surface_voxels);
/* voxblox EsdfIntegrator::Config (esdf_max_distance / esdf_min_distance: config/coxgraph_client.yaml:68-69) */
typedef struct cox_esdf_config {
  float max_distance_m;     /* the wavefront stops here */
  float min_distance_m;     /* TSDF voxels closer than this to the surface are copied ("fixed") and seed the wavefront */
  float default_distance_m; /* observed voxels start at +- this */
  float min_weight;         /* TSDF voxels below this weight are unobserved */
} cox_esdf_config;
void cox_esdf_config_default(cox_esdf_config* cfg);
/* VoxgraphSubmap::finishSubmap() -> generateEsdf (EsdfIntegrator::updateFromTsdfLayerBatch): the reading side of a
 * registration constraint interpolates the ESDF when use_esdf_distance is set (voxgraph's default).  The result is a layer of
 * its own in TSDF wire layout -- distance = ESDF distance, weight = 1 for observed voxels, colour word = 1 for fixed voxels --
 * so it can be the `reading` of cox_reg_create, be downloaded or handed to another GPU like any layer. */
int cox_esdf_from_tsdf(const cox_layer_t* tsdf, const cox_esdf_config* cfg, cox_layer_t** esdf_out);
/* number of points in a set */
int cox_regpoints_size(const cox_regpoints_t* pts, uint64_t* n);
/* RegistrationConstraint::Config{first_submap_ptr, second_submap_ptr, registration{...}} as set up
 * by PoseGraphInterface::addForceRegistrationConstraint (src/server/pose_graph_interface.cpp:88-105):
 * reference = first submap's registration points, reading = second submap's TSDF layer. */
int cox_reg_create(const cox_regpoints_t* reference, const cox_layer_t* reading, const cox_reg_config* cfg, cox_reg_t** out);
void cox_reg_destroy(cox_reg_t* reg);
/* ceres::CostFunction::Evaluate(parameters, residuals, jacobians): parameter blocks {4,4} =
 * (reference pose, reading pose); residuals[n_res]; jacobians row-major n_res x 4, either may be
 * NULL.  sample_idx = n_res indices into the registration points (the weighted sampler's draws,
 * made explicit so both sides use the same ones) or NULL for "all points in order" (then n_res
 * must equal the number of points, i.e. sampling_ratio = -1). Host output buffers. */
int cox_reg_evaluate(cox_reg_t* reg, const double pose_ref[4], const double pose_read[4], const uint32_t* sample_idx, uint64_t n_res,
                     double* residuals, double* jac_ref, double* jac_read);
/* Fused form: H = J^T J (8x8 row-major, J = [J_ref J_read]), b = J^T r (8), cost = 0.5*|r|^2,
 * n_corr = residuals with a correspondence.  Block-parallel reduction on the GPU; only these
 * 74 numbers cross PCIe. */
int cox_reg_normal_eq(cox_reg_t* reg, const double pose_ref[4], const double pose_read[4], const uint32_t* sample_idx, uint64_t n_res,
                      double H[64], double b[8], double* cost, uint64_t* n_corr);
/* The same evaluation without waiting for it: begin enqueues the kernels on the handle's own stream, finish waits and
 * returns the 74 numbers.  A pose-graph evaluation begins all of its constraints (their kernels overlap on the GPU),
 * then finishes them: one round of launch + PCIe latency per solver iteration instead of one per constraint.  One
 * begin may be outstanding per handle. */
int cox_reg_normal_eq_begin(cox_reg_t* reg, const double pose_ref[4], const double pose_read[4], const uint32_t* sample_idx, uint64_t n_res);
int cox_reg_normal_eq_finish(cox_reg_t* reg, double H[64], double b[8], double* cost, uint64_t* n_corr);
/* All the registration constraints of ONE pose-graph evaluation in ONE launch (Ceres evaluates every residual block of the problem
 * per iteration, include/coxgraph/server/backend/pose_graph.h:52-73; BASELINE configs[4] has 28 of them): regs[c] with the poses
 * poses_ref[4c..], poses_read[4c..]; every handle uses its stored sample set (cox_reg_set_samples / cox_reg_draw_samples) or, without
 * one, all its points in order.  Outputs per constraint as cox_reg_normal_eq: H[64c..], b[8c..], cost[c], n_corr[c] (may be NULL).
 * All handles on one GPU; bit-identical to n calls of cox_reg_normal_eq.  Synchronous. */
int cox_reg_normal_eq_batch(cox_reg_t* const* regs, uint64_t n, const double* poses_ref, const double* poses_read, double* H, double* b, double* cost,
                            uint64_t* n_corr);
/* Keep a set of sample indices on the GPU: later calls that pass sample_idx = NULL with this n_res use them (no 4*n_res
 * byte upload per evaluation).  sample_idx = NULL here drops the stored set (NULL then means "all points in order" again). */
int cox_reg_set_samples(cox_reg_t* reg, const uint32_t* sample_idx, uint64_t n_res);
/* WeightedSampler<RegistrationPoint>::getRandomItem for a whole evaluation, on the GPU: voxgraph's cost function draws its
 * n_res = sampling_ratio * |set| points with replacement, weight-proportionally, with an unseeded std::mt19937 (not
 * reproducible); here draw i = splitmix64(seed, i) scaled into the exact fixed-point prefix sums of the weights.  The draws
 * become the stored set (as after cox_reg_set_samples); a caller that wants voxgraph's "fresh draw per Evaluate" passes a
 * new seed before each evaluation.  cox_reg_get_samples copies the stored set to the host. */
int cox_reg_draw_samples(cox_reg_t* reg, uint64_t n_res, uint64_t seed);
int cox_reg_get_samples(cox_reg_t* reg, uint32_t* sample_idx, uint64_t cap, uint64_t* n_res);
/* HIP-event time of the registration kernel since last reset (bench.py) */
int cox_reg_kernel_time(cox_reg_t* reg, double* ms, uint64_t* launches, int reset);

/* ---- multi-GPU: RCCL behind the C ABI ---------------------------------------------------------------
 * One rank = one process = one GPU (a coxgraph client, or the server's share of the constraints).  The reference's star
 * topology -- the server pulls whole submaps over TCPROS one by one (src/server/client_handler.cpp:82-104,
 * src/server/coxgraph_server.cpp:119-128,253-258), Ceres sums all residual blocks in one process
 * (include/coxgraph/server/backend/pose_graph.h:52-73) -- becomes two collectives over xGMI:
 *   cox_comm_allgather_dev   the submap exchange: every rank's device-resident wire arrays (cox_layer_export_dev,
 *                            cox_regpoints_data_dev) to every rank; receivers rebuild with cox_layer_upload_dev / cox_regpoints_create_dev
 *   cox_comm_allreduce_f64   the packed (4N)^2 + 4N + 1 doubles of one pose-graph evaluation, summed with ONE all-reduce
 * rank 0 makes the id (cox_comm_unique_id) and hands it to the others by whatever channel the application has (the
 * reference would use a ROS topic); librccl.so is loaded on first use, COX_ERR_UNSUPPORTED without it. */
#define COX_COMM_ID_BYTES 128
typedef struct cox_comm cox_comm_t;
int cox_comm_unique_id(uint8_t id[COX_COMM_ID_BYTES]);
int cox_comm_init_rank(int device, int rank, int world, const uint8_t id[COX_COMM_ID_BYTES], cox_comm_t** out);
void cox_comm_destroy(cox_comm_t* comm);
int cox_comm_rank(const cox_comm_t* comm, int* rank, int* world);
/* host buffer in, host buffer out (the payload is KBs and the solver reads it on the host) */
int cox_comm_allreduce_f64(cox_comm_t* comm, double* buf, uint64_t n);
/* recv_dev holds world * bytes_per_rank bytes, rank r's part at offset r * bytes_per_rank; device pointers.  The collective runs on
 * the communicator's own stream behind an event on the stream that produced send_dev (_on: the caller names it; without: the
 * legacy default stream, which orders behind the engine's synchronous exports and behind torch's default stream) -- never behind
 * a device-wide synchronisation.  Returns when the gathered bytes are in recv_dev. */
int cox_comm_allgather_dev(cox_comm_t* comm, const void* send_dev, void* recv_dev, uint64_t bytes_per_rank);
int cox_comm_allgather_dev_on(cox_comm_t* comm, const void* send_dev, void* recv_dev, uint64_t bytes_per_rank, void* producer_stream);

/* ---- process set-up ---------------------------------------------------------------------------------------
 * An integrator keeps frames in flight on four HIP streams, and the ROCm runtime multiplexes a process's streams onto
 * GPU_MAX_HW_QUEUES hardware queues (4 by default, shared with every other HIP user of the process): stages that share a queue
 * run back to back (6.1 k instead of 7.5 k frames/s at 5 cm).  The variable is read when the HIP runtime initialises, so a host
 * calls cox_runtime_prepare() FIRST -- before its first HIP call, before it loads anything that makes one (torch, ...).  It sets
 * GPU_MAX_HW_QUEUES=16 unless the variable is already set, touches nothing else, and returns 1 when the runtime of this process
 * was not yet initialised by this library's knowledge (0: too late to have an effect, or the caller's own value stands).  The
 * library no longer does this behind the host's back when it is loaded (round 2 did: a setenv from a static constructor). */
int cox_runtime_prepare(void);

/* ---- recover mode: mesh-with-history -> per-pose point clouds -> integrator ------------------ */
/* voxblox::MeshConverter (coxgraph/include/coxgraph/map_comm/mesh_converter.h:22-289), the front end of the only
 * in-tree integrator call (TsdfRecover::processMesh, map_comm/tsdf_recover.h:59-99).  A voxblox_msgs/Mesh with
 * per-triangle observation history and the client's trajectory, flattened: */
typedef struct cox_mesh_msg {
  float block_edge_length;          /* Mesh.block_edge_length */
  uint64_t n_blocks;                /* Mesh.mesh_blocks.size() */
  const int64_t* block_index;       /* 3 per block: MeshBlock.index */
  const uint64_t* vertex_begin;     /* n_blocks+1: block b owns vertices [vertex_begin[b], vertex_begin[b+1]); every
                                     * count must be a multiple of 3 (the reference CHECKs x.size()/3 == history.size()) */
  const uint16_t *x, *y, *z;        /* MeshBlock.x/y/z, fixed point in units of 2*block_edge_length/65535 */
  const uint8_t *r, *g, *b;         /* MeshBlock.r/g/b */
  const uint8_t* block_has_history; /* n_blocks: 0 = MeshBlock.history.empty() -> block skipped (mesh_converter.h:87) */
  const uint64_t* history_begin;    /* n_vertices/3 + 1: triangle t (global vertex index / 3) owns history[begin[t], begin[t+1]) */
  const uint32_t* history;          /* ObsHistory.history: inclusive [first, last] frame-id runs, even length */
  uint64_t n_poses;                 /* Mesh.trajectory.poses.size() */
  const uint32_t* stamp_sec;        /* PoseStamped.header.stamp */
  const uint32_t* stamp_nsec;
  const float* T_G_C;               /* 7 per pose (qw qx qy qz tx ty tz): the pose after tf::poseMsgToKindr + cast<float>
                                     * (mesh_converter.h:64-69; that conversion is minkindr's and stays with the caller) */
} cox_mesh_msg;

typedef struct cox_meshconv cox_meshconv_t;
/* MeshConverter(nh_private): interpolate_voxel_size is the rosparam of mesh_converter.h:36-41 (default 0.20) */
int cox_meshconv_create(int device, float interpolate_voxel_size, cox_meshconv_t** out);
void cox_meshconv_destroy(cox_meshconv_t* conv);
/* setMesh(mesh) + setTrajectory(mesh.trajectory)  (mesh_converter.h:55-72): uploads the message. A mesh with an empty
 * trajectory is ignored like the reference does (returns COX_OK, nothing stored). */
int cox_meshconv_set_mesh(cox_meshconv_t* conv, const cox_mesh_msg* mesh);
/* convertToPointCloud(&recovered_pointcloud)  (mesh_converter.h:74-168): decodes the vertices, interpolates every
 * triangle (interpolateTriangle, :212-277), expands the run-length histories and builds one cloud per frame id
 * (uint8 key, as the reference's std::map<uint8_t,...>) in HBM; also transforms each pose's cloud into its camera frame
 * (what getNextPointcloud does per call).  *converted = 0 when the mesh has no blocks (the reference returns false). */
int cox_meshconv_convert(cox_meshconv_t* conv, uint64_t* n_recovered_points, int* converted);
/* the recovered_pointcloud of convertToPointCloud (one XYZRGB point per mesh vertex of blocks with history): n*3 floats, n*3 bytes */
int cox_meshconv_recovered(cox_meshconv_t* conv, float* xyz, uint8_t* rgb, uint64_t cap, uint64_t* n);
/* getNextPointcloud(&i, &T_G_C, &points_C, &colors)  (mesh_converter.h:183-210): device pointers to pose *i's cloud in
 * the camera frame (valid until set_mesh/clear/destroy), *i is advanced; *has_next = 0 when *i is past the trajectory */
int cox_meshconv_next(cox_meshconv_t* conv, int32_t* i, float T_G_C[7], const float** xyz_dev, const uint8_t** rgba_dev, uint64_t* n,
                      int* has_next);
/* host copy of pose i's cloud (tests): does not advance anything */
int cox_meshconv_download(cox_meshconv_t* conv, int32_t i, float* xyz, uint8_t* rgba, uint64_t cap, uint64_t* n);
/* clear()  (mesh_converter.h:170-181) */
int cox_meshconv_clear(cox_meshconv_t* conv);
/* TsdfRecover::processMesh(mesh_msg, &layer_msg, &recovered_pointcloud)  (tsdf_recover.h:59-99) up to the
 * serialisation: removeAllBlocks, setMesh, convertToPointCloud, then one integratePointCloud(T_G_C, points_C, colors,
 * false) per pose with a non-empty cloud, clear().  Everything stays on the GPU; the layer is then read with
 * cox_layer_download.  n_integrated = number of integratePointCloud calls made. */
int cox_recover_process_mesh(cox_meshconv_t* conv, cox_integrator_t* integ, const cox_mesh_msg* mesh, uint64_t* n_recovered_points,
                             uint64_t* n_integrated);

#ifdef __cplusplus
}
#endif
#endif /* COXGRAPH_HIP_H_ */
