/*
 * tl3d.h -- C-ABI of the MI355X depth-fusion back end (libtl3d.so).
 *
 * The reference (kamalnath26/textureless-3d-reconstruction) is pure Python and has no FFI of its
 * own; the seam this library sits under is the Python method surface of its dense back end
 * (SURVEY.md section 8b).  Each entry point below names the reference code it replaces
 * (file:line relative to the reference checkout).  INTEGRATION.md shows the ctypes stubs a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns int: 0 = TL3D_OK, negative = error; no exception crosses the boundary;
 *     tl3d_last_error() returns a thread-local message for the last failing call.
 *   - the caller owns every host buffer it passes in or out for the duration of the call; the library
 *     owns all device memory inside a tl3d_ctx.  "hd" pointers may be host OR device pointers
 *     (copied with hipMemcpyDefault).
 *   - arrays are C-contiguous row-major: depth [H][W], bgr [H][W][3], poses fp64 row-major.
 *   - a pose is world->camera  X_cam = R X_world + t  (depth_to_reconstruction.py:373-376, 543-546).
 *   - one tl3d_ctx per GPU per process; a ctx is not thread-safe; calls enqueue on the ctx's HIP
 *     stream and return, every call with a host output blocks until that output is ready.
 */
#ifndef TL3D_H
#define TL3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TL3D_ABI_VERSION 5

/* error codes */
#define TL3D_OK 0
#define TL3D_E_INVALID (-1)     /* bad argument / shape / slot                       */
#define TL3D_E_HIP (-2)         /* a HIP runtime call failed                         */
#define TL3D_E_NOMEM (-3)       /* device allocation failed                          */
#define TL3D_E_CAPACITY (-4)    /* output buffer too small; *out_n holds the need    */
#define TL3D_E_STATE (-5)       /* call not valid in this state (channel disabled..) */
#define TL3D_E_NODEVICE (-6)    /* no usable GPU                                     */

/* grid channels */
#define TL3D_CH_TSDF 1u         /* {int32 sum of quantised tsdf, int32 weight}         8 B/voxel  */
#define TL3D_CH_CENTROID 2u     /* {sx|sy<<32, sz|n<<32, sr|sg<<32, sb} u64 x4         32 B/voxel */
#define TL3D_CH_FREE 4u         /* the per-brick free-space counts (uint32 [nx ny nz / 512]; tl3d_integrate).  tl3d_grid_device_ptr: the
                                   counts as they stand, pending ones included -- a merge sums them like records (afterwards they are
                                   pending on every rank and fold into the records at the next read).  OR-ed into the channel mask
                                   of tl3d_grid_touched_bricks / _pack_bricks / _unpack_bricks: pending counts stay pending and mark
                                   no brick, i.e. the free-space observations travel as 4 bytes per brick, not as 4 KB of records */
#define TL3D_CH_SUB 8u          /* with the merge helpers below: the unit is a 4x4x4 SUB-BRICK (64 contiguous records), ids = brick * 8 + sub-brick */

/* fixed-point formats of the accumulators (exact, order-free sums => bit-identical multi-GPU merge) */
#define TL3D_TSDF_QSCALE 32767          /* tsdf in [-1,1] -> rint(tsdf * 32767)                  */
#define TL3D_TSDF_MAX_WEIGHT 65536      /* observations one voxel may hold: |sum| <= weight * 32767 < 2^31.  Free-space
                                           voxels seen by every camera reach it first.  tl3d_integrate and tl3d_grid_add
                                           return TL3D_E_STATE instead of wrapping; a merge done outside the library
                                           (all-reduce on tl3d_grid_device_ptr memory) must check the sum of the ranks'
                                           tl3d_grid_max_weight itself (tl3d.distributed does).  The centroid channel's
                                           limit is 2^20 points per voxel.                                        */
#define TL3D_CENTROID_FRAC_BITS 12      /* in-voxel offset in units of voxel/4096                */
#define TL3D_BRICK 8                    /* grid is stored brick-major, 8x8x8 voxels per brick    */

/* depth upload kinds (depth_to_reconstruction.py:80-97) */
#define TL3D_DEPTH_F32_M 0      /* float32 metres (or relative units), as np.load().astype(f32) */
#define TL3D_DEPTH_U16_MM 1     /* uint16 millimetres; converted on device as f32(u16)/1000.0f  */

/* flags of tl3d_backproject / tl3d_accumulate_centroid */
#define TL3D_F_SCALE_F64 1u     /* depth*scale and the range compares run in fp64 (numpy-2 promotion when
                                   `scale` is an np.float64: depth_to_reconstruction.py:356 with :323) */
#define TL3D_F_NO_POSE 2u       /* pose=None: points stay in the camera frame (D2R:377-378)            */

/* extraction modes */
#define TL3D_EXTRACT_CENTROID 0 /* one point per occupied voxel = centroid (Open3D voxel_down_sample)   */
#define TL3D_EXTRACT_TSDF 1     /* zero crossings of the TSDF along +x,+y,+z edges                      */

typedef struct tl3d_ctx tl3d_ctx;

/* Replaces ReconstructionConfig (depth_to_reconstruction.py:45-73) and CameraIntrinsics
 * (depth_enhanced_reconstruction.py:57-80) for the device path, plus the grid geometry. */
typedef struct tl3d_config {
    int32_t abi_version;        /* = TL3D_ABI_VERSION                                              */
    int32_t width, height;      /* W, H of every frame of this ctx                                 */
    double fx, fy, cx, cy;      /* intrinsics; pixel centres sit on integer (u,v) (D2R:291-293)    */
    double min_depth, max_depth;/* strict validity range (D2R:359-361); 0.1/50 D2R, 0.1/100 DER    */
    int32_t n_slots;            /* resident frame slots                                            */
    uint32_t channels;          /* TL3D_CH_* bit mask; 0 = no grid (back-projection / ICP only)    */
    int32_t nx, ny, nz;         /* voxels per axis, multiples of TL3D_BRICK                        */
    double origin[3];           /* world position of the min corner of voxel (0,0,0)               */
    double voxel_size;          /* metres (D2R:64: 0.005)                                          */
    double sdf_trunc;           /* metres                                                          */
    void *ext_tsdf;             /* optional caller-owned device memory for the grids (e.g. a torch */
    void *ext_centroid;         /*   tensor's data_ptr, so torch.distributed can all-reduce it)    */
    void *stream;               /* optional hipStream_t to enqueue on; NULL = library-owned stream */
    int64_t pool_bricks_tsdf;   /* SPARSE grid: the channel holds records for at most this many 8^3 bricks (4 KB each), handed out */
    int64_t pool_bricks_centroid; /* on first touch through a brick table; 0 = dense (every brick has records, 16 KB each for the
                                   centroid channel).  What the reference's hash-map merge gives for free (any extent at any voxel
                                   size, D2R:404-410): nx ny nz may describe a volume far larger than memory.  Bricks that are only
                                   ever free space hold a 4-byte count, no records.  When the pool runs out further new bricks are
                                   refused and counted (tl3d_stats.pool_refused): nothing is written out of bounds.             */
    int64_t voxel_offset[3];    /* the grid is a BLOCK of a larger voxel lattice: its voxel (0,0,0) is voxel voxel_offset of the lattice that
                                   starts at `origin` (multiples of TL3D_BRICK).  Voxel indices are computed against `origin` as ever (Open3D:
                                   floor((p - origin) / voxel), D2R:404-410) and the offset is subtracted: a lattice of more than 2^32
                                   voxels is fused block by block with identical voxels (DenseReconstructor.merge_pointclouds does).  Centroid
                                   channel only: a grid with a TSDF channel must have offset 0.                                      */
} tl3d_config;

/* Result of an ICP run (device solve, read back once at the end). */
typedef struct tl3d_icp_result {
    double T[16];               /* src-camera -> tgt-camera, row-major 4x4                         */
    double fitness;             /* correspondences / valid source samples, at T                    */
    double rmse;                /* sqrt(mean r^2) over correspondences, at T                       */
    int64_t n_corr;             /* correspondences at T                                            */
    int64_t n_src;              /* valid source samples                                            */
    int32_t iters_run;          /* iterations that produced an update                              */
    int32_t status;             /* 0 ok, 1 converged early, 2 singular system (T from last good)   */
    double scale;               /* metric scale of the source depth at the end: the caller's scale_src, or the estimate
                                   of a run with estimate_scale (replaces the SfM scale of D2R:297-326 / DER:659-697) */
} tl3d_icp_result;

typedef struct tl3d_icp_params {
    int32_t iters;              /* maximum Gauss-Newton iterations                                 */
    int32_t stride;             /* source pixel stride                                             */
    double max_dist;            /* correspondence gate |p-q| (metres)                              */
    double damping;             /* Levenberg factor: A += damping * trace(A)/6 * I                 */
    double eps;                 /* stop when |update|_inf < eps                                    */
    double eig_rel;             /* drop directions with eigenvalue < eig_rel * largest (unobservable DOFs) */
    int32_t estimate_scale;     /* 1: Sim(3) -- the metric scale of the SOURCE depth is a 7th unknown (sigma <- sigma
                                   exp(alpha), Jacobian column n . (R sigma p)), solved with the pose under the same
                                   eigenvalue cutoff: where the geometry does not observe it, it keeps scale_src    */
    int32_t reserved;
} tl3d_icp_params;

/* per-launch statistics of the fusion kernels, for roofline accounting */
typedef struct tl3d_stats {
    uint64_t tsdf_launches;
    uint64_t tsdf_records_read;      /* voxel records loaded by tl3d_integrate kernels (counting mode) */
    uint64_t tsdf_records_written;
    uint64_t tsdf_bricks_visited;    /* bricks that passed culling                                      */
    uint64_t tsdf_bricks_free;       /* of those: free-space bricks (no depth lookups)                   */
    uint64_t tsdf_bricks_free_counted; /* of those: handled by ONE add to the brick's free-space counter instead of a
                                          read-modify-write of its 512 records (8 B instead of 8 KB; counting mode)   */
    uint64_t centroid_launches;
    uint64_t centroid_points;        /* points accumulated                                              */
    uint64_t centroid_dropped;       /* valid points that fell outside the grid                         */
    double tsdf_kernel_ms;           /* summed hipEvent time of the integrate kernels (profile mode)    */
    uint64_t tsdf_kernel_timed;      /* launches contributing to tsdf_kernel_ms                         */
    uint64_t tsdf_batch_bricks;      /* bricks the update launches visited (a brick once per batch of frames; counting mode) */
    uint64_t bp_lookback_retries;    /* tl3d_backproject calls repeated in dynamic tile order after a look-back time-out */
    uint64_t icp_batch_timeouts;     /* batched registrations whose in-launch barrier timed out ...                  */
    uint64_t icp_batch_fallback_pairs; /* ... and the pairs re-registered through the per-iteration kernel instead   */
    uint64_t merge_bricks_sent;      /* tl3d_allreduce_grid: bricks whose records went over the wire (all merges so far) ... */
    uint64_t merge_bricks_total;     /* ... of this many bricks in the grid                                              */
    uint64_t pool_slots_tsdf;        /* sparse grids: brick slots handed out so far, per channel (dense: every brick)             */
    uint64_t pool_slots_centroid;
    uint64_t pool_refused;           /* first touches refused because a pool was full: > 0 means the result lacks those bricks   */
    uint64_t centroid_record_updates; /* 32-B centroid records added to in the grid (counted: one per distinct voxel and tile of samples) */
} tl3d_stats;

const char *tl3d_last_error(void);
int tl3d_version(void);
int tl3d_device_count(int *n);
/* HIP_VERSION the library was compiled with, and the runtime / driver versions it is running on (the binding refuses a
 * different major version: PyTorch-ROCm wheels bundle their own runtime and both must resolve to one copy). */
int tl3d_runtime_info(int *hip_compiled, int *hip_runtime, int *hip_driver);
/* Measurement aid: n_streams one-wave kernels of spin_ms each, one per fresh stream; elapsed_ms ~ ceil(n_streams / Q) *
 * spin_ms where Q is the number of hardware queues the runtime really multiplexes streams onto (GPU_MAX_HW_QUEUES is
 * read when the HIP runtime initialises -- setting it later has no effect; the ICP lanes and prep streams want >= 20). */
int tl3d_probe_hw_queues(int device, int n_streams, double spin_ms, double *elapsed_ms);

/* lifetime */
int tl3d_create(const tl3d_config *cfg, int device, tl3d_ctx **out);
int tl3d_destroy(tl3d_ctx *ctx);
int tl3d_sync(tl3d_ctx *ctx);
/* The hipStream_t the context enqueues on (the caller's, tl3d_config.stream, or the library's own): so that a caller can put ITS device
 * work -- a collective on the grid memory -- in the same order instead of waiting for the device (tl3d.distributed does). */
int tl3d_get_stream(tl3d_ctx *ctx, void **stream);
/* The frame buffers of a destroyed context stay in a process-wide cache (by device and size, at most 64 GiB) for the next context
 * of the same shape: the reference's process reconstructs one sequence per run (D2R:705-808), a service reconstructs many, and
 * memory the driver has just taken back is slow to come out of it again.  This call returns all of it to the driver (a host that
 * shares the GPU with another allocator calls it between batches of sequences); an allocation that fails does so by itself. */
int tl3d_release_cached_memory(void);

/* a2: frames.  Replaces DepthImageLoader.load_depth's dtype handling (D2R:80-97) and the in-RAM frame
 * lists self.images/self.depths (D2R:434-437).  bgr may be NULL (colour (0,0,0)). */
int tl3d_upload_frame(tl3d_ctx *ctx, int slot, const void *depth_hd, int depth_kind, const uint8_t *bgr_hd);
int tl3d_download_depth(tl3d_ctx *ctx, int slot, float *depth_out_hd);
/* f2: decode/upload pipeline (replaces holding every decoded frame in host RAM, D2R:434-437, 469-470).
 * Pinned staging buffers + an upload that returns at once: the host buffers must stay untouched until
 * tl3d_slot_wait(slot) returns.  Kernels that read the slot are ordered after the copy on the device. */
int tl3d_pinned_alloc(size_t bytes, void **out);
int tl3d_pinned_free(void *p);
int tl3d_upload_frame_async(tl3d_ctx *ctx, int slot, const void *depth_hd, int depth_kind, const uint8_t *bgr_hd);
int tl3d_slot_wait(tl3d_ctx *ctx, int slot);

/* a3+a4/a5: DenseReconstructor.depth_to_pointcloud (depth_to_reconstruction.py:328-384) and
 * DensePointCloudGenerator.depth_to_pointcloud (depth_enhanced_reconstruction.py:554-613).
 * Writes the surviving points in row-major pixel order.  cap = capacity in points; on
 * TL3D_E_CAPACITY *out_n is the required count.  out_* may be host or device pointers. */
int tl3d_backproject(tl3d_ctx *ctx, int slot, const double R[9], const double t[3], double scale,
                     uint32_t flags, int subsample, double min_depth, double max_depth,
                     float *out_xyz_hd, uint8_t *out_rgb_hd, int64_t cap, int64_t *out_n);

/* The same, asynchronous and device-only: points, colours AND the count go to device memory (out_n_dev: one int64),
 * nothing is read back and the call returns as soon as its one kernel is enqueued.  cap is the capacity of the buffers in
 * points (ceil(H/s) * ceil(W/s) always suffices); points beyond cap are not written, the count is the true one. */
int tl3d_backproject_device(tl3d_ctx *ctx, int slot, const double R[9], const double t[3], double scale,
                            uint32_t flags, int subsample, double min_depth, double max_depth,
                            float *out_xyz_dev, uint8_t *out_rgb_dev, int64_t cap, int64_t *out_n_dev);

/* Extent of the points tl3d_backproject would emit for this frame, without emitting them: out_min / out_max = component-wise
 * min / max of the float32 points (+inf / -inf when no pixel survives).  What `p.min(0)`, `p.max(0)` give the reference when it
 * bounds a cloud for Open3D's voxel origin (D2R:404-410). */
int tl3d_frame_bounds(tl3d_ctx *ctx, int slot, const double R[9], const double t[3], double scale, uint32_t flags,
                      int subsample, double min_depth, double max_depth, double out_min[3], double out_max[3],
                      int64_t *out_reserved /* may be NULL */);

/* The same over n_frames frames with ONE read-back per 16 frames: R = n_frames x 9, t = n_frames x 3 (NULL with
 * TL3D_F_NO_POSE), scales = n_frames entries (NULL = 1.0).  The scene-bounding pass of the pipeline (D2R:404-410 bounds the
 * merged cloud; the extent of the union is the union of the extents). */
int tl3d_frames_bounds(tl3d_ctx *ctx, int n_frames, const int32_t *slots, const double *R, const double *t, const double *scales,
                       uint32_t flags, int subsample, double min_depth, double max_depth, double out_min[3], double out_max[3]);

/* How many 8^3 bricks a fusion of these frames into the grid `grid` describes (its geometry fields only: channels, nx ny nz, origin,
 * voxel_size, sdf_trunc; no grid need be attached) WOULD give records to, per channel: the TSDF classification of every frame
 * and the bricks the frames' samples (stride centroid_subsample; < 1: not counted) fall into, without touching a record.  What
 * the pools of a sparse grid must hold (tl3d_config.pool_bricks_*): Open3D's hash map sizes itself as the merged cloud is inserted
 * (D2R:404-410); a pool is allocated before the first frame, and this is how to know its size -- a few microseconds per frame.
 * Exact: the fusion of the same frames takes exactly these many slots (one per brick, however many waves touch it first). */
int tl3d_count_bricks(tl3d_ctx *ctx, const tl3d_config *grid, int n_frames, const int32_t *slots, const double *R, const double *t,
                      const double *scales, int centroid_subsample, double min_depth, double max_depth, int64_t *bricks_tsdf,
                      int64_t *bricks_centroid);

/* a7 (fusion half): accumulate the same points straight into the centroid channel, no point list
 * (replaces np.vstack + Open3D voxel_down_sample's hash-map insert, D2R:401-410). */
int tl3d_accumulate_centroid(tl3d_ctx *ctx, int slot, const double R[9], const double t[3], double scale,
                             uint32_t flags, int subsample, double min_depth, double max_depth);
/* same, from an explicit point list (merge_pointclouds' call shape, D2R:386-420) */
int tl3d_accumulate_points(tl3d_ctx *ctx, const float *xyz_hd, const uint8_t *rgb_hd, int64_t n);
/* min/max bound of a point list (Open3D's voxel origin = min_bound - voxel/2) */
int tl3d_points_bounds(tl3d_ctx *ctx, const float *xyz_hd, int64_t n, double out_min[3], double out_max[3]);

/* a11: TSDF integration of one frame (no reference code; convention in DESIGN.md).
 * Free space: a brick (8^3 voxels) that lies wholly in front of everything the frame sees would get (+32767, +1) on each
 * of its 512 records; the library adds 1 to a per-brick counter instead and folds the pending counts into the records
 * before anything reads the TSDF channel's records (tl3d_grid_device_ptr, tl3d_grid_download, tl3d_grid_add, tl3d_extract,
 * tl3d_sync), so the channel's contents are the same bit for bit (tl3d_grid_max_weight and the TL3D_CH_FREE forms of the merge
 * calls work on records and counts as they stand).
 * The frame joins a pending BATCH (up to 32 frames, one depth kind): the batch's classification kernels run on a side stream,
 * then ONE update kernel on the context's stream reads and writes every touched record once for all its frames.  A batch is
 * issued when it is full and whenever any other call touches the grid or a slot (tl3d_sync, tl3d_event_record and
 * tl3d_grid_device_ptr included), so results never depend on the batching; a caller that works on a grid pointer obtained
 * EARLIER must call tl3d_grid_device_ptr (or tl3d_sync) again before using it. */
int tl3d_integrate(tl3d_ctx *ctx, int slot, const double R[9], const double t[3], double scale);
/* The fusion loop of a sequence in one call (replaces D2R:625-659 per view): for i in 0..n-1, tl3d_integrate (when the TSDF
 * channel exists) and, when centroid_subsample >= 1 and the centroid channel exists, tl3d_accumulate_centroid of slots[i] with
 * pose (R + 9 i, t + 3 i) and scales[i] (NULL = 1.0).  Same results as the per-frame calls in that order. */
int tl3d_fuse_frames(tl3d_ctx *ctx, int n, const int32_t *slots, const double *R, const double *t, const double *scales,
                     uint32_t flags, int centroid_subsample, double min_depth, double max_depth);

/* a10: vertex/normal map + point-to-plane ICP (no reference code; replaces the SIFT/essential-matrix
 * pose front end D2R:144-215 as the pose source, output in the convention of D2R:618-620). */
int tl3d_build_normals(tl3d_ctx *ctx, int slot, double scale, double depth_jump);
/* the same for n slots in one call (scales: n entries or NULL = 1.0): what a host loop over the frames of a sequence does
 * before registering them (D2R:573 per frame), without a foreign call per frame */
int tl3d_build_normals_many(tl3d_ctx *ctx, int n, const int32_t *slots, const double *scales, double depth_jump);
int tl3d_download_normals(tl3d_ctx *ctx, int slot, float *nmap_out_hd /* [H][W][4] */);
int tl3d_icp_p2plane(tl3d_ctx *ctx, int slot_src, double scale_src, int slot_tgt, const double T_init[16],
                     const tl3d_icp_params *prm, tl3d_icp_result *out);
/* The same registration, asynchronous: one ICP run is a chain of short dependent kernels (latency-bound), but runs
 * for different frame pairs are independent, so up to TL3D_ICP_LANES of them may be in flight, each on its own
 * stream.  enqueue returns at once; collect blocks for that lane's result.  A lane holds one run at a time. */
#define TL3D_ICP_LANES 16
/* A run reads slot_src's depth and slot_tgt's normal map until it is collected.  tl3d_upload_frame* into slot_src and
 * tl3d_build_normals of slot_tgt issued meanwhile are ordered behind the run on the device (they do not corrupt it). */
int tl3d_icp_enqueue(tl3d_ctx *ctx, int lane, int slot_src, double scale_src, int slot_tgt, const double T_init[16],
                     const tl3d_icp_params *prm);
int tl3d_icp_collect(tl3d_ctx *ctx, int lane, tl3d_icp_result *out);

/* a2 (host side of the decode pipeline; no GPU call): the decode workers of a host copy image rows into pinned staging
 * buffers without holding their interpreter's lock (a foreign call releases it).  rows = `height` pointers to image rows as an
 * image library keeps them: R,G,B,X 4-byte pixels -> packed B,G,R (what cv2.imread hands the reference, D2R:454), or rows
 * of row_bytes bytes copied as they are (16-bit depth PNGs, D2R:86-90). */
int tl3d_host_pack_bgr_rows(uint8_t *dst, const uint8_t *const *rows, int height, int width);
int tl3d_host_copy_rows(uint8_t *dst, const uint8_t *const *rows, int height, size_t row_bytes);

/* a10, batched: n_pairs independent registrations, each through ALL of `levels` (coarse to fine: a level starts from the
 * pose the previous one ended with; a pair stops early when a level fails or ends with fewer than 8 correspondences, as
 * the per-level calls above are used by a host) in ONE kernel launch: no host round trip and no launch per iteration.
 * The result of a pair is that of its last level run; it does not depend on which batch the pair is in.  One batch may be
 * in flight per context (own stream); uploads and normal maps issued before the enqueue precede it, slot rewrites issued
 * after it wait for it.  What replaces the reference's per-pair detect_and_match + compute_pose calls (D2R:573-596). */
#define TL3D_ICP_MAX_LEVELS 4
typedef struct tl3d_icp_pair {
    int32_t slot_src, slot_tgt;
    double scale_src;           /* metric scale of the source depth                                */
    double T_init[16];          /* initial src-camera -> tgt-camera pose, row-major 4x4            */
} tl3d_icp_pair;
int tl3d_icp_batch_enqueue(tl3d_ctx *ctx, const tl3d_icp_pair *pairs, int n_pairs, const tl3d_icp_params *levels, int n_levels);
int tl3d_icp_batch_collect(tl3d_ctx *ctx, tl3d_icp_result *out /* [n_pairs] */, int n_pairs);

/* grids */
/* Give a context created with channels = 0 its grid later (geometry fields of cfg: channels, nx, ny, nz, origin,
 * voxel_size, sdf_trunc, ext_*): frames stay resident while poses and scene bounds are still being computed. */
int tl3d_attach_grid(tl3d_ctx *ctx, const tl3d_config *cfg);
int tl3d_grid_reset(tl3d_ctx *ctx);
int tl3d_grid_device_ptr(tl3d_ctx *ctx, uint32_t channel, void **ptr, size_t *bytes);
int tl3d_grid_download(tl3d_ctx *ctx, uint32_t channel, void *out_hd, size_t bytes);
int tl3d_grid_upload(tl3d_ctx *ctx, uint32_t channel, const void *in_hd, size_t bytes);
int tl3d_grid_add(tl3d_ctx *ctx, uint32_t channel, const void *other_hd, size_t bytes);   /* grid += other (merge) */
/* largest number of observations any voxel of the TSDF channel holds (one reduction over the grid, blocks) */
int tl3d_grid_max_weight(tl3d_ctx *ctx, int64_t *out);

/* Sparse form of the merge (a frame-sharded run touches a few per cent of a large grid): which bricks hold anything, and their
 * records as one contiguous block.  tl3d_grid_touched_bricks ORs 1 into map[b] (one byte per brick, nx ny nz / 512 of them, device
 * memory; the caller zeroes it) for every brick with a TSDF weight or a centroid count in the selected channels; after a MAX
 * all-reduce of the map every rank holds the same brick set.  tl3d_grid_pack_bricks copies the records of bricks[0 .. n) (device
 * memory, ascending brick indices) of ONE channel into `packed` (n x 4 KB for TL3D_CH_TSDF, n x 16 KB for TL3D_CH_CENTROID, device
 * memory); tl3d_grid_unpack_bricks writes such a block back (after the SUM all-reduce).  tl3d.distributed.merge_context_grids and
 * tl3d_allreduce_grid use them when fewer than half of the bricks are touched; tl3d.distributed passes TL3D_CH_FREE with the TSDF
 * channel and sums the counts separately (config-5 shape, 32 frames: 29 % of the bricks hold free-space counts, a few per cent records). */
int tl3d_grid_touched_bricks(tl3d_ctx *ctx, uint32_t channels, uint8_t *map_dev, int64_t n_bricks);
int tl3d_grid_pack_bricks(tl3d_ctx *ctx, uint32_t channel, const uint32_t *bricks_dev, int64_t n, void *packed_dev);
int tl3d_grid_unpack_bricks(tl3d_ctx *ctx, uint32_t channel, const uint32_t *bricks_dev, int64_t n, const void *packed_dev);

/* e: the merge step of the multi-GPU path for hosts WITHOUT torch.distributed (frames shard across
 * ranks, one sum all-reduce of the per-GPU grids at merge time).  One process per GPU; rank 0 obtains an id and hands it
 * to the others by any means (file, socket, MPI); every rank then joins and merges.  RCCL is loaded at run time
 * (librccl.so, the copy already in the process if there is one), so libtl3d.so itself does not depend on it.  The
 * all-reduce runs on the context's stream, in place on the grid memory (int32 / uint64 sums: the merged grid is
 * bit-identical to a single-GPU run); the TSDF channel's int32 headroom (TL3D_TSDF_MAX_WEIGHT) is checked over all ranks
 * first and the merge refused with TL3D_E_STATE if it could wrap.  Python hosts use tl3d.distributed (same operations
 * through torch.distributed on tl3d_grid_device_ptr memory). */
#define TL3D_RCCL_ID_BYTES 128
int tl3d_rccl_unique_id(uint8_t id_out[TL3D_RCCL_ID_BYTES]);
int tl3d_rccl_init(tl3d_ctx *ctx, int world, int rank, const uint8_t id[TL3D_RCCL_ID_BYTES]);
int tl3d_allreduce_grid(tl3d_ctx *ctx, uint32_t channels /* TL3D_CH_* mask; 0 = every channel the grid has */);

/* a7 (read-back half) + N4: fused grid -> point list. min_count: centroid occupancy threshold;
 * tsdf gate (centroid mode, only if the TSDF channel exists and min_weight > 0): keep voxels with
 * weight >= min_weight and |mean tsdf| <= max_abs_tsdf. */
int tl3d_extract(tl3d_ctx *ctx, int mode, int min_count, int min_weight, double max_abs_tsdf,
                 float *out_xyz_hd, uint8_t *out_rgb_hd, int64_t cap, int64_t *out_n);

/* f1: statistical outlier removal on a point list (Open3D remove_statistical_outlier, D2R:412-415) */
int tl3d_statistical_outlier(tl3d_ctx *ctx, const float *xyz_hd, int64_t n, int nb_neighbors, double std_ratio,
                             double cell_size, uint8_t *keep_out_hd, int64_t *out_kept);

/* measurement */
int tl3d_set_profile(tl3d_ctx *ctx, int count_records, int time_kernels);
/* Noise-robust registration: normal maps built after this call come from the depth averaged over a (2 radius + 1)^2 window
 * -- the HARMONIC mean (mean of 1 / z, which is linear in the pixel coordinates on any plane) over the centre pixel and the
 * pixel pairs (u + du, v + dv), (u - du, v - dv) that are both valid and within depth_jump of the centre (a symmetric set: no
 * average runs across a depth edge, and none is pulled to one side at the edge of a surface) -- with the tangent vectors
 * `radius` pixels to either side; registrations then read that averaged depth as their SOURCE too (tl3d_icp_*: a slot whose
 * normal map was built smoothed).  0 (default) = central differences of the depth image itself.  With 1 mm of depth noise the
 * frame-to-frame chain of config 2 drifts 7 x less at radius 1 (tests/test_gpu_baseline_configs.py). */
int tl3d_set_normal_smoothing(tl3d_ctx *ctx, int radius);

/* tl3d_integrate collects up to 32 frames per update launch (the bricks they see are read and written once for all of them;
 * the grid is the same bit for bit).  on = 0: one frame per launch.  Default: on. */
int tl3d_set_tsdf_pairing(tl3d_ctx *ctx, int on);
int tl3d_get_stats(tl3d_ctx *ctx, tl3d_stats *out);
int tl3d_reset_stats(tl3d_ctx *ctx);
int tl3d_event_record(tl3d_ctx *ctx, int which /* 0 or 1 */);
int tl3d_event_elapsed_ms(tl3d_ctx *ctx, float *ms);      /* time between event 0 and event 1, blocks */

#ifdef __cplusplus
}
#endif
#endif /* TL3D_H */
