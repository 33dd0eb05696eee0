// tl3d_internal.h -- shared declarations of libtl3d.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tl3d.h"

namespace tl3d {

// ---- parameter blocks passed to kernels by value ---------------------------------------------
struct Cam {
    int W, H;
    float fx, fy, cx, cy;        // f32 path (TSDF, normals, ICP)
    double fxd, fyd, cxd, cyd;   // fp64 path (reference-exact back-projection)
};

// PHASE-MAJOR ROWS (normal maps and window-averaged depth: what the registration gathers from).  Pixel u of a row sits at
// (u & 3) * W4 + (u >> 2), W4 = ceil(W / 4): the row is kept as its four pixel phases one after the other.  A registration level
// samples every stride-th pixel (stride 4 and 2 in the pipeline's schedule) and the projected targets of neighbouring samples keep
// the stride over long runs, so in a row-major map every sample pays a 64-B sector of its own for a 16-B entry (8.3 MB fetched
// per 1080p pair-iteration at stride 4 for 2.1 MB used); here samples 4 px apart are neighbours in memory: four entries per
// sector.  Stride-1 accesses cost the same sectors as before (64 consecutive pixels = four runs of 16 entries).  Same values,
// another address: every result is bit for bit what the row-major map gave.
__host__ __device__ __forceinline__ int pm_w4(int W) { return (W + 3) >> 2; }
__host__ __device__ __forceinline__ size_t pm_pixels(int W, int H) { return (size_t)4 * pm_w4(W) * H; }
__host__ __device__ __forceinline__ size_t pm_index(int u, int v, int w4) { return (size_t)v * (size_t)(4 * w4) + (size_t)((u & 3) * w4 + (u >> 2)); }

struct Grid {
    int nx, ny, nz;              // voxels
    int nbx, nby, nbz;           // bricks (8^3 voxels)
    float ox, oy, oz, vs;        // f32 origin / voxel size (TSDF path)
    double oxd, oyd, ozd, vsd;   // fp64 (centroid path, Open3D index semantics)
    double offx, offy, offz;     // the grid's voxel (0,0,0) is voxel (offx, offy, offz) of the lattice that starts at the origin (integers; centroid path)
    float trunc, inv_trunc;
    // Brick tables: the records of virtual brick b of a channel sit in pool slot table[b] (512 records each).  A dense grid has
    // the identity table and a pool of every brick; a SPARSE grid starts with an empty table and hands out slots on first
    // touch (brick_slot_ensure) until its pool is full.  cursors: [0] TSDF slots handed out, [1] TSDF refusals (pool full),
    // [2] / [3] the same for the centroid channel.
    unsigned *tsdf_tab, *cen_tab, *cursors;
    unsigned tsdf_cap, cen_cap;
    const unsigned *free_cnt;    // per-brick free-space counts not yet folded into records (readers add them: see tsdf_record)
};
constexpr unsigned SLOT_EMPTY = 0xffffffffu;    // no records yet
constexpr unsigned SLOT_PENDING = 0xfffffffeu;  // a wave is drawing the brick's slot right now (inside the allocating kernel only)
constexpr unsigned SLOT_FULL = 0xfffffffdu;     // the pool had no slot left when the brick was first touched: its updates are dropped (counted)
                                                // (every value >= SLOT_FULL reads as "no records")

struct PoseF {                   // world->camera, f32 (or src->tgt for ICP)
    float r[9];
    float t[3];
};

struct PoseD {                   // fp64 R (row-major) and ct = R^T t, for the back-projection path
    double r[9];
    double ct[3];
};

struct BpArgs {                  // back-projection arguments common to count / write / centroid kernels
    int sub, Ws, Hs;             // stride and subsampled extent (ceil)
    unsigned flags;
    double scale, min_d, max_d;
    unsigned long long zero;     // 0 the compiler cannot see: fetch_add(p, zero) stays a read-modify-write (a fresh read, see bp look-back)
};

// one frame of a batch of voxel-centroid accumulations (kernels_centroid.hip)
constexpr int TL3D_CEN_MAXBATCH = 32;
struct CenFrame {
    PoseD p;
    BpArgs a;
    const float *depth;
    const uint8_t *bgr;
};

// frustum side planes for brick culling: inside iff nx*x + nz*z >= -rad (left/right), ny*y + nz*z >= -rad
struct Frustum {
    float lx, lz, rx, rz, ty, tz, by, bz;
};

struct IcpState {                // lives in device memory for the whole ICP run (no host round trips)
    double T[16];
    double sums[40];             // last reduced sums: a[21] b[6] e cnt nsrc . . c[6] cc bc (scale column, Sim(3) runs only)
    double scale;                // metric scale of the source depth: the caller's value, updated by runs that estimate it
    int done;                    // set on convergence or failure: remaining iteration kernels exit at once
    int status;
    int iters_run;
    int over;                    // batched runs: no further level follows (last level done, failed, or < 8 correspondences)
};

struct IcpRun {                  // per-run arguments of the ICP kernels, read from device memory so that a lane's
    const float *depth_src;      // kernel chain has constant launch arguments and can be replayed as a hipGraph
    const float4 *nmap_tgt;
    float scale, md2, mind, maxd;
    int stride, Ws, Hs, est_scale;   // est_scale: the source depth's scale is a 7th unknown (Sim(3))
    double damping, eps, eig_rel;
    int src_pm, pad;             // src_pm: depth_src is a window-averaged depth in phase-major rows (pm_index), else the frame's row-major image
};

// Batched registration (icp_batch_kernel): every pair of a batch runs ALL its levels and iterations inside one launch.
constexpr int ICP_MAX_LEVELS = 4;
constexpr int ICP_BATCH_MAX_MEMBERS = 64;    // workgroups that may share one pair (what the buffers are sized for)
// Samples per member and pass.  A member's pass costs ~4.5 us whatever it accumulates (wave + workgroup reduction, partial store,
// arrival, barriers, poll) plus ~0.55 us per sample and thread; measured on 255 pairs of 1080p frames (tools/bench_icp.py, round 4,
// two slots per workgroup): stride 4 at 32 / 16 / 8 members per pair 1.35 / 1.03 / 0.92 us per pair-iteration, the two-level
// schedule 57 / 66 / 65 k pairs/s (45 k with the 64 members of round 3).  16 members at 1080p stride 4, at most 32: a pair
// registered alone still spreads over 16-32 CUs.  A function of the level geometry only, never of the batch: a pair's sums (and
// pose) do not depend on the batch it is in.
constexpr int ICP_BATCH_SAMPLES_PER_MEMBER = 8192;
constexpr int ICP_BATCH_MEMBERS_CAP = 32;
struct IcpBatchPair { const float *depth_src; const float4 *nmap_tgt; float scale; int src_pm; };    // src_pm: as IcpRun::src_pm
struct IcpLevel { float md2; int stride, Ws, Hs, iters, est_scale; double damping, eps, eig_rel; };
struct IcpBatchArgs {
    const IcpBatchPair *pairs;   // [n_pairs]
    IcpState *states;            // [n_pairs] initial pose in, result out
    unsigned *sync;              // two arrays of 64 x sync_rows 64-B lines: arrival counters, then generation lines (zero at launch);
                                 // pair p owns line (p % 64) * sync_rows + p / 64 of each
    double *slab;                // [n_pairs][members][ICP_SLAB] partial sums of the current pass
    unsigned *ctl;               // [0] workgroup tickets handed out, [1] error (a wait timed out); zero at launch
    int n_pairs, members, n_levels, sync_rows, zero;    // zero: 0 the compiler cannot see (an add of it stays a read-modify-write)
    float mind, maxd;
    unsigned *stage;             // experiments: [n_pairs * members][4] progress markers (null in production)
    unsigned long long *dbg;     // experiments: [members][16 passes][8] timestamps of pair 0 (null in production)
    IcpLevel lv[ICP_MAX_LEVELS];
};

// Frame buffers come from slabs, not one hipMalloc per buffer: a 1000-frame context used to make (and, slower, free) 4000
// allocations.  One pool per buffer kind (equal-sized blocks); a slab holds up to 64 blocks and lives until tl3d_destroy.
constexpr int FRAME_SLAB_BLOCKS = 64;
struct FramePool {
    size_t block = 0;             // bytes per block, rounded up to 256
    int remaining = 0;            // blocks this pool may still hand out (= frame slots without a buffer of this kind)
    char *cur = nullptr;          // next free block of the newest slab
    int cur_left = 0;
    void **slabs = nullptr;       // [max_slabs]
    size_t *slab_bytes = nullptr; // [max_slabs] size of each slab (they go back to the process-wide slab cache by size)
    int n_slabs = 0, max_slabs = 0;
    int device = 0;
};

struct Slot {
    float *depth = nullptr;      // [H][W] f32
    uint16_t *depth_u16 = nullptr;   // [H][W] millimetres, kept when the frame was uploaded as TL3D_DEPTH_U16_MM (TSDF gathers read it)
    bool has_u16 = false;
    uint8_t *bgr = nullptr;      // [H][W][3]
    float4 *nmap = nullptr;      // [H][4 * W4] (nx,ny,nz,d) in phase-major rows (pm_index), lazily allocated
    float *sdepth = nullptr;     // [H][4 * W4], phase-major rows: window-averaged depth the normal map was taken from (tl3d_set_normal_smoothing > 0); registration reads it as the source depth too
    int smooth_radius = 0;       // radius sdepth / nmap were built with (0: nmap from the depth image itself)
    hipEvent_t ev_upload = nullptr;   // recorded on the main stream after the slot's last upload
    hipEvent_t ev_normals = nullptr;  // recorded on the main stream after the slot's normal map was built
    bool has_color = false;
    bool loaded = false;
    bool has_normals = false;
};

constexpr int ICP_MAX_BLOCKS = 256;     // one reduce workgroup per CU; the solve sums the slab serially
constexpr int ICP_SLAB = 40;     // doubles per block partial: 30 sums of the pose system (+ 2 unused), 8 of the scale column

}  // namespace tl3d

// TSDF updates are issued in batches of up to TL3D_TSDF_MAXBATCH frames (one bit per frame in a brick's frame mask): one prep
// chain (5 launches) for the whole batch on the side stream, then ONE update launch on the main stream that reads and writes
// every touched record once per batch.  Four batch scratch buffers are taken in turn (three prep streams), so the prep chains of
// the next batches run beside the update of batch k.
#define TL3D_TSDF_MAXBATCH 32
constexpr int TSDF_SCRATCHES = 4;

struct tl3d_ctx {
    tl3d_config cfg;
    int device;
    hipStream_t stream;
    bool own_stream;
    tl3d::Cam cam;
    tl3d::Grid grid;
    size_t nvox;
    tl3d::Slot *slots;
    tl3d::FramePool pool_depth, pool_u16, pool_bgr, pool_nmap, pool_sdepth;
    int normal_radius;           // tl3d_set_normal_smoothing: window radius of the next normal maps (0: none)
    int2 *tsdf;                  // record pool of the TSDF channel: [tsdf_cap bricks][512] {sum_q, weight}; a dense grid's pool is the grid
    unsigned long long *centroid;// record pool of the centroid channel: [cen_cap bricks][512][4]
    bool own_tsdf, own_centroid;
    bool sparse;                 // pools smaller than the grid, slots handed out on first touch (tl3d_config.pool_bricks_*)
    unsigned *brick_tabs;        // one allocation: TSDF table [nbricks], centroid table [nbricks], cursors [4]
    // scratch
    // TSDF integration is double-buffered over two streams: the tile/pyramid/cull kernels of frame i+1 run on
    // prep_stream while the update kernel of frame i streams the grid on the main stream.
    hipStream_t prep_stream[4];  // consecutive BATCHES take them in turn (three by default: one per batch scratch)
    int n_prep_streams;
    void *tsdf_scratch[TSDF_SCRATCHES];   // batch scratch (descriptors, tile pyramids, brick lists, frame masks, sub-brick masks): three batches in flight
                                          // (the update of batch k, the prep chains of batches k+1 and k+2)
    void *tsdf_scratch_slab;              // the one allocation they are carved from
    bool tsdf_pairing;                    // frames of a batch share ONE update launch (tl3d_set_tsdf_pairing(ctx, 0): one frame per launch)
    bool pend_u16;                        // depth kind of the pending batch (a batch holds one kind)
    hipEvent_t ev_prep[TSDF_SCRATCHES];                // prep of the batch using scratch h is done (recorded on its prep stream)
    hipEvent_t ev_upd[TSDF_SCRATCHES];                 // the update of the last batch that used scratch h is done (main stream)
    bool upd_recorded[TSDF_SCRATCHES];
    bool tsdf_use_u16;                    // gather from the millimetre image when the slot has one (env TL3D_U16_GATHER=0: never)
    int tsdf_batch;                       // frames per batch (env TL3D_TSDF_BATCH, default 32; 1 = no deferral)
    unsigned tsdf_seq, tsdf_batch_no;
    struct PendingUpdate { int slot; tl3d::PoseF pose; float scale; } pend[TL3D_TSDF_MAXBATCH];
    tl3d::CenFrame cen_pend[tl3d::TL3D_CEN_MAXBATCH];   // voxel-centroid accumulations not yet launched (one stride per batch)
    int n_cen_pend = 0;
    tl3d::CenFrame *d_cen_frames = nullptr;              // their descriptors on the device
    int n_pend;                           // frames of the batch being collected
    // extraction is called twice (size query, then with buffers): the block counts of the query are kept while nothing
    // has touched the grids in between (every grid-modifying or pointer-exposing call bumps grid_epoch)
    unsigned long long grid_epoch, ext_epoch, ext_total;
    int ext_mode, ext_min_count, ext_min_weight;
    double ext_max_abs;
    bool ext_valid;
    unsigned long long *bp_state;        // one-launch back-projection: ticket, error word, per-tile granules, [bp_state_words - 1] = total
    size_t bp_state_words;
    float *bp_stage_xyz;                 // staging for host-side outputs, sized for a full frame at subsample 1
    uint8_t *bp_stage_rgb;
    bool bp_async_pending;
    double *bp_factors;                  // projection factors (D2R:287-295): [0, W) xf[u] = (u - cx) / fx, [W, W + H) yf[v] = (v - cy) / fy
    unsigned *block_counts;      // compaction counts
    unsigned long long *block_offsets;
    size_t scratch_blocks;
    // int32 headroom of the TSDF sums: |sum_q| <= weight * 32767 stays below 2^31 while weight <= TL3D_TSDF_MAX_WEIGHT.
    // tsdf_w_upper bounds the largest voxel weight from above (+1 per integrated frame); when it reaches the limit, or after
    // a merge the library could not see (grid upload, grid pointer handed out), it is re-measured by a reduction over the grid
    // Free-space counters: a brick that is wholly free space in a frame gets +1 here (one integer add by the classification
    // kernel) instead of (+32767, +1) on each of its 512 records; the pending counts are folded into the records before
    // anything reads the TSDF channel.  TL3D_FREE_COUNTERS=0 keeps the round-1 behaviour (records streamed every frame).
    unsigned *free_cnt;
    bool free_dirty;
    hipEvent_t ev_free;                   // recorded on the main stream behind its last write to free_cnt (clear, fold); prep chains wait for it
    bool ev_free_recorded;
    int tsdf_max_blocks, tsdf_xcd_group;  // launch geometry of the update kernel (1536 workgroups, XCD-grouped lists)
    bool tsdf_single_stream;
    void *rccl_comm;             // ncclComm_t of tl3d_rccl_init (RCCL is dlopen'ed: tl3d_api.hip)
    int rccl_world;
    long long tsdf_w_upper;
    bool tsdf_w_unknown;
    int *d_maxw;
    unsigned long long *d_counters;   // device counters [16]
    unsigned long long *d_cen_counters;   // centroid statistics, sharded: [256 lines][8] (points kept, points dropped)
    struct IcpLane {             // one in-flight ICP run: own stream, device state, partial-sum slab, pinned read-back
        hipStream_t stream;
        double *slab;            // [ICP_MAX_BLOCKS][ICP_SLAB]
        unsigned *ticket;        // arrival counter of the iteration kernel (64-B block of its own; 0 between launches)
        tl3d::IcpState *state;
        tl3d::IcpState *host;    // pinned: initial state in, final state out
        tl3d::IcpRun *run;       // device descriptor of the current run
        tl3d::IcpRun *run_host;  // pinned
        // captured chains: descriptor + state upload, ticket re-arm, (iters+1) iteration kernels, state download -- one per
        // iteration count used lately (a coarse-to-fine caller alternates between two or three)
        hipGraphExec_t graphs[4];
        int graph_iters[4];
        int graph_next;          // slot the next new iteration count replaces
        bool busy;               // a run was enqueued and not collected yet
        hipEvent_t ev_done;      // recorded on the lane's stream behind the run: writers of the slots it reads wait on it
        int src_slot, tgt_slot;  // the run reads slots[src].depth and slots[tgt].nmap
    };
    IcpLane icp_lanes[TL3D_ICP_LANES];
    struct IcpBatch {            // one batch of registrations in flight, on its own stream
        hipStream_t stream;
        hipEvent_t ev_ready, ev_done;
        int cap_pairs; size_t cap_slab;          // capacities of the buffers below
        int sync_rows;                           // geometry of `sync` (see IcpBatchArgs)
        tl3d::IcpBatchPair *pairs, *pairs_host;  // device / pinned
        tl3d::IcpState *states, *states_host;
        unsigned *sync, *ctl, *ctl_host;
        double *slab;
        int n_pairs;             // of the run in flight
        tl3d_icp_pair *req_pairs;                // the request as the caller made it: the fallback of a timed-out launch re-registers from it
        int req_cap, req_n_levels;
        tl3d_icp_params req_levels[TL3D_ICP_MAX_LEVELS];
        unsigned *stage; size_t stage_n;  // experiments (TL3D_ICP_STAGES)
        unsigned long long *dbg; // experiments (TL3D_ICP_TRACE)
        int dbg_members;
        bool busy;
    } icp_batch;
    float *bounds_slab;
    // stats / profiling
    tl3d_stats stats;
    bool count_records, time_kernels;
    hipEvent_t ev[2];
    hipEvent_t kev0, kev1;
    struct KTimer { hipEvent_t a, b; int launches; };   // one event pair around `launches` back-to-back update kernels
    KTimer *ktimers;
    int n_ktimers, ktimers_used;
};

namespace tl3d {

int set_err(int code, const char *fmt, ...);

#define TL3D_HIP(x)                                                                          \
    do {                                                                                     \
        hipError_t e__ = (x);                                                                \
        if (e__ != hipSuccess) return tl3d::set_err(TL3D_E_HIP, "%s failed: %s (%s:%d)", #x, \
                                                    hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ size_t brick_base(int bx, int by, int bz, int nbx, int nby) {
    return (((size_t)bz * (size_t)nby + (size_t)by) * (size_t)nbx + (size_t)bx) << 9;
}
// Record order inside a brick: eight 4x4x4 sub-bricks (sub-brick = x >> 2 | (y >> 2) << 1 | (z >> 2) << 2) of 64 contiguous records
// each, x fastest inside a sub-brick.  A sub-brick is the unit the TSDF update classifies and touches: 64 records = one 8-B-per-lane
// access of a wave = 512 contiguous bytes of the TSDF channel (2 KB of the centroid channel).
__host__ __device__ __forceinline__ int in_brick_index(int i, int j, int k) {
    return ((k & 4) << 6) | ((j & 4) << 5) | ((i & 4) << 4) | ((k & 3) << 4) | ((j & 3) << 2) | (i & 3);
}
__host__ __device__ __forceinline__ void in_brick_coords(int l, int &i, int &j, int &k) {
    i = ((l >> 4) & 4) | (l & 3);
    j = ((l >> 5) & 4) | ((l >> 2) & 3);
    k = ((l >> 6) & 4) | ((l >> 4) & 3);
}
__device__ __forceinline__ size_t vox_index(int i, int j, int k, int nbx, int nby) {
    return brick_base(i >> 3, j >> 3, k >> 3, nbx, nby) + (size_t)in_brick_index(i, j, k);
}

// ---- brick tables ---------------------------------------------------------------------------------------------------------
// The table is read and written by kernels that run side by side: agent-scope accesses only (a CU's L1 would go on showing
// SLOT_EMPTY after another CU -- or this one, through the L2 -- filled the entry, and every such stale read would take, and
// leak, a fresh slot).
__device__ __forceinline__ unsigned brick_slot(const unsigned *table, unsigned brick) {
    return __hip_atomic_load(table + brick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Slot of `brick`, handing one out on first touch -- exactly one per brick: the wave that turns the entry from EMPTY to PENDING
// draws the slot and publishes it; a wave that finds PENDING waits for that store (the drawing wave waits for nobody: an atomic add
// and a store, so the wait ends; it is bounded all the same -- ~30 ms -- and then gives up as if the pool were full, counted).
// Until round 3 every contender drew a slot and the losers of the compare-and-swap lost theirs to the pool: with 32 frames per
// accumulation launch, first touches of one brick by a dozen workgroups at once, a tenth of the centroid pool leaked, and a pool
// sized by a count (tl3d_count_bricks) must not leak.  SLOT_FULL when the pool is exhausted (sticky: later touches see it).
__device__ __forceinline__ unsigned brick_slot_ensure(unsigned *table, unsigned *cursor, unsigned cap, unsigned brick) {
    for (unsigned spin = 0; spin < (1u << 20); ++spin) {
        const unsigned s = brick_slot(table, brick);
        if (s < SLOT_PENDING) return s;                            // a slot, or SLOT_FULL
        if (s == SLOT_EMPTY) {
            const unsigned old = atomicCAS(table + brick, SLOT_EMPTY, SLOT_PENDING);
            if (old == SLOT_EMPTY) {
                unsigned n = atomicAdd(cursor, 1u);
                if (n >= cap) {
                    n = SLOT_FULL;
                    atomicAdd(cursor + 1, 1u);
                }
                __hip_atomic_store(table + brick, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return n;
            }
            if (old < SLOT_PENDING) return old;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    atomicAdd(cursor + 1, 1u);
    return SLOT_FULL;
}
// the same for every lane of a wave at once (want: the lane needs its brick's slot): one leader per distinct brick draws, the
// lanes that share the brick take its answer.  Every lane of the wave must call it.
__device__ __forceinline__ unsigned wave_slots(unsigned *table, unsigned *cursor, unsigned cap, unsigned brick, bool want) {
    unsigned slot = want ? brick_slot(table, brick) : 0u;          // the common case: every brick has its slot already
    unsigned long long todo = __ballot(want && slot >= SLOT_PENDING);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)brick, leader);
        unsigned s = 0;
        if ((threadIdx.x & 63) == leader) s = brick_slot_ensure(table, cursor, cap, b);
        s = (unsigned)__builtin_amdgcn_readlane((int)s, leader);
        const bool mine = want && slot >= SLOT_PENDING && brick == b;
        todo &= ~__ballot(mine);
        if (mine) slot = s;
    }
    return slot;
}
// TSDF record of virtual record index idx as a reader must see it: the stored sums plus the brick's pending free-space count
// (count x (32767, 1)); a brick without records reads as free space only
__device__ __forceinline__ int2 tsdf_record(const Grid &g, const int2 *__restrict__ pool, size_t idx) {
    const unsigned brick = (unsigned)(idx >> 9);
    const unsigned s = brick_slot(g.tsdf_tab, brick);
    int2 r = make_int2(0, 0);
    if (s < SLOT_FULL) r = pool[((size_t)s << 9) | (idx & 511)];
    if (g.free_cnt) {
        const unsigned c = g.free_cnt[brick];
        r.x += (int)(c * 32767u);
        r.y += (int)c;
    }
    return r;
}
// centroid record (4 words) of virtual record index idx, nullptr when its brick has none
__device__ __forceinline__ const unsigned long long *cen_record(const Grid &g, const unsigned long long *__restrict__ pool, size_t idx) {
    const unsigned s = brick_slot(g.cen_tab, (unsigned)(idx >> 9));
    return s < SLOT_FULL ? pool + 4 * (((size_t)s << 9) | (idx & 511)) : nullptr;
}

// ---- kernel launchers (one per .hip file) -----------------------------------------------------
// frames
int launch_u16_to_f32(hipStream_t s, const uint16_t *in, float *out, size_t n);
// back-projection
int launch_scan(hipStream_t s, const unsigned *counts, unsigned long long *offsets, int n, unsigned long long *total);
int launch_bp_bounds(hipStream_t s, const Cam &cam, const BpArgs &a, const PoseD &p, const float *depth, const double *xf, const double *yf,
                     float *slab, int nblocks);
int bp_fused_tiles(const BpArgs &a);
int bp_state_words(const BpArgs &a);
int launch_bp_fused(hipStream_t s, const Cam &cam, const BpArgs &a, const PoseD &p, const float *depth, const uint8_t *bgr,
                    const double *xf, const double *yf, unsigned long long *state, float *xyz, uint8_t *rgb, unsigned long long cap,
                    unsigned long long *total_out, bool force_dynamic = false);
// centroid
int launch_centroid_batch(hipStream_t s, const Cam &cam, const Grid &g, int n, const CenFrame *host, CenFrame *dev, const double *xf, const double *yf,
                          unsigned long long *grid, unsigned long long *counters);
int launch_centroid_points(hipStream_t s, const Grid &g, const float *xyz, const uint8_t *rgb, long long n,
                           unsigned long long *grid, unsigned long long *counters);
int launch_bounds(hipStream_t s, const float *xyz, long long n, float *slab, int nblocks);
// tsdf
size_t tsdf_batch_scratch_bytes(const Cam &cam, const Grid &g, int max_frames);
void tsdf_batch_scratch_zero_range(const Cam &cam, const Grid &g, int max_frames, size_t *off, size_t *bytes);
int launch_tsdf_prepare(hipStream_t s, const Cam &cam, const Grid &g, int n, int max_frames, const PoseF *p, const Frustum &fr,
                        const void *const *depth, bool depth_u16, const float *scale, float mind, float maxd, void *scratch, unsigned *free_cnt,
                        bool classify_only = false);
int launch_centroid_mark(hipStream_t s, const Cam &cam, const Grid &g, const BpArgs &a, const PoseD &p, const float *depth, const double *xf, const double *yf);
#ifdef TL3D_EXPERIMENTS
void tsdf_debug_print_spans(int nwaves);
#endif
int launch_tsdf_update(hipStream_t s, const Cam &cam, const Grid &g, int n, int max_frames, bool depth_u16, float mind, float maxd, int2 *grid,
                       void *scratch, unsigned long long *counters, bool count, int max_blocks, int xcd_group);
int launch_fold_free(hipStream_t s, const Grid &g, int2 *grid, unsigned *free_cnt);
// normals + icp
int launch_nmap_rowmajor(hipStream_t s, const Cam &cam, const float4 *nmap, float4 *out);
int launch_normals(hipStream_t s, const Cam &cam, const float *depth, float scale, float mind, float maxd, float jump, int radius, float *sdepth,
                   float4 *nmap);
int launch_icp_batch(hipStream_t s, const Cam &cam, const IcpBatchArgs &a);
int launch_icp_iteration(hipStream_t s, const Cam &cam, const IcpRun *run, int final_pass, double *slab, IcpState *state, int nblocks,
                         unsigned *ticket);
// extraction
int launch_extract_count(hipStream_t s, const Grid &g, int mode, int min_count, int min_weight, double max_abs,
                         const int2 *tsdf, const unsigned long long *cen, unsigned *block_counts, int nblocks);
int launch_extract_write(hipStream_t s, const Grid &g, int mode, int min_count, int min_weight, double max_abs,
                         const int2 *tsdf, const unsigned long long *cen, const unsigned long long *offsets, int nblocks,
                         float *xyz, uint8_t *rgb, unsigned long long cap);
// grids
int launch_max_weight(hipStream_t s, const Grid &g, const int2 *pool, int *d_out);
int launch_max_weight_dense(hipStream_t s, const int2 *grid, size_t nvox, int *d_out);
int launch_touched_bricks(hipStream_t s, const Grid &g, const int2 *tsdf, const unsigned long long *cen, unsigned nbricks, unsigned char *map, bool sub = false);
int launch_brick_rows(hipStream_t s, const Grid &g, int mode, bool is_tsdf, void *pool, const unsigned *idx, long long n, void *rows, bool add_free, bool sub = false);
int launch_iota(hipStream_t s, unsigned *t, unsigned n);
int probe_hw_queues(int n_streams, double spin_ms, double *elapsed_ms);
int launch_add_i32(hipStream_t s, int *dst, const int *src, size_t n);
int launch_add_u64(hipStream_t s, unsigned long long *dst, const unsigned long long *src, size_t n);
// outlier filter
int sor_run(tl3d_ctx *ctx, const float *xyz_dev, long long n, int k, double std_ratio, double cell, uint8_t *keep_dev, long long *kept);

constexpr int EXTRACT_CHUNK = 2048;   // records per block in the extraction kernels

}  // namespace tl3d
