// tl3d_api.hip -- the extern "C" surface declared in include/tl3d.h: context, frame slots, launch glue.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>
#include <math.h>

#include <mutex>
#include <new>
#include <vector>

#include "tl3d_internal.h"

using namespace tl3d;

static thread_local char g_err[512] = "";

namespace tl3d {
int set_err(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace tl3d

// Tuning knobs (TL3D_PREP_STREAMS, TL3D_TSDF_BATCH, ...) are read from the environment only by the experiments flavour of the
// library (csrc/build.sh with TL3D_FLAVOUR=experiments -> libtl3d_exp.so); the shipped library ignores them.
static int env_int(const char *name, int dflt) {
#ifdef TL3D_EXPERIMENTS
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

#define REQUIRE(cond, code, ...)                           \
    do {                                                   \
        if (!(cond)) return set_err((code), __VA_ARGS__);  \
    } while (0)

// The five RCCL entry points the merge needs, resolved at run time; enum values from rccl.h (ncclInt32 = 2, ncclInt64 = 4,
// ncclUint64 = 5, ncclSum = 0).
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, const void *, int) = nullptr;      // ncclUniqueId is passed BY VALUE in C: see rccl_init_rank
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;            // optional
    const char *(*GetErrorString)(int) = nullptr;
};
struct RcclId { char bytes[TL3D_RCCL_ID_BYTES]; };
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.lib) return TL3D_OK;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return set_err(TL3D_E_STATE, "RCCL (librccl.so) could not be loaded: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(void *))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void **, int, const void *, int))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void *))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
    g_rccl.CommAbort = (int (*)(void *))dlsym(g_rccl.lib, "ncclCommAbort");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
        g_rccl.lib = nullptr;
        return set_err(TL3D_E_STATE, "librccl.so lacks the expected entry points");
    }
    return TL3D_OK;
}
const char *rccl_msg(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"; }
}  // namespace


// hipMalloc for the large allocations of a grid: if the device is out of memory while the frame-slab cache (pool_release) holds
// some, give that back and try once more
static hipError_t device_malloc_big(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    (void)tl3d_release_cached_memory();
    e = hipMalloc(p, bytes);
    if (e != hipSuccess) (void)hipGetLastError();
    return e;
}

static bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

static PoseF make_pose_f(const double R[9], const double t[3]) {
    PoseF p;
    for (int i = 0; i < 9; ++i) p.r[i] = (float)R[i];
    for (int i = 0; i < 3; ++i) p.t[i] = (float)t[i];
    return p;
}

static PoseD make_pose_d(const double R[9], const double t[3], bool no_pose) {
    PoseD p;
    memset(&p, 0, sizeof(p));
    if (no_pose) {
        p.r[0] = p.r[4] = p.r[8] = 1.0;
        return p;
    }
    for (int i = 0; i < 9; ++i) p.r[i] = R[i];
    for (int i = 0; i < 3; ++i) p.ct[i] = (R[0 + i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2];   // (R^T t)_i, D2R:376
    return p;
}

static Frustum make_frustum(const Cam &c) {
    // inside: aL <= x/z <= aR, aT <= y/z <= aB, widened by one pixel on every side
    const double aL = (-1.5 - c.cxd) / c.fxd, aR = ((double)c.W + 0.5 - c.cxd) / c.fxd;
    const double aT = (-1.5 - c.cyd) / c.fyd, aB = ((double)c.H + 0.5 - c.cyd) / c.fyd;
    Frustum f;
    double n;
    n = sqrt(1.0 + aL * aL); f.lx = (float)(1.0 / n);  f.lz = (float)(-aL / n);
    n = sqrt(1.0 + aR * aR); f.rx = (float)(-1.0 / n); f.rz = (float)(aR / n);
    n = sqrt(1.0 + aT * aT); f.ty = (float)(1.0 / n);  f.tz = (float)(-aT / n);
    n = sqrt(1.0 + aB * aB); f.by = (float)(-1.0 / n); f.bz = (float)(aB / n);
    return f;
}

static int ensure_scratch_blocks(tl3d_ctx *ctx, size_t nblocks) {
    if (nblocks <= ctx->scratch_blocks) return TL3D_OK;
    if (ctx->block_counts) (void)hipFree(ctx->block_counts);
    if (ctx->block_offsets) (void)hipFree(ctx->block_offsets);
    ctx->block_counts = nullptr;
    ctx->block_offsets = nullptr;
    ctx->scratch_blocks = 0;
    if (hipMalloc(&ctx->block_counts, nblocks * sizeof(unsigned)) != hipSuccess) return set_err(TL3D_E_NOMEM, "scratch alloc failed");
    if (hipMalloc(&ctx->block_offsets, nblocks * sizeof(unsigned long long)) != hipSuccess) return set_err(TL3D_E_NOMEM, "scratch alloc failed");
    ctx->scratch_blocks = nblocks;
    return TL3D_OK;
}

static int check_slot(tl3d_ctx *ctx, int slot, bool need_loaded) {
    REQUIRE(ctx != nullptr, TL3D_E_INVALID, "null ctx");
    REQUIRE(slot >= 0 && slot < ctx->cfg.n_slots, TL3D_E_INVALID, "slot %d out of range [0,%d)", slot, ctx->cfg.n_slots);
    if (need_loaded) REQUIRE(ctx->slots[slot].loaded, TL3D_E_STATE, "slot %d holds no frame", slot);
    return TL3D_OK;
}

static int upload_impl(tl3d_ctx *ctx, int slot, const void *depth_hd, int depth_kind, const uint8_t *bgr_hd, bool wait);
static void icp_lane_free(tl3d_ctx::IcpLane &ln);
static void icp_batch_free(tl3d_ctx::IcpBatch &b);
static int bp_check_error(tl3d_ctx *ctx, unsigned long long err);
static int flush_updates(tl3d_ctx *ctx);
static int flush_centroid(tl3d_ctx *ctx);
// every call that reads or writes the TSDF grid, re-uses a frame slot, synchronises or time-stamps first issues the deferred updates
#define FLUSH_UPDATES(ctx_)                      \
    do {                                         \
        const int rc_ = flush_updates(ctx_);     \
        if (rc_) return rc_;                     \
    } while (0)

// ICP lanes read slots[src].depth and slots[tgt].nmap on their own streams.  A call that rewrites one of those buffers on
// the main stream (a new upload into the slot, a rebuilt normal map) is ordered behind every uncollected run that reads it.
static int order_after_lanes(tl3d_ctx *ctx, int slot, bool rewrites_depth, bool rewrites_nmap) {
    for (int l = 0; l < TL3D_ICP_LANES; ++l) {
        tl3d_ctx::IcpLane &ln = ctx->icp_lanes[l];
        if (!ln.stream || !ln.busy || !ln.ev_done) continue;
        if ((rewrites_depth && ln.src_slot == slot) || (rewrites_nmap && ln.tgt_slot == slot))
            TL3D_HIP(hipStreamWaitEvent(ctx->stream, ln.ev_done, 0));
    }
    if (ctx->icp_batch.busy && ctx->icp_batch.ev_done)      // (coarse: a batch may read any slot)
        TL3D_HIP(hipStreamWaitEvent(ctx->stream, ctx->icp_batch.ev_done, 0));
    return TL3D_OK;
}

// exact largest voxel weight of the TSDF channel (blocks; the deferred updates must have been issued)
static int measure_max_weight(tl3d_ctx *ctx, const int2 *grid, long long *out) {
    if (!ctx->d_maxw && hipMalloc(&ctx->d_maxw, sizeof(int)) != hipSuccess) return set_err(TL3D_E_NOMEM, "alloc failed");
    // the context's own channel: the pool slots in use + the counts of bricks without records; anything else: a dense array
    int rc = grid == ctx->tsdf ? launch_max_weight(ctx->stream, ctx->grid, ctx->tsdf, ctx->d_maxw)
                               : launch_max_weight_dense(ctx->stream, grid, ctx->nvox, ctx->d_maxw);
    if (rc) return rc;
    int h = 0;
    TL3D_HIP(hipMemcpyAsync(&h, ctx->d_maxw, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    TL3D_HIP(hipStreamSynchronize(ctx->stream));
    *out = (long long)h;
    return TL3D_OK;
}

// The free-space counters are incremented by the classification kernels on the prep streams and cleared / folded on the main
// stream.  Every main-stream write to them is followed by this marker; the next prep chains wait for it (flush_updates),
// so a count can never land in front of a clear that was issued before it (a reset followed at once by integrate calls).
static int mark_free_cnt_write(tl3d_ctx *ctx) {
    if (!ctx->ev_free && hipEventCreateWithFlags(&ctx->ev_free, hipEventDisableTiming) != hipSuccess) return set_err(TL3D_E_HIP, "event create failed");
    TL3D_HIP(hipEventRecord(ctx->ev_free, ctx->stream));
    ctx->ev_free_recorded = true;
    return TL3D_OK;
}

// pending free-space counts -> records (main stream; the deferred updates must have been issued: their classification
// kernels, which increment the counters on the side streams, are ordered before the main stream by flush_updates)
static int fold_free(tl3d_ctx *ctx) {
    if (!ctx->free_cnt || !ctx->free_dirty) return TL3D_OK;
    ctx->free_dirty = false;
    const int rc = launch_fold_free(ctx->stream, ctx->grid, ctx->tsdf, ctx->free_cnt);
    if (rc) return rc;
    return mark_free_cnt_write(ctx);
}
#define FLUSH_AND_FOLD(ctx_)                     \
    do {                                         \
        FLUSH_UPDATES(ctx_);                     \
        const int frc_ = fold_free(ctx_);        \
        if (frc_) return frc_;                   \
    } while (0)

static int validate_grid(const tl3d_config *cfg) {
    REQUIRE((cfg->channels & ~(TL3D_CH_TSDF | TL3D_CH_CENTROID)) == 0 && cfg->channels != 0, TL3D_E_INVALID, "bad channel bits 0x%x", cfg->channels);
    REQUIRE(cfg->nx > 0 && cfg->ny > 0 && cfg->nz > 0 && cfg->nx % TL3D_BRICK == 0 && cfg->ny % TL3D_BRICK == 0 &&
                cfg->nz % TL3D_BRICK == 0,
            TL3D_E_INVALID, "grid dims %dx%dx%d must be positive multiples of %d", cfg->nx, cfg->ny, cfg->nz, TL3D_BRICK);
    REQUIRE((double)cfg->nx * cfg->ny * cfg->nz <= 4294967296.0, TL3D_E_INVALID, "grid larger than 2^32 voxels");
    REQUIRE(cfg->voxel_size > 0, TL3D_E_INVALID, "voxel_size must be positive");
    if (cfg->channels & TL3D_CH_TSDF) REQUIRE(cfg->sdf_trunc > 0, TL3D_E_INVALID, "sdf_trunc must be positive");
    REQUIRE(cfg->pool_bricks_tsdf >= 0 && cfg->pool_bricks_centroid >= 0, TL3D_E_INVALID, "negative brick pool size");
    for (int a = 0; a < 3; ++a)
        REQUIRE(cfg->voxel_offset[a] >= 0 && cfg->voxel_offset[a] % TL3D_BRICK == 0 && cfg->voxel_offset[a] < (1ll << 40), TL3D_E_INVALID,
                "voxel_offset must be non-negative multiples of %d", TL3D_BRICK);
    if (cfg->channels & TL3D_CH_TSDF)
        REQUIRE(cfg->voxel_offset[0] == 0 && cfg->voxel_offset[1] == 0 && cfg->voxel_offset[2] == 0, TL3D_E_INVALID,
                "a grid with a TSDF channel cannot be a block of a larger lattice (voxel_offset must be 0)");
    if (cfg->pool_bricks_tsdf > 0 || cfg->pool_bricks_centroid > 0)
        REQUIRE(cfg->ext_tsdf == nullptr && cfg->ext_centroid == nullptr, TL3D_E_INVALID, "a sparse grid (pool_bricks_*) cannot live in caller-owned dense memory");
    return TL3D_OK;
}

// brick tables and cursors of a fresh (or reset) grid, on the main stream: a dense channel's table is the identity and its
// cursor sits at the end of the pool; a sparse channel starts empty
static int reset_brick_tables(tl3d_ctx *ctx) {
    Grid &g = ctx->grid;
    const size_t nbr = ctx->nvox >> 9;
    unsigned cur[4] = {0u, 0u, 0u, 0u};
    if (g.tsdf_cap >= nbr) {
        const int rc = launch_iota(ctx->stream, g.tsdf_tab, (unsigned)nbr);
        if (rc) return rc;
        cur[0] = (unsigned)nbr;
    } else {
        TL3D_HIP(hipMemsetAsync(g.tsdf_tab, 0xff, nbr * sizeof(unsigned), ctx->stream));
    }
    if (g.cen_cap >= nbr) {
        const int rc = launch_iota(ctx->stream, g.cen_tab, (unsigned)nbr);
        if (rc) return rc;
        cur[2] = (unsigned)nbr;
    } else {
        TL3D_HIP(hipMemsetAsync(g.cen_tab, 0xff, nbr * sizeof(unsigned), ctx->stream));
    }
    TL3D_HIP(hipMemcpyAsync(g.cursors, cur, sizeof(cur), hipMemcpyHostToDevice, ctx->stream));
    TL3D_HIP(hipStreamSynchronize(ctx->stream));           // `cur` is on this stack; the prep streams do not wait for the main stream
    return TL3D_OK;
}

static void grid_geometry(Grid &g, const tl3d_config *cfg) {
    g.nx = cfg->nx; g.ny = cfg->ny; g.nz = cfg->nz;
    g.nbx = cfg->nx / 8; g.nby = cfg->ny / 8; g.nbz = cfg->nz / 8;
    g.oxd = cfg->origin[0]; g.oyd = cfg->origin[1]; g.ozd = cfg->origin[2]; g.vsd = cfg->voxel_size;
    g.offx = (double)cfg->voxel_offset[0]; g.offy = (double)cfg->voxel_offset[1]; g.offz = (double)cfg->voxel_offset[2];
    g.ox = (float)g.oxd; g.oy = (float)g.oyd; g.oz = (float)g.ozd; g.vs = (float)g.vsd;
    g.trunc = (float)cfg->sdf_trunc;
    g.inv_trunc = (cfg->channels & TL3D_CH_TSDF) ? 1.0f / g.trunc : 0.0f;
}

// allocates (or borrows) the grid channels of cfg and the TSDF side stream / scratch; ctx->stream must exist
static int alloc_grid(tl3d_ctx *ctx, const tl3d_config *cfg) {
    Grid &g = ctx->grid;
    grid_geometry(g, cfg);
    ctx->nvox = (size_t)g.nx * g.ny * g.nz;
    ctx->tsdf_w_upper = 0;
    ctx->tsdf_w_unknown = false;
    {   // brick tables: identity for a dense channel, empty for a sparse one (slots handed out on first touch)
        const size_t nbr = ctx->nvox >> 9;
        ctx->sparse = cfg->pool_bricks_tsdf > 0 || cfg->pool_bricks_centroid > 0;
        g.tsdf_cap = (unsigned)((cfg->pool_bricks_tsdf > 0 && (size_t)cfg->pool_bricks_tsdf < nbr) ? (size_t)cfg->pool_bricks_tsdf : nbr);
        g.cen_cap = (unsigned)((cfg->pool_bricks_centroid > 0 && (size_t)cfg->pool_bricks_centroid < nbr) ? (size_t)cfg->pool_bricks_centroid : nbr);
        if (hipMalloc(&ctx->brick_tabs, (2 * nbr + 64) * sizeof(unsigned)) != hipSuccess) return set_err(TL3D_E_NOMEM, "brick table alloc failed");
        g.tsdf_tab = ctx->brick_tabs;
        g.cen_tab = ctx->brick_tabs + nbr;
        g.cursors = ctx->brick_tabs + 2 * nbr;
        g.free_cnt = nullptr;
        const int trc = reset_brick_tables(ctx);
        if (trc) return trc;
    }
    if (cfg->channels & TL3D_CH_TSDF) {
        if (cfg->ext_tsdf) {
            ctx->tsdf = (int2 *)cfg->ext_tsdf;
            ctx->tsdf_w_unknown = true;                 // caller-owned memory: contents unknown
        } else {
            const size_t pool_b = (size_t)g.tsdf_cap << 12;
            if (device_malloc_big((void **)&ctx->tsdf, pool_b) != hipSuccess) return set_err(TL3D_E_NOMEM, "TSDF record pool alloc (%zu B) failed", pool_b);
            ctx->own_tsdf = true;
            if (hipMemsetAsync(ctx->tsdf, 0, pool_b, ctx->stream) != hipSuccess) return set_err(TL3D_E_HIP, "memset failed");
        }
        {   // Side streams for the prep chains of the batches (descriptors, tiles, pyramid, brick classes, cost order, sub-brick
            // classes): the chains of the next batches run beside the update kernel of batch k.  Three streams, taken in turn, and
            // TSDF_SCRATCHES = 4 batch scratches.  Beside an update kernel a chain's big kernels stretch over a whole update period each,
            // so a chain is three periods long and three are in flight; a fourth stream with six scratches was slower (74.1 k against
            // 76.5 k frames/s: the step is bound by the sum of the work), two streams too (77.3 k against 79.2 k).
            // (A higher stream priority for the chains changes nothing measurable: 55.4k against 55.3k frames/s.)
            ctx->n_prep_streams = env_int("TL3D_PREP_STREAMS", 3);
            if (ctx->n_prep_streams < 1) ctx->n_prep_streams = 1;
            if (ctx->n_prep_streams > 4) ctx->n_prep_streams = 4;
            for (int q = 0; q < ctx->n_prep_streams; ++q)
                if (!ctx->prep_stream[q] && hipStreamCreateWithFlags(&ctx->prep_stream[q], hipStreamNonBlocking) != hipSuccess)
                    return set_err(TL3D_E_HIP, "stream create failed");
        }
        {
            const size_t nbr = (size_t)g.nbx * g.nby * g.nbz;
            ctx->free_dirty = false;
            if (hipMalloc(&ctx->free_cnt, nbr * sizeof(unsigned)) != hipSuccess) return set_err(TL3D_E_NOMEM, "free-space counter alloc failed");
            g.free_cnt = ctx->free_cnt;
            if (hipMemsetAsync(ctx->free_cnt, 0, nbr * sizeof(unsigned), ctx->stream) != hipSuccess) return set_err(TL3D_E_HIP, "memset failed");
            const int mrc = mark_free_cnt_write(ctx);
            if (mrc) return mrc;
        }
        ctx->tsdf_use_u16 = env_int("TL3D_U16_GATHER", 1) != 0;
        ctx->tsdf_pairing = env_int("TL3D_TSDF_PAIR", 1) != 0;
        ctx->tsdf_batch = env_int("TL3D_TSDF_BATCH", 32);
        if (ctx->tsdf_batch < 1) ctx->tsdf_batch = 1;
        if (ctx->tsdf_batch > TL3D_TSDF_MAXBATCH) ctx->tsdf_batch = TL3D_TSDF_MAXBATCH;
        {   // Workgroups of the update kernel (the pair form: ~56 vector registers, eight waves per SIMD, tasks by ticket): 12 per CU.
            // Round 4, same box, interleaved: 1536 / 2048 / 3072 workgroups 74.8 / 76.5 / 77.8 k frames/s; 4096 and 6144 inside
            // the spread of 3072 (77.9 / 78.2 k against 77.5 k).  (Round 3's frame-major kernel, 128 registers, wanted three per CU.)
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus < 1) cus = 256;
            ctx->tsdf_max_blocks = env_int("TL3D_UPDATE_BLOCKS", 12 * cus);
        }
        if (ctx->tsdf_max_blocks < 8) ctx->tsdf_max_blocks = 8;
        ctx->tsdf_xcd_group = env_int("TL3D_XCD_GROUP", 1);       // consecutive list entries per ticket chunk (1: best balance; 16: 41.8k against 43.8k frames/s)
        ctx->tsdf_single_stream = env_int("TL3D_SINGLE_STREAM", 0) != 0;
        {   // the scratch of the batches in flight (TSDF_SCRATCHES): one allocation; the frame masks start out zero (the update re-arms them)
            const size_t each = (tsdf_batch_scratch_bytes(ctx->cam, g, ctx->tsdf_batch) + 255) & ~(size_t)255;
            void *slab = nullptr;
            if (device_malloc_big(&slab, each * TSDF_SCRATCHES) != hipSuccess) return set_err(TL3D_E_NOMEM, "TSDF scratch alloc (%zu B) failed", each * TSDF_SCRATCHES);
            ctx->tsdf_scratch_slab = slab;
            size_t zoff = 0, zbytes = 0;
            tsdf_batch_scratch_zero_range(ctx->cam, g, ctx->tsdf_batch, &zoff, &zbytes);
            for (int h = 0; h < TSDF_SCRATCHES; ++h) {
                ctx->tsdf_scratch[h] = (char *)slab + each * (size_t)h;
                if (hipMemsetAsync((char *)ctx->tsdf_scratch[h] + zoff, 0, zbytes, ctx->stream) != hipSuccess) return set_err(TL3D_E_HIP, "memset failed");
                if (!ctx->ev_prep[h] && hipEventCreateWithFlags(&ctx->ev_prep[h], hipEventDisableTiming) != hipSuccess) return set_err(TL3D_E_HIP, "event create failed");
            }
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) return set_err(TL3D_E_HIP, "sync failed");     // the prep streams do not wait for this memset
        }
        for (int h = 0; h < TSDF_SCRATCHES; ++h)
            if (hipEventCreateWithFlags(&ctx->ev_upd[h], hipEventDisableTiming) != hipSuccess) return set_err(TL3D_E_HIP, "event create failed");
    }
    if (cfg->channels & TL3D_CH_CENTROID) {
        if (cfg->ext_centroid) {
            ctx->centroid = (unsigned long long *)cfg->ext_centroid;
        } else {
            const size_t pool_b = (size_t)g.cen_cap << 14;
            if (device_malloc_big((void **)&ctx->centroid, pool_b) != hipSuccess) return set_err(TL3D_E_NOMEM, "centroid record pool alloc (%zu B) failed", pool_b);
            ctx->own_centroid = true;
            if (hipMemsetAsync(ctx->centroid, 0, pool_b, ctx->stream) != hipSuccess) return set_err(TL3D_E_HIP, "memset failed");
        }
    }
    ctx->cfg.channels = cfg->channels;
    ctx->cfg.nx = cfg->nx; ctx->cfg.ny = cfg->ny; ctx->cfg.nz = cfg->nz;
    for (int i = 0; i < 3; ++i) ctx->cfg.origin[i] = cfg->origin[i];
    ctx->cfg.voxel_size = cfg->voxel_size;
    ctx->cfg.sdf_trunc = cfg->sdf_trunc;
    return TL3D_OK;
}

extern "C" {

const char *tl3d_last_error(void) { return g_err; }
int tl3d_version(void) { return TL3D_ABI_VERSION; }

int tl3d_runtime_info(int *hip_compiled, int *hip_runtime, int *hip_driver) {
    REQUIRE(hip_compiled && hip_runtime && hip_driver, TL3D_E_INVALID, "null out pointer");
    *hip_compiled = HIP_VERSION;
    *hip_runtime = *hip_driver = 0;
    if (hipRuntimeGetVersion(hip_runtime) != hipSuccess) (void)hipGetLastError();
    if (hipDriverGetVersion(hip_driver) != hipSuccess) (void)hipGetLastError();
    return TL3D_OK;
}

int tl3d_probe_hw_queues(int device, int n_streams, double spin_ms, double *elapsed_ms) {
    REQUIRE(elapsed_ms != nullptr, TL3D_E_INVALID, "null out pointer");
    REQUIRE(n_streams >= 1 && n_streams <= 256 && spin_ms > 0 && spin_ms <= 50, TL3D_E_INVALID, "bad probe arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return set_err(TL3D_E_NODEVICE, "no HIP device visible");
    }
    REQUIRE(device >= 0 && device < ndev, TL3D_E_INVALID, "device %d out of range [0,%d)", device, ndev);
    TL3D_HIP(hipSetDevice(device));
    return probe_hw_queues(n_streams, spin_ms, elapsed_ms);
}

int tl3d_device_count(int *n) {
    REQUIRE(n != nullptr, TL3D_E_INVALID, "null out pointer");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return TL3D_OK;
}

int tl3d_create(const tl3d_config *cfg, int device, tl3d_ctx **out) {
    REQUIRE(cfg && out, TL3D_E_INVALID, "null argument");
    REQUIRE(cfg->abi_version == TL3D_ABI_VERSION, TL3D_E_INVALID, "ABI version %d != %d", cfg->abi_version, TL3D_ABI_VERSION);
    REQUIRE(cfg->width > 0 && cfg->height > 0 && cfg->width <= 32768 && cfg->height <= 32768, TL3D_E_INVALID,
            "bad frame size %dx%d", cfg->width, cfg->height);
    REQUIRE(cfg->fx > 0 && cfg->fy > 0, TL3D_E_INVALID, "focal lengths must be positive");
    REQUIRE(cfg->n_slots >= 1 && cfg->n_slots <= (1 << 20), TL3D_E_INVALID, "n_slots %d out of range", cfg->n_slots);
    if (cfg->channels) {
        const int vrc = validate_grid(cfg);
        if (vrc) return vrc;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return set_err(TL3D_E_NODEVICE, "no HIP device visible (libtl3d has no CPU fallback)");
    }
    REQUIRE(device >= 0 && device < ndev, TL3D_E_INVALID, "device %d out of range [0,%d)", device, ndev);
    TL3D_HIP(hipSetDevice(device));

    tl3d_ctx *ctx = new (std::nothrow) tl3d_ctx();
    REQUIRE(ctx != nullptr, TL3D_E_NOMEM, "host allocation failed");
    memset(ctx, 0, sizeof(*ctx));
    ctx->cfg = *cfg;
    ctx->device = device;
    ctx->slots = new (std::nothrow) Slot[cfg->n_slots];
    if (!ctx->slots) { delete ctx; return set_err(TL3D_E_NOMEM, "host allocation failed"); }
    {
        const size_t npx = (size_t)cfg->width * cfg->height;
        const size_t npm = pm_pixels(cfg->width, cfg->height);      // normal maps and averaged depth: phase-major rows of 4 * ceil(W / 4) entries
        const size_t bytes[5] = {npx * sizeof(float), npx * sizeof(uint16_t), npx * 3, npm * sizeof(float4), npm * sizeof(float)};
        FramePool *pools[5] = {&ctx->pool_depth, &ctx->pool_u16, &ctx->pool_bgr, &ctx->pool_nmap, &ctx->pool_sdepth};
        for (int k = 0; k < 5; ++k) {
            FramePool &fp = *pools[k];
            fp.block = (bytes[k] + 16 + 255) & ~(size_t)255;      // 16 B of slack: a pixel's 3 colour bytes are read as one 4-byte word
            fp.remaining = cfg->n_slots;
            fp.max_slabs = (cfg->n_slots + FRAME_SLAB_BLOCKS - 1) / FRAME_SLAB_BLOCKS;
            fp.slabs = (void **)calloc((size_t)fp.max_slabs, sizeof(void *));
            fp.slab_bytes = (size_t *)calloc((size_t)fp.max_slabs, sizeof(size_t));
            fp.device = device;
            if (!fp.slabs || !fp.slab_bytes) { (void)tl3d_destroy(ctx); return set_err(TL3D_E_NOMEM, "host allocation failed"); }
        }
    }

    Cam &c = ctx->cam;
    c.W = cfg->width; c.H = cfg->height;
    c.fxd = cfg->fx; c.fyd = cfg->fy; c.cxd = cfg->cx; c.cyd = cfg->cy;
    c.fx = (float)cfg->fx; c.fy = (float)cfg->fy; c.cx = (float)cfg->cx; c.cy = (float)cfg->cy;
    Grid &g = ctx->grid;
    memset(&g, 0, sizeof(g));
    int rc = TL3D_OK;
    auto fail = [&](int code) { tl3d_destroy(ctx); return code; };

    if (cfg->stream) {
        ctx->stream = (hipStream_t)cfg->stream;
        ctx->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return fail(set_err(TL3D_E_HIP, "stream create failed"));
        ctx->own_stream = true;
    }
    if (cfg->channels) {
        const int grc = alloc_grid(ctx, cfg);
        if (grc) return fail(grc);
    }
    if (hipMalloc(&ctx->d_counters, 16 * sizeof(unsigned long long)) != hipSuccess) return fail(set_err(TL3D_E_NOMEM, "alloc failed"));
    if (hipMemsetAsync(ctx->d_counters, 0, 16 * sizeof(unsigned long long), ctx->stream) != hipSuccess) return fail(set_err(TL3D_E_HIP, "memset failed"));
    if (hipMalloc(&ctx->d_cen_counters, 256 * 8 * sizeof(unsigned long long)) != hipSuccess) return fail(set_err(TL3D_E_NOMEM, "alloc failed"));
    if (hipMemsetAsync(ctx->d_cen_counters, 0, 256 * 8 * sizeof(unsigned long long), ctx->stream) != hipSuccess) return fail(set_err(TL3D_E_HIP, "memset failed"));
    if (hipMalloc(&ctx->bounds_slab, 1024 * 6 * sizeof(float)) != hipSuccess) return fail(set_err(TL3D_E_NOMEM, "alloc failed"));
    {   // projection-factor tables: the two cached maps of the reference (D2R:287-295) are separable, so W + H doubles hold
        // them; computed on the host with the same fp64 expression the kernels used per pixel
        std::vector<double> f((size_t)c.W + c.H);
        for (int u = 0; u < c.W; ++u) f[u] = ((double)u - c.cxd) / c.fxd;
        for (int v = 0; v < c.H; ++v) f[(size_t)c.W + v] = ((double)v - c.cyd) / c.fyd;
        if (hipMalloc(&ctx->bp_factors, f.size() * sizeof(double)) != hipSuccess) return fail(set_err(TL3D_E_NOMEM, "alloc failed"));
        if (hipMemcpy(ctx->bp_factors, f.data(), f.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(set_err(TL3D_E_HIP, "copy failed"));
    }
    for (int i = 0; i < 2; ++i)
        if (hipEventCreate(&ctx->ev[i]) != hipSuccess) return fail(set_err(TL3D_E_HIP, "event create failed"));
    (void)rc;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(set_err(TL3D_E_HIP, "sync failed"));
    *out = ctx;
    return TL3D_OK;
}

// One block of a frame pool (nullptr when the device is out of memory).  A slab is sized for the slots still without a
// buffer of this kind, at most FRAME_SLAB_BLOCKS of them, so a context never holds more than it was created for.
// Frame slabs outlive their context in a PROCESS-WIDE CACHE (by device and size): a process that reconstructs one sequence after
// another creates a context per sequence, and memory the driver has just taken back comes out of hipMalloc at ~26 GB/s -- the
// second reconstruct() of a 1000-frame 720p sequence spent 0.7 s re-allocating the 25 GB its predecessor had freed, five times its
// whole first run (tools/run_config.py --config 4, round 4).  A released slab is kept (nothing on the device refers to it any
// more: tl3d_destroy has waited for the device) and handed to the next pool that asks for exactly that size; at most
// SLAB_CACHE_LIMIT bytes stay cached, the rest is freed at once; tl3d_release_cached_memory() frees everything (a host that
// shares the GPU with another allocator calls it when it is done with a batch of sequences).
namespace {
struct CachedSlab { int device; size_t bytes; void *p; };
std::mutex g_slab_mutex;
std::vector<CachedSlab> g_slab_cache;
size_t g_slab_cached = 0;
constexpr size_t SLAB_CACHE_LIMIT = (size_t)64 << 30;

void *slab_alloc(int device, size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(g_slab_mutex);
        for (size_t i = g_slab_cache.size(); i-- > 0;)
            if (g_slab_cache[i].device == device && g_slab_cache[i].bytes == bytes) {
                void *p = g_slab_cache[i].p;
                g_slab_cached -= bytes;
                g_slab_cache.erase(g_slab_cache.begin() + (long)i);
                return p;
            }
    }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) == hipSuccess) return p;
    (void)hipGetLastError();
    // out of memory with slabs of other sizes in the cache: give them back and try once more
    if (tl3d_release_cached_memory() == TL3D_OK && hipMalloc(&p, bytes) == hipSuccess) return p;
    (void)hipGetLastError();
    return nullptr;
}

void slab_free(int device, size_t bytes, void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_slab_mutex);
        if (g_slab_cached + bytes <= SLAB_CACHE_LIMIT) {
            g_slab_cache.push_back({device, bytes, p});
            g_slab_cached += bytes;
            return;
        }
    }
    (void)hipFree(p);
}
}  // namespace

static void *pool_take(FramePool &fp) {
    if (fp.cur_left == 0) {
        if (fp.remaining <= 0 || fp.n_slabs >= fp.max_slabs) return nullptr;
        const int nb = fp.remaining < FRAME_SLAB_BLOCKS ? fp.remaining : FRAME_SLAB_BLOCKS;
        void *p = slab_alloc(fp.device, fp.block * (size_t)nb);
        if (!p) return nullptr;
        fp.slab_bytes[fp.n_slabs] = fp.block * (size_t)nb;
        fp.slabs[fp.n_slabs++] = p;
        fp.cur = (char *)p;
        fp.cur_left = nb;
    }
    void *out = fp.cur;
    fp.cur += fp.block;
    --fp.cur_left;
    --fp.remaining;
    return out;
}

// (the caller has waited for the device: nothing in flight refers to the slabs)
static void pool_release(FramePool &fp) {
    for (int i = 0; i < fp.n_slabs; ++i) slab_free(fp.device, fp.slab_bytes[i], fp.slabs[i]);
    free(fp.slabs);
    free(fp.slab_bytes);
    fp.slabs = nullptr;
    fp.slab_bytes = nullptr;
    fp.n_slabs = fp.max_slabs = 0;
}

int tl3d_release_cached_memory(void) {
    std::vector<CachedSlab> take;
    {
        std::lock_guard<std::mutex> lk(g_slab_mutex);
        take.swap(g_slab_cache);
        g_slab_cached = 0;
    }
    int dev0 = -1;
    (void)hipGetDevice(&dev0);
    for (const CachedSlab &c : take) {
        (void)hipSetDevice(c.device);
        (void)hipFree(c.p);
    }
    if (dev0 >= 0) (void)hipSetDevice(dev0);
    return TL3D_OK;
}

int tl3d_destroy(tl3d_ctx *ctx) {
    if (!ctx) return TL3D_OK;
    (void)hipSetDevice(ctx->device);
    (void)flush_updates(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipDeviceSynchronize();                           // the frame slabs go to the process-wide cache, not back to the driver: nothing may still read them
    if (ctx->slots) {
        for (int i = 0; i < ctx->cfg.n_slots; ++i) {
            if (ctx->slots[i].ev_upload) (void)hipEventDestroy(ctx->slots[i].ev_upload);
            if (ctx->slots[i].ev_normals) (void)hipEventDestroy(ctx->slots[i].ev_normals);
        }
        delete[] ctx->slots;
    }
    pool_release(ctx->pool_depth);
    pool_release(ctx->pool_u16);
    pool_release(ctx->pool_bgr);
    pool_release(ctx->pool_nmap);
    pool_release(ctx->pool_sdepth);
    if (ctx->brick_tabs) (void)hipFree(ctx->brick_tabs);
    if (ctx->own_tsdf && ctx->tsdf) (void)hipFree(ctx->tsdf);
    if (ctx->own_centroid && ctx->centroid) (void)hipFree(ctx->centroid);
    for (int q = 0; q < 4; ++q)
        if (ctx->prep_stream[q]) (void)hipStreamSynchronize(ctx->prep_stream[q]);
    if (ctx->tsdf_scratch_slab) (void)hipFree(ctx->tsdf_scratch_slab);
    for (int b = 0; b < TSDF_SCRATCHES; ++b) {
        if (ctx->ev_prep[b]) (void)hipEventDestroy(ctx->ev_prep[b]);
        if (ctx->ev_upd[b]) (void)hipEventDestroy(ctx->ev_upd[b]);
    }
    for (int q = 0; q < 4; ++q)
        if (ctx->prep_stream[q]) (void)hipStreamDestroy(ctx->prep_stream[q]);
    if (ctx->block_counts) (void)hipFree(ctx->block_counts);
    if (ctx->block_offsets) (void)hipFree(ctx->block_offsets);
    if (ctx->bp_state) (void)hipFree(ctx->bp_state);
    if (ctx->bp_factors) (void)hipFree(ctx->bp_factors);
    if (ctx->bp_stage_xyz) (void)hipFree(ctx->bp_stage_xyz);
    if (ctx->bp_stage_rgb) (void)hipFree(ctx->bp_stage_rgb);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_cen_frames) (void)hipFree(ctx->d_cen_frames);
    if (ctx->d_cen_counters) (void)hipFree(ctx->d_cen_counters);
    if (ctx->rccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->rccl_comm);
    if (ctx->d_maxw) (void)hipFree(ctx->d_maxw);
    if (ctx->free_cnt) (void)hipFree(ctx->free_cnt);
    if (ctx->ev_free) (void)hipEventDestroy(ctx->ev_free);
    for (int l = 0; l < TL3D_ICP_LANES; ++l) {
        tl3d_ctx::IcpLane &ln = ctx->icp_lanes[l];
        if (ln.stream) (void)hipStreamSynchronize(ln.stream);
        icp_lane_free(ln);
    }
    if (ctx->icp_batch.stream) (void)hipStreamSynchronize(ctx->icp_batch.stream);
    icp_batch_free(ctx->icp_batch);
    if (ctx->bounds_slab) (void)hipFree(ctx->bounds_slab);
    for (int i = 0; i < 2; ++i)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->ktimers) {
        for (int i = 0; i < ctx->n_ktimers; ++i) {
            (void)hipEventDestroy(ctx->ktimers[i].a);
            (void)hipEventDestroy(ctx->ktimers[i].b);
        }
        delete[] ctx->ktimers;
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return TL3D_OK;
}

int tl3d_sync(tl3d_ctx *ctx) {
    REQUIRE(ctx != nullptr, TL3D_E_INVALID, "null ctx");
    FLUSH_AND_FOLD(ctx);
    TL3D_HIP(hipSetDevice(ctx->device));
    for (int q = 0; q < 4; ++q)
        if (ctx->prep_stream[q]) TL3D_HIP(hipStreamSynchronize(ctx->prep_stream[q]));
    for (int l = 0; l < TL3D_ICP_LANES; ++l)
        if (ctx->icp_lanes[l].stream) TL3D_HIP(hipStreamSynchronize(ctx->icp_lanes[l].stream));
    if (ctx->icp_batch.stream) TL3D_HIP(hipStreamSynchronize(ctx->icp_batch.stream));
    TL3D_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->bp_async_pending && ctx->bp_state) {
        ctx->bp_async_pending = false;
        unsigned long long err = 0;
        TL3D_HIP(hipMemcpy(&err, ctx->bp_state + 1, sizeof(err), hipMemcpyDeviceToHost));
        const int erc = bp_check_error(ctx, err);
        if (erc) return erc;
    }
    return TL3D_OK;
}

// ------------------------------------------------------------------------------------------- frames
int tl3d_get_stream(tl3d_ctx *ctx, void **stream) {
    REQUIRE(ctx && stream, TL3D_E_INVALID, "null argument");
    *stream = (void *)ctx->stream;
    return TL3D_OK;
}

int tl3d_upload_frame(tl3d_ctx *ctx, int slot, const void *depth_hd, int depth_kind, const uint8_t *bgr_hd) {
    return upload_impl(ctx, slot, depth_hd, depth_kind, bgr_hd, true);
}

}  // extern "C"

static int upload_impl(tl3d_ctx *ctx, int slot, const void *depth_hd, int depth_kind, const uint8_t *bgr_hd, bool wait) {
    int rc = check_slot(ctx, slot, false);
    if (rc) return rc;
    REQUIRE(depth_hd != nullptr, TL3D_E_INVALID, "null depth");
    REQUIRE(depth_kind == TL3D_DEPTH_F32_M || depth_kind == TL3D_DEPTH_U16_MM, TL3D_E_INVALID, "bad depth_kind %d", depth_kind);
    FLUSH_UPDATES(ctx);
    TL3D_HIP(hipSetDevice(ctx->device));
    rc = order_after_lanes(ctx, slot, true, false);     // an uncollected ICP run may still read this slot's depth
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    const size_t npx = (size_t)ctx->cam.W * ctx->cam.H;
    if (!s.depth && !(s.depth = (float *)pool_take(ctx->pool_depth))) return set_err(TL3D_E_NOMEM, "frame alloc failed");
    if (depth_kind == TL3D_DEPTH_F32_M) {
        TL3D_HIP(hipMemcpyAsync(s.depth, depth_hd, npx * sizeof(float), hipMemcpyDefault, ctx->stream));
        s.has_u16 = false;
    } else {
        // the millimetre image stays beside its f32 conversion: the TSDF kernels gather from it (half the cache lines
        // under a brick's footprint) and convert with the same IEEE division, every other kernel reads the f32 copy
        if (!s.depth_u16 && !(s.depth_u16 = (uint16_t *)pool_take(ctx->pool_u16))) return set_err(TL3D_E_NOMEM, "frame alloc failed");
        TL3D_HIP(hipMemcpyAsync(s.depth_u16, depth_hd, npx * sizeof(uint16_t), hipMemcpyDefault, ctx->stream));
        rc = launch_u16_to_f32(ctx->stream, s.depth_u16, s.depth, npx);
        if (rc) return rc;
        s.has_u16 = true;
    }
    if (bgr_hd) {
        if (!s.bgr && !(s.bgr = (uint8_t *)pool_take(ctx->pool_bgr))) return set_err(TL3D_E_NOMEM, "frame alloc failed");
        TL3D_HIP(hipMemcpyAsync(s.bgr, bgr_hd, npx * 3, hipMemcpyDefault, ctx->stream));
        s.has_color = true;
    } else {
        s.has_color = false;
    }
    s.loaded = true;
    s.has_normals = false;
    s.smooth_radius = 0;                                 // (the averaged depth belongs to the previous frame of this slot)
    if (!s.ev_upload) TL3D_HIP(hipEventCreateWithFlags(&s.ev_upload, hipEventDisableTiming));
    TL3D_HIP(hipEventRecord(s.ev_upload, ctx->stream));       // the prep stream waits on this, not on the whole main stream
    // pageable host sources are consumed before hipMemcpyAsync returns only for small copies; make the hand-over explicit
    if (wait && (!is_device_ptr(depth_hd) || (bgr_hd && !is_device_ptr(bgr_hd)))) TL3D_HIP(hipStreamSynchronize(ctx->stream));
    return TL3D_OK;
}

extern "C" {

int tl3d_attach_grid(tl3d_ctx *ctx, const tl3d_config *cfg) {
    REQUIRE(ctx && cfg, TL3D_E_INVALID, "null argument");
    REQUIRE(ctx->tsdf == nullptr && ctx->centroid == nullptr, TL3D_E_STATE, "context already has a grid");
    int rc = validate_grid(cfg);
    if (rc) return rc;
    TL3D_HIP(hipSetDevice(ctx->device));
    rc = alloc_grid(ctx, cfg);
    if (rc) return rc;
    TL3D_HIP(hipStreamSynchronize(ctx->stream));
    return TL3D_OK;
}

int tl3d_pinned_alloc(size_t bytes, void **out) {
    REQUIRE(out != nullptr && bytes > 0, TL3D_E_INVALID, "bad argument");
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return set_err(TL3D_E_NOMEM, "pinned allocation of %zu B failed", bytes);
    }
    return TL3D_OK;
}

int tl3d_pinned_free(void *p) {
    if (p && hipHostFree(p) != hipSuccess) return set_err(TL3D_E_HIP, "hipHostFree failed");
    return TL3D_OK;
}

int tl3d_upload_frame_async(tl3d_ctx *ctx, int slot, const void *depth_hd, int depth_kind, const uint8_t *bgr_hd) {
    return upload_impl(ctx, slot, depth_hd, depth_kind, bgr_hd, false);
}

int tl3d_slot_wait(tl3d_ctx *ctx, int slot) {
    int rc = check_slot(ctx, slot, true);
    if (rc) return rc;
    TL3D_HIP(hipSetDevice(ctx->device));
    if (ctx->slots[slot].ev_upload) TL3D_HIP(hipEventSynchronize(ctx->slots[slot].ev_upload));
    return TL3D_OK;
}

int tl3d_download_depth(tl3d_ctx *ctx, int slot, float *out) {
    int rc = check_slot(ctx, slot, true);
    if (rc) return rc;
    REQUIRE(out != nullptr, TL3D_E_INVALID, "null out");
    TL3D_HIP(hipSetDevice(ctx->device));
    TL3D_HIP(hipMemcpyAsync(out, ctx->slots[slot].depth, (size_t)ctx->cam.W * ctx->cam.H * sizeof(float), hipMemcpyDefault, ctx->stream));
    TL3D_HIP(hipStreamSynchronize(ctx->stream));
    return TL3D_OK;
}

// ------------------------------------------------------------------------------------------- back-projection
static int make_bp_args(tl3d_ctx *ctx, double scale, uint32_t flags, int subsample, double min_d, double max_d, BpArgs *a) {
    REQUIRE(subsample >= 1, TL3D_E_INVALID, "subsample must be >= 1");
    REQUIRE((flags & ~(TL3D_F_SCALE_F64 | TL3D_F_NO_POSE)) == 0, TL3D_E_INVALID, "unknown flags 0x%x", flags);
    a->sub = subsample;
    a->Ws = (ctx->cam.W + subsample - 1) / subsample;
    a->Hs = (ctx->cam.H + subsample - 1) / subsample;
    a->flags = flags;
    a->scale = scale;
    a->min_d = min_d;
    a->max_d = max_d;
    a->zero = 0ull;
    return TL3D_OK;
}

// scratch of the one-launch back-projection (layout: kernels_backproject.hip), + one word at the end for the total.  Zeroed
// here once; every launch leaves it zero again (the tile that finishes last re-arms it), except the error word.
static int bp_ensure_state(tl3d_ctx *ctx, const BpArgs &a) {
    const size_t need = (size_t)bp_state_words(a) + 2;
    if (need <= ctx->bp_state_words) return TL3D_OK;
    if (ctx->bp_state) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->bp_state);
    }
    ctx->bp_state = nullptr;
    ctx->bp_state_words = 0;
    const size_t words = need * 2;                                  // headroom: a later, larger subsample-1 call re-uses it
    if (hipMalloc(&ctx->bp_state, words * sizeof(unsigned long long)) != hipSuccess) return set_err(TL3D_E_NOMEM, "back-projection scratch alloc failed");
    if (hipMemsetAsync(ctx->bp_state, 0, words * sizeof(unsigned long long), ctx->stream) != hipSuccess) return set_err(TL3D_E_HIP, "memset failed");
    ctx->bp_state_words = words;
    return TL3D_OK;
}

// a look-back time-out leaves the scratch in an unknown state: report it and start from zeroes again
static int bp_check_error(tl3d_ctx *ctx, unsigned long long err) {
    if (err == 0) return TL3D_OK;
    (void)hipMemsetAsync(ctx->bp_state, 0, ctx->bp_state_words * sizeof(unsigned long long), ctx->stream);
    return set_err(TL3D_E_HIP, "back-projection look-back timed out (a predecessor tile never published)");
}

int tl3d_backproject(tl3d_ctx *ctx, int slot, const double R[9], const double t[3], double scale, uint32_t flags,
                     int subsample, double min_depth, double max_depth, float *out_xyz, uint8_t *out_rgb, int64_t cap,
                     int64_t *out_n) {
    int rc = check_slot(ctx, slot, true);
    if (rc) return rc;
    REQUIRE(out_n != nullptr, TL3D_E_INVALID, "null out_n");
    REQUIRE((flags & TL3D_F_NO_POSE) || (R && t), TL3D_E_INVALID, "pose required unless TL3D_F_NO_POSE");
    BpArgs a;
    rc = make_bp_args(ctx, scale, flags, subsample, min_depth, max_depth, &a);
    if (rc) return rc;
    TL3D_HIP(hipSetDevice(ctx->device));
    const Slot &s = ctx->slots[slot];
    rc = bp_ensure_state(ctx, a);
    if (rc) return rc;
    unsigned long long *d_total = ctx->bp_state + (ctx->bp_state_words - 1);
    const PoseD p = make_pose_d(R, t, (flags & TL3D_F_NO_POSE) != 0);
    const bool want = out_xyz && out_rgb;
    const bool direct = want && is_device_ptr(out_xyz) && is_device_ptr(out_rgb);
    const unsigned long long ns = (unsigned long long)a.Ws * (unsigned long long)a.Hs;
    unsigned long long cap_eff = cap < 0 ? 0ull : (unsigned long long)cap;
    float *dxyz = out_xyz;
    uint8_t *drgb = out_rgb;
    if (want && !direct) {
        // host outputs: stage on the device (buffers kept for the life of the context, sized for a full frame)
        const size_t npx = (size_t)ctx->cam.W * ctx->cam.H;
        if (!ctx->bp_stage_xyz && hipMalloc(&ctx->bp_stage_xyz, npx * 3 * sizeof(float)) != hipSuccess) return set_err(TL3D_E_NOMEM, "output staging alloc failed");
        if (!ctx->bp_stage_rgb && hipMalloc(&ctx->bp_stage_rgb, npx * 3) != hipSuccess) return set_err(TL3D_E_NOMEM, "output staging alloc failed");
        dxyz = ctx->bp_stage_xyz;
        drgb = ctx->bp_stage_rgb;
        if (cap_eff > ns) cap_eff = ns;
    }
    // ONE launch: count, ordered offsets (decoupled look-back) and LDS-staged writes; without buffers: the count only.
    // A look-back time-out (static tile order and a predecessor that never got a slot: see launch_bp_fused) is repeated
    // once in dynamic order, which cannot starve.
    unsigned long long h[2] = {0, 0};
    for (int attempt = 0; attempt < 2; ++attempt) {
        rc = launch_bp_fused(ctx->stream, ctx->cam, a, p, s.depth, s.has_color ? s.bgr : nullptr, ctx->bp_factors, ctx->bp_factors + ctx->cam.W,
                             ctx->bp_state, want ? dxyz : nullptr, want ? drgb : nullptr, cap_eff, d_total, attempt == 1);
        if (rc) return rc;
        TL3D_HIP(hipMemcpyAsync(&h[0], d_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        TL3D_HIP(hipMemcpyAsync(&h[1], ctx->bp_state + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        TL3D_HIP(hipStreamSynchronize(ctx->stream));
        rc = bp_check_error(ctx, h[1]);
        if (rc == TL3D_OK) break;
        if (attempt == 1) return rc;
        ctx->stats.bp_lookback_retries++;                               // (never seen so far; tl3d_get_stats shows it if it happens)
    }
    const unsigned long long total = h[0];
    *out_n = (int64_t)total;
    if (!want) return TL3D_OK;                                      // size query
    if ((int64_t)total > cap) return set_err(TL3D_E_CAPACITY, "need %llu points, capacity %lld", total, (long long)cap);
    if (total == 0 || direct) return TL3D_OK;
    TL3D_HIP(hipMemcpyAsync(out_xyz, dxyz, total * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    TL3D_HIP(hipMemcpyAsync(out_rgb, drgb, total * 3, hipMemcpyDeviceToHost, ctx->stream));
    TL3D_HIP(hipStreamSynchronize(ctx->stream));
    return TL3D_OK;
}

int tl3d_backproject_device(tl3d_ctx *ctx, int slot, const double R[9], const double t[3], double scale, uint32_t flags,
                            int subsample, double min_depth, double max_depth, float *out_xyz_dev, uint8_t *out_rgb_dev,
                            int64_t cap, int64_t *out_n_dev) {
    int rc = check_slot(ctx, slot, true);
    if (rc) return rc;
    REQUIRE(out_xyz_dev && out_rgb_dev && out_n_dev, TL3D_E_INVALID, "null output pointer");
    REQUIRE(is_device_ptr(out_xyz_dev) && is_device_ptr(out_rgb_dev) && is_device_ptr(out_n_dev), TL3D_E_INVALID,
            "tl3d_backproject_device writes device memory only");
    REQUIRE(cap >= 0, TL3D_E_INVALID, "negative capacity");
    REQUIRE((flags & TL3D_F_NO_POSE) || (R && t), TL3D_E_INVALID, "pose required unless TL3D_F_NO_POSE");
    BpArgs a;
    rc = make_bp_args(ctx, scale, flags, subsample, min_depth, max_depth, &a);
    if (rc) return rc;
    TL3D_HIP(hipSetDevice(ctx->device));
    const Slot &s = ctx->slots[slot];
    rc = bp_ensure_state(ctx, a);
    if (rc) return rc;
    const PoseD p = make_pose_d(R, t, (flags & TL3D_F_NO_POSE) != 0);
    ctx->bp_async_pending = true;                       // tl3d_sync reports a look-back time-out of these launches
    // nobody reads the error word back between these launches: an earlier time-out must not make every later launch give up
    // (the word is cleared IN STREAM ORDER; tl3d_sync still reports a time-out of the last launch), and with a batched
    // registration spinning on the chip the static tile order's "every tile is resident" cannot be taken for granted
    TL3D_HIP(hipMemsetAsync(ctx->bp_state + 1, 0, sizeof(unsigned long long), ctx->stream));
    return launch_bp_fused(ctx->stream, ctx->cam, a, p, s.depth, s.has_color ? s.bgr : nullptr, ctx->bp_factors, ctx->bp_factors + ctx->cam.W,
                           ctx->bp_state, out_xyz_dev, out_rgb_dev, (unsigned long long)cap, reinterpret_cast<unsigned long long *>(out_n_dev),
                           ctx->icp_batch.busy);
}

int tl3d_frames_bounds(tl3d_ctx *ctx, int n_frames, const int32_t *slots, const double *R, const double *t, const double *scales,
                       uint32_t flags, int subsample, double min_depth, double max_depth, double out_min[3], double out_max[3]) {
    REQUIRE(ctx && slots && out_min && out_max, TL3D_E_INVALID, "null argument");
    REQUIRE(n_frames >= 1, TL3D_E_INVALID, "n_frames must be positive");
    REQUIRE((flags & TL3D_F_NO_POSE) || (R && t), TL3D_E_INVALID, "poses required unless TL3D_F_NO_POSE");
    for (int i = 0; i < n_frames; ++i) {
        const int rc = check_slot(ctx, slots[i], true);
        if (rc) return rc;
    }
    BpArgs a;
    int rc = make_bp_args(ctx, scales ? scales[0] : 1.0, flags, subsample, min_depth, max_depth, &a);
    if (rc) return rc;
    TL3D_HIP(hipSetDevice(ctx->device));
    const long long ns = (long long)a.Ws * a.Hs;
    int nb = (int)((ns + 255) / 256);
    if (nb > 64) nb = 64;                                  // 16 frames share the 1024-row slab between two read-backs
    const int per_round = 1024 / nb;
    for (int k = 0; k < 3; ++k) { out_min[k] = INFINITY; out_max[k] = -INFINITY; }
    std::vector<float> h((size_t)1024 * 6);
    for (int i0 = 0; i0 < n_frames; i0 += per_round) {
        const int cnt = n_frames - i0 < per_round ? n_frames - i0 : per_round;
        for (int j = 0; j < cnt; ++j) {
            const int i = i0 + j;
            if (scales) {
                rc = make_bp_args(ctx, scales[i], flags, subsample, min_depth, max_depth, &a);
                if (rc) return rc;
            }
            const PoseD p = make_pose_d(R ? R + (size_t)9 * i : nullptr, t ? t + (size_t)3 * i : nullptr, (flags & TL3D_F_NO_POSE) != 0);
            rc = launch_bp_bounds(ctx->stream, ctx->cam, a, p, ctx->slots[slots[i]].depth, ctx->bp_factors, ctx->bp_factors + ctx->cam.W,
                                  ctx->bounds_slab + (size_t)j * nb * 6, nb);
            if (rc) return rc;
        }
        TL3D_HIP(hipMemcpyAsync(h.data(), ctx->bounds_slab, (size_t)cnt * nb * 6 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        TL3D_HIP(hipStreamSynchronize(ctx->stream));
        for (int b = 0; b < cnt * nb; ++b)
            for (int k = 0; k < 3; ++k) {
                out_min[k] = fmin(out_min[k], (double)h[(size_t)b * 6 + k]);
                out_max[k] = fmax(out_max[k], (double)h[(size_t)b * 6 + 3 + k]);
            }
    }
    return TL3D_OK;
}

int tl3d_frame_bounds(tl3d_ctx *ctx, int slot, const double R[9], const double t[3], double scale, uint32_t flags, int subsample,
                      double min_depth, double max_depth, double out_min[3], double out_max[3], int64_t *out_reserved) {
    const int32_t s = slot;
    if (out_reserved) *out_reserved = 0;
    return tl3d_frames_bounds(ctx, 1, &s, R, t, &scale, flags, subsample, min_depth, max_depth, out_min, out_max);
}

// ------------------------------------------------------------------------------------------- occupancy of a planned grid
// Which bricks of a grid (geometry only: no records, no pools) a fusion of these frames WOULD give records to: the TSDF
// classification of every frame (tiles, pyramid, brick classes: the update's own prep chain, sub-brick masks and update left out)
// and the bricks the frames' samples fall into, against scratch brick tables whose cursors count the distinct bricks.  What a
// sparse grid's pools must hold (tl3d_config.pool_bricks_*): the reference's hash-map merge sizes itself (D2R:404-410); a pool is
// allocated up front, and this is how to know how large -- a few microseconds per frame instead of a guess.
int tl3d_count_bricks(tl3d_ctx *ctx, const tl3d_config *cfg, int n_frames, const int32_t *slots, const double *R, const double *t,
                      const double *scales, int centroid_subsample, double min_depth, double max_depth, int64_t *bricks_tsdf,
                      int64_t *bricks_centroid) {
    REQUIRE(ctx && cfg && slots && R && t && bricks_tsdf && bricks_centroid, TL3D_E_INVALID, "null argument");
    REQUIRE(n_frames >= 1, TL3D_E_INVALID, "n_frames must be positive");
    int rc = validate_grid(cfg);
    if (rc) return rc;
    for (int i = 0; i < n_frames; ++i) {
        rc = check_slot(ctx, slots[i], true);
        if (rc) return rc;
    }
    FLUSH_UPDATES(ctx);
    TL3D_HIP(hipSetDevice(ctx->device));
    Grid g;
    memset(&g, 0, sizeof(g));
    grid_geometry(g, cfg);
    const size_t nbr = ((size_t)g.nx * g.ny * g.nz) >> 9;
    unsigned *tabs = nullptr, *free_cnt = nullptr;
    void *scratch = nullptr;
    auto done = [&](int code) {
        (void)hipStreamSynchronize(ctx->stream);
        if (tabs) (void)hipFree(tabs);
        if (free_cnt) (void)hipFree(free_cnt);
        if (scratch) (void)hipFree(scratch);
        return code;
    };
    if (hipMalloc(&tabs, (2 * nbr + 64) * sizeof(unsigned)) != hipSuccess) return done(set_err(TL3D_E_NOMEM, "brick table alloc failed"));
    if (hipMemsetAsync(tabs, 0xff, 2 * nbr * sizeof(unsigned), ctx->stream) != hipSuccess || hipMemsetAsync(tabs + 2 * nbr, 0, 64 * sizeof(unsigned), ctx->stream) != hipSuccess)
        return done(set_err(TL3D_E_HIP, "memset failed"));
    g.tsdf_tab = tabs; g.cen_tab = tabs + nbr; g.cursors = tabs + 2 * nbr;
    g.tsdf_cap = g.cen_cap = (unsigned)nbr;
    const float mind = (float)min_depth, maxd = (float)max_depth;
    if (cfg->channels & TL3D_CH_TSDF) {
        const int batch = TL3D_TSDF_MAXBATCH;
        const size_t sb = tsdf_batch_scratch_bytes(ctx->cam, g, batch);
        if (hipMalloc(&free_cnt, nbr * sizeof(unsigned)) != hipSuccess || hipMalloc(&scratch, sb) != hipSuccess) return done(set_err(TL3D_E_NOMEM, "scratch alloc (%zu B) failed", sb));
        size_t zoff = 0, zbytes = 0;
        tsdf_batch_scratch_zero_range(ctx->cam, g, batch, &zoff, &zbytes);
        if (hipMemsetAsync(free_cnt, 0, nbr * sizeof(unsigned), ctx->stream) != hipSuccess) return done(set_err(TL3D_E_HIP, "memset failed"));
        const Frustum fr = make_frustum(ctx->cam);
        for (int i0 = 0; i0 < n_frames; i0 += batch) {
            const int m = n_frames - i0 < batch ? n_frames - i0 : batch;
            PoseF poses[TL3D_TSDF_MAXBATCH];
            const void *depths[TL3D_TSDF_MAXBATCH];
            float sc[TL3D_TSDF_MAXBATCH];
            for (int j = 0; j < m; ++j) {
                poses[j] = make_pose_f(R + (size_t)9 * (i0 + j), t + (size_t)3 * (i0 + j));
                depths[j] = ctx->slots[slots[i0 + j]].depth;
                sc[j] = (float)(scales ? scales[i0 + j] : 1.0);
            }
            if (hipMemsetAsync((char *)scratch + zoff, 0, zbytes, ctx->stream) != hipSuccess) return done(set_err(TL3D_E_HIP, "memset failed"));    // (no update re-arms the frame masks)
            rc = launch_tsdf_prepare(ctx->stream, ctx->cam, g, m, batch, poses, fr, depths, false, sc, mind, maxd, scratch, free_cnt, true);
            if (rc) return done(rc);
        }
    }
    if ((cfg->channels & TL3D_CH_CENTROID) && centroid_subsample >= 1) {
        for (int i = 0; i < n_frames; ++i) {
            BpArgs a;
            rc = make_bp_args(ctx, scales ? scales[i] : 1.0, 0, centroid_subsample, min_depth, max_depth, &a);
            if (rc) return done(rc);
            const PoseD p = make_pose_d(R + (size_t)9 * i, t + (size_t)3 * i, false);
            rc = launch_centroid_mark(ctx->stream, ctx->cam, g, a, p, ctx->slots[slots[i]].depth, ctx->bp_factors, ctx->bp_factors + ctx->cam.W);
            if (rc) return done(rc);
        }
    }
    unsigned cur[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(cur, g.cursors, sizeof(cur), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return done(set_err(TL3D_E_HIP, "copy failed"));
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return done(set_err(TL3D_E_HIP, "sync failed"));
    *bricks_tsdf = (int64_t)(cur[0] < nbr ? cur[0] : nbr);
    *bricks_centroid = (int64_t)(cur[2] < nbr ? cur[2] : nbr);
    return done(TL3D_OK);
}

// ------------------------------------------------------------------------------------------- centroid accumulation
int tl3d_accumulate_centroid(tl3d_ctx *ctx, int slot, const double R[9], const double t[3], double scale, uint32_t flags,
                             int subsample, double min_depth, double max_depth) {
    int rc = check_slot(ctx, slot, true);
    if (rc) return rc;
    REQUIRE(ctx->centroid != nullptr, TL3D_E_STATE, "centroid channel not enabled");
    REQUIRE((flags & TL3D_F_NO_POSE) || (R && t), TL3D_E_INVALID, "pose required unless TL3D_F_NO_POSE");
    BpArgs a;
    rc = make_bp_args(ctx, scale, flags, subsample, min_depth, max_depth, &a);
    if (rc) return rc;
    const Slot &s = ctx->slots[slot];
    // The frame joins the pending batch (one stride per batch); the batch goes out when it is full, or when any call needs the grid,
    // a slot, a sync or a time stamp (flush_updates).  Integer sums: the grid does not depend on where the batches are cut.
    if (ctx->n_cen_pend > 0 && ctx->cen_pend[0].a.sub != a.sub) {
        rc = flush_centroid(ctx);
        if (rc) return rc;
    }
    CenFrame &f = ctx->cen_pend[ctx->n_cen_pend++];
    f.p = make_pose_d(R, t, (flags & TL3D_F_NO_POSE) != 0);
    f.a = a;
    f.depth = s.depth;
    f.bgr = s.has_color ? s.bgr : nullptr;
    if (ctx->n_cen_pend >= TL3D_CEN_MAXBATCH) return flush_centroid(ctx);
    return TL3D_OK;
}

int tl3d_accumulate_points(tl3d_ctx *ctx, const float *xyz, const uint8_t *rgb, int64_t n) {
    REQUIRE(ctx != nullptr, TL3D_E_INVALID, "null ctx");
    REQUIRE(ctx->centroid != nullptr, TL3D_E_STATE, "centroid channel not enabled");
    REQUIRE(n >= 0 && (n == 0 || (xyz && rgb)), TL3D_E_INVALID, "bad point list");
    if (n == 0) return TL3D_OK;
    TL3D_HIP(hipSetDevice(ctx->device));
    const bool direct = is_device_ptr(xyz) && is_device_ptr(rgb);
    const float *dxyz = xyz;
    const uint8_t *drgb = rgb;
    float *tx = nullptr;
    uint8_t *tc = nullptr;
    if (!direct) {
        if (hipMalloc(&tx, (size_t)n * 12) != hipSuccess) return set_err(TL3D_E_NOMEM, "point staging alloc failed");
        if (hipMalloc(&tc, (size_t)n * 3) != hipSuccess) { (void)hipFree(tx); return set_err(TL3D_E_NOMEM, "point staging alloc failed"); }
        hipError_t e = hipMemcpyAsync(tx, xyz, (size_t)n * 12, hipMemcpyDefault, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(tc, rgb, (size_t)n * 3, hipMemcpyDefault, ctx->stream);
        if (e != hipSuccess) { (void)hipFree(tx); (void)hipFree(tc); return set_err(TL3D_E_HIP, "point upload failed: %s", hipGetErrorString(e)); }
        dxyz = tx;
        drgb = tc;
    }
    ctx->grid_epoch++;
    int rc = launch_centroid_points(ctx->stream, ctx->grid, dxyz, drgb, n, ctx->centroid, ctx->d_cen_counters);
    if (!direct) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(tx);
        (void)hipFree(tc);
        if (rc == TL3D_OK && e != hipSuccess) return set_err(TL3D_E_HIP, "sync failed: %s", hipGetErrorString(e));
    }
    if (rc) return rc;
    ctx->stats.centroid_launches++;
    return TL3D_OK;
}

int tl3d_points_bounds(tl3d_ctx *ctx, const float *xyz, int64_t n, double out_min[3], double out_max[3]) {
    REQUIRE(ctx && xyz && out_min && out_max, TL3D_E_INVALID, "null argument");
    REQUIRE(n > 0, TL3D_E_INVALID, "empty point list has no bounds");
    TL3D_HIP(hipSetDevice(ctx->device));
    const bool direct = is_device_ptr(xyz);
    const float *d = xyz;
    float *tx = nullptr;
    if (!direct) {
        if (hipMalloc(&tx, (size_t)n * 12) != hipSuccess) return set_err(TL3D_E_NOMEM, "point staging alloc failed");
        hipError_t e = hipMemcpyAsync(tx, xyz, (size_t)n * 12, hipMemcpyDefault, ctx->stream);
        if (e != hipSuccess) { (void)hipFree(tx); return set_err(TL3D_E_HIP, "point upload failed"); }
        d = tx;
    }
    int nb = (int)((n + 255) / 256);
    if (nb > 1024) nb = 1024;
    int rc = launch_bounds(ctx->stream, d, n, ctx->bounds_slab, nb);
    std::vector<float> h((size_t)nb * 6);
    hipError_t e = hipMemcpyAsync(h.data(), ctx->bounds_slab, h.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (tx) (void)hipFree(tx);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return set_err(TL3D_E_HIP, "bounds read-back failed");
    for (int a = 0; a < 3; ++a) { out_min[a] = INFINITY; out_max[a] = -INFINITY; }
    for (int b = 0; b < nb; ++b)
        for (int a = 0; a < 3; ++a) {
            out_min[a] = fmin(out_min[a], (double)h[(size_t)b * 6 + a]);
            out_max[a] = fmax(out_max[a], (double)h[(size_t)b * 6 + 3 + a]);
        }
    return TL3D_OK;
}

// ------------------------------------------------------------------------------------------- TSDF
static int ktimer_begin(tl3d_ctx *ctx) {
    if (!ctx->time_kernels) return -1;
    if (!ctx->ktimers) {
        ctx->n_ktimers = 1024;
        ctx->ktimers = new (std::nothrow) tl3d_ctx::KTimer[ctx->n_ktimers];
        if (!ctx->ktimers) return -1;
        for (int i = 0; i < ctx->n_ktimers; ++i) {
            (void)hipEventCreate(&ctx->ktimers[i].a);
            (void)hipEventCreate(&ctx->ktimers[i].b);
        }
        ctx->ktimers_used = 0;
    }
    if (ctx->ktimers_used == ctx->n_ktimers) {          // drain
        (void)hipStreamSynchronize(ctx->stream);
        for (int i = 0; i < ctx->ktimers_used; ++i) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ctx->ktimers[i].a, ctx->ktimers[i].b) == hipSuccess) {
                ctx->stats.tsdf_kernel_ms += ms;
                ctx->stats.tsdf_kernel_timed += (uint64_t)ctx->ktimers[i].launches;
            }
        }
        ctx->ktimers_used = 0;
    }
    const int id = ctx->ktimers_used++;
    (void)hipEventRecord(ctx->ktimers[id].a, ctx->stream);
    return id;
}

// Issues the pending batch: its prep chain on a side stream (it needs the frames' uploads, the batch scratch whose previous
// user -- three batches ago -- has been updated, and the last clear / fold of the free-space counters), then ONE update launch on
// the main stream behind it.  Results never depend on where the batch boundaries fall (integer sums).
// the pending voxel-centroid accumulations: one launch for all of them (they share a stride)
static int flush_centroid(tl3d_ctx *ctx) {
    if (ctx->n_cen_pend == 0) return TL3D_OK;
    const int n = ctx->n_cen_pend;
    ctx->n_cen_pend = 0;                                // whatever happens below, the batch is consumed
    TL3D_HIP(hipSetDevice(ctx->device));
    if (!ctx->d_cen_frames && hipMalloc(&ctx->d_cen_frames, sizeof(CenFrame) * TL3D_CEN_MAXBATCH) != hipSuccess)
        return set_err(TL3D_E_NOMEM, "centroid descriptor alloc failed");
    ctx->grid_epoch++;
    const int rc = launch_centroid_batch(ctx->stream, ctx->cam, ctx->grid, n, ctx->cen_pend, ctx->d_cen_frames, ctx->bp_factors, ctx->bp_factors + ctx->cam.W,
                                         ctx->centroid, ctx->d_cen_counters);
    if (rc) return rc;
    ctx->stats.centroid_launches++;
    return TL3D_OK;
}

static int flush_updates(tl3d_ctx *ctx) {
    {
        const int crc = flush_centroid(ctx);
        if (crc) return crc;
    }
    if (ctx->n_pend == 0) return TL3D_OK;
    ctx->grid_epoch++;
    TL3D_HIP(hipSetDevice(ctx->device));
    const float mind = (float)ctx->cfg.min_depth, maxd = (float)ctx->cfg.max_depth;
    const Frustum fr = make_frustum(ctx->cam);
    const int n = ctx->n_pend;
    ctx->n_pend = 0;                                    // whatever happens below, the batch is consumed
    const int half = (int)(ctx->tsdf_batch_no % (unsigned)TSDF_SCRATCHES);
    const bool u16 = ctx->pend_u16;
    hipStream_t ps = ctx->tsdf_single_stream ? ctx->stream : ctx->prep_stream[ctx->tsdf_batch_no % (unsigned)ctx->n_prep_streams];
    PoseF poses[TL3D_TSDF_MAXBATCH];
    const void *depths[TL3D_TSDF_MAXBATCH];
    float scales[TL3D_TSDF_MAXBATCH];
    for (int i = 0; i < n; ++i) {
        const tl3d_ctx::PendingUpdate &w = ctx->pend[i];
        const Slot &ws = ctx->slots[w.slot];
        if (ps != ctx->stream && ws.ev_upload) TL3D_HIP(hipStreamWaitEvent(ps, ws.ev_upload, 0));
        poses[i] = w.pose;
        depths[i] = u16 ? (const void *)ws.depth_u16 : (const void *)ws.depth;
        scales[i] = w.scale;
    }
    if (ps != ctx->stream) {
        if (ctx->upd_recorded[half]) TL3D_HIP(hipStreamWaitEvent(ps, ctx->ev_upd[half], 0));
        if (ctx->ev_free_recorded) TL3D_HIP(hipStreamWaitEvent(ps, ctx->ev_free, 0));      // clears / folds of the free-space counters
    }
    int rc = launch_tsdf_prepare(ps, ctx->cam, ctx->grid, n, ctx->tsdf_batch, poses, fr, depths, u16, scales, mind, maxd, ctx->tsdf_scratch[half], ctx->free_cnt);
    if (rc) return rc;
    if (ps != ctx->stream) {
        TL3D_HIP(hipEventRecord(ctx->ev_prep[half], ps));
        TL3D_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_prep[half], 0));
    }
    // profiling mode: one event pair around the update kernel
    const int kt = ktimer_begin(ctx);
    rc = launch_tsdf_update(ctx->stream, ctx->cam, ctx->grid, n, ctx->tsdf_batch, u16, mind, maxd, ctx->tsdf, ctx->tsdf_scratch[half], ctx->d_counters,
                            ctx->count_records, ctx->tsdf_max_blocks, ctx->tsdf_xcd_group);
    if (rc == TL3D_OK) ctx->stats.tsdf_launches++;
    if (kt >= 0) {
        ctx->ktimers[kt].launches = rc == TL3D_OK ? 1 : 0;
        (void)hipEventRecord(ctx->ktimers[kt].b, ctx->stream);
    }
    ctx->tsdf_batch_no++;
    TL3D_HIP(hipEventRecord(ctx->ev_upd[half], ctx->stream));
    ctx->upd_recorded[half] = true;
    return rc;
}

int tl3d_integrate(tl3d_ctx *ctx, int slot, const double R[9], const double t[3], double scale) {
    int rc = check_slot(ctx, slot, true);
    if (rc) return rc;
    REQUIRE(ctx->tsdf != nullptr, TL3D_E_STATE, "TSDF channel not enabled");
    REQUIRE(R && t, TL3D_E_INVALID, "null pose");
    TL3D_HIP(hipSetDevice(ctx->device));
    const PoseF p = make_pose_f(R, t);
    // int32 headroom: one more observation must keep every |sum_q| <= weight * 32767 below 2^31
    if (ctx->tsdf_w_unknown || ctx->tsdf_w_upper + 1 > TL3D_TSDF_MAX_WEIGHT) {
        FLUSH_AND_FOLD(ctx);
        long long w = 0;
        rc = measure_max_weight(ctx, ctx->tsdf, &w);
        if (rc) return rc;
        ctx->tsdf_w_upper = w;
        ctx->tsdf_w_unknown = false;
        REQUIRE(w + 1 <= TL3D_TSDF_MAX_WEIGHT, TL3D_E_STATE,
                "a voxel already holds %lld observations: one more could overflow its int32 TSDF sum (limit %d); extract or reset the grid",
                w, TL3D_TSDF_MAX_WEIGHT);
    }
    ctx->tsdf_w_upper++;
    ctx->free_dirty = true;
    const Slot &sl = ctx->slots[slot];
    const bool u16 = sl.has_u16 && ctx->tsdf_use_u16;
    // The frame joins the pending batch (one depth kind per batch); the batch goes out when it is full, or when any call needs
    // the grid, a slot, a sync or a time stamp (flush_updates).
    if (ctx->n_pend > 0 && ctx->pend_u16 != u16) FLUSH_UPDATES(ctx);
    ctx->pend_u16 = u16;
    tl3d_ctx::PendingUpdate &u = ctx->pend[ctx->n_pend++];
    u.slot = slot; u.pose = p; u.scale = (float)scale;
    if (ctx->n_pend >= (ctx->tsdf_pairing ? ctx->tsdf_batch : 1)) return flush_updates(ctx);
    return TL3D_OK;
}

int tl3d_fuse_frames(tl3d_ctx *ctx, int n, const int32_t *slots, const double *R, const double *t, const double *scales,
                     uint32_t flags, int centroid_subsample, double min_depth, double max_depth) {
    REQUIRE(ctx && slots && R && t, TL3D_E_INVALID, "null argument");
    REQUIRE(n >= 0, TL3D_E_INVALID, "n must not be negative");
    for (int i = 0; i < n; ++i) {
        const double sc = scales ? scales[i] : 1.0;
        if (ctx->tsdf) {
            const int rc = tl3d_integrate(ctx, slots[i], R + (size_t)9 * i, t + (size_t)3 * i, sc);
            if (rc) return rc;
        }
        if (ctx->centroid && centroid_subsample >= 1) {
            const int rc = tl3d_accumulate_centroid(ctx, slots[i], R + (size_t)9 * i, t + (size_t)3 * i, sc, flags, centroid_subsample, min_depth, max_depth);
            if (rc) return rc;
        }
    }
    return TL3D_OK;
}

// ------------------------------------------------------------------------------------------- normals + ICP
int tl3d_build_normals(tl3d_ctx *ctx, int slot, double scale, double depth_jump) {
    int rc = check_slot(ctx, slot, true);
    if (rc) return rc;
    TL3D_HIP(hipSetDevice(ctx->device));
    Slot &s = ctx->slots[slot];
    if (!s.nmap && !(s.nmap = (float4 *)pool_take(ctx->pool_nmap))) return set_err(TL3D_E_NOMEM, "normal map alloc failed");
    rc = order_after_lanes(ctx, slot, false, true);     // an uncollected ICP run may still read this slot's normal map
    if (rc) return rc;
    const int radius = ctx->normal_radius;
    if (radius > 0 && !s.sdepth && !(s.sdepth = (float *)pool_take(ctx->pool_sdepth))) return set_err(TL3D_E_NOMEM, "smoothed depth alloc failed");
    if (radius > 0 || s.smooth_radius > 0) {
        rc = order_after_lanes(ctx, slot, true, false);  // ... or this slot's (smoothed) depth as its source
        if (rc) return rc;
    }
    rc = launch_normals(ctx->stream, ctx->cam, s.depth, (float)scale, (float)ctx->cfg.min_depth, (float)ctx->cfg.max_depth,
                        (float)depth_jump, radius, s.sdepth, s.nmap);
    if (rc) return rc;
    s.smooth_radius = radius;
    s.has_normals = true;
    if (!s.ev_normals) TL3D_HIP(hipEventCreateWithFlags(&s.ev_normals, hipEventDisableTiming));
    TL3D_HIP(hipEventRecord(s.ev_normals, ctx->stream));      // ICP lanes wait on this, not on the whole main stream
    return TL3D_OK;
}

int tl3d_download_normals(tl3d_ctx *ctx, int slot, float *out) {
    int rc = check_slot(ctx, slot, true);
    if (rc) return rc;
    REQUIRE(out != nullptr, TL3D_E_INVALID, "null out");
    REQUIRE(ctx->slots[slot].has_normals, TL3D_E_STATE, "slot %d has no normal map (call tl3d_build_normals)", slot);
    TL3D_HIP(hipSetDevice(ctx->device));
    // the map lives in phase-major rows (tl3d_internal.h: pm_index); the caller gets it row-major
    if (is_device_ptr(out)) {
        const int rc2 = launch_nmap_rowmajor(ctx->stream, ctx->cam, ctx->slots[slot].nmap, (float4 *)out);
        if (rc2) return rc2;
        TL3D_HIP(hipStreamSynchronize(ctx->stream));
        return TL3D_OK;
    }
    const int W = ctx->cam.W, H = ctx->cam.H, w4 = pm_w4(W);
    float4 *tmp = (float4 *)malloc(pm_pixels(W, H) * sizeof(float4));
    REQUIRE(tmp != nullptr, TL3D_E_NOMEM, "host allocation failed");
    hipError_t e = hipMemcpyAsync(tmp, ctx->slots[slot].nmap, pm_pixels(W, H) * sizeof(float4), hipMemcpyDefault, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { free(tmp); TL3D_HIP(e); }
    float4 *o4 = (float4 *)out;
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) o4[(size_t)v * W + u] = tmp[pm_index(u, v, w4)];
    free(tmp);
    return TL3D_OK;
}

int tl3d_build_normals_many(tl3d_ctx *ctx, int n, const int32_t *slots, const double *scales, double depth_jump) {
    REQUIRE(ctx && slots, TL3D_E_INVALID, "null argument");
    for (int i = 0; i < n; ++i) {
        const int rc = tl3d_build_normals(ctx, slots[i], scales ? scales[i] : 1.0, depth_jump);
        if (rc) return rc;
    }
    return TL3D_OK;
}

static void icp_lane_free(tl3d_ctx::IcpLane &ln) {
    if (ln.slab) (void)hipFree(ln.slab);
    if (ln.ticket) (void)hipFree(ln.ticket);
    if (ln.state) (void)hipFree(ln.state);
    for (int g = 0; g < 4; ++g)
        if (ln.graphs[g]) (void)hipGraphExecDestroy(ln.graphs[g]);
    if (ln.host) (void)hipHostFree(ln.host);
    if (ln.run) (void)hipFree(ln.run);
    if (ln.run_host) (void)hipHostFree(ln.run_host);
    if (ln.ev_done) (void)hipEventDestroy(ln.ev_done);
    if (ln.stream) (void)hipStreamDestroy(ln.stream);
    memset(&ln, 0, sizeof(ln));
}

// all or nothing: a lane whose stream exists has every buffer (a half-built lane would launch kernels on null pointers)
static int icp_lane_init(tl3d_ctx *ctx, int lane) {
    tl3d_ctx::IcpLane &ln = ctx->icp_lanes[lane];
    if (ln.stream) return TL3D_OK;
    bool ok = hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc(&ln.slab, (size_t)ICP_MAX_BLOCKS * ICP_SLAB * sizeof(double)) == hipSuccess;
    ok = ok && hipMalloc(&ln.ticket, 64) == hipSuccess;
    ok = ok && hipMalloc(&ln.state, sizeof(IcpState)) == hipSuccess;
    ok = ok && hipHostMalloc(&ln.host, sizeof(IcpState), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc(&ln.run, sizeof(IcpRun)) == hipSuccess;
    ok = ok && hipHostMalloc(&ln.run_host, sizeof(IcpRun), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ln.ev_done, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        icp_lane_free(ln);
        return set_err(TL3D_E_NOMEM, "ICP lane %d: stream / buffer allocation failed", lane);
    }
    for (int g = 0; g < 4; ++g) { ln.graphs[g] = nullptr; ln.graph_iters[g] = -1; }
    ln.graph_next = 0;
    ln.busy = false;
    ln.src_slot = ln.tgt_slot = -1;
    return TL3D_OK;
}

int tl3d_icp_enqueue(tl3d_ctx *ctx, int lane, int slot_src, double scale_src, int slot_tgt, const double T_init[16],
                     const tl3d_icp_params *prm) {
    int rc = check_slot(ctx, slot_src, true);
    if (rc) return rc;
    rc = check_slot(ctx, slot_tgt, true);
    if (rc) return rc;
    REQUIRE(lane >= 0 && lane < TL3D_ICP_LANES, TL3D_E_INVALID, "lane %d out of range [0,%d)", lane, TL3D_ICP_LANES);
    REQUIRE(prm != nullptr, TL3D_E_INVALID, "null argument");
    REQUIRE(prm->iters >= 0 && prm->iters <= 1000, TL3D_E_INVALID, "iters out of range");
    REQUIRE(prm->stride >= 1, TL3D_E_INVALID, "stride must be >= 1");
    REQUIRE(prm->max_dist > 0, TL3D_E_INVALID, "max_dist must be positive");
    REQUIRE(ctx->slots[slot_tgt].has_normals, TL3D_E_STATE, "target slot %d has no normal map (call tl3d_build_normals)", slot_tgt);
    TL3D_HIP(hipSetDevice(ctx->device));
    rc = icp_lane_init(ctx, lane);
    if (rc) return rc;
    tl3d_ctx::IcpLane &ln = ctx->icp_lanes[lane];
    REQUIRE(!ln.busy, TL3D_E_STATE, "ICP lane %d still holds an uncollected run", lane);
    // order this run after the frames it reads became valid on the main stream
    Slot &ss = ctx->slots[slot_src], &st = ctx->slots[slot_tgt];
    if (ss.ev_upload) TL3D_HIP(hipStreamWaitEvent(ln.stream, ss.ev_upload, 0));
    if (st.ev_upload) TL3D_HIP(hipStreamWaitEvent(ln.stream, st.ev_upload, 0));
    if (st.ev_normals) TL3D_HIP(hipStreamWaitEvent(ln.stream, st.ev_normals, 0));
    if (ss.smooth_radius > 0 && ss.ev_normals) TL3D_HIP(hipStreamWaitEvent(ln.stream, ss.ev_normals, 0));     // the averaged depth is written with the source's normal map
    IcpState *h = ln.host;
    memset(h, 0, sizeof(*h));
    if (T_init) {
        memcpy(h->T, T_init, sizeof(h->T));
    } else {
        h->T[0] = h->T[5] = h->T[10] = h->T[15] = 1.0;
    }
    IcpRun *r = ln.run_host;
    r->depth_src = ss.smooth_radius > 0 ? ss.sdepth : ss.depth;      // the window-averaged depth when the slot's normals were built smoothed
    r->nmap_tgt = st.nmap;
    r->src_pm = ss.smooth_radius > 0 ? 1 : 0;
    r->scale = (float)scale_src;
    r->md2 = (float)prm->max_dist * (float)prm->max_dist;
    r->mind = (float)ctx->cfg.min_depth;
    r->maxd = (float)ctx->cfg.max_depth;
    r->stride = prm->stride;
    r->Ws = (ctx->cam.W + prm->stride - 1) / prm->stride;
    r->Hs = (ctx->cam.H + prm->stride - 1) / prm->stride;
    r->est_scale = prm->estimate_scale ? 1 : 0;
    h->scale = scale_src;
    r->damping = prm->damping;
    r->eps = prm->eps;
    r->eig_rel = prm->eig_rel;
    // The chain's launch arguments never change (everything per-run sits behind ln.run / ln.state / pinned buffers), so
    // it is captured once per iteration count and replayed: one host call per registration instead of ~25.
    int gi = -1;
    for (int g = 0; g < 4; ++g)
        if (ln.graphs[g] && ln.graph_iters[g] == prm->iters) gi = g;
    if (gi < 0) {
        gi = ln.graph_next;
        ln.graph_next = (ln.graph_next + 1) & 3;
        if (ln.graphs[gi]) { (void)hipGraphExecDestroy(ln.graphs[gi]); ln.graphs[gi] = nullptr; }
        hipGraph_t g = nullptr;
        TL3D_HIP(hipStreamBeginCapture(ln.stream, hipStreamCaptureModeThreadLocal));
        hipError_t ce = hipMemcpyAsync(ln.run, ln.run_host, sizeof(IcpRun), hipMemcpyHostToDevice, ln.stream);
        if (ce == hipSuccess) ce = hipMemcpyAsync(ln.state, ln.host, sizeof(IcpState), hipMemcpyHostToDevice, ln.stream);
        if (ce == hipSuccess) ce = hipMemsetAsync(ln.ticket, 0, 64, ln.stream);      // polled words are re-armed by every replay
        int lrc = TL3D_OK;
        for (int it = 0; ce == hipSuccess && lrc == TL3D_OK && it <= prm->iters; ++it)
            lrc = launch_icp_iteration(ln.stream, ctx->cam, ln.run, it == prm->iters, ln.slab, ln.state, ICP_MAX_BLOCKS, ln.ticket);
        if (ce == hipSuccess && lrc == TL3D_OK) ce = hipMemcpyAsync(ln.host, ln.state, sizeof(IcpState), hipMemcpyDeviceToHost, ln.stream);
        hipError_t ee = hipStreamEndCapture(ln.stream, &g);
        if (ce != hipSuccess || ee != hipSuccess || lrc != TL3D_OK || !g) {
            if (g) (void)hipGraphDestroy(g);
            return set_err(TL3D_E_HIP, "ICP graph capture failed: %s", hipGetErrorString(ce != hipSuccess ? ce : ee));
        }
        hipError_t ie = hipGraphInstantiate(&ln.graphs[gi], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ie != hipSuccess) { ln.graphs[gi] = nullptr; return set_err(TL3D_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie)); }
        ln.graph_iters[gi] = prm->iters;
    }
    TL3D_HIP(hipGraphLaunch(ln.graphs[gi], ln.stream));
    TL3D_HIP(hipEventRecord(ln.ev_done, ln.stream));
    ln.busy = true;
    ln.src_slot = slot_src;
    ln.tgt_slot = slot_tgt;
    return TL3D_OK;
}

int tl3d_icp_collect(tl3d_ctx *ctx, int lane, tl3d_icp_result *out) {
    REQUIRE(ctx && out, TL3D_E_INVALID, "null argument");
    REQUIRE(lane >= 0 && lane < TL3D_ICP_LANES, TL3D_E_INVALID, "lane %d out of range [0,%d)", lane, TL3D_ICP_LANES);
    tl3d_ctx::IcpLane &ln = ctx->icp_lanes[lane];
    REQUIRE(ln.stream && ln.busy, TL3D_E_STATE, "ICP lane %d holds no run", lane);
    TL3D_HIP(hipSetDevice(ctx->device));
    TL3D_HIP(hipStreamSynchronize(ln.stream));
    ln.busy = false;
    const IcpState &h = *ln.host;
    memcpy(out->T, h.T, sizeof(out->T));
    out->n_corr = (int64_t)h.sums[28];
    out->n_src = (int64_t)h.sums[29];
    out->fitness = h.sums[29] > 0 ? h.sums[28] / h.sums[29] : 0.0;
    out->rmse = h.sums[28] > 0 ? sqrt(h.sums[27] / h.sums[28]) : 0.0;
    out->iters_run = h.iters_run;
    out->status = h.status;
    out->scale = h.scale;
    return TL3D_OK;
}

int tl3d_icp_p2plane(tl3d_ctx *ctx, int slot_src, double scale_src, int slot_tgt, const double T_init[16],
                     const tl3d_icp_params *prm, tl3d_icp_result *out) {
    REQUIRE(out != nullptr, TL3D_E_INVALID, "null argument");
    int rc = tl3d_icp_enqueue(ctx, 0, slot_src, scale_src, slot_tgt, T_init, prm);
    if (rc) return rc;
    return tl3d_icp_collect(ctx, 0, out);
}

// ---- host helpers of the decode pipeline (no device work) -----------------------------------------------------------------
int tl3d_host_pack_bgr_rows(uint8_t *dst, const uint8_t *const *rows, int height, int width) {
    REQUIRE(dst && rows && height >= 0 && width >= 0, TL3D_E_INVALID, "bad argument");
    for (int y = 0; y < height; ++y) {
        const uint8_t *__restrict__ s = rows[y];
        uint8_t *__restrict__ d = dst + (size_t)y * width * 3;
        REQUIRE(s != nullptr, TL3D_E_INVALID, "null row %d", y);
        for (int x = 0; x < width; ++x) {
            d[3 * x + 0] = s[4 * x + 2];
            d[3 * x + 1] = s[4 * x + 1];
            d[3 * x + 2] = s[4 * x + 0];
        }
    }
    return TL3D_OK;
}

int tl3d_host_copy_rows(uint8_t *dst, const uint8_t *const *rows, int height, size_t row_bytes) {
    REQUIRE(dst && rows && height >= 0, TL3D_E_INVALID, "bad argument");
    for (int y = 0; y < height; ++y) {
        REQUIRE(rows[y] != nullptr, TL3D_E_INVALID, "null row %d", y);
        memcpy(dst + (size_t)y * row_bytes, rows[y], row_bytes);
    }
    return TL3D_OK;
}

// ---- batched registration: all pairs, all levels, all iterations in one launch (icp_batch_kernel) ----------------------
static void icp_batch_free(tl3d_ctx::IcpBatch &b) {
    if (b.pairs) (void)hipFree(b.pairs);
    if (b.states) (void)hipFree(b.states);
    if (b.sync) (void)hipFree(b.sync);
    if (b.ctl) (void)hipFree(b.ctl);
    if (b.slab) (void)hipFree(b.slab);
    if (b.dbg) (void)hipFree(b.dbg);
    if (b.stage) (void)hipFree(b.stage);
    if (b.pairs_host) (void)hipHostFree(b.pairs_host);
    if (b.states_host) (void)hipHostFree(b.states_host);
    if (b.ctl_host) (void)hipHostFree(b.ctl_host);
    free(b.req_pairs);
    if (b.ev_ready) (void)hipEventDestroy(b.ev_ready);
    if (b.ev_done) (void)hipEventDestroy(b.ev_done);
    if (b.stream) (void)hipStreamDestroy(b.stream);
    memset(&b, 0, sizeof(b));
}

static int icp_batch_reserve(tl3d_ctx *ctx, int n_pairs, size_t slab_doubles) {
    tl3d_ctx::IcpBatch &b = ctx->icp_batch;
    bool ok = true;
    if (!b.stream) {
        ok = hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&b.ev_ready, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipMalloc(&b.ctl, 64) == hipSuccess;
        ok = ok && hipHostMalloc(&b.ctl_host, 64, hipHostMallocDefault) == hipSuccess;
    }
    if (ok && n_pairs > b.cap_pairs) {
        const int cap = n_pairs < 64 ? 64 : n_pairs;
        if (b.pairs) (void)hipFree(b.pairs);
        if (b.states) (void)hipFree(b.states);
        if (b.sync) (void)hipFree(b.sync);
        if (b.pairs_host) (void)hipHostFree(b.pairs_host);
        if (b.states_host) (void)hipHostFree(b.states_host);
        b.pairs = nullptr; b.states = nullptr; b.sync = nullptr; b.pairs_host = nullptr; b.states_host = nullptr;
        b.cap_pairs = 0;
        ok = hipMalloc(&b.pairs, (size_t)cap * sizeof(IcpBatchPair)) == hipSuccess;
        ok = ok && hipMalloc(&b.states, (size_t)cap * sizeof(IcpState)) == hipSuccess;
        int rows = (cap + 63) / 64;
        if (rows < 68) rows = 68;                            // neighbours' lines 4352 B apart at least
        ok = ok && hipMalloc(&b.sync, (size_t)2 * 64 * rows * 64) == hipSuccess;
        if (ok) b.sync_rows = rows;
        ok = ok && hipHostMalloc(&b.pairs_host, (size_t)cap * sizeof(IcpBatchPair), hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc(&b.states_host, (size_t)cap * sizeof(IcpState), hipHostMallocDefault) == hipSuccess;
        if (ok) b.cap_pairs = cap;
    }
    if (ok && slab_doubles > b.cap_slab) {
        if (b.slab) (void)hipFree(b.slab);
        b.slab = nullptr;
        b.cap_slab = 0;
        ok = hipMalloc(&b.slab, slab_doubles * sizeof(double)) == hipSuccess;
        if (ok) b.cap_slab = slab_doubles;
    }
    if (!ok) {
        (void)hipGetLastError();
        icp_batch_free(b);
        return set_err(TL3D_E_NOMEM, "ICP batch: stream / buffer allocation failed");
    }
    return TL3D_OK;
}

int tl3d_icp_batch_enqueue(tl3d_ctx *ctx, const tl3d_icp_pair *pairs, int n_pairs, const tl3d_icp_params *levels, int n_levels) {
    REQUIRE(ctx && pairs && levels, TL3D_E_INVALID, "null argument");
    REQUIRE(n_pairs >= 1 && n_pairs <= (1 << 20), TL3D_E_INVALID, "n_pairs %d out of range", n_pairs);
    REQUIRE(n_levels >= 1 && n_levels <= TL3D_ICP_MAX_LEVELS, TL3D_E_INVALID, "n_levels %d out of range [1,%d]", n_levels, TL3D_ICP_MAX_LEVELS);
    REQUIRE(!ctx->icp_batch.busy, TL3D_E_STATE, "an ICP batch is still uncollected");
    IcpBatchArgs a;
    memset(&a, 0, sizeof(a));
    long long ns_max = 1;
    for (int l = 0; l < n_levels; ++l) {
        const tl3d_icp_params &p = levels[l];
        REQUIRE(p.iters >= 0 && p.iters <= 1000, TL3D_E_INVALID, "iters out of range");
        REQUIRE(p.stride >= 1, TL3D_E_INVALID, "stride must be >= 1");
        REQUIRE(p.max_dist > 0, TL3D_E_INVALID, "max_dist must be positive");
        IcpLevel &L = a.lv[l];
        L.md2 = (float)p.max_dist * (float)p.max_dist;
        L.stride = p.stride;
        L.Ws = (ctx->cam.W + p.stride - 1) / p.stride;
        L.Hs = (ctx->cam.H + p.stride - 1) / p.stride;
        L.iters = p.iters;
        L.est_scale = p.estimate_scale ? 1 : 0;
        L.damping = p.damping;
        L.eps = p.eps;
        L.eig_rel = p.eig_rel;
        const long long ns = (long long)L.Ws * L.Hs;
        if (ns > ns_max) ns_max = ns;
    }
    for (int i = 0; i < n_pairs; ++i) {
        int rc = check_slot(ctx, pairs[i].slot_src, true);
        if (rc) return rc;
        rc = check_slot(ctx, pairs[i].slot_tgt, true);
        if (rc) return rc;
        REQUIRE(ctx->slots[pairs[i].slot_tgt].has_normals, TL3D_E_STATE, "target slot %d has no normal map (call tl3d_build_normals)", pairs[i].slot_tgt);
    }
    // workgroups per pair: a function of the level geometry only, so a pair's sums (and pose) do not depend on the batch it is in
    long long members = (ns_max + ICP_BATCH_SAMPLES_PER_MEMBER - 1) / ICP_BATCH_SAMPLES_PER_MEMBER;
    if (members > ICP_BATCH_MEMBERS_CAP) members = ICP_BATCH_MEMBERS_CAP;
#ifdef TL3D_EXPERIMENTS
    if (const char *e = getenv("TL3D_ICP_MEMBERS")) {
        const long long m = atoll(e);
        if (m >= 1 && m <= ICP_BATCH_MAX_MEMBERS) members = m;
    }
#endif
    TL3D_HIP(hipSetDevice(ctx->device));
    int rc = icp_batch_reserve(ctx, n_pairs, (size_t)n_pairs * (size_t)members * ICP_SLAB);
    if (rc) return rc;
    tl3d_ctx::IcpBatch &b = ctx->icp_batch;
    if (n_pairs > b.req_cap) {
        free(b.req_pairs);
        b.req_pairs = (tl3d_icp_pair *)malloc((size_t)n_pairs * sizeof(tl3d_icp_pair));
        b.req_cap = b.req_pairs ? n_pairs : 0;
        REQUIRE(b.req_pairs != nullptr, TL3D_E_NOMEM, "host allocation failed");
    }
    memcpy(b.req_pairs, pairs, (size_t)n_pairs * sizeof(tl3d_icp_pair));
    memcpy(b.req_levels, levels, (size_t)n_levels * sizeof(tl3d_icp_params));
    b.req_n_levels = n_levels;
    for (int i = 0; i < n_pairs; ++i) {
        const Slot &ss = ctx->slots[pairs[i].slot_src], &st = ctx->slots[pairs[i].slot_tgt];
        b.pairs_host[i].depth_src = ss.smooth_radius > 0 ? ss.sdepth : ss.depth;
        b.pairs_host[i].nmap_tgt = st.nmap;
        b.pairs_host[i].scale = (float)pairs[i].scale_src;
        b.pairs_host[i].src_pm = ss.smooth_radius > 0 ? 1 : 0;
        IcpState &h = b.states_host[i];
        memset(&h, 0, sizeof(h));
        memcpy(h.T, pairs[i].T_init, sizeof(h.T));
        h.scale = pairs[i].scale_src;
    }
    // uploads and normal maps are issued on the main stream: everything issued there so far precedes the batch
    TL3D_HIP(hipEventRecord(b.ev_ready, ctx->stream));
    TL3D_HIP(hipStreamWaitEvent(b.stream, b.ev_ready, 0));
    TL3D_HIP(hipMemcpyAsync(b.pairs, b.pairs_host, (size_t)n_pairs * sizeof(IcpBatchPair), hipMemcpyHostToDevice, b.stream));
    TL3D_HIP(hipMemcpyAsync(b.states, b.states_host, (size_t)n_pairs * sizeof(IcpState), hipMemcpyHostToDevice, b.stream));
    TL3D_HIP(hipMemsetAsync(b.sync, 0, (size_t)2 * 64 * b.sync_rows * 64, b.stream));
    TL3D_HIP(hipMemsetAsync(b.ctl, 0, 64, b.stream));
    a.pairs = b.pairs;
    a.states = b.states;
    a.sync = b.sync;
    a.slab = b.slab;
    a.ctl = b.ctl;
    a.n_pairs = n_pairs;
    a.members = (int)members;
    a.n_levels = n_levels;
    a.sync_rows = b.sync_rows;
    a.mind = (float)ctx->cfg.min_depth;
    a.maxd = (float)ctx->cfg.max_depth;
#ifdef TL3D_EXPERIMENTS
    if (getenv("TL3D_ICP_TRACE")) {                        // per-pass timestamps of pair 0
        if (b.dbg) (void)hipFree(b.dbg);
        b.dbg = nullptr;
        TL3D_HIP(hipMalloc(&b.dbg, (size_t)members * 16 * 8 * sizeof(unsigned long long)));
        TL3D_HIP(hipMemsetAsync(b.dbg, 0, (size_t)members * 16 * 8 * sizeof(unsigned long long), b.stream));
        b.dbg_members = (int)members;
        a.dbg = b.dbg;
    }
    if (getenv("TL3D_ICP_STAGES")) {
        if (b.stage) (void)hipFree(b.stage);
        b.stage = nullptr;
        b.stage_n = (size_t)n_pairs * members;
        TL3D_HIP(hipMalloc(&b.stage, b.stage_n * 16));
        TL3D_HIP(hipMemsetAsync(b.stage, 0, b.stage_n * 16, b.stream));
        a.stage = b.stage;
    }
#endif
    rc = launch_icp_batch(b.stream, ctx->cam, a);
    if (rc) return rc;
    TL3D_HIP(hipMemcpyAsync(b.states_host, b.states, (size_t)n_pairs * sizeof(IcpState), hipMemcpyDeviceToHost, b.stream));
    TL3D_HIP(hipMemcpyAsync(b.ctl_host, b.ctl, 64, hipMemcpyDeviceToHost, b.stream));
    TL3D_HIP(hipEventRecord(b.ev_done, b.stream));
    b.n_pairs = n_pairs;
    b.busy = true;
    return TL3D_OK;
}

// The batch of the last enqueue, registered by the per-iteration kernel: up to TL3D_ICP_LANES pairs side by side, level
// after level; a pair's next level starts from the pose (and scale) its previous level ended with, and a pair stops when a
// level fails or ends with fewer than 8 correspondences -- the rule icp_batch_kernel applies between its levels.
static int icp_batch_fallback(tl3d_ctx *ctx, tl3d_icp_result *out, int n_out) {
    tl3d_ctx::IcpBatch &b = ctx->icp_batch;
    for (int l = 0; l < TL3D_ICP_LANES; ++l)
        REQUIRE(!ctx->icp_lanes[l].busy, TL3D_E_STATE, "the batched registration timed out and ICP lane %d holds an uncollected run: cannot fall back", l);
    std::vector<char> over((size_t)n_out, 0);
    for (int i = 0; i < n_out; ++i) {
        memset(&out[i], 0, sizeof(out[i]));
        memcpy(out[i].T, b.req_pairs[i].T_init, sizeof(out[i].T));
        out[i].scale = b.req_pairs[i].scale_src;
    }
    for (int lv = 0; lv < b.req_n_levels; ++lv)
        for (int i0 = 0; i0 < n_out; i0 += TL3D_ICP_LANES) {
            const int m = n_out - i0 < TL3D_ICP_LANES ? n_out - i0 : TL3D_ICP_LANES;
            for (int k = 0; k < m; ++k) {
                const int i = i0 + k;
                if (over[i]) continue;
                const int rc = tl3d_icp_enqueue(ctx, k, b.req_pairs[i].slot_src, out[i].scale, b.req_pairs[i].slot_tgt, out[i].T, &b.req_levels[lv]);
                if (rc) return rc;
            }
            for (int k = 0; k < m; ++k) {
                const int i = i0 + k;
                if (over[i]) continue;
                const int rc = tl3d_icp_collect(ctx, k, &out[i]);
                if (rc) return rc;
                if (out[i].status == 2 || out[i].n_corr < 8) over[i] = 1;
            }
        }
    ctx->stats.icp_batch_fallback_pairs += (uint64_t)n_out;
    return TL3D_OK;
}

int tl3d_icp_batch_collect(tl3d_ctx *ctx, tl3d_icp_result *out, int n_out) {
    REQUIRE(ctx && out, TL3D_E_INVALID, "null argument");
    tl3d_ctx::IcpBatch &b = ctx->icp_batch;
    REQUIRE(b.stream && b.busy, TL3D_E_STATE, "no ICP batch in flight");
    REQUIRE(n_out == b.n_pairs, TL3D_E_INVALID, "the batch in flight has %d pairs, not %d", b.n_pairs, n_out);
    TL3D_HIP(hipSetDevice(ctx->device));
    TL3D_HIP(hipStreamSynchronize(b.stream));
    b.busy = false;
#ifdef TL3D_EXPERIMENTS
    if (b.dbg && getenv("TL3D_ICP_TRACE")) {
        std::vector<unsigned long long> h((size_t)b.dbg_members * 16 * 8);
        TL3D_HIP(hipMemcpy(h.data(), b.dbg, h.size() * 8, hipMemcpyDeviceToHost));
        FILE *f = fopen(getenv("TL3D_ICP_TRACE"), "w");
        if (f) {
            unsigned long long t0 = ~0ull;
            for (size_t i = 0; i < h.size(); ++i)
                if ((i & 7) < 7 && h[i] && h[i] < t0) t0 = h[i];
            for (int m = 0; m < b.dbg_members; ++m)
                for (int g = 0; g < 16; ++g) {
                    const unsigned long long *r = &h[((size_t)m * 16 + g) * 8];
                    if (!r[0]) continue;
                    fprintf(f, "%d %d %llu", m, g, r[7]);
                    for (int k = 0; k < 7; ++k) fprintf(f, " %lld", r[k] ? (long long)(r[k] - t0) : -1ll);
                    fprintf(f, "\n");
                }
            fclose(f);
        }
    }
    if (b.stage && b.ctl_host[1] != 0 && getenv("TL3D_ICP_STAGES")) {
        std::vector<unsigned> h(b.stage_n * 4);
        TL3D_HIP(hipMemcpy(h.data(), b.stage, h.size() * 4, hipMemcpyDeviceToHost));
        const size_t mem = b.stage_n / (size_t)b.n_pairs;
        size_t hist[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < b.stage_n; ++i) hist[h[i * 4] & 3u]++;
        fprintf(stderr, "stages: never started %zu, started %zu, accumulated %zu, arrived %zu\n", hist[0], hist[1], hist[2], hist[3]);
        const size_t p0 = b.ctl_host[4];
        for (size_t m = 0; m < mem; ++m) {
            const unsigned *r = &h[(p0 * mem + m) * 4];
            fprintf(stderr, "  pair %zu ticket-member %zu: stage %u pass %u block %u arrived-as %u hwid %08x\n", p0, m, r[0] & 15u, r[0] >> 4, r[1], r[2], r[3]);
        }
    }
#endif
    bool timed_out = b.ctl_host[1] != 0;
#ifdef TL3D_EXPERIMENTS
    if (getenv("TL3D_ICP_FORCE_TIMEOUT")) timed_out = true;              // rehearsal of the fallback below
#endif
    if (timed_out) {
        // A wait inside the launch ran into its time bound (the waits need a pair's other workgroups to be running; other
        // work on the chip can, in principle, keep them off it).  The launch has ended; its results are discarded and the
        // whole batch is registered again by the per-iteration kernel on the ICP lanes, which waits for nobody inside a
        // launch: same levels, same chaining rule.
        ctx->stats.icp_batch_timeouts++;
        fprintf(stderr, "libtl3d: batched registration timed out inside its launch (pair %u member %u pass %u level %u iteration %u: %u of its "
                        "workgroups had arrived, %u workgroups started); re-registering %d pairs on the per-iteration kernel\n",
                b.ctl_host[4], b.ctl_host[5], b.ctl_host[6], b.ctl_host[10], b.ctl_host[11], b.ctl_host[8], b.ctl_host[0], n_out);
        return icp_batch_fallback(ctx, out, n_out);
    }
    for (int i = 0; i < n_out; ++i) {
        const IcpState &h = b.states_host[i];
        memcpy(out[i].T, h.T, sizeof(out[i].T));
        out[i].n_corr = (int64_t)h.sums[28];
        out[i].n_src = (int64_t)h.sums[29];
        out[i].fitness = h.sums[29] > 0 ? h.sums[28] / h.sums[29] : 0.0;
        out[i].rmse = h.sums[28] > 0 ? sqrt(h.sums[27] / h.sums[28]) : 0.0;
        out[i].iters_run = h.iters_run;
        out[i].status = h.status;
        out[i].scale = h.scale;
    }
    return TL3D_OK;
}

// ------------------------------------------------------------------------------------------- grids
static int grid_sel(tl3d_ctx *ctx, uint32_t channel, void **p, size_t *bytes) {
    REQUIRE(ctx != nullptr, TL3D_E_INVALID, "null ctx");
    if (channel == TL3D_CH_TSDF) {
        REQUIRE(ctx->tsdf != nullptr, TL3D_E_STATE, "TSDF channel not enabled");
        *p = ctx->tsdf;
        *bytes = ctx->nvox * sizeof(int2);
    } else if (channel == TL3D_CH_CENTROID) {
        REQUIRE(ctx->centroid != nullptr, TL3D_E_STATE, "centroid channel not enabled");
        *p = ctx->centroid;
        *bytes = ctx->nvox * 32;
    } else {
        return set_err(TL3D_E_INVALID, "channel must be exactly one of TL3D_CH_TSDF / TL3D_CH_CENTROID");
    }
    return TL3D_OK;
}

int tl3d_grid_reset(tl3d_ctx *ctx) {
    REQUIRE(ctx != nullptr, TL3D_E_INVALID, "null ctx");
    FLUSH_UPDATES(ctx);
    ctx->grid_epoch++;
    TL3D_HIP(hipSetDevice(ctx->device));
    if (ctx->tsdf) TL3D_HIP(hipMemsetAsync(ctx->tsdf, 0, (size_t)ctx->grid.tsdf_cap << 12, ctx->stream));
    if (ctx->centroid) TL3D_HIP(hipMemsetAsync(ctx->centroid, 0, (size_t)ctx->grid.cen_cap << 14, ctx->stream));
    if (ctx->sparse) {
        const int trc = reset_brick_tables(ctx);
        if (trc) return trc;
    }
    if (ctx->free_cnt) {
        TL3D_HIP(hipMemsetAsync(ctx->free_cnt, 0, (ctx->nvox >> 9) * sizeof(unsigned), ctx->stream));
        const int mrc = mark_free_cnt_write(ctx);
        if (mrc) return mrc;
    }
    ctx->free_dirty = false;
    ctx->tsdf_w_upper = 0;
    ctx->tsdf_w_unknown = false;
    return TL3D_OK;
}

int tl3d_grid_device_ptr(tl3d_ctx *ctx, uint32_t channel, void **ptr, size_t *bytes) {
    REQUIRE(ptr && bytes, TL3D_E_INVALID, "null out pointer");
    if (ctx && channel == TL3D_CH_FREE) {
        REQUIRE(ctx->free_cnt != nullptr, TL3D_E_STATE, "TSDF channel not enabled");
        FLUSH_UPDATES(ctx);                             // the counts as they stand (folded into the records or not: readers add what is pending)
        ctx->free_dirty = true;                         // the caller may write through the pointer (a merge sums them)
        ctx->tsdf_w_unknown = true;
        ctx->grid_epoch++;
        *ptr = ctx->free_cnt;
        *bytes = (ctx->nvox >> 9) * sizeof(unsigned);
        return TL3D_OK;
    }
    if (ctx) REQUIRE(!ctx->sparse, TL3D_E_STATE, "a sparse grid has no dense layout to point at: use tl3d_grid_download / tl3d_grid_pack_bricks");
    if (ctx) {
        FLUSH_AND_FOLD(ctx);
        ctx->grid_epoch++;                  // the caller may write through the pointer (all-reduce)
        if (channel == TL3D_CH_TSDF) ctx->tsdf_w_unknown = true;
    }
    return grid_sel(ctx, channel, ptr, bytes);
}

int tl3d_grid_download(tl3d_ctx *ctx, uint32_t channel, void *out, size_t bytes) {
    void *p;
    size_t nb;
    int rc = grid_sel(ctx, channel, &p, &nb);
    if (rc) return rc;
    FLUSH_AND_FOLD(ctx);
    REQUIRE(out && bytes == nb, TL3D_E_INVALID, "buffer is %zu B, grid channel is %zu B", bytes, nb);
    TL3D_HIP(hipSetDevice(ctx->device));
    if (ctx->sparse) {
        // the dense image of the channel (what a dense grid would hold, free-space counts folded in): bricks gathered through the table
        void *tmp = nullptr;
        void *dst = out;
        const bool dev_out = is_device_ptr(out);
        if (!dev_out) {
            if (hipMalloc(&tmp, nb) != hipSuccess) return set_err(TL3D_E_NOMEM, "dense image of the sparse grid (%zu B) does not fit", nb);
            dst = tmp;
        }
        rc = launch_brick_rows(ctx->stream, ctx->grid, 0, channel == TL3D_CH_TSDF, p, nullptr, (long long)(ctx->nvox >> 9), dst, true);
        hipError_t e = hipSuccess;
        if (rc == TL3D_OK && !dev_out) e = hipMemcpyAsync(out, tmp, nb, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (tmp) (void)hipFree(tmp);
        if (rc) return rc;
        TL3D_HIP(e);
        return TL3D_OK;
    }
    TL3D_HIP(hipMemcpyAsync(out, p, nb, hipMemcpyDefault, ctx->stream));
    TL3D_HIP(hipStreamSynchronize(ctx->stream));
    return TL3D_OK;
}

int tl3d_grid_upload(tl3d_ctx *ctx, uint32_t channel, const void *in, size_t bytes) {
    void *p;
    size_t nb;
    int rc = grid_sel(ctx, channel, &p, &nb);
    if (rc) return rc;
    FLUSH_UPDATES(ctx);
    ctx->grid_epoch++;
    REQUIRE(in && bytes == nb, TL3D_E_INVALID, "buffer is %zu B, grid channel is %zu B", bytes, nb);
    TL3D_HIP(hipSetDevice(ctx->device));
    if (channel == TL3D_CH_TSDF) {
        ctx->tsdf_w_unknown = true;
        if (ctx->free_cnt) {                                  // the upload replaces the channel
            TL3D_HIP(hipMemsetAsync(ctx->free_cnt, 0, (ctx->nvox >> 9) * sizeof(unsigned), ctx->stream));
            const int mrc = mark_free_cnt_write(ctx);
            if (mrc) return mrc;
        }
        ctx->free_dirty = false;
    }
    if (ctx->sparse) {
        // the channel becomes the dense image `in`: bricks that hold anything get records, the others' records are cleared
        void *tmp = nullptr;
        const void *src = in;
        if (!is_device_ptr(in)) {
            if (hipMalloc(&tmp, nb) != hipSuccess) return set_err(TL3D_E_NOMEM, "grid staging alloc failed");
            hipError_t e = hipMemcpyAsync(tmp, in, nb, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) { (void)hipFree(tmp); return set_err(TL3D_E_HIP, "grid upload failed"); }
            src = tmp;
        }
        rc = launch_brick_rows(ctx->stream, ctx->grid, 1, channel == TL3D_CH_TSDF, p, nullptr, (long long)(ctx->nvox >> 9), const_cast<void *>(src), false);
        (void)hipStreamSynchronize(ctx->stream);
        if (tmp) (void)hipFree(tmp);
        return rc;
    }
    TL3D_HIP(hipMemcpyAsync(p, in, nb, hipMemcpyDefault, ctx->stream));
    TL3D_HIP(hipStreamSynchronize(ctx->stream));
    return TL3D_OK;
}

int tl3d_grid_add(tl3d_ctx *ctx, uint32_t channel, const void *other, size_t bytes) {
    void *p;
    size_t nb;
    int rc = grid_sel(ctx, channel, &p, &nb);
    if (rc) return rc;
    FLUSH_AND_FOLD(ctx);
    ctx->grid_epoch++;
    REQUIRE(other && bytes == nb, TL3D_E_INVALID, "buffer is %zu B, grid channel is %zu B", bytes, nb);
    TL3D_HIP(hipSetDevice(ctx->device));
    const void *src = other;
    void *tmp = nullptr;
    if (!is_device_ptr(other)) {
        if (hipMalloc(&tmp, nb) != hipSuccess) return set_err(TL3D_E_NOMEM, "grid staging alloc failed");
        hipError_t e = hipMemcpyAsync(tmp, other, nb, hipMemcpyDefault, ctx->stream);
        if (e != hipSuccess) { (void)hipFree(tmp); return set_err(TL3D_E_HIP, "grid upload failed"); }
        src = tmp;
    }
    if (channel == TL3D_CH_TSDF) {
        // the merged weights must keep the int32 sums in range: largest weight here + largest weight there
        long long wa = ctx->tsdf_w_upper, wb = 0;
        if (ctx->tsdf_w_unknown) rc = measure_max_weight(ctx, ctx->tsdf, &wa);
        if (rc == TL3D_OK) rc = measure_max_weight(ctx, (const int2 *)src, &wb);
        if (rc == TL3D_OK && wa + wb > TL3D_TSDF_MAX_WEIGHT)
            rc = set_err(TL3D_E_STATE, "merging grids with up to %lld + %lld observations per voxel could overflow the int32 TSDF sums (limit %d)",
                         wa, wb, TL3D_TSDF_MAX_WEIGHT);
        if (rc == TL3D_OK) {
            ctx->tsdf_w_upper = wa + wb;
            ctx->tsdf_w_unknown = false;
            rc = ctx->sparse ? launch_brick_rows(ctx->stream, ctx->grid, 2, true, p, nullptr, (long long)(ctx->nvox >> 9), const_cast<void *>(src), false)
                             : launch_add_i32(ctx->stream, (int *)p, (const int *)src, nb / 4);
        }
    } else
        rc = ctx->sparse ? launch_brick_rows(ctx->stream, ctx->grid, 2, false, p, nullptr, (long long)(ctx->nvox >> 9), const_cast<void *>(src), false)
                         : launch_add_u64(ctx->stream, (unsigned long long *)p, (const unsigned long long *)src, nb / 8);
    if (tmp) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(tmp);
    }
    return rc;
}

int tl3d_grid_touched_bricks(tl3d_ctx *ctx, uint32_t channels, uint8_t *map_dev, int64_t n_bricks) {
    REQUIRE(ctx && map_dev, TL3D_E_INVALID, "null argument");
    const bool counts_apart = (channels & TL3D_CH_FREE) != 0;      // the free-space counts travel on their own (tl3d.h)
    const bool sub = (channels & TL3D_CH_SUB) != 0;                 // one byte per 4x4x4 sub-brick
    channels &= ~(TL3D_CH_FREE | TL3D_CH_SUB);
    if (channels == 0) channels = (ctx->tsdf ? TL3D_CH_TSDF : 0u) | (ctx->centroid ? TL3D_CH_CENTROID : 0u);
    REQUIRE((channels & ~(TL3D_CH_TSDF | TL3D_CH_CENTROID)) == 0, TL3D_E_INVALID, "bad channel mask 0x%x", channels);
    if (channels & TL3D_CH_TSDF) REQUIRE(ctx->tsdf != nullptr, TL3D_E_STATE, "TSDF channel not enabled");
    if (channels & TL3D_CH_CENTROID) REQUIRE(ctx->centroid != nullptr, TL3D_E_STATE, "centroid channel not enabled");
    REQUIRE(n_bricks == (int64_t)(ctx->nvox >> 9) * (sub ? 8 : 1), TL3D_E_INVALID, "the grid has %zu %sbricks, the map %lld", (ctx->nvox >> 9) * (sub ? 8 : 1), sub ? "sub-" : "",
            (long long)n_bricks);
    REQUIRE(is_device_ptr(map_dev), TL3D_E_INVALID, "the brick map must be device memory");
    if (counts_apart) FLUSH_UPDATES(ctx);
    else FLUSH_AND_FOLD(ctx);
    TL3D_HIP(hipSetDevice(ctx->device));
    return launch_touched_bricks(ctx->stream, ctx->grid, (channels & TL3D_CH_TSDF) ? ctx->tsdf : nullptr, (channels & TL3D_CH_CENTROID) ? ctx->centroid : nullptr,
                                 (unsigned)(ctx->nvox >> 9), map_dev, sub);
}

static int brick_rows(tl3d_ctx *ctx, uint32_t channel, const uint32_t *bricks_dev, int64_t n, void *packed_dev, bool pack) {
    void *p;
    size_t nb;
    const bool counts_apart = (channel & TL3D_CH_FREE) != 0;       // records only: pending free-space counts stay pending
    const bool sub = (channel & TL3D_CH_SUB) != 0;                  // rows are 4x4x4 sub-bricks, ids = brick * 8 + sub-brick
    channel &= ~(TL3D_CH_FREE | TL3D_CH_SUB);
    int rc = grid_sel(ctx, channel, &p, &nb);
    if (rc) return rc;
    REQUIRE(n >= 0 && n <= (int64_t)(ctx->nvox >> 9) * (sub ? 8 : 1), TL3D_E_INVALID, "brick count %lld out of range", (long long)n);
    if (n == 0) return TL3D_OK;
    REQUIRE(bricks_dev && packed_dev && is_device_ptr(bricks_dev) && is_device_ptr(packed_dev), TL3D_E_INVALID, "brick list and block must be device memory");
    if (counts_apart) FLUSH_UPDATES(ctx);
    else FLUSH_AND_FOLD(ctx);
    ctx->grid_epoch++;
    TL3D_HIP(hipSetDevice(ctx->device));
    if (!pack && channel == TL3D_CH_TSDF) ctx->tsdf_w_unknown = true;      // the records now hold what the caller summed
    return launch_brick_rows(ctx->stream, ctx->grid, pack ? 0 : 1, channel == TL3D_CH_TSDF, p, bricks_dev, n, packed_dev, false, sub);
}

int tl3d_grid_pack_bricks(tl3d_ctx *ctx, uint32_t channel, const uint32_t *bricks_dev, int64_t n, void *packed_dev) {
    return brick_rows(ctx, channel, bricks_dev, n, packed_dev, true);
}

int tl3d_grid_unpack_bricks(tl3d_ctx *ctx, uint32_t channel, const uint32_t *bricks_dev, int64_t n, const void *packed_dev) {
    return brick_rows(ctx, channel, bricks_dev, n, const_cast<void *>(packed_dev), false);
}

int tl3d_grid_max_weight(tl3d_ctx *ctx, int64_t *out) {
    REQUIRE(ctx && out, TL3D_E_INVALID, "null argument");
    REQUIRE(ctx->tsdf != nullptr, TL3D_E_STATE, "TSDF channel not enabled");
    FLUSH_UPDATES(ctx);                                 // (pending free-space counts are added per brick by the kernel: no fold needed)
    TL3D_HIP(hipSetDevice(ctx->device));
    long long w = 0;
    const int rc = measure_max_weight(ctx, ctx->tsdf, &w);
    if (rc) return rc;
    ctx->tsdf_w_upper = w;
    ctx->tsdf_w_unknown = false;
    *out = (int64_t)w;
    return TL3D_OK;
}

// ------------------------------------------------------------------------------------------- RCCL merge (no torch)
int tl3d_rccl_unique_id(uint8_t id_out[TL3D_RCCL_ID_BYTES]) {
    REQUIRE(id_out != nullptr, TL3D_E_INVALID, "null out pointer");
    int rc = rccl_load();
    if (rc) return rc;
    RcclId id;
    memset(&id, 0, sizeof(id));
    const int nrc = g_rccl.GetUniqueId(&id);
    REQUIRE(nrc == 0, TL3D_E_HIP, "ncclGetUniqueId failed: %s", rccl_msg(nrc));
    memcpy(id_out, id.bytes, TL3D_RCCL_ID_BYTES);
    return TL3D_OK;
}

int tl3d_rccl_init(tl3d_ctx *ctx, int world, int rank, const uint8_t id_in[TL3D_RCCL_ID_BYTES]) {
    REQUIRE(ctx && id_in, TL3D_E_INVALID, "null argument");
    REQUIRE(world >= 1 && rank >= 0 && rank < world, TL3D_E_INVALID, "bad rank %d of %d", rank, world);
    REQUIRE(ctx->rccl_comm == nullptr, TL3D_E_STATE, "context already joined a communicator");
    int rc = rccl_load();
    if (rc) return rc;
    TL3D_HIP(hipSetDevice(ctx->device));
    RcclId id;
    memcpy(id.bytes, id_in, TL3D_RCCL_ID_BYTES);
    // ncclResult_t ncclCommInitRank(ncclComm_t*, int nranks, ncclUniqueId commId /* 128-byte struct BY VALUE */, int rank)
    typedef int (*init_fn)(void **, int, RcclId, int);
    void *comm = nullptr;
    const int nrc = ((init_fn)(void *)g_rccl.CommInitRank)(&comm, world, id, rank);
    REQUIRE(nrc == 0 && comm, TL3D_E_HIP, "ncclCommInitRank failed: %s", rccl_msg(nrc));
    ctx->rccl_comm = comm;
    ctx->rccl_world = world;
    return TL3D_OK;
}

// A rank that fails BETWEEN the collectives of a merge must not simply return: its peers would sit in the next all-reduce for ever.
// The communicator is aborted (the peers' pending and later collectives on it end with an error) and the context forgets it.
static int rccl_abandon(tl3d_ctx *ctx, int code) {
    if (ctx->rccl_comm) {
        if (g_rccl.CommAbort) (void)g_rccl.CommAbort(ctx->rccl_comm);
        else if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->rccl_comm);
        ctx->rccl_comm = nullptr;
    }
    return code;
}
#define REQUIRE_OR_ABANDON(cond_, code_, ...) \
    do {                                       \
        if (!(cond_)) return rccl_abandon(ctx, set_err(code_, __VA_ARGS__)); \
    } while (0)

int tl3d_allreduce_grid(tl3d_ctx *ctx, uint32_t channels) {
    REQUIRE(ctx != nullptr, TL3D_E_INVALID, "null ctx");
    REQUIRE(ctx->rccl_comm != nullptr, TL3D_E_STATE, "call tl3d_rccl_init first");
    if (channels == 0) channels = (ctx->tsdf ? TL3D_CH_TSDF : 0u) | (ctx->centroid ? TL3D_CH_CENTROID : 0u);
    REQUIRE((channels & ~(TL3D_CH_TSDF | TL3D_CH_CENTROID)) == 0, TL3D_E_INVALID, "bad channel mask 0x%x", channels);
    if (channels & TL3D_CH_TSDF) REQUIRE(ctx->tsdf != nullptr, TL3D_E_STATE, "TSDF channel not enabled");
    if (channels & TL3D_CH_CENTROID) REQUIRE(ctx->centroid != nullptr, TL3D_E_STATE, "centroid channel not enabled");
    FLUSH_AND_FOLD(ctx);
    TL3D_HIP(hipSetDevice(ctx->device));
    ctx->grid_epoch++;
    if (channels & TL3D_CH_TSDF) {
        // int32 headroom over all ranks: the sum of the ranks' largest voxel weights bounds the merged grid's
        long long w = 0;
        int rc = measure_max_weight(ctx, ctx->tsdf, &w);
        if (rc) return rccl_abandon(ctx, rc);
        long long *d_w = nullptr;
        REQUIRE_OR_ABANDON(hipMalloc(&d_w, sizeof(long long)) == hipSuccess, TL3D_E_NOMEM, "alloc failed");
        hipError_t e = hipMemcpyAsync(d_w, &w, sizeof(w), hipMemcpyHostToDevice, ctx->stream);
        int nrc = e == hipSuccess ? g_rccl.AllReduce(d_w, d_w, 1, 4 /* ncclInt64 */, 0 /* ncclSum */, ctx->rccl_comm, ctx->stream) : -1;
        long long total = 0;
        if (nrc == 0) e = hipMemcpyAsync(&total, d_w, sizeof(total), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_w);
        REQUIRE_OR_ABANDON(nrc == 0 && e == hipSuccess, TL3D_E_HIP, "weight all-reduce failed: %s", nrc ? rccl_msg(nrc) : hipGetErrorString(e));
        REQUIRE_OR_ABANDON(total <= TL3D_TSDF_MAX_WEIGHT, TL3D_E_STATE,
                "the merged TSDF grid could hold %lld observations per voxel (limit %d): merge more often or extract between scans", total,
                TL3D_TSDF_MAX_WEIGHT);
        ctx->tsdf_w_upper = total;
        ctx->tsdf_w_unknown = false;
    }
    // Which bricks does ANY rank hold something in?  One byte per brick, MAX all-reduce: every rank gets the same set.  When it
    // is less than half of the grid only those bricks' records travel (packed, summed, unpacked); a frame-sharded run of a
    // few dozen frames per rank touches a few per cent of a 1024^3 grid.
    const size_t nbr = ctx->nvox >> 9;
    unsigned char *d_map = nullptr;
    REQUIRE_OR_ABANDON(hipMalloc(&d_map, nbr) == hipSuccess, TL3D_E_NOMEM, "brick map alloc failed");
    std::vector<unsigned char> h_map(nbr);
    std::vector<unsigned> h_idx;
    int rc = TL3D_OK;
    {
        hipError_t e = hipMemsetAsync(d_map, 0, nbr, ctx->stream);
        if (e == hipSuccess)
            rc = launch_touched_bricks(ctx->stream, ctx->grid, (channels & TL3D_CH_TSDF) ? ctx->tsdf : nullptr, (channels & TL3D_CH_CENTROID) ? ctx->centroid : nullptr,
                                       (unsigned)nbr, d_map);
        int nrc = (e == hipSuccess && rc == TL3D_OK) ? g_rccl.AllReduce(d_map, d_map, nbr, 1 /* ncclUint8 */, 2 /* ncclMax */, ctx->rccl_comm, ctx->stream) : -1;
        if (nrc == 0) e = hipMemcpyAsync(h_map.data(), d_map, nbr, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_map);
        REQUIRE_OR_ABANDON(rc == TL3D_OK && nrc == 0 && e == hipSuccess, TL3D_E_HIP, "brick-map all-reduce failed: %s", nrc > 0 ? rccl_msg(nrc) : hipGetErrorString(e));
    }
    for (size_t b = 0; b < nbr; ++b)
        if (h_map[b]) h_idx.push_back((unsigned)b);
    const bool sparse = ctx->sparse || h_idx.size() * 2 < nbr;
    if (ctx->sparse && (channels & TL3D_CH_TSDF) && ctx->free_cnt) {
        // bricks without records carry their free-space observations as a count: summed like the records (4 B per brick)
        const int nrc = g_rccl.AllReduce(ctx->free_cnt, ctx->free_cnt, nbr, 3 /* ncclUint32 */, 0, ctx->rccl_comm, ctx->stream);
        REQUIRE_OR_ABANDON(nrc == 0, TL3D_E_HIP, "ncclAllReduce (free-space counts) failed: %s", rccl_msg(nrc));
    }
    ctx->stats.merge_bricks_sent += sparse ? h_idx.size() : nbr;
    ctx->stats.merge_bricks_total += nbr;
    if (!sparse) {
        if (channels & TL3D_CH_TSDF) {
            const int nrc = g_rccl.AllReduce(ctx->tsdf, ctx->tsdf, ctx->nvox * 2, 2 /* ncclInt32 */, 0, ctx->rccl_comm, ctx->stream);
            REQUIRE_OR_ABANDON(nrc == 0, TL3D_E_HIP, "ncclAllReduce (TSDF) failed: %s", rccl_msg(nrc));
        }
        if (channels & TL3D_CH_CENTROID) {
            const int nrc = g_rccl.AllReduce(ctx->centroid, ctx->centroid, ctx->nvox * 4, 5 /* ncclUint64 */, 0, ctx->rccl_comm, ctx->stream);
            REQUIRE_OR_ABANDON(nrc == 0, TL3D_E_HIP, "ncclAllReduce (centroid) failed: %s", rccl_msg(nrc));
        }
        REQUIRE_OR_ABANDON(hipStreamSynchronize(ctx->stream) == hipSuccess, TL3D_E_HIP, "sync failed");
        return TL3D_OK;
    }
    if (h_idx.empty()) {                                 // (the free-space counts' all-reduce may still be in flight)
        REQUIRE_OR_ABANDON(hipStreamSynchronize(ctx->stream) == hipSuccess, TL3D_E_HIP, "sync failed");
        return TL3D_OK;
    }
    unsigned *d_idx = nullptr;
    void *d_pack = nullptr;
    const size_t n = h_idx.size();
    const size_t row = (channels & TL3D_CH_CENTROID) ? 16384 : 4096;
    if (hipMalloc(&d_idx, n * sizeof(unsigned)) != hipSuccess || hipMalloc(&d_pack, n * row) != hipSuccess) {
        (void)hipGetLastError();
        if (d_idx) (void)hipFree(d_idx);
        return rccl_abandon(ctx, set_err(TL3D_E_NOMEM, "merge staging alloc (%zu B) failed", n * row));
    }
    hipError_t e = hipMemcpyAsync(d_idx, h_idx.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream);
    int nrc = 0;
    if (e == hipSuccess && (channels & TL3D_CH_TSDF)) {
        rc = launch_brick_rows(ctx->stream, ctx->grid, 0, true, ctx->tsdf, d_idx, (long long)n, d_pack, false);
        if (rc == TL3D_OK) nrc = g_rccl.AllReduce(d_pack, d_pack, n * 1024, 2 /* ncclInt32 */, 0, ctx->rccl_comm, ctx->stream);
        if (rc == TL3D_OK && nrc == 0) rc = launch_brick_rows(ctx->stream, ctx->grid, 1, true, ctx->tsdf, d_idx, (long long)n, d_pack, false);
    }
    if (e == hipSuccess && rc == TL3D_OK && nrc == 0 && (channels & TL3D_CH_CENTROID)) {
        rc = launch_brick_rows(ctx->stream, ctx->grid, 0, false, ctx->centroid, d_idx, (long long)n, d_pack, false);
        if (rc == TL3D_OK) nrc = g_rccl.AllReduce(d_pack, d_pack, n * 2048, 5 /* ncclUint64 */, 0, ctx->rccl_comm, ctx->stream);
        if (rc == TL3D_OK && nrc == 0) rc = launch_brick_rows(ctx->stream, ctx->grid, 1, false, ctx->centroid, d_idx, (long long)n, d_pack, false);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_idx);
    (void)hipFree(d_pack);
    REQUIRE_OR_ABANDON(rc == TL3D_OK && nrc == 0 && e == hipSuccess, TL3D_E_HIP, "sparse grid all-reduce failed: %s", nrc ? rccl_msg(nrc) : hipGetErrorString(e));
    return TL3D_OK;
}

// ------------------------------------------------------------------------------------------- extraction
int tl3d_extract(tl3d_ctx *ctx, int mode, int min_count, int min_weight, double max_abs_tsdf, float *out_xyz,
                 uint8_t *out_rgb, int64_t cap, int64_t *out_n) {
    REQUIRE(ctx && out_n, TL3D_E_INVALID, "null argument");
    REQUIRE(mode == TL3D_EXTRACT_CENTROID || mode == TL3D_EXTRACT_TSDF, TL3D_E_INVALID, "bad mode %d", mode);
    if (mode == TL3D_EXTRACT_CENTROID) REQUIRE(ctx->centroid != nullptr, TL3D_E_STATE, "centroid channel not enabled");
    if (mode == TL3D_EXTRACT_TSDF) REQUIRE(ctx->tsdf != nullptr, TL3D_E_STATE, "TSDF channel not enabled");
    FLUSH_AND_FOLD(ctx);
    TL3D_HIP(hipSetDevice(ctx->device));
    const int nblocks = (int)((ctx->nvox + EXTRACT_CHUNK - 1) / EXTRACT_CHUNK);
    int rc = ensure_scratch_blocks(ctx, (size_t)nblocks + 1);
    if (rc) return rc;
    unsigned long long total = 0;
    const bool reuse = ctx->ext_valid && ctx->ext_epoch == ctx->grid_epoch && ctx->ext_mode == mode && ctx->ext_min_count == min_count &&
                       ctx->ext_min_weight == min_weight && ctx->ext_max_abs == max_abs_tsdf;
    if (reuse) {                                        // the size query just before this call already counted and scanned
        total = ctx->ext_total;
    } else {
        ctx->ext_valid = false;
        rc = launch_extract_count(ctx->stream, ctx->grid, mode, min_count, min_weight, max_abs_tsdf, ctx->tsdf, ctx->centroid, ctx->block_counts, nblocks);
        if (rc) return rc;
        rc = launch_scan(ctx->stream, ctx->block_counts, ctx->block_offsets, nblocks, ctx->block_offsets + nblocks);
        if (rc) return rc;
        TL3D_HIP(hipMemcpyAsync(&total, ctx->block_offsets + nblocks, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
        TL3D_HIP(hipStreamSynchronize(ctx->stream));
        ctx->ext_valid = true; ctx->ext_epoch = ctx->grid_epoch; ctx->ext_total = total;
        ctx->ext_mode = mode; ctx->ext_min_count = min_count; ctx->ext_min_weight = min_weight; ctx->ext_max_abs = max_abs_tsdf;
    }
    *out_n = (int64_t)total;
    if (!out_xyz || !out_rgb) return TL3D_OK;
    if ((int64_t)total > cap) return set_err(TL3D_E_CAPACITY, "need %llu points, capacity %lld", total, (long long)cap);
    if (total == 0) return TL3D_OK;
    const bool direct = is_device_ptr(out_xyz) && is_device_ptr(out_rgb);
    float *dxyz = out_xyz;
    uint8_t *drgb = out_rgb;
    if (!direct) {
        if (hipMalloc(&dxyz, total * 12) != hipSuccess) return set_err(TL3D_E_NOMEM, "output staging alloc failed");
        if (hipMalloc(&drgb, total * 3) != hipSuccess) { (void)hipFree(dxyz); return set_err(TL3D_E_NOMEM, "output staging alloc failed"); }
    }
    rc = launch_extract_write(ctx->stream, ctx->grid, mode, min_count, min_weight, max_abs_tsdf, ctx->tsdf, ctx->centroid,
                              ctx->block_offsets, nblocks, dxyz, drgb, total);
    hipError_t e = hipSuccess;
    if (rc == TL3D_OK && !direct) {
        e = hipMemcpyAsync(out_xyz, dxyz, total * 12, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_rgb, drgb, total * 3, hipMemcpyDeviceToHost, ctx->stream);
    }
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (!direct) { (void)hipFree(dxyz); (void)hipFree(drgb); }
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return set_err(TL3D_E_HIP, "extract copy/sync failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return TL3D_OK;
}

// ------------------------------------------------------------------------------------------- outlier filter
int tl3d_statistical_outlier(tl3d_ctx *ctx, const float *xyz, int64_t n, int nb_neighbors, double std_ratio,
                             double cell_size, uint8_t *keep_out, int64_t *out_kept) {
    REQUIRE(ctx && keep_out && out_kept, TL3D_E_INVALID, "null argument");
    REQUIRE(n >= 0 && (n == 0 || xyz), TL3D_E_INVALID, "bad point list");
    REQUIRE(nb_neighbors >= 1 && nb_neighbors <= 64, TL3D_E_INVALID, "nb_neighbors must be in [1,64]");
    REQUIRE(cell_size > 0, TL3D_E_INVALID, "cell_size must be positive");
    *out_kept = 0;
    if (n == 0) return TL3D_OK;
    TL3D_HIP(hipSetDevice(ctx->device));
    const bool din = is_device_ptr(xyz), dout = is_device_ptr(keep_out);
    float *tx = nullptr;
    uint8_t *tk = nullptr;
    const float *dx = xyz;
    uint8_t *dk = keep_out;
    if (!din) {
        if (hipMalloc(&tx, (size_t)n * 12) != hipSuccess) return set_err(TL3D_E_NOMEM, "staging alloc failed");
        if (hipMemcpyAsync(tx, xyz, (size_t)n * 12, hipMemcpyDefault, ctx->stream) != hipSuccess) { (void)hipFree(tx); return set_err(TL3D_E_HIP, "upload failed"); }
        dx = tx;
    }
    if (!dout) {
        if (hipMalloc(&tk, (size_t)n) != hipSuccess) { if (tx) (void)hipFree(tx); return set_err(TL3D_E_NOMEM, "staging alloc failed"); }
        dk = tk;
    }
    long long kept = 0;
    int rc = sor_run(ctx, dx, n, nb_neighbors, std_ratio, cell_size, dk, &kept);
    if (rc == TL3D_OK && !dout) {
        if (hipMemcpyAsync(keep_out, dk, (size_t)n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = set_err(TL3D_E_HIP, "download failed");
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (tx) (void)hipFree(tx);
    if (tk) (void)hipFree(tk);
    if (rc) return rc;
    *out_kept = kept;
    return TL3D_OK;
}

// ------------------------------------------------------------------------------------------- measurement
int tl3d_set_profile(tl3d_ctx *ctx, int count_records, int time_kernels) {
    REQUIRE(ctx != nullptr, TL3D_E_INVALID, "null ctx");
    FLUSH_UPDATES(ctx);
    ctx->count_records = count_records != 0;
    ctx->time_kernels = time_kernels != 0;
    return TL3D_OK;
}

int tl3d_set_normal_smoothing(tl3d_ctx *ctx, int radius) {
    REQUIRE(ctx != nullptr, TL3D_E_INVALID, "null ctx");
    REQUIRE(radius >= 0 && radius <= 8, TL3D_E_INVALID, "smoothing radius %d out of range [0, 8]", radius);
    ctx->normal_radius = radius;
    return TL3D_OK;
}

int tl3d_set_tsdf_pairing(tl3d_ctx *ctx, int on) {
    REQUIRE(ctx != nullptr, TL3D_E_INVALID, "null ctx");
    FLUSH_UPDATES(ctx);
    ctx->tsdf_pairing = on != 0;
    return TL3D_OK;
}

int tl3d_get_stats(tl3d_ctx *ctx, tl3d_stats *out) {
    REQUIRE(ctx && out, TL3D_E_INVALID, "null argument");
    FLUSH_UPDATES(ctx);
    TL3D_HIP(hipSetDevice(ctx->device));
    unsigned long long h[16];
    std::vector<unsigned long long> hc(256 * 8);
    TL3D_HIP(hipMemcpyAsync(h, ctx->d_counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    TL3D_HIP(hipMemcpyAsync(hc.data(), ctx->d_cen_counters, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    TL3D_HIP(hipStreamSynchronize(ctx->stream));
    h[0] = h[1] = 0;
    unsigned long long cen_updates = 0;
    for (int i = 0; i < 256; ++i) { h[0] += hc[(size_t)i * 8]; h[1] += hc[(size_t)i * 8 + 1]; cen_updates += hc[(size_t)i * 8 + 2]; }
    for (int i = 0; i < ctx->ktimers_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->ktimers[i].a, ctx->ktimers[i].b) == hipSuccess) {
            ctx->stats.tsdf_kernel_ms += ms;
            ctx->stats.tsdf_kernel_timed += (uint64_t)ctx->ktimers[i].launches;
        }
    }
    ctx->ktimers_used = 0;
    ctx->stats.centroid_points = h[0];
    ctx->stats.centroid_dropped = h[1];
    ctx->stats.centroid_record_updates = cen_updates;
    ctx->stats.tsdf_records_read = h[2];
    ctx->stats.tsdf_records_written = h[3];
    ctx->stats.tsdf_bricks_visited = h[4];
    ctx->stats.tsdf_bricks_free = h[5];
    ctx->stats.tsdf_bricks_free_counted = h[6];
    ctx->stats.tsdf_batch_bricks = h[7];
#ifdef TL3D_EXPERIMENTS
    if (h[14]) fprintf(stderr, "[tl3d exp] update kernel, per wave (s_memtime ticks): setup %.0f  pair loop %.0f  record update %.0f  lifetime %.0f;  bricks per wave %.2f, pairs per brick %.1f, waves %llu; longest wave %.0f x launches\n",
                       (double)h[8] / h[14], (double)h[9] / h[14], (double)h[10] / h[14], (double)h[11] / h[14], (double)h[12] / h[14], h[12] ? (double)h[13] / h[12] : 0.0, h[14], (double)h[15]);
    if (h[14]) tsdf_debug_print_spans(ctx->tsdf_max_blocks * 4 > 16384 ? 16384 : ctx->tsdf_max_blocks * 4);
#endif
    if (ctx->brick_tabs) {
        unsigned cur[4] = {0, 0, 0, 0};
        TL3D_HIP(hipMemcpy(cur, ctx->grid.cursors, sizeof(cur), hipMemcpyDeviceToHost));
        ctx->stats.pool_slots_tsdf = ctx->tsdf ? (cur[0] < ctx->grid.tsdf_cap ? cur[0] : ctx->grid.tsdf_cap) : 0;
        ctx->stats.pool_slots_centroid = ctx->centroid ? (cur[2] < ctx->grid.cen_cap ? cur[2] : ctx->grid.cen_cap) : 0;
        ctx->stats.pool_refused = (uint64_t)cur[1] + cur[3];
    }
    *out = ctx->stats;
    return TL3D_OK;
}

int tl3d_reset_stats(tl3d_ctx *ctx) {
    REQUIRE(ctx != nullptr, TL3D_E_INVALID, "null ctx");
    FLUSH_UPDATES(ctx);
    TL3D_HIP(hipSetDevice(ctx->device));
    TL3D_HIP(hipStreamSynchronize(ctx->stream));
    TL3D_HIP(hipMemsetAsync(ctx->d_counters, 0, 16 * sizeof(unsigned long long), ctx->stream));
    TL3D_HIP(hipMemsetAsync(ctx->d_cen_counters, 0, 256 * 8 * sizeof(unsigned long long), ctx->stream));
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    ctx->ktimers_used = 0;
    return TL3D_OK;
}

int tl3d_event_record(tl3d_ctx *ctx, int which) {
    REQUIRE(ctx != nullptr && (which == 0 || which == 1), TL3D_E_INVALID, "bad argument");
    FLUSH_UPDATES(ctx);
    TL3D_HIP(hipSetDevice(ctx->device));
    TL3D_HIP(hipEventRecord(ctx->ev[which], ctx->stream));
    return TL3D_OK;
}

int tl3d_event_elapsed_ms(tl3d_ctx *ctx, float *ms) {
    REQUIRE(ctx && ms, TL3D_E_INVALID, "null argument");
    TL3D_HIP(hipSetDevice(ctx->device));
    TL3D_HIP(hipEventSynchronize(ctx->ev[1]));
    TL3D_HIP(hipEventElapsedTime(ms, ctx->ev[0], ctx->ev[1]));
    return TL3D_OK;
}

}  // extern "C"
