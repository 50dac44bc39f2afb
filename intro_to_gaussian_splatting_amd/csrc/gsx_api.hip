// C ABI of libgsx.so (see include/gsx.h): argument validation, workspace carving and the
// stream-ordered launch sequence.  No device memory is allocated here and nothing is retained
// between calls; errors are returned as codes with a thread-local message.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "gsx_internal.h"

namespace {

thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

#define GSX_HIP(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(GSX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr size_t kAlign = 256;
constexpr int64_t kMaxPairs = ((int64_t)1 << 31) - 1;
inline size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

// Number of tiles along an axis.  REF_CPU iterates range(0, extent - tile, tile)
// (splat/gaussian_scene.py:208,214): the last row/column is never rendered.
// REF_CUDA covers the frame (splat/c/render.cu:119-120).
inline int32_t tiles_along(int32_t extent, int32_t tile, int semantics) {
    if (semantics == GSX_SEM_REF_CPU) return extent > tile ? (extent - tile + tile - 1) / tile : 0;
    return (extent + tile - 1) / tile;
}

struct Carve {
    size_t keys0, keys1, vals0, vals1;      // n x u32: depth keys / original indices (ping-pong)
    size_t rec, rect, rrect, bbox;          // per Gaussian: record, tile rectangle by index / by depth rank
    size_t tkeys0, tkeys1, tvals0, tvals1;  // cap x u32: tile ids / Gaussian indices (ping-pong)
    size_t ranges, longs, counters, temp, temp_bytes, total;
};

// The 64-byte `counters` block: what the kernels of one frame hand to each other on the device.
//   u32 [0] Gaussians behind the cull plane   [1] Gaussians kept by the depth sort (M)
//       [2] min(D, 2^32 - 1)                   [3] long tiles found (LongTiles.count)
//   i64 at byte 16: n_visible, D (the first two fields of a GsxFrameStats)
enum { kCtrCulled = 0, kCtrKept = 1, kCtrPairs = 2, kCtrLong = 3 };

Carve carve(int64_t n, int64_t cap, int64_t max_tiles, size_t temp_bytes) {
    Carve c;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off = align_up(off + bytes);
        return at;
    };
    size_t nn = (size_t)(n > 0 ? n : 1), cc = (size_t)(cap > 0 ? cap : 1);
    c.keys0 = take(nn * 4); c.keys1 = take(nn * 4); c.vals0 = take(nn * 4); c.vals1 = take(nn * 4);
    c.rec = take(nn * sizeof(gsx::Record));
    c.rect = take(nn * sizeof(gsx::TileRect));
    c.rrect = take(nn * sizeof(gsx::TileRect));
    c.bbox = take(nn * sizeof(float4));
    c.tkeys0 = take(cc * 4); c.tkeys1 = take(cc * 4); c.tvals0 = take(cc * 4); c.tvals1 = take(cc * 4);
    c.ranges = take((size_t)(max_tiles > 0 ? max_tiles : 1) * sizeof(uint2));
    c.longs = take(gsx::kMaxLongTiles * sizeof(uint32_t));
    c.counters = take(64);
    c.temp = take(temp_bytes);
    c.temp_bytes = temp_bytes;
    c.total = off;
    return c;
}

int64_t max_tiles_of(int32_t width, int32_t height, int32_t tile) {
    return (int64_t)((width + tile - 1) / tile) * ((height + tile - 1) / tile);
}

// Largest instance capacity whose carve fits `bytes` (temp size depends weakly on capacity).
int64_t capacity_for(size_t bytes, int64_t n, int64_t max_tiles) {
    Carve fixed = carve(n, 1, max_tiles, gsx::binning_temp_bytes(n, 1));
    if (fixed.total > bytes) return -1;
    int64_t cap = (int64_t)((bytes - fixed.total) / 16);
    for (int it = 0; it < 16 && cap > 0; ++it) {
        Carve c = carve(n, cap, max_tiles, gsx::binning_temp_bytes(n, cap));
        if (c.total <= bytes) return cap;
        int64_t over = (int64_t)((c.total - bytes + 15) / 16) + 64;
        cap = cap > over ? cap - over : 0;
    }
    return 0;
}

struct Plan {
    gsx::TileGrid grid;
    gsx::OutDesc out;
    int semantics;
    bool timing;
    bool no_sync;  // GSX_FLAG_NO_SYNC: nothing waits for the device
    bool generic;  // GSX_FLAG_GENERIC_KERNELS
    bool tight;    // GSX_SEM_STD_3DGS without GSX_FLAG_PUBLISHED_RECTS
    bool split;    // long tiles on four waves (not GSX_FLAG_NO_LONG_TILE_SPLIT)
    const GsxCamera *camera_device;
    uint32_t *tile_counts;
    const float *sh;
    int sh_degree;   // -1: RGB colours
    float background[3];
};

// Optional per-stage timing with HIP events on the launch stream (GSX_FLAG_TIMING).
struct StageTimer {
    bool on = false;
    hipStream_t s = nullptr;
    hipEvent_t ev[8];
    int n = 0;
    void begin(bool enable, hipStream_t stream) {
        on = enable;
        s = stream;
        mark();
    }
    void mark() {
        if (!on || n >= 8) return;
        if (hipEventCreate(&ev[n]) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(ev[n], s);
        ++n;
    }
    // marks: 0 start | 1 project (+keys) | 2 depth sort | 3 scan | 4 bin | 5 blend
    void finish(GsxFrameStats *st) {
        if (n > 0) (void)hipEventSynchronize(ev[n - 1]);
        if (st) {
            for (int i = 0; i < 8; ++i) st->stage_ms[i] = 0.0f;
            for (int i = 1; i < n; ++i) {
                float ms = 0.0f;
                (void)hipEventElapsedTime(&ms, ev[i - 1], ev[i]);
                st->stage_ms[i - 1] = ms;
            }
            if (n > 1) {
                float ms = 0.0f;
                (void)hipEventElapsedTime(&ms, ev[0], ev[n - 1]);
                st->stage_ms[GSX_STAGE_TOTAL] = ms;
            }
        }
        for (int i = 0; i < n; ++i) (void)hipEventDestroy(ev[i]);
        n = 0;
    }
    ~StageTimer() { finish(nullptr); }
};

int make_plan(int32_t width, int32_t height, int32_t tile, float *out_image, const GsxParams *params, Plan &p) {
    GsxParams d;
    gsx_default_params(&d);
    if (params) d = *params;
    if (width <= 0 || height <= 0) return fail(GSX_ERR_INVALID_ARGUMENT, "image size %dx%d is not positive", width, height);
    if (tile <= 0 || tile > 1024) return fail(GSX_ERR_INVALID_ARGUMENT, "tile size %d out of range [1,1024]", tile);
    if (d.semantics != GSX_SEM_REF_CPU && d.semantics != GSX_SEM_REF_CUDA && d.semantics != GSX_SEM_STD_3DGS)
        return fail(GSX_ERR_UNSUPPORTED, "unknown semantics %d", d.semantics);
    for (int i = 0; i < 3; ++i) p.background[i] = d.background[i];
    p.camera_device = d.camera_device;
    p.tile_counts = d.tile_counts;
    p.sh = d.sh;
    p.sh_degree = d.sh ? d.sh_degree : -1;
    if (d.sh && (d.sh_degree < 0 || d.sh_degree > 3)) return fail(GSX_ERR_INVALID_ARGUMENT, "SH degree %d outside [0,3]", d.sh_degree);
    if (d.layout != GSX_LAYOUT_WH3 && d.layout != GSX_LAYOUT_HW3) return fail(GSX_ERR_INVALID_ARGUMENT, "unknown layout %d", d.layout);
    if (!out_image) return fail(GSX_ERR_INVALID_ARGUMENT, "out_image is NULL");
    p.semantics = d.semantics;
    p.timing = (d.flags & GSX_FLAG_TIMING) != 0;
    p.no_sync = (d.flags & GSX_FLAG_NO_SYNC) != 0 && !p.timing;
    p.generic = (d.flags & GSX_FLAG_GENERIC_KERNELS) != 0;
    p.tight = d.semantics == GSX_SEM_STD_3DGS && (d.flags & GSX_FLAG_PUBLISHED_RECTS) == 0;
    p.split = (d.flags & GSX_FLAG_NO_LONG_TILE_SPLIT) == 0;
    gsx::TileGrid &g = p.grid;
    g.tile = tile;
    g.ntx = tiles_along(width, tile, d.semantics);
    g.nty = tiles_along(height, tile, d.semantics);
    g.width = width;
    g.height = height;
    if (g.ntx > 65535 || g.nty > 65535) return fail(GSX_ERR_UNSUPPORTED, "more than 65535 tiles along an axis");
    g.wx0 = d.tile_x0 < 0 ? 0 : d.tile_x0;
    g.wy0 = d.tile_y0 < 0 ? 0 : d.tile_y0;
    // tile_x1 / tile_y1 < 0: "to the end" (the default); x1 == x0 is an EMPTY window, also at tile 0
    g.wx1 = (d.tile_x1 < 0 || d.tile_x1 > g.ntx) ? g.ntx : d.tile_x1;
    g.wy1 = (d.tile_y1 < 0 || d.tile_y1 > g.nty) ? g.nty : d.tile_y1;
    if (g.wx0 > g.wx1) g.wx0 = g.wx1;
    if (g.wy0 > g.wy1) g.wy0 = g.wy1;
    gsx::OutDesc &o = p.out;
    o.ptr = out_image;
    o.x0 = d.out_w > 0 ? d.out_x0 : 0;
    o.y0 = d.out_h > 0 ? d.out_y0 : 0;
    o.w = d.out_w > 0 ? d.out_w : width;
    o.h = d.out_h > 0 ? d.out_h : height;
    if (d.layout == GSX_LAYOUT_WH3) {
        o.stride_x = (int64_t)o.h * 3;
        o.stride_y = 3;
    } else {
        o.stride_x = 3;
        o.stride_y = (int64_t)o.w * 3;
    }
    if (g.count() > 0) {
        // every rendered tile must lie inside the output buffer
        int64_t px0 = (int64_t)g.wx0 * tile, px1 = (int64_t)g.wx1 * tile, py0 = (int64_t)g.wy0 * tile, py1 = (int64_t)g.wy1 * tile;
        px1 = px1 > width ? width : px1;   // partial edge tiles (REF_CUDA) end at the frame border
        py1 = py1 > height ? height : py1;
        if (px0 < o.x0 || px1 > (int64_t)o.x0 + o.w || py0 < o.y0 || py1 > (int64_t)o.y0 + o.h)
            return fail(GSX_ERR_INVALID_ARGUMENT, "tile window [%d,%d)x[%d,%d) does not fit the %dx%d output buffer at (%d,%d)",
                        g.wx0, g.wx1, g.wy0, g.wy1, o.w, o.h, o.x0, o.y0);
    }
    return GSX_OK;
}

// The pixels of the output buffer that no tile of the window covers (the compositing kernel writes every
// pixel of every tile it owns, empty tiles included): up to four rectangles, zeroed by extra workgroups
// of the compositing launch instead of a whole-frame memset (25 MB at 1080p).  whole = true: the entire
// buffer (nothing is rendered).
gsx::ClearPlan make_clear_plan(const Plan &p, bool whole) {
    const gsx::OutDesc &o = p.out;
    const int T = p.grid.tile;
    gsx::ClearPlan cp;
    memset(&cp, 0, sizeof cp);
    // window in buffer-local pixel coordinates along (slow, fast) memory axes
    const bool wh3 = o.stride_y < o.stride_x;  // x is the slow axis
    const int64_t slow_n = wh3 ? o.w : o.h, fast_n = wh3 ? o.h : o.w;
    cp.pitch = fast_n * 3;
    auto rect = [&](int64_t s0, int64_t s1, int64_t f0, int64_t f1) {
        if (s1 <= s0 || f1 <= f0) return;
        const int i = cp.n++;
        cp.s0[i] = (int32_t)s0; cp.rows[i] = (int32_t)(s1 - s0);
        cp.f0[i] = (int32_t)f0; cp.fw[i] = (int32_t)(f1 - f0);
        cp.first[i + 1] = cp.first[i] + gsx::clear_blocks_for(s1 - s0, f1 - f0);
    };
    if (whole || p.grid.count() == 0) {
        // row by row blocks of at most 2^31 floats each: one rectangle per quarter keeps 32-bit indices safe
        const int64_t q = (slow_n + 3) / 4;
        for (int k = 0; k < 4; ++k) rect(k * q, (k + 1) * q < slow_n ? (k + 1) * q : slow_n, 0, fast_n);
        return cp;
    }
    int64_t ws0 = (int64_t)(wh3 ? p.grid.wx0 : p.grid.wy0) * T - (wh3 ? o.x0 : o.y0);
    int64_t ws1 = (int64_t)(wh3 ? p.grid.wx1 : p.grid.wy1) * T;
    int64_t wf1 = (int64_t)(wh3 ? p.grid.wy1 : p.grid.wx1) * T;
    ws1 = (ws1 > (wh3 ? p.grid.width : p.grid.height) ? (wh3 ? p.grid.width : p.grid.height) : ws1) - (wh3 ? o.x0 : o.y0);
    wf1 = (wf1 > (wh3 ? p.grid.height : p.grid.width) ? (wh3 ? p.grid.height : p.grid.width) : wf1) - (wh3 ? o.y0 : o.x0);
    int64_t wf0 = (int64_t)(wh3 ? p.grid.wy0 : p.grid.wx0) * T - (wh3 ? o.y0 : o.x0);
    rect(0, ws0, 0, fast_n);
    rect(ws1, slow_n, 0, fast_n);
    rect(ws0, ws1, 0, wf0);
    rect(ws0, ws1, wf1, fast_n);
    return cp;
}

// Steps shared by both render entry points once records and rank-ordered tile rectangles exist.
// `order` = Gaussian index of each depth rank (nullptr: rows are already in compositing order);
// `m_dev` = number of ranks on the device (nullptr: n); `culled_dev` = Gaussians behind the cull plane.
// The pair count D never has to reach the host for the frame to be enqueued: the kernels read
// it from device memory and their grids are sized by the workspace capacity.  The normal call
// synchronises ONCE, after the last launch, to report the counts and to detect D > capacity;
// GSX_FLAG_NO_SYNC skips even that (the counts then arrive in pinned memory on their own).
int bin_and_blend(const Plan &p, const Carve &c, char *ws, int64_t n, int64_t cap, const gsx::TileRect *rrect,
                  const uint32_t *order, const uint32_t *m_dev, const uint32_t *culled_dev, GsxFrameStats *stats,
                  StageTimer &tm, hipStream_t s) {
    uint32_t *counters = (uint32_t *)(ws + c.counters);
    void *temp = ws + c.temp;
    int64_t *dev2 = (int64_t *)(counters + 4);
    uint2 *ranges = (uint2 *)(ws + c.ranges);
    bool counts_on_device = false, counts_in_host = false;
    if (p.grid.count() > 0 && n == 0 && p.semantics == GSX_SEM_STD_3DGS) {
        // no Gaussians: every pixel of the window is the background colour
        tm.mark();  // 3: scan + emit
        GSX_HIP(gsx::launch_zero_words((uint32_t *)ranges, (size_t)p.grid.count() * 2, s));
        GSX_HIP(gsx::launch_blend((const gsx::Record *)(ws + c.rec), nullptr, (const uint32_t *)(ws + c.tvals0), ranges,
                                  p.grid, p.out, p.semantics, p.background, p.generic, make_clear_plan(p, false),
                                  gsx::LongTiles{nullptr, nullptr, 0u}, s));
    } else if (n == 0) {
        tm.mark();
        GSX_HIP(gsx::launch_clear(make_clear_plan(p, true), p.out.ptr, s));
    } else {
        // the counts are produced by the emit kernel even when no tile is rendered (an empty window
        // still reports n_visible; its pair count is 0 because every rectangle was clamped away)
        gsx::BinCounts bc{dev2, nullptr, counters + kCtrPairs, counters + kCtrLong, culled_dev, n};
        if (p.no_sync && stats) {
            // pinned host memory is device-visible: the emit kernel stores the two counts there itself
            void *alias = nullptr;
            if (hipHostGetDevicePointer(&alias, stats, 0) == hipSuccess && alias) {
                bc.stats2_host = (int64_t *)alias;
                counts_in_host = true;
            } else {
                (void)hipGetLastError();
            }
        }
        GSX_HIP(gsx::emit_instances(temp, rrect, order, m_dev, n, cap, p.grid, ws + c.tkeys0, (uint32_t *)(ws + c.tvals0),
                                    ranges, bc, s));
        counts_on_device = true;
        tm.mark();  // 3: scan + emit
        if (p.grid.count() == 0) {
            GSX_HIP(gsx::launch_clear(make_clear_plan(p, true), p.out.ptr, s));
        } else {
            const uint32_t *sorted_vals = nullptr;
            const bool split = p.split && cap > 0 && gsx::blend_splits_long_tiles(p.grid, p.semantics, p.generic);
            const gsx::LongTiles lt{counters + kCtrLong, (uint32_t *)(ws + c.longs), split ? gsx::kMaxLongTiles : 0u};
            GSX_HIP(gsx::sort_instances(temp, cap, p.grid, ws + c.tkeys0, ws + c.tkeys1, (uint32_t *)(ws + c.tvals0),
                                        (uint32_t *)(ws + c.tvals1), ranges, counters + kCtrPairs, lt, &sorted_vals, s));
            if (p.tile_counts) GSX_HIP(gsx::launch_tile_counts(ranges, p.grid.count(), p.tile_counts, s));
            tm.mark();  // 4: tile sort
            GSX_HIP(gsx::launch_blend((const gsx::Record *)(ws + c.rec), (const float4 *)(ws + c.bbox), sorted_vals,
                                      ranges, p.grid, p.out, p.semantics, p.background, p.generic,
                                      make_clear_plan(p, false), lt, s));
            tm.mark();  // 5: blend
        }
    }
    if (p.no_sync) {
        if (stats) {
            // stats must be pinned host memory; the two counts land when the stream gets here
            if (!counts_on_device) {
                stats->n_visible = 0;
                stats->n_instances = 0;
            } else if (!counts_in_host) {
                GSX_HIP(hipMemcpyAsync(stats, dev2, 16, hipMemcpyDeviceToHost, s));
            }
            stats->n_tiles = p.grid.count();
            stats->reserved = cap;  // > 0: counts are delivered asynchronously; value = pair capacity used
        }
        return GSX_OK;
    }
    int64_t host2[2] = {0, 0};
    if (counts_on_device) GSX_HIP(hipMemcpyAsync(host2, dev2, 16, hipMemcpyDeviceToHost, s));
    GSX_HIP(hipStreamSynchronize(s));
    if (stats) {
        stats->n_visible = host2[0];
        stats->n_instances = host2[1];
        stats->n_tiles = p.grid.count();
        stats->reserved = 0;
    }
    tm.finish(stats);
    if (host2[1] > cap)
        return fail(GSX_ERR_WORKSPACE_TOO_SMALL, "frame needs %lld tile instances, workspace holds %lld",
                    (long long)host2[1], (long long)cap);
    return GSX_OK;
}

int check_workspace(void *workspace, size_t bytes, int64_t n, int64_t max_tiles, Carve &c, int64_t &cap) {
    if (!workspace) return fail(GSX_ERR_INVALID_ARGUMENT, "workspace is NULL");
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0) return fail(GSX_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
    cap = capacity_for(bytes, n, max_tiles);
    // the kernels index pairs with 32 bits and gsx_workspace_bytes sizes for < 2^31 pairs: a larger
    // buffer (a 288 GB part can hand over 68 GB and more) must not raise the capacity beyond that
    if (cap > kMaxPairs) cap = kMaxPairs;
    if (cap < 0)
        return fail(GSX_ERR_WORKSPACE_TOO_SMALL, "workspace of %zu bytes cannot hold %lld Gaussians", bytes, (long long)n);
    c = carve(n, cap, max_tiles, gsx::binning_temp_bytes(n, cap));
    return GSX_OK;
}

}  // namespace

extern "C" {

int gsx_version(void) { return GSX_VERSION; }

const char *gsx_last_error(void) { return g_error; }

void gsx_default_params(GsxParams *params) {
    if (!params) return;
    memset(params, 0, sizeof *params);
    params->semantics = GSX_SEM_REF_CPU;
    params->layout = GSX_LAYOUT_WH3;
    params->tile_x1 = -1;
    params->tile_y1 = -1;
}

size_t gsx_workspace_bytes(int64_t n, int32_t width, int32_t height, int32_t tile, int64_t max_instances) {
    if (n < 0 || width <= 0 || height <= 0 || tile <= 0 || max_instances < 0) return 0;
    if (n >= (int64_t)1 << 31 || max_instances >= (int64_t)1 << 31) return 0;
    size_t temp = gsx::binning_temp_bytes(n, max_instances);
    if (temp == 0) return 0;
    return carve(n, max_instances, max_tiles_of(width, height, tile), temp).total;
}

int gsx_preprocess(const GsxCamera *camera, const float *means3d, const float *scales, const float *quats,
                   const float *opacity_logit, const float *colors, int64_t n, float *points_xy, float *colors_out,
                   float *covariance_2d, float *depths, float *inverse_covariance_2d, float *radius, float *min_x,
                   float *max_x, float *min_y, float *max_y, float *sigmoid_opacity, int32_t *order,
                   int64_t *n_visible_host, const GsxParams *params, void *workspace, size_t workspace_bytes,
                   void *stream) {
    (void)params;
    hipStream_t s = (hipStream_t)stream;
    if (!camera) return fail(GSX_ERR_INVALID_ARGUMENT, "camera is NULL");
    if (n < 0 || n >= (int64_t)1 << 31) return fail(GSX_ERR_INVALID_ARGUMENT, "n = %lld out of range", (long long)n);
    if (n_visible_host) *n_visible_host = 0;
    if (n == 0) return GSX_OK;
    if (!means3d || !scales || !quats || !opacity_logit || !colors) return fail(GSX_ERR_INVALID_ARGUMENT, "an input array is NULL");
    if (!points_xy || !colors_out || !covariance_2d || !depths || !inverse_covariance_2d || !radius || !min_x || !max_x ||
        !min_y || !max_y || !sigmoid_opacity)
        return fail(GSX_ERR_INVALID_ARGUMENT, "an output array is NULL");
    Carve c;
    int64_t cap;
    int rc = check_workspace(workspace, workspace_bytes, n, 1, c, cap);
    if (rc != GSX_OK) return rc;
    char *ws = (char *)workspace;
    uint32_t *k0 = (uint32_t *)(ws + c.keys0), *k1 = (uint32_t *)(ws + c.keys1);
    uint32_t *v0 = (uint32_t *)(ws + c.vals0), *v1 = (uint32_t *)(ws + c.vals1);
    uint32_t *counters = (uint32_t *)(ws + c.counters);
    GSX_HIP(hipMemsetAsync(counters, 0, 64, s));
    GSX_HIP(gsx::launch_depth_keys(*camera, means3d, n, k0, v0, s));
    GSX_HIP(gsx::radix_sort_pairs_u32(ws + c.temp, k0, k1, v0, v1, nullptr, n, 32, s));
    GSX_HIP(gsx::launch_count_visible(k0, n, counters, s));
    gsx::GaussiansIn in{means3d, scales, quats, opacity_logit, colors};
    gsx::StageOneOut out{points_xy, colors_out, covariance_2d, depths, inverse_covariance_2d, radius,
                         min_x, max_x, min_y, max_y, sigmoid_opacity, order};
    GSX_HIP(gsx::launch_project_full(*camera, in, k0, v0, n, out, s));
    uint32_t nv = 0;
    GSX_HIP(hipMemcpyAsync(&nv, counters, 4, hipMemcpyDeviceToHost, s));
    GSX_HIP(hipStreamSynchronize(s));
    if (n_visible_host) *n_visible_host = nv;
    return GSX_OK;
}

int gsx_render_preprocessed(int32_t image_height, int32_t image_width, int32_t tile_size, const float *point_means,
                            const float *point_colors, const float *inverse_covariance_2d, const float *min_x,
                            const float *max_x, const float *min_y, const float *max_y, const float *opacity,
                            int64_t n, float *out_image, const GsxParams *params, GsxFrameStats *stats_host,
                            void *workspace, size_t workspace_bytes, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    Plan p;
    int rc = make_plan(image_width, image_height, tile_size, out_image, params, p);
    if (rc != GSX_OK) return rc;
    if (p.semantics == GSX_SEM_STD_3DGS)
        return fail(GSX_ERR_UNSUPPORTED, "GSX_SEM_STD_3DGS has its own stage 1: use gsx_render_forward");
    if (n < 0 || n >= (int64_t)1 << 31) return fail(GSX_ERR_INVALID_ARGUMENT, "n = %lld out of range", (long long)n);
    if (n > 0 && (!point_means || !point_colors || !inverse_covariance_2d || !min_x || !max_x || !min_y || !max_y || !opacity))
        return fail(GSX_ERR_INVALID_ARGUMENT, "an input array is NULL");
    Carve c;
    int64_t cap;
    rc = check_workspace(workspace, workspace_bytes, n, max_tiles_of(image_width, image_height, tile_size), c, cap);
    if (rc != GSX_OK) return rc;
    char *ws = (char *)workspace;
    StageTimer tm;
    tm.begin(p.timing, s);
    tm.mark();  // 1: (no depth sort on this entry point)
    gsx::PreprocessedIn in{point_means, point_colors, inverse_covariance_2d, min_x, max_x, min_y, max_y, opacity};
    GSX_HIP(gsx::launch_pack_preprocessed(in, n, p.grid, p.semantics, (gsx::Record *)(ws + c.rec),
                                          (gsx::TileRect *)(ws + c.rect),
                                          p.semantics == GSX_SEM_REF_CUDA ? (float4 *)(ws + c.bbox) : nullptr, s));
    tm.mark();  // 2: pack
    // rows are already in compositing order: the rectangles by row ARE the rectangles by rank
    return bin_and_blend(p, c, ws, n, cap, (const gsx::TileRect *)(ws + c.rect), nullptr, nullptr, nullptr, stats_host,
                         tm, s);
}

int gsx_render_forward(const GsxCamera *camera, const float *means3d, const float *scales, const float *quats,
                       const float *opacity_logit, const float *colors, int64_t n, int32_t tile_size,
                       float *out_image, const GsxParams *params, GsxFrameStats *stats_host, void *workspace,
                       size_t workspace_bytes, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!camera) return fail(GSX_ERR_INVALID_ARGUMENT, "camera is NULL");
    Plan p;
    int rc = make_plan(camera->width, camera->height, tile_size, out_image, params, p);
    if (rc != GSX_OK) return rc;
    if (n < 0 || n >= (int64_t)1 << 31) return fail(GSX_ERR_INVALID_ARGUMENT, "n = %lld out of range", (long long)n);
    if (n > 0 && (!means3d || !scales || !quats || !opacity_logit || (!colors && !p.sh)))
        return fail(GSX_ERR_INVALID_ARGUMENT, "an input array is NULL");
    Carve c;
    int64_t cap;
    rc = check_workspace(workspace, workspace_bytes, n, max_tiles_of(camera->width, camera->height, tile_size), c, cap);
    if (rc != GSX_OK) return rc;
    char *ws = (char *)workspace;
    uint32_t *k0 = (uint32_t *)(ws + c.keys0), *k1 = (uint32_t *)(ws + c.keys1);
    uint32_t *v0 = (uint32_t *)(ws + c.vals0), *v1 = (uint32_t *)(ws + c.vals1);
    StageTimer tm;
    tm.begin(p.timing, s);
    gsx::GaussiansIn in{means3d, scales, quats, opacity_logit, p.sh ? p.sh : colors};
    uint32_t *counters = (uint32_t *)(ws + c.counters);
    GSX_HIP(gsx::launch_project_pack(*camera, p.camera_device, in, n, p.grid, p.semantics, p.tight, p.sh_degree, k0, (gsx::Record *)(ws + c.rec),
                                     (gsx::TileRect *)(ws + c.rect), counters,
                                     p.semantics == GSX_SEM_REF_CUDA ? (float4 *)(ws + c.bbox) : nullptr, s));
    tm.mark();  // 1: project (+ depth keys)
    GSX_HIP(gsx::sort_depth_compact(ws + c.temp, k0, k1, v0, v1, n, counters + kCtrKept, counters + kCtrCulled,
                                    (const gsx::TileRect *)(ws + c.rect), (gsx::TileRect *)(ws + c.rrect), s));
    tm.mark();  // 2: depth sort (drops what reaches no tile, leaves the rectangles in rank order)
    return bin_and_blend(p, c, ws, n, cap, (const gsx::TileRect *)(ws + c.rrect), v0, counters + kCtrKept,
                         counters + kCtrCulled, stats_host, tm, s);
}

// Test hook (not part of include/gsx.h): the pipeline's radix sort on caller-provided pairs.
// keys / vals: n 32-bit words each, sorted in place; key16 != 0 sorts uint16 keys.  scratch must
// hold 2 * 4n bytes + gsx radix table (use gsx_workspace_bytes(n, 16, 16, 16, n)).  count_dev
// (may be NULL) = device pointer to the element count, as the tile sort uses it.
int gsx_debug_sort_pairs(void *keys, uint32_t *vals, int64_t n, int32_t key_bits, int32_t key16,
                         const uint32_t *count_dev, void *scratch, size_t scratch_bytes, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0 || !keys || !vals || !scratch) return fail(GSX_ERR_INVALID_ARGUMENT, "bad arguments");
    size_t need = align_up((size_t)n * 4) * 2 + gsx::radix_temp_bytes(n);
    if (scratch_bytes < need) return fail(GSX_ERR_WORKSPACE_TOO_SMALL, "scratch needs %zu bytes", need);
    char *sc = (char *)scratch;
    uint32_t *valt = (uint32_t *)(sc + align_up((size_t)n * 4));
    void *temp = sc + 2 * align_up((size_t)n * 4);
    uint32_t *vc = vals, *va = valt;
    if (key16) {
        uint16_t *kc = (uint16_t *)keys, *ka = (uint16_t *)sc;
        GSX_HIP(gsx::radix_sort_pairs_u16(temp, kc, ka, vc, va, count_dev, n, key_bits, s));
        if (kc != (uint16_t *)keys) GSX_HIP(hipMemcpyAsync(keys, kc, (size_t)n * 2, hipMemcpyDeviceToDevice, s));
    } else {
        uint32_t *kc = (uint32_t *)keys, *ka = (uint32_t *)sc;
        GSX_HIP(gsx::radix_sort_pairs_u32(temp, kc, ka, vc, va, count_dev, n, key_bits, s));
        if (kc != (uint32_t *)keys) GSX_HIP(hipMemcpyAsync(keys, kc, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    }
    if (vc != vals) GSX_HIP(hipMemcpyAsync(vals, vc, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return GSX_OK;
}

int gsx_sh_to_rgb(const float *means3d, const float *sh, int32_t degree, int64_t n, const float *camera_center_host,
                  float *colors_out, void *stream) {
    if (degree < 0 || degree > 3) return fail(GSX_ERR_INVALID_ARGUMENT, "SH degree %d outside [0,3]", degree);
    if (n < 0) return fail(GSX_ERR_INVALID_ARGUMENT, "n is negative");
    if (!camera_center_host) return fail(GSX_ERR_INVALID_ARGUMENT, "camera_center_host is NULL");
    if (n > 0 && (!means3d || !sh || !colors_out)) return fail(GSX_ERR_INVALID_ARGUMENT, "an array is NULL");
    GSX_HIP(gsx::launch_sh_to_rgb(means3d, sh, degree, n, camera_center_host, colors_out, (hipStream_t)stream));
    return GSX_OK;
}

int gsx_covariance_3d(const float *scales, const float *quats, int64_t n, float *covariance_out, void *stream) {
    if (n < 0) return fail(GSX_ERR_INVALID_ARGUMENT, "n is negative");
    if (n > 0 && (!scales || !quats || !covariance_out)) return fail(GSX_ERR_INVALID_ARGUMENT, "an array is NULL");
    GSX_HIP(gsx::launch_covariance3d(scales, quats, n, covariance_out, (hipStream_t)stream));
    return GSX_OK;
}

int gsx_covariance_2d(const GsxCamera *camera, const float *points, const float *covariance_3d, int64_t n,
                      float *covariance_2d_out, void *stream) {
    if (!camera) return fail(GSX_ERR_INVALID_ARGUMENT, "camera is NULL");
    if (n < 0) return fail(GSX_ERR_INVALID_ARGUMENT, "n is negative");
    if (n > 0 && (!points || !covariance_3d || !covariance_2d_out)) return fail(GSX_ERR_INVALID_ARGUMENT, "an array is NULL");
    GSX_HIP(gsx::launch_covariance2d(*camera, points, covariance_3d, n, covariance_2d_out, (hipStream_t)stream));
    return GSX_OK;
}

int gsx_project_points(const GsxCamera *camera, const float *means3d, int64_t n, float *points_out,
                       uint8_t *in_view_out, void *stream) {
    if (!camera) return fail(GSX_ERR_INVALID_ARGUMENT, "camera is NULL");
    if (n < 0) return fail(GSX_ERR_INVALID_ARGUMENT, "n is negative");
    if (n > 0 && (!means3d || !points_out || !in_view_out)) return fail(GSX_ERR_INVALID_ARGUMENT, "an array is NULL");
    GSX_HIP(gsx::launch_project_points(*camera, means3d, n, points_out, in_view_out, (hipStream_t)stream));
    return GSX_OK;
}

}  // extern "C"
