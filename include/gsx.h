/*
 * gsx.h -- C ABI of libgsx.so, the MI355X (gfx950) forward Gaussian-splat rasteriser.
 *
 * This is the drop-in boundary for the hot path of dcaustin33/intro_to_gaussian_splatting:
 * plain pointers and sizes, no torch types, no C++ exceptions.  Every entry point cites the
 * reference interface it replaces (paths relative to the reference repository).
 *
 * Conventions
 *   - All array pointers are DEVICE pointers to contiguous float32 (or the stated integer type)
 *     on the current HIP device, unless the name ends in `_host`.
 *   - All work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *     Calls that report a host-side count (n_visible / n_instances) synchronise that stream
 *     once, after their last launch; nothing else blocks.
 *   - The library never allocates, frees or retains device memory: the caller owns inputs,
 *     outputs and the workspace (size from gsx_workspace_bytes).
 *   - Return value: GSX_OK (0) or a negative GsxStatus; gsx_last_error() gives a thread-local
 *     message for the last failing call on the calling thread.
 */
#ifndef GSX_H_
#define GSX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSX_VERSION 305 /* major*10000 + minor*100 + patch.  305: GsxParams.stats_size, .original_index, .block_bounds and .row_of_index (160 bytes): GsxFrameStats is written only as far
                         * as the caller says its struct reaches (64 bytes -- the ABI-300 struct, no n_redo -- when params is NULL or ends
                         * before the field); GSX_FLAG_PLAIN_FOOTPRINTS needs stats_size >= 72.  Binaries built against the 302 .. 304
                         * headers must be REBUILT (their 128-byte GsxParams is still read, but their 72-byte GsxFrameStats gets no
                         * n_redo, and a 302 binary that fills its struct through the exported function gsx_default_params has its
                         * n_substrips fields ignored -- see there).  304: one compositing launch; GSX_FLAG_SKIP_REDO (303) is now
                         * GSX_FLAG_PLAIN_FOOTPRINTS, same value and contract; gsx_hints_bytes is smaller.  303: GsxFrameStats.n_redo (72 bytes);
                         * gsx_default_params_sized (gsx_default_params is a macro over
                         * it; the exported function of that name serves ABI 300 / 301 binaries), GSX_FLAG_SMALL_BATCH / _ONE_VISIBLE,
                         * stage 1 in the operation order torch executes.  302: GsxParams.n_substrips .. substrip_events (appended;
                         * a struct_size of 104 -- or 0 -- still means the ABI-300 struct).  301: GsxParams.struct_size (in
                         * reserved0's place), a larger schedule region in gsx_hints_bytes.  300: GsxParams.kept_hint, GsxFrameStats.n_kept,
                           * tile_counts zeroed when nothing is rendered, tile_x1 == tile_x0 is an empty window,
                           * GsxCamera.camera_center; the library exports exactly the functions declared here */

/* Every entry point below is exported; nothing else is (the library is built with -fvisibility=hidden). */
#if defined(__GNUC__)
#define GSX_API __attribute__((visibility("default")))
#else
#define GSX_API
#endif

typedef enum GsxStatus {
    GSX_OK = 0,
    GSX_ERR_INVALID_ARGUMENT = -1,
    GSX_ERR_WORKSPACE_TOO_SMALL = -2, /* stats->n_instances holds the count that is needed */
    GSX_ERR_HIP = -3,                 /* a HIP runtime call or kernel launch failed        */
    GSX_ERR_UNSUPPORTED = -4
} GsxStatus;

/* Compositing semantics. */
typedef enum GsxSemantics {
    /* splat/gaussian_scene.py:146-238: sigmoid applied twice, no alpha clamp, stop when
     * T(1-alpha) < 1e-6 before accumulating, every binned Gaussian evaluated at every pixel of
     * the tile, last tile row/column never rendered, pixel centres at integers. */
    GSX_SEM_REF_CPU = 0,
    /* splat/c/render.cu:21-87: single sigmoid, alpha clamped to 0.99, stop at 0.001, per-pixel
     * inclusive bbox cull, int-truncated means, all tiles rendered. */
    GSX_SEM_REF_CUDA = 1,
    /* Build extension (not in the reference -- parity unpinned): the forward
     * pass of the published 3D Gaussian Splatting rasteriser (Kerbl et al. 2023).  Stage 1: cull
     * z_view <= 0.2, quaternion normalised once, focal = extent / (2 tan(fov/2)), pixel =
     * ((ndc + 1) extent - 1) / 2, 0.3 added to the diagonal of the 2D covariance, symmetric conic,
     * Gaussians with det == 0 or an empty tile rectangle dropped, single sigmoid.  Tiles: every tile
     * whose index lies in [(int)((p - r) / T), (int)((p + r + T - 1) / T)) clamped to the grid, all
     * tiles of the frame (partial edge tiles included).  Per pixel: skip when the exponent is > 0,
     * alpha = min(0.99, opacity * exp(exponent)), skip when alpha < 1/255, stop (before
     * accumulating) when T (1 - alpha) < 1e-4, out = C + T_final * background.  Whole-path entry
     * point only (gsx_render_forward). */
    GSX_SEM_STD_3DGS = 2
} GsxSemantics;

/* Memory layout of the output frame. */
typedef enum GsxLayout {
    GSX_LAYOUT_WH3 = 0, /* out[x][y][c], what render_image returns (gaussian_scene.py:206) */
    GSX_LAYOUT_HW3 = 1  /* out[y][x][c], what render_image_cuda returns (render.cu:116)     */
} GsxLayout;

/*
 * Already-computed float32 camera constants, i.e. the attributes of the reference's
 * GaussianImage that preprocess() reads (splat/image.py:28-66).  Matrices are row-major and in
 * the reference's row-vector convention (p_row @ M).
 */
typedef struct GsxCamera {
    float world2view[16]; /* splat/image.py:51-53  */
    float full_proj[16];  /* splat/image.py:61-65  */
    float tan_fovx;       /* splat/image.py:42     */
    float tan_fovy;       /* splat/image.py:43     */
    float fx;             /* splat/image.py:28     */
    float fy;             /* splat/image.py:29     */
    int32_t width;        /* splat/image.py:38     */
    int32_t height;       /* splat/image.py:37     */
    float camera_center[3]; /* splat/image.py:66 (world2view.inverse()[3,:3]); read only when GsxParams.sh is set */
} GsxCamera;

/*
 * Options.  gsx_default_params() fills the reference's behaviour; a NULL params pointer means
 * the defaults.  The tile window selects which tiles this call renders (multi-GPU strips):
 * tiles [tile_x0, tile_x1) x [tile_y0, tile_y1); tile_x1 / tile_y1 < 0 (the default, -1) means "to the
 * end", tile_x1 == tile_x0 (or y) is an empty window: nothing is rendered, `out` is zeroed.
 * `out` always addresses a buffer of out_w x out_h pixels whose pixel (0,0) is frame pixel
 * (out_x0, out_y0); out_w / out_h <= 0 means the whole frame.
 */
typedef struct GsxParams {
    int32_t semantics; /* GsxSemantics, default GSX_SEM_REF_CPU */
    int32_t layout;    /* GsxLayout, default GSX_LAYOUT_WH3     */
    int32_t tile_x0, tile_x1, tile_y0, tile_y1;
    int32_t out_x0, out_y0, out_w, out_h;
    int32_t flags; /* GSX_FLAG_* */
    float background[3]; /* GSX_SEM_STD_3DGS only: colour behind the last Gaussian (default 0) */
    /* gsx_render_forward only.  NULL (default): the camera constants are the `camera` argument, read at
     * call time.  Otherwise a GsxCamera in DEVICE memory that the projection kernel reads when it RUNS
     * -- the call (and a hipGraph that recorded it) then follows whatever the buffer holds at that
     * moment, e.g. a camera that moves between replays of a captured frame.  `camera` must still be
     * given: its width / height size the frame on the host and must equal the buffer's. */
    const GsxCamera *camera_device;
    /* Optional DEVICE array of one uint32 per tile of the window (window-local id = (tx - tile_x0) *
     * window_height_in_tiles + (ty - tile_y0)): the length of every tile's Gaussian list, written by the
     * frame (all zero when the scene is empty; an empty window has no entries).  What a multi-GPU caller balances
     * its strips with (strips.balanced_plan).  NULL: not reported. */
    uint32_t *tile_counts;
    /* gsx_render_forward only, build extension (the reference has no spherical harmonics).  NULL (default): the
     * `colors` argument is (n,3) RGB.  Otherwise DEVICE coefficients (n, (sh_degree+1)^2, 3), sh_degree in 0..3,
     * in the published 3DGS convention (see gsx_sh_to_rgb); the view-dependent colour is evaluated inside the
     * projection kernel with GsxCamera.camera_center (of `camera`, or of camera_device when that is set), the
     * `colors` argument is ignored and may be NULL. */
    const float *sh;
    int32_t sh_degree;
    /* sizeof(GsxParams) as the CALLER compiled it; gsx_default_params() -- a macro that passes that sizeof -- fills it
     * in.  The fields below it were appended in ABI 300: a struct_size that ends before one of them makes the library
     * ignore that field (a client built against an older header hands over a shorter struct -- whatever lies behind it
     * is neither written by gsx_default_params nor read by any call).  0 = not stated: the struct is taken to be the
     * ABI-300 one, which ends behind `hints` (its callers zeroed this word).  Callers should also check
     * gsx_version() == GSX_VERSION once: the version is bumped whenever a struct or a signature changes. */
    int32_t struct_size;
    /* gsx_render_forward only.  How many Gaussians are expected to reach a tile of the window
     * (GsxFrameStats.n_kept of an earlier frame of this view and window); 0 (default) = unknown, assume all n.
     * A hint, never a bound: it only selects the depth-sort route (the sample-partitioned routes sort what is
     * KEPT inside LDS buckets, so a rank that owns 1/8 of a 5M-Gaussian frame takes the fast route although
     * n is large); a wrong value costs time, not correctness. */
    int64_t kept_hint;
    /* gsx_render_forward only.  NULL (default) or a DEVICE buffer of gsx_hints_bytes() bytes that the caller keeps
     * for ONE view (camera, window, tile size) from frame to frame, zero-filled before its first use.  Two things a
     * frame computes for its own use are correct whatever their values -- the splitters of the depth sort and the
     * order in which tiles are handed to the SIMDs -- so with GSX_FLAG_HINTS_VALID a frame takes both from what the
     * PREVIOUS frame of this view left in the buffer instead of computing them on its own critical path (two
     * dependent kernels, ~20 us of a 0.42 ms frame), and every frame given the buffer leaves fresh ones for the next
     * (computed by spare workgroups of launches that run anyway).  The buffer also holds what every tile COST in the
     * previous frame, which decides which tiles are composited by four waves instead of one (the same pixels either
     * way).  Stale hints (the camera moved) cost time -- unevenly filled sort buckets, an unbalanced hand-out --
     * never a pixel: the frame is the same bit for bit (tested).  Frames in flight on different streams need a
     * buffer each.  The buffer must be 256-byte aligned (GSX_ERR_INVALID_ARGUMENT otherwise). */
    void *hints;
    /* gsx_render_forward only (ABI 302).  Compositing in PARTS, for a caller that wants to start moving a finished part
     * of the window -- a multi-GPU rank sending its strip to the rank that assembles the frame -- while the rest is still
     * being composited: projection, depth order and binning run ONCE for the window, then the compositing launch is
     * issued n_substrips times, part k covering the tiles whose coordinate along `substrip_axis` (0: tile column x, 1:
     * tile row y) lies in [substrip_bounds[k], substrip_bounds[k + 1]), and after the launch of part k the library
     * records substrip_events[k] (a hipEvent_t the caller created) on `stream`.  The pixels are the same bit for bit
     * as those of the one-launch frame.  substrip_bounds: HOST array of n_substrips + 1 absolute tile coordinates,
     * ascending, [0] = the window's first and [n_substrips] = its last + 1 tile along that axis; substrip_events: HOST
     * array of n_substrips events.  n_substrips 0 or 1 (default): one launch, nothing recorded.  At most 16 parts.
     * Not while `stream` is being captured into a hipGraph (GSX_ERR_UNSUPPORTED): the events are recorded on the stream.
     * Rule sets / tile sizes without a partial launch composite in one launch and record every event behind it. */
    int32_t n_substrips;
    int32_t substrip_axis;
    const int32_t *substrip_bounds;
    void *const *substrip_events;
    /* sizeof(GsxFrameStats) as the CALLER compiled it (ABI 305); gsx_default_params() fills it in.  The render calls write
     * `stats_host` only that far: 64 bytes (n_visible .. n_kept, the struct of ABI 300) or 72 (+ n_redo).  A caller whose
     * GsxParams ends before this field, or who passes params == NULL, is taken to own the 64-byte struct and gets no n_redo
     * (GsxFrameStats grew once, in ABI 303, with nothing to tell the two apart: INTEGRATION.md's stub was overrun by 8 bytes).
     * Any other value is GSX_ERR_INVALID_ARGUMENT.  Every later field of GsxFrameStats will be reported the same way. */
    int32_t stats_size;
    int32_t reserved1; /* 0 */
    /* gsx_render_forward / gsx_preprocess (ABI 305).  NULL (default): Gaussian i of the input arrays IS Gaussian i.  Otherwise
     * a DEVICE array of n int32, a permutation of 0 .. n-1: the caller has REORDERED its five parameter arrays (and `sh`) -- e.g.
     * along a space-filling curve, so that the Gaussians of one part of the frame lie together in memory and a rank that
     * renders a strip reads AND WRITES whole cache lines of survivors (Gaussians.spatially_ordered() of the Python surface)
     * -- and row i of them holds the Gaussian whose ORIGINAL index is original_index[i].  The frame is then the one the
     * original arrays give, bit for bit: the depth keys are filed under the original index and the depth sort enumerates
     * them in that order, so equal depths still composite in original-index order; `order` and every reported index is the
     * original one.  gsx_render_forward also needs the inverse, row_of_index (below): records and rectangles stay in ROW
     * order -- written and, tile by tile, read back as neighbours -- and the sort's first pass turns an original index into
     * the row it names. */
    const int32_t *original_index;
    /* gsx_render_forward, with original_index (ABI 305).  NULL (default) or a DEVICE array of 8 floats per block of
     * GSX_BOUNDS_ROWS consecutive ROWS of the (reordered) arrays -- (min x, min y, min z, largest |scale|, max x, max y, max z,
     * 0) over the block's means and scales; ceil(n / GSX_BOUNDS_ROWS) blocks.  A call that renders a PART of the frame then
     * drops a whole block after reading these 32 bytes when the projected box, grown by the largest footprint radius any of
     * its Gaussians can have, misses the tile window and lies in front of the cull plane -- instead of reading 24 bytes of
     * every Gaussian to find the same thing out row by row.  Conservative: a block is only dropped when each of its rows
     * would be; the frame is the same bit for bit.  The bounds describe the arrays AS THEY ARE: whoever moves a mean or
     * grows a scale recomputes them (the Python surface does, from torch's in-place version counters:
     * Gaussians.current_block_bounds()). */
    const float *block_bounds;
    /* gsx_render_forward, with original_index: DEVICE array of n int32, the inverse permutation -- row_of_index[original_index[i]]
     * == i.  Required there (GSX_ERR_INVALID_ARGUMENT without it); gsx_preprocess does not read it. */
    const int32_t *row_of_index;
} GsxParams;
#define GSX_BOUNDS_ROWS 256

/* Record per-stage GPU times with HIP events on `stream` into GsxFrameStats.stage_ms (the call
 * then waits for the frame to finish).  Off by default: timing is measurement, not product. */
#define GSX_FLAG_TIMING 1

/* Enqueue the frame without waiting for the device at all.  Normally a render call synchronises
 * the stream once, after its last launch, to fill stats_host and to detect a pair count beyond
 * the workspace capacity.  With this flag it returns as soon as the launches are queued:
 * stats_host must then be PINNED host memory that stays valid until the stream has passed this
 * frame; n_visible / n_instances arrive by an asynchronous copy (stats_host->reserved holds the
 * pair capacity the frame was enqueued with, > 0); read them after synchronising the stream.  If
 * n_instances turns out larger than that capacity, pairs were dropped and the frame must be
 * rendered again with a larger workspace. */
#define GSX_FLAG_NO_SYNC 2

/* Composite with the any-tile-size kernels (one pixel per lane) even when tile == 16, where the
 * specialised 4-pixels-per-lane kernels would run.  Same arithmetic, same pixels bit for bit; exists
 * so that tests can hold the two kernel families against each other. */
#define GSX_FLAG_GENERIC_KERNELS 4

/* GSX_SEM_STD_3DGS only.  By default a Gaussian is binned into the tiles met by the bounding box of
 * the ellipse on which its alpha reaches 1/255 (with a 1 % safety margin on alpha), intersected with
 * the published rectangle (the square of the 3-sigma radius): a pixel outside that ellipse is skipped
 * by the alpha < 1/255 rule anyway, so the frame is the same bit for bit while the tile lists get
 * shorter (24 % fewer pairs on the benchmark scene; a Gaussian whose opacity is below 1/255 is in
 * no tile at all).  This flag bins with the published rectangles instead, e.g. to compare
 * GsxFrameStats.n_instances with another implementation of the published algorithm. */
#define GSX_FLAG_PUBLISHED_RECTS 8

/* GSX_SEM_REF_CPU, tile 16.  By default a tile whose Gaussian list is more than 4x the frame's average (and
 * more than 1024 entries) long is composited by four waves -- a quarter of its pixels each, one pixel per
 * lane -- instead of one, so that a few very long lists (trained scenes are heavy-tailed) do not outlast
 * the rest of the frame.  Same arithmetic, same pixels bit for bit; this flag keeps every tile on one wave,
 * for tests that hold the two paths against each other. */
#define GSX_FLAG_NO_LONG_TILE_SPLIT 16

/* Tile 16, GSX_SEM_REF_CPU / GSX_SEM_STD_3DGS.  All the tile workgroups of a 1080p frame are resident at once, so
 * a SIMD is busy for as long as the lists of its own tiles take; frames of >= 300 000 Gaussians therefore run one
 * more small kernel that ranks the tiles by list length (windows of more than 2048 tiles), and the compositing launch hands them out so that every
 * SIMD gets its share of every length class.  Which workgroup composites which tile does
 * not touch a pixel: the frame is the same bit for bit.  GSX_FLAG_TILE_SCHEDULE asks for the schedule whatever
 * the size of the scene, GSX_FLAG_NO_TILE_SCHEDULE never builds it (tests hold the two against each other). */
#define GSX_FLAG_TILE_SCHEDULE 32
#define GSX_FLAG_NO_TILE_SCHEDULE 64

/* GsxParams.hints holds what an earlier frame of the same view (same buffer, same n, same window and tile size) left
 * there: use it.  Without the flag a frame given a hints buffer only fills it. */
#define GSX_FLAG_HINTS_VALID 128

/* gsx_render_forward / gsx_preprocess, GSX_SEM_REF_CPU and GSX_SEM_REF_CUDA: how many Gaussians the reference has
 * VISIBLE when that is at most three.  Its matrix products over the visible Gaussians -- [p,1] @ world2view
 * (splat/gaussian_scene.py:79-85, splat/utils.py:333) and ... @ W.T (splat/utils.py:354) -- are single BLAS calls, and
 * the BLAS it runs on (MKL under torch) sums an output's products as a sequential FMA chain from four rows up, as
 * (k0 + k2) + (k1 + k3) unfused on two or three rows and in a third order on one (oracle/probe_torch_order.py).  A
 * call does not know N_vis while it projects: it assumes all n Gaussians are visible (n <= 3 selects the few-row
 * orders by itself).  A frame that reports another class (GsxFrameStats.n_visible = 1, or 2..3, out of more) differs
 * from the reference in the last bit of up to three depths and 2D covariances; a caller that wants the reference's
 * bits there too issues the call again with the flag that names the class -- the Python surface does. */
#define GSX_FLAG_SMALL_BATCH 256  /* two or three Gaussians are visible */
#define GSX_FLAG_ONE_VISIBLE 512  /* exactly one is */

/* gsx_render_forward / gsx_render_preprocessed, GSX_SEM_REF_CPU at tile size 16.  The compositing launch evaluates
 * ill-conditioned footprints (axis ratios from ~20:1, on the tiles along their ridge) in the reference's own float32
 * operation order; how many tiles and long-tile quarters held one is GsxFrameStats.n_redo.  With
 * this flag the call runs the instance of that launch that CANNOT evaluate them -- half the registers, twice the waves
 * per SIMD, ~6 % faster on a 4K frame of 5M Gaussians, no faster at 1080p -- for a caller that knows from an earlier frame
 * of the same view that n_redo was 0.  The frame's own n_redo says whether that held: if it is > 0, that many tiles /
 * quarters were NOT composited (their pixels are unspecified) and the frame must be rendered again without the flag.
 * A call that could not report n_redo -- stats_host NULL, or GsxParams.stats_size < 72 -- is refused with the flag set
 * (GSX_ERR_INVALID_ARGUMENT): the caller MUST be able to read it. */
#define GSX_FLAG_PLAIN_FOOTPRINTS 1024

/* Indices into GsxFrameStats.stage_ms (milliseconds). */
enum {
    GSX_STAGE_PROJECT = 0,    /* projection, depth keys, record packing         */
    GSX_STAGE_DEPTH_SORT = 1, /* stable sort of the depth keys (drops Gaussians that reach no tile) */
    GSX_STAGE_SCAN = 2,       /* tile-count sums + (tile, Gaussian) pair emission */
    GSX_STAGE_BIN = 3,        /* tile sort, tile ranges (+ the compositing schedule) */
    GSX_STAGE_BLEND = 4,      /* the compositing kernel alone                   */
    GSX_STAGE_TOTAL = 5
};

/* Host-side counts of one frame (N_vis and D). */
typedef struct GsxFrameStats {
    int64_t n_visible;   /* Gaussians that pass the z_view >= 0.2 cull         */
    int64_t n_instances; /* (Gaussian, tile) pairs binned inside the tile window */
    int64_t n_tiles;     /* tiles inside the window                             */
    int64_t reserved;
    float stage_ms[6];   /* filled only with GSX_FLAG_TIMING                     */
    int64_t n_kept;      /* Gaussians that reach a tile of the window (what the depth sort keeps): GsxParams.kept_hint
                          * of the next frame of this view                       */
    int64_t n_redo;      /* tiles (and long-tile quarters) that held an ill-conditioned footprint; see
                          * GSX_FLAG_PLAIN_FOOTPRINTS (ABI 303: the struct has 72 bytes).  Written only when
                          * GsxParams.stats_size >= 72 says the caller's struct has it */
} GsxFrameStats;
#define GSX_FRAME_STATS_BYTES_ABI300 64 /* n_visible .. n_kept: what a call writes when it is not told the struct's size */

GSX_API int gsx_version(void);
GSX_API const char *gsx_last_error(void);
/* Fills the first `struct_size` bytes of *params -- sizeof(GsxParams) as the caller compiled it -- with the reference's
 * behaviour and states that size in GsxParams.struct_size.  Nothing behind those bytes is touched. */
GSX_API void gsx_default_params_sized(GsxParams *params, size_t struct_size);
/* What callers write: the size is this header's.  (The exported FUNCTION of this name exists for binaries built
 * against the ABI 300 / 301 headers, where the struct had 104 bytes: it fills those and states struct_size = 104 -- so
 * the fields appended since (n_substrips .., stats_size) are NOT read from a struct filled through it.  A binary built
 * against the 302 header, the only one that had those fields AND called this function, must be rebuilt: its
 * substrip_events would never be recorded.  Language bindings call gsx_default_params_sized.) */
GSX_API void gsx_default_params(GsxParams *params);
#define gsx_default_params(params) gsx_default_params_sized((params), sizeof(GsxParams))

/*
 * Bytes of device workspace needed by any entry point below for up to `n` Gaussians, a
 * width x height frame, tile size `tile` and at most `max_instances` (Gaussian, tile) pairs.
 * Returns 0 on invalid arguments.
 */
GSX_API size_t gsx_workspace_bytes(int64_t n, int32_t width, int32_t height, int32_t tile, int64_t max_instances);

/* Bytes of a GsxParams.hints buffer for frames of width x height at tile size `tile` (0 on invalid arguments). */
GSX_API size_t gsx_hints_bytes(int32_t width, int32_t height, int32_t tile);

/*
 * Stage 1.  Replaces GaussianScene.preprocess (splat/gaussian_scene.py:70-144) with
 * Gaussians.get_3d_covariance_matrix (splat/gaussians.py:54-69) and the helpers of
 * splat/utils.py:293-423 fused in.  Inputs are the Gaussians attributes (splat/gaussians.py:19-33):
 * means3d (n,3), scales (n,3) linear, quats (n,4) (w,x,y,z), opacity_logit (n,1), colors (n,3).
 * Outputs are the PreprocessedScene fields (splat/schema.py:13-25), each with room for n rows,
 * depth-sorted (ascending view z, ties by original index); rows >= *n_visible_host are unspecified.
 * order (n) int32: original index of each sorted row (may be NULL).  Synchronises `stream`.
 */
GSX_API int gsx_preprocess(const GsxCamera *camera, const float *means3d, const float *scales, const float *quats,
                   const float *opacity_logit, const float *colors, int64_t n,
                   float *points_xy, float *colors_out, float *covariance_2d, float *depths,
                   float *inverse_covariance_2d, float *radius, float *min_x, float *max_x,
                   float *min_y, float *max_y, float *sigmoid_opacity, int32_t *order,
                   int64_t *n_visible_host, const GsxParams *params, void *workspace,
                   size_t workspace_bytes, void *stream);

/*
 * Stage 2 on stage-1 arrays.  Argument-for-argument mirror of the reference's native entry
 * point `torch::Tensor render_image(int image_height, int image_width, int tile_size, point_means,
 * point_colors, inverse_covariance_2d, min_x, max_x, min_y, max_y, opacity)`
 * (splat/c/render.cu:90-101, declared at splat/gaussian_scene.py:244-257), with the output
 * buffer caller-allocated instead of returned.  Rows must be in compositing (depth) order.
 * `opacity` is PreprocessedScene.sigmoid_opacity (n,1).  Under GSX_SEM_REF_CPU this computes what
 * GaussianScene.render_image (splat/gaussian_scene.py:200-238) computes from the same arrays.
 * out_image is fully written (pixels outside rendered tiles are set to 0).
 */
GSX_API int gsx_render_preprocessed(int32_t image_height, int32_t image_width, int32_t tile_size,
                            const float *point_means, const float *point_colors,
                            const float *inverse_covariance_2d, const float *min_x, const float *max_x,
                            const float *min_y, const float *max_y, const float *opacity, int64_t n,
                            float *out_image, const GsxParams *params, GsxFrameStats *stats_host,
                            void *workspace, size_t workspace_bytes, void *stream);

/*
 * Whole hot path: projection -> depth order -> tile binning -> compositing.  Replaces
 * GaussianScene.render_image (splat/gaussian_scene.py:200-238) and, with GSX_SEM_REF_CUDA /
 * GSX_LAYOUT_HW3, GaussianScene.render_image_cuda (splat/gaussian_scene.py:263-285); with
 * GSX_SEM_STD_3DGS (build extension) it is the forward pass of the published 3DGS rasteriser.
 * stats_host may be NULL.  Synchronises `stream` once, after the last launch, to report the counts
 * and to detect n_instances > capacity (GSX_ERR_WORKSPACE_TOO_SMALL; stats_host->n_instances then
 * holds the count to size the workspace for); with GSX_FLAG_NO_SYNC it does not synchronise at all.
 */
GSX_API int gsx_render_forward(const GsxCamera *camera, const float *means3d, const float *scales, const float *quats,
                       const float *opacity_logit, const float *colors, int64_t n, int32_t tile_size,
                       float *out_image, const GsxParams *params, GsxFrameStats *stats_host,
                       void *workspace, size_t workspace_bytes, void *stream);

/*
 * Replaces Gaussians.get_3d_covariance_matrix (splat/gaussians.py:54-69): covariance_out (n,3,3) =
 * (R S)(R S)^T from linear scales (n,3) and quaternions (n,4) (w,x,y,z), the quaternion
 * normalised twice as the reference does.  The render entry points compute this inline; the
 * function exists for callers of the reference's method.
 */
GSX_API int gsx_covariance_3d(const float *scales, const float *quats, int64_t n, float *covariance_out, void *stream);

/*
 * Replaces GaussianScene.get_2d_covariance (splat/gaussian_scene.py:53-68 -> compute_2d_covariance,
 * splat/utils.py:320-354): the EWA 2D covariance (n,2,2) of caller-given points (n,3) with 3D covariances
 * (n,3,3) under `camera` -- view-space point clamped at 1.3 tan(fov/2), J from the COLMAP focal lengths,
 * (((J W) Sigma) W^T) J^T left to right, top-left 2x2.  No cull: every row is projected.
 */
GSX_API int gsx_covariance_2d(const GsxCamera *camera, const float *points, const float *covariance_3d, int64_t n,
                      float *covariance_2d_out, void *stream);

/*
 * Debug helper on the same projection: replaces GaussianScene.render_points_image
 * (splat/gaussian_scene.py:44-51, splat/image.py:72-89).  Writes (x_pix, y_pix, ndc_z) for every
 * Gaussian in input order into points_out (n,3) and 1/0 into in_view_out (n) (uint8).
 */
GSX_API int gsx_project_points(const GsxCamera *camera, const float *means3d, int64_t n, float *points_out,
                       uint8_t *in_view_out, void *stream);

/*
 * Build extension (no counterpart in the reference, whose colour is the stored rgb/256,
 * splat/gaussians.py:20-22): view-dependent colour from real spherical harmonics of degree 0..3 in
 * the published 3D Gaussian Splatting convention, colour = max(0, 0.5 + sum_k Y_k(d) sh[k]) with
 * d = normalize(mean - camera centre).  sh is (n, (degree+1)^2, 3); camera_center_host is 3 floats
 * in HOST memory (GaussianImage.camera_center, splat/image.py:66); colors_out (n,3) then feeds the
 * `colors` argument of gsx_render_forward.
 */
GSX_API int gsx_sh_to_rgb(const float *means3d, const float *sh, int32_t degree, int64_t n,
                  const float *camera_center_host, float *colors_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSX_H_ */
