// Stage 1 on gfx950: 3D -> 2D EWA projection, conic, extent, bounding box, tile rectangle.
//
// THIS TRANSLATION UNIT IS COMPILED WITH -ffp-contract=off.  The float32 operation order below
// is what torch executes for the reference's expressions (oracle/probe_torch_order.py): the
// products torch folds into one sgemm -- [p,1] @ M and (N,3,3) @ (3,3) -- are sequential FMA
// chains, spelled as explicit fmaf here; batched 3x3 products round every product and sum.  With
// no other FMA formed and correctly rounded divide/sqrt, view depth, radius and bounding box --
// the quantities whose rounding decides sort order and tile membership -- come out bit-identical
// to the reference's (0 differing bits at N = 1e5 and 1e6, tests/golden/stage1_*) and to the CPU
// restatement's.
//
// Reference behaviour restated here (paths relative to the reference repository):
//   splat/gaussian_scene.py:70-144  preprocess
//   splat/gaussians.py:54-69        Sigma = (R S)(R S)^T
//   splat/utils.py:132-155          quaternion -> rotation (normalises again)
//   splat/utils.py:293-310          z_view >= 0.2 cull
//   splat/utils.py:313-317          NDC -> pixel, (v + 1)(dim - 1)/2
//   splat/utils.py:320-354          EWA 2D covariance, clamp at 1.3 tan(fov/2)
//   splat/utils.py:368-393          inverse with determinant floored at 1e-3
//   splat/utils.py:409-423          r = ceil(3 sqrt(lambda_max)), discriminant floored at 0.1
//   splat/gaussian_scene.py:209-217 tile membership test (min <= x0 + T and max >= x0)
#include "gsx_internal.h"
#include "gsx_sample_device.h"
#include "gsx_schedule_device.h"
#include "gsx_sh_device.h"

namespace gsx {
namespace {

constexpr int kBlock = 256;
static_assert(kBlock == sh::kBlock, "the projection kernel stages SH coefficients with the SH kernel's block size");

struct Projected {
    float x, y;            // pixel position
    float ca, cb, cc, cd;  // 2D covariance [[ca, cb], [cc, cd]]
    float q00, q01, q10, q11;
    float radius, depth;
    float min_x, max_x, min_y, max_y;
};

// Column `col` of [p,1] @ M as torch's (N,4) @ (4,4) executes it: one sgemm, a sequential FMA chain over k
// (gaussian_scene.py:79-90, utils.py:305-307, 333); the last step fma(1, M3, acc) is a plain add.
__device__ __forceinline__ float row4(float p0, float p1, float p2, const float *M, int col) {
    float acc = p0 * M[0 + col];
    acc = fmaf(p1, M[4 + col], acc);
    acc = fmaf(p2, M[8 + col], acc);
    return acc + M[12 + col];
}

// ... when M is world2view -- a TRANSPOSED view in the reference (splat/image.py:51-53), which its BLAS is told about --
// and the product has at most three rows, other MKL kernels run (oracle/probe_torch_order.py, "few rows"):
//   kRowsOne       ((p0 M0 + p1 M1 fused) + M3) + p2 M2, the last product rounded on its own
//   kRowsTwoThree  (p0 M0 + p2 M2) + (p1 M1 + M3), nothing fused
// (full_proj_transform is a contiguous bmm result: always the chain above.)  How many rows the reference multiplies:
// ALL n points in the cull (utils.py:305-307), the N_vis visible ones everywhere else (gaussian_scene.py:79-85,
// utils.py:333) -- which is what the callers' two "rows" classes are.
enum { kRowsMany = 0, kRowsOne = 1, kRowsTwoThree = 2 };
__host__ __device__ __forceinline__ int rows_class(int64_t rows) { return rows == 1 ? kRowsOne : (rows <= 3 ? kRowsTwoThree : kRowsMany); }
__device__ __forceinline__ float row4_view(float p0, float p1, float p2, const float *M, int col, int rows) {
    if (rows == kRowsOne) return (fmaf(p1, M[4 + col], p0 * M[0 + col]) + M[12 + col]) + p2 * M[8 + col];
    if (rows == kRowsTwoThree) return (p0 * M[0 + col] + p2 * M[8 + col]) + (p1 * M[4 + col] + M[12 + col]);
    return row4(p0, p1, p2, M, col);
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

// torch.sigmoid on the opacity array (splat/gaussian_scene.py:143) as torch executes it on an AVX-512 host: 1 / (1 + e),
// e = the SIMD exponential of its vector library -- Sleef's 1.0-ulp expf: round-to-nearest reduction by ln 2 in two FMA
// steps, a degree-5 polynomial, FMA throughout -- restated here operation for operation (oracle/raster_cpu.c: vexpf_;
// probed against torch: 0 differing bits in 1e6 values).  torch hands the last < 32 elements of every thread's chunk to
// libm's expf instead; WHICH elements those are depends on N_vis and on the reference run's thread count, which a
// projection kernel knows nothing about: up to 31 values per thread can differ from the reference's in their last bit.
__device__ __forceinline__ float sigmoid_torch(float v) {
    const float d = 0.0f - v;
    const int q = (int)rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    float s = fmaf((float)q, -0.693145751953125f, d), u;
    s = fmaf((float)q, -1.428606765330187045e-06f, s);
    u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    u = u * __uint_as_float((uint32_t)((q >> 1) + 0x7f) << 23) * __uint_as_float((uint32_t)((q - (q >> 1)) + 0x7f) << 23);
    u = d < -104.0f ? 0.0f : u;
    u = d > 100.0f ? __builtin_inff() : u;
    return 1.0f / (1.0f + u);
}

// Sigma = (R S)(R S)^T with the quaternion normalised twice (F.normalize, then build_rotation's
// own division): splat/gaussians.py:54-69, splat/utils.py:132-155.
// `twice` = false (GSX_SEM_STD_3DGS): normalised once, as the 3DGS model's rotation activation does.
__device__ __forceinline__ void covariance3d(float s0, float s1, float s2, float qw, float qx, float qy, float qz,
                                             float (&S)[3][3], bool twice = true) {
    float n1 = sqrtf(((qw * qw + qx * qx) + qy * qy) + qz * qz);
    n1 = fmaxf(n1, 1e-12f);
    float a0 = qw / n1, a1 = qx / n1, a2 = qy / n1, a3 = qz / n1;
    float w = a0, x = a1, y = a2, z = a3;
    if (twice) {
        float n2 = sqrtf(((a0 * a0 + a1 * a1) + a2 * a2) + a3 * a3);
        w = a0 / n2; x = a1 / n2; y = a2 / n2; z = a3 / n2;
    }
    float R[3][3];
    R[0][0] = 1.0f - 2.0f * (y * y + z * z);
    R[0][1] = 2.0f * (x * y - w * z);
    R[0][2] = 2.0f * (x * z + w * y);
    R[1][0] = 2.0f * (x * y + w * z);
    R[1][1] = 1.0f - 2.0f * (x * x + z * z);
    R[1][2] = 2.0f * (y * z - w * x);
    R[2][0] = 2.0f * (x * z - w * y);
    R[2][1] = 2.0f * (y * z + w * x);
    R[2][2] = 1.0f - 2.0f * (x * x + y * y);
    float M[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        M[i][0] = R[i][0] * s0;
        M[i][1] = R[i][1] * s1;
        M[i][2] = R[i][2] * s2;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) S[i][j] = (M[i][0] * M[j][0] + M[i][1] * M[j][1]) + M[i][2] * M[j][2];
}

// EWA 2D covariance ((((J W) Sigma) W^T) J^T)[:2,:2], left to right (splat/utils.py:320-354), with
// the view-space point clamped to 1.3 tan(fov/2).  Rows 2 of J (all zero) and the structural
// zeros J01, J10 are skipped: adding an exact zero -- or fusing a product with one -- does not change a float32
// sum.  J @ W and ... @ W^T multiply the whole batch by ONE 3x3: torch folds each into an sgemm (sequential FMA
// over k); the two products between batched matrices round every product and sum.
// vis: the rows class of the reference's N_vis (rows_class): with at most three visible Gaussians its BLAS also
// evaluates ... @ W^T -- W^T is world2view[:3,:3], a column-major view -- as (k0 + k2) + k1 with nothing fused
// (oracle/probe_torch_order.py).
__device__ __forceinline__ void ewa_covariance(const GsxCamera &cam, float fx, float fy, float p0, float p1, float p2,
                                               float tz, const float (&S)[3][3], Projected &o, int vis) {
    const float *V = cam.world2view;
    const bool small_batch = vis != kRowsMany;
    float tx = row4_view(p0, p1, p2, V, 0, vis), ty = row4_view(p0, p1, p2, V, 1, vis);
    float limx = 1.3f * cam.tan_fovx, limy = 1.3f * cam.tan_fovy;
    float cx = fminf(fmaxf(tx / tz, -limx), limx) * tz;
    float cy = fminf(fmaxf(ty / tz, -limy), limy) * tz;
    float j00 = fx / tz;
    float j02 = -(fx * cx) / (tz * tz);
    float j11 = fy / tz;
    float j12 = -(fy * cy) / (tz * tz);
    // A = J @ Wm, Wm[i][j] = V[j*4+i]
    float A[2][3], B[2][3], C[2][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        A[0][j] = fmaf(j02, V[j * 4 + 2], j00 * V[j * 4 + 0]);
        A[1][j] = fmaf(j12, V[j * 4 + 2], j11 * V[j * 4 + 1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) B[i][j] = (A[i][0] * S[0][j] + A[i][1] * S[1][j]) + A[i][2] * S[2][j];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[i][j] = small_batch ? (B[i][0] * V[0 * 4 + j] + B[i][2] * V[2 * 4 + j]) + B[i][1] * V[1 * 4 + j]
                                  : fmaf(B[i][2], V[2 * 4 + j], fmaf(B[i][1], V[1 * 4 + j], B[i][0] * V[0 * 4 + j]));
    o.ca = C[0][0] * j00 + C[0][2] * j02;
    o.cb = C[0][1] * j11 + C[0][2] * j12;
    o.cc = C[1][0] * j00 + C[1][2] * j02;
    o.cd = C[1][1] * j11 + C[1][2] * j12;
}

// What stage 1 derives from the pixel position and the 2D covariance alone (o.x, o.y, o.ca .. o.cd, tz): inverse
// with the determinant floored at 1e-3, radius with the discriminant floored at 0.1, bounding box.  One sequence
// of float32 operations, used by every caller: the same inputs give the same bits.
__device__ __forceinline__ void finish_projection(float tz, Projected &o) {
    float det = o.ca * o.cd - o.cb * o.cc;
    det = fmaxf(det, 1e-3f);
    o.q00 = o.cd / det;
    o.q01 = -o.cb / det;
    o.q10 = -o.cc / det;
    o.q11 = o.ca / det;

    float mid = 0.5f * (o.ca + o.cd);
    float det2 = o.ca * o.cd - o.cb * o.cb;
    float m = fmaxf(mid * mid - det2, 0.1f);
    float root = sqrtf(m);
    float lam = fmaxf(mid + root, mid - root);
    o.radius = ceilf(3.0f * sqrtf(lam));
    o.depth = tz;
    o.min_x = floorf(o.x - o.radius);
    o.max_x = ceilf(o.x + o.radius);
    o.min_y = floorf(o.y - o.radius);
    o.max_y = ceilf(o.y + o.radius);
}

// Everything of stage 1 for one visible Gaussian.
__device__ __forceinline__ void project(const GsxCamera &cam, float p0, float p1, float p2, float tz,
                                        float s0, float s1, float s2, float qw, float qx, float qy, float qz,
                                        Projected &o, int vis) {
    const float *F = cam.full_proj;
    float S[3][3];
    covariance3d(s0, s1, s2, qw, qx, qy, qz, S);

    // pixel position
    float cw = row4(p0, p1, p2, F, 3);
    float ndcx = row4(p0, p1, p2, F, 0) / cw;
    float ndcy = row4(p0, p1, p2, F, 1) / cw;
    o.x = (ndcx + 1.0f) * ((float)cam.width - 1.0f) * 0.5f;
    o.y = (ndcy + 1.0f) * ((float)cam.height - 1.0f) * 0.5f;

    ewa_covariance(cam, cam.fx, cam.fy, p0, p1, p2, tz, S, o, vis);
    finish_projection(tz, o);
}

// GSX_SEM_STD_3DGS stage 1 (published 3DGS forward pass; see include/gsx.h).  Returns false when
// the Gaussian is dropped (det == 0).  o.q00, o.q01 = o.q10, o.q11 = the symmetric conic.
__device__ __forceinline__ bool project_std(const GsxCamera &cam, float p0, float p1, float p2, float tz,
                                            float s0, float s1, float s2, float qw, float qx, float qy, float qz,
                                            Projected &o) {
    const float *F = cam.full_proj;
    float S[3][3];
    covariance3d(s0, s1, s2, qw, qx, qy, qz, S, false);
    float pw = 1.0f / (row4(p0, p1, p2, F, 3) + 0.0000001f);
    float ndcx = row4(p0, p1, p2, F, 0) * pw, ndcy = row4(p0, p1, p2, F, 1) * pw;
    o.x = ((ndcx + 1.0f) * (float)cam.width - 1.0f) * 0.5f;
    o.y = ((ndcy + 1.0f) * (float)cam.height - 1.0f) * 0.5f;
    float fx = (float)cam.width / (2.0f * cam.tan_fovx), fy = (float)cam.height / (2.0f * cam.tan_fovy);
    ewa_covariance(cam, fx, fy, p0, p1, p2, tz, S, o, kRowsMany);
    o.ca = o.ca + 0.3f;
    o.cd = o.cd + 0.3f;
    o.cc = o.cb;
    float det = o.ca * o.cd - o.cb * o.cb;
    if (det == 0.0f) return false;
    float det_inv = 1.0f / det;
    o.q00 = o.cd * det_inv;
    o.q01 = -o.cb * det_inv;
    o.q10 = o.q01;
    o.q11 = o.ca * det_inv;
    float mid = 0.5f * (o.ca + o.cd);
    float root = sqrtf(fmaxf(0.1f, mid * mid - det));
    float lam = fmaxf(mid + root, mid - root);
    o.radius = ceilf(3.0f * sqrtf(lam));
    o.depth = tz;
    o.min_x = o.x - o.radius; o.max_x = o.x + o.radius;
    o.min_y = o.y - o.radius; o.max_y = o.y + o.radius;
    return true;
}

// GSX_SEM_STD_3DGS tile range along one axis: [(int)((p - r)/T), (int)((p + r + T - 1)/T)) clamped
// to [0, nt], then to the window.  (int) truncates toward zero; NaN -> no tile.
__device__ __forceinline__ void axis_range_std(float p, float r, int T, int nt, int w0, int w1, int &lo, int &hi) {
    const float big = 1073741824.0f;
    float a = (p - r) / (float)T;
    float b = ((p + r + (float)T) - 1.0f) / (float)T;
    if (!(a == a) || !(b == b)) {
        lo = 1;
        hi = 0;
        return;
    }
    int ia = (int)fminf(fmaxf(a, -big), big), ib = (int)fminf(fmaxf(b, -big), big);
    ia = ia < 0 ? 0 : (ia > nt ? nt : ia);
    ib = ib < 0 ? 0 : (ib > nt ? nt : ib);
    lo = ia < w0 ? w0 : ia;
    hi = (ib > w1 ? w1 : ib) - 1;
}

// Tight variant (default under GSX_SEM_STD_3DGS, see GSX_FLAG_PUBLISHED_RECTS in include/gsx.h):
// tiles holding a pixel with |pixel - p| <= e, intersected with the published range.
__device__ __forceinline__ void axis_range_std_tight(float p, float e, int T, int &lo, int &hi) {
    const float big = 1073741824.0f;
    float a = floorf((p - e) / (float)T), b = floorf((p + e) / (float)T);
    if (!(a == a) || !(b == b)) {
        lo = 1;
        hi = 0;
        return;
    }
    int ia = (int)fminf(fmaxf(a, -big), big), ib = (int)fminf(fmaxf(b, -big), big);
    lo = ia > lo ? ia : lo;
    hi = ib < hi ? ib : hi;
}

// Tile index range along one axis for the reference's test `mn <= t*T + T and mx >= t*T`
// (gaussian_scene.py:209-217), clamped to the window [w0, w1).  Exact for every finite input;
// NaN compares false in the reference, i.e. the Gaussian is in no tile.
__device__ __forceinline__ void axis_range(float mn, float mx, int T, int w0, int w1, int &lo, int &hi) {
    if (!(mn == mn) || !(mx == mx)) {
        lo = 1;
        hi = 0;
        return;
    }
    const float big = 1073741824.0f;  // 2^30
    int imn = (int)ceilf(fminf(fmaxf(mn, -big), big));
    int imx = (int)floorf(fminf(fmaxf(mx, -big), big));
    // smallest t with t*T + T >= imn  <=>  t >= (imn - T) / T ; largest t with t*T <= imx
    int a = imn - T;
    lo = a >= 0 ? (a + T - 1) / T : -((-a) / T);
    hi = imx >= 0 ? imx / T : -((-imx + T - 1) / T);
    lo = lo < w0 ? w0 : lo;
    hi = hi > w1 - 1 ? w1 - 1 : hi;
}

// REF_CUDA: a Gaussian matters to the pixels p with mn <= p <= mx (splat/c/render.cu:55-60), i.e. to
// the tiles holding an integer pixel of [ceil(mn), floor(mx)] inside the frame.
__device__ __forceinline__ void axis_range_pixels(float mn, float mx, int T, int extent, int w0, int w1, int &lo,
                                                  int &hi) {
    if (!(mn == mn) || !(mx == mx)) {
        lo = 1;
        hi = 0;
        return;
    }
    const float big = 1073741824.0f;
    int p0 = (int)ceilf(fminf(fmaxf(mn, -big), big));
    int p1 = (int)floorf(fminf(fmaxf(mx, -big), big));
    p0 = p0 < 0 ? 0 : p0;
    p1 = p1 > extent - 1 ? extent - 1 : p1;
    if (p0 > p1) {
        lo = 1;
        hi = 0;
        return;
    }
    lo = p0 / T;
    hi = p1 / T;
    lo = lo < w0 ? w0 : lo;
    hi = hi > w1 - 1 ? w1 - 1 : hi;
}

__device__ __forceinline__ uint32_t tile_rect(float mnx, float mxx, float mny, float mxy, const TileGrid &g,
                                              int semantics, TileRect &r) {
    int lx, hx, ly, hy;
    if (semantics == GSX_SEM_STD_3DGS) {  // arguments are (x, radius, y, radius)
        axis_range_std(mnx, mxx, g.tile, g.ntx, g.wx0, g.wx1, lx, hx);
        axis_range_std(mny, mxy, g.tile, g.nty, g.wy0, g.wy1, ly, hy);
    } else if (semantics == GSX_SEM_REF_CUDA) {
        axis_range_pixels(mnx, mxx, g.tile, g.width, g.wx0, g.wx1, lx, hx);
        axis_range_pixels(mny, mxy, g.tile, g.height, g.wy0, g.wy1, ly, hy);
    } else {
        axis_range(mnx, mxx, g.tile, g.wx0, g.wx1, lx, hx);
        axis_range(mny, mxy, g.tile, g.wy0, g.wy1, ly, hy);
    }
    if (lx > hx || ly > hy) {
        r.x0 = 1; r.x1 = 0; r.y0 = 1; r.y1 = 0;
        return 0u;
    }
    r.x0 = (uint16_t)lx; r.x1 = (uint16_t)hx; r.y0 = (uint16_t)ly; r.y1 = (uint16_t)hy;
    return (uint32_t)(hx - lx + 1) * (uint32_t)(hy - ly + 1);
}

// ------------------------------------------------------------------------------------ kernels

// gsx_preprocess, first kernel: one thread per Gaussian in ORIGINAL order (coalesced reads of the five parameter
// arrays): the depth key for the sort (z >= 0.2 > 0: the IEEE bits of z are monotone in z; kCulledKey behind the cull
// plane -- dropped and counted by the sort's first step) and, for a visible Gaussian, what the rank-ordered output
// kernel cannot cheaply recompute -- pixel position, colour, 2D covariance, depth, sigmoid(opacity): 11 floats in the
// Gaussian's 48-byte record slot.  (Round 3: gathering the five input arrays by depth rank instead -- five random
// 64-byte sectors per Gaussian -- took 124 us at 1M.)  Also zeroes the sort's device counters (no memset node).
__global__ void __launch_bounds__(kBlock)
    project_stage_kernel(GsxCamera cam, GaussiansIn in, int64_t n, uint32_t *__restrict__ keys, Record *__restrict__ stage,
                         uint32_t *__restrict__ counters, int vis) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < 4) counters[i] = 0u;
    if (i >= n) return;
    const int64_t slot = in.original_index ? (int64_t)in.original_index[i] : i;   // GsxParams.original_index: filed under the original index
    const float *p = in.means3d + 3 * i;
    const float p0 = p[0], p1 = p[1], p2 = p[2];
    if (!(row4_view(p0, p1, p2, cam.world2view, 2, rows_class(n)) >= 0.2f)) {      // utils.py:293-310: all n rows at once
        keys[slot] = kCulledKey;
        return;
    }
    const float tz = row4_view(p0, p1, p2, cam.world2view, 2, vis);                // gaussian_scene.py:79-85: the visible rows
    keys[slot] = __float_as_uint(tz);
    const float *s = in.scales + 3 * i, *q = in.quats + 4 * i, *c = in.colors + 3 * i;
    Projected o;
    project(cam, p0, p1, p2, tz, s[0], s[1], s[2], q[0], q[1], q[2], q[3], o, vis);
    Record r;
    r.a = make_float4(o.x, o.y, c[0], c[1]);
    r.b = make_float4(c[2], tz, sigmoid_torch(in.opacity_logit[i]), o.ca);
    r.c = make_float4(o.cb, o.cc, o.cd, 0.0f);
    stage[slot] = r;
}

// Opacity factor and conic as the compositing kernel consumes them.
//   op  : sigmoid(opacity), applied twice under the CPU semantics (gaussian_scene.py:143 and :164);
//         stored as log2(op) so that alpha = exp2(d Q'' d^T + log2 op) needs no multiply
//   Q'' : Q * (-1/2 log2 e), so that the weight is exp2(d Q'' d^T); the -1/2 is exact, log2 e costs
//         one rounding per entry (relative 6e-8, far inside the 1e-4 pixel tolerance)
// REF_CUDA (splat/c/render.cu:5-19, 61-68): the mean is truncated to int by the device function's
// signature and the quadratic form is a dx^2 + 2 b dx dy + c dy^2 with b = Q01 only.
__device__ __forceinline__ float trunc_to_int(float v) {
    const float big = 2147483520.0f;
    return (float)(int)fminf(fmaxf(v, -big), big);
}

__device__ __forceinline__ void pack_record(int semantics, float x, float y, float q00, float q01, float q10,
                                            float q11, float op, float cr, float cg, float cb, float depth,
                                            Record &out, float4 *qraw) {
    const float k = -0.5f * 1.44269504088896340736f;
    if (semantics == GSX_SEM_REF_CUDA) {
        x = trunc_to_int(x);
        y = trunc_to_int(y);
        q10 = q01;
    }
    const float Q00 = q00 * k, Qs = (q01 + q10) * k, Q11 = q11 * k;     // Q'' = -1/2 log2(e) Q, float32
    if (semantics == GSX_SEM_REF_CPU) {
        // gsx_blend.hip evaluates e Q'' e^T with the square completed in y (see there):
        //   M = -Q'',  r11 = sqrt(M11),  h = M01 / r11,  D1 = M00 - h^2   (float64 from the float32 entries)
        // Whenever that does not exist as finite numbers (M11 <= 0, NaN, inf: caller-given inverse
        // covariances on the stage-2 entry) the record keeps the monomial coefficients, flagged in c.z.
        // from the UNSCALED float32 entries: a thin footprint's conic is nearly singular (det ~ 1e-4 of the
        // product of its diagonal), and rounding each entry of Q'' = k Q separately already moves the flat
        // direction's curvature by a part in 1e3 -- the scale factor is applied in float64 instead
        const double kd = 0.5 * 1.44269504088896340736;
        const double m00 = kd * (double)q00, m01 = kd * 0.5 * ((double)q01 + (double)q10), m11 = kd * (double)q11;
        const double r11 = sqrt(m11), h = m01 / r11, d1 = m00 - h * h;
        const float fr = (float)r11, fh = (float)h, fd = (float)d1;
        const bool ok = m11 > 0.0 && isfinite(fr) && isfinite(fh) && isfinite(fd) && fr > 0.0f;
        // ILL-CONDITIONED footprints keep the reference's own float32 evaluation (c.z = 2, kRefOrderFlag).  The
        // reference sums d Q d^T as four float32 products (splat/utils.py:363-364); where they cancel, their rounding
        // is part of its result: with S = the sum of their magnitudes the exponent is off by up to ~4 ulp(S), and
        // S <= rho |exponent|, rho = (1 + c) / (1 - c), c = |Q01 + Q10| / (2 sqrt(Q00 Q11)) (the largest generalised
        // eigenvalue of |Q| against Q; rho = k^2 for an axis ratio k at 45 degrees).  Weighted by alpha = op exp(-x),
        // x e^-x <= 1/e: the reference's alpha differs from the exact one by up to 8.8e-8 op rho anywhere on the
        // footprint (measured: a seventh of that on the 250:1 needles of tests/golden/needle_160x160_n110, 5.4e-4).
        // Beyond 3e-5 -- rho op > 340, axis ratios from ~20:1 -- the completed square, which is exact to 1e-5, is no
        // longer "the reference's result to 1e-4": the record is flagged, its raw float32 conic goes to the Gaussian's
        // slot of the side array (`qraw`: the workspace's per-Gaussian float4, which only REF_CUDA uses otherwise) and
        // the compositing kernels execute the reference's operations on it, one for one (alpha_ref in gsx_blend.hip).
        // (float32 and the approximate reciprocal square root are plenty for a threshold)
        const float cc = 0.5f * fabsf(q01 + q10) * __builtin_amdgcn_rsqf(q00 * q11);
        // (c >= 1: the float32 conic is not even positive definite -- the right-hand side is <= 0 and the record is flagged)
        const bool ref_order = ok && q00 > 0.0f && q11 > 0.0f && 8.8e-8f * op * (1.0f + cc) > 3e-5f * (1.0f - cc);
        if (ref_order && qraw) {
            // the completed square as for any record (the skip bounds of the compositing kernels are computed from it),
            // the opacity factor itself where the depth would be, and the raw conic in the Gaussian's side slot
            out.a = make_float4(x, y, fd, fh);
            out.b = make_float4(fr, log2f(op), cr, cg);
            out.c = make_float4(cb, op, 2.0f, 0.0f);
            *qraw = make_float4(q00, q01, q10, q11);
            return;
        }
        out.a = ok ? make_float4(x, y, fd, fh) : make_float4(x, y, Q00, Qs);
        out.b = make_float4(ok ? fr : Q11, log2f(op), cr, cg);
        out.c = make_float4(cb, depth, ok ? 0.0f : 1.0f, 0.0f);
        return;
    }
    out.a = make_float4(x, y, Q00, Qs);
    // STD_3DGS multiplies by the opacity after the exponential (its exponent is tested on its own)
    out.b = make_float4(Q11, semantics == GSX_SEM_STD_3DGS ? op : log2f(op), cr, cg);
    out.c = make_float4(cb, depth, 0.0f, 0.0f);
}

// A spare workgroup of the projection launch ranking its share of the depth sort's sample (SampleHint, gsx_sample_device.h):
// the key of sample index j is computed here -- the cull plane and the view depth of that Gaussian; whether it reaches a
// tile of the window is not known yet, and need not be: any monotone set of splitters sorts correctly, these divide the
// frustum's depths evenly.  lds: ns + 4 words.
struct SampleJob {
    uint32_t *splitters;
    unsigned long long *chunk_sums;
    uint32_t nsums, ns, wgs;
    const uint32_t *row_of;
};
__device__ __forceinline__ void presample_depths(const GsxCamera &cam, const GaussiansIn &in, int64_t n, int semantics,
                                                 const SampleJob &job, uint32_t wg, uint32_t *lds) {
    const bool std3dgs = semantics == GSX_SEM_STD_3DGS;
    sample_rank_body<kSortBins>(
        [&](uint32_t j) -> uint32_t {
            const int64_t row = job.row_of ? (int64_t)job.row_of[j] : (int64_t)j;
            const float *p = in.means3d + 3 * row;
            const float tz = row4(p[0], p[1], p[2], cam.world2view, 2);
            return (std3dgs ? !(tz > 0.2f) : !(tz >= 0.2f)) ? kCulledKey : __float_as_uint(tz);
        },
        (uint32_t)n, job.ns, job.splitters, job.chunk_sums, job.nsums, wg, job.wgs, lds, lds + job.ns);
}

// The exact projection of ONE Gaussian: row g of the input arrays; record and rectangle go to row g, the depth key to `slot`
// (g itself unless GsxParams.original_index reorders the rows: then the key is filed under the original index, which is the
// order the stable depth sort enumerates the keys in).  CULL: the caller has not applied the cull plane yet (the
// whole-frame kernel); the windowed kernel's survivors have passed it.  PRE: the caller fetched opacity logit and RGB colour
// together with everything else (pre_op, cr / cg / cb) -- a survivor of the window test nearly always needs them, and
// fetched here, behind the arithmetic, they are two more dependent trips to memory.  SHDEG >= 0: cr / cg / cb hold the
// colour evaluated from spherical harmonics.  Shared by both kernels: same operations, same bits.
template <int SHDEG, bool CULL, bool PRE>
__device__ __forceinline__ void project_one(const GsxCamera &cam, const GaussiansIn &in, int64_t g, int64_t slot, bool remapped,
                                            int64_t n, const TileGrid &grid, int semantics, bool tight, int vis, float cr,
                                            float cg, float cb, float pre_op, float p0, float p1, float p2, float s0, float s1,
                                            float s2, float q0, float q1, float q2, float q3, uint32_t *__restrict__ keys,
                                            Record *__restrict__ rec, TileRect *__restrict__ rect, float4 *__restrict__ bbox) {
    const bool std3dgs = semantics == GSX_SEM_STD_3DGS;
    // the cull multiplies all n points at once (utils.py:305-307), everything after it the visible ones (its rows class
    // is `vis`): below four rows the two products differ in their last bit (row4_view)
    if (CULL) {
        const float tz_cull = row4_view(p0, p1, p2, cam.world2view, 2, std3dgs ? (int)kRowsMany : rows_class(n));
        if (std3dgs ? !(tz_cull > 0.2f) : !(tz_cull >= 0.2f)) {               // utils.py:293-310
            keys[slot] = kCulledKey;
            return;
        }
    }
    const float tz = row4_view(p0, p1, p2, cam.world2view, 2, std3dgs ? (int)kRowsMany : vis);
    Projected o;
    bool keep = true;
    if (std3dgs)
        keep = project_std(cam, p0, p1, p2, tz, s0, s1, s2, q0, q1, q2, q3, o);
    else
        project(cam, p0, p1, p2, tz, s0, s1, s2, q0, q1, q2, q3, o, vis);
    TileRect tr;
    uint32_t cnt = std3dgs ? tile_rect(o.x, o.radius, o.y, o.radius, grid, semantics, tr)
                           : tile_rect(o.min_x, o.max_x, o.min_y, o.max_y, grid, semantics, tr);
    float op = 0.0f;
    if (keep && cnt) {
        // (the reference's first sigmoid runs over the whole array -- torch's SIMD form --, its second on one element at a
        // time, which torch computes with libm's expf: gaussian_scene.py:143 and :164)
        const float logit = PRE ? pre_op : in.opacity_logit[g];
        op = std3dgs ? sigmoidf(logit) : sigmoid_torch(logit);
        if (semantics == GSX_SEM_REF_CPU) op = sigmoidf(op);
    }
    if (std3dgs && tight && keep && cnt) {
        // alpha = op exp(power) >= 1/255  <=>  -power <= log(255 op); on the level set of a Gaussian with
        // covariance (ca, cb; cb, cd) the coordinates reach sqrt(2 log(255 op) ca) and sqrt(... cd).
        // The margin of 0.01 on the logarithm (1 % on alpha) is far above any rounding of alpha itself.
        const float lim = logf(255.0f * op) + 0.01f;
        if (!(lim > 0.0f)) {
            keep = false;  // alpha < 1/255 everywhere
        } else {
            int lx = tr.x0, hx = tr.x1, ly = tr.y0, hy = tr.y1;
            axis_range_std_tight(o.x, sqrtf(2.0f * lim * o.ca) * 1.0001f, grid.tile, lx, hx);
            axis_range_std_tight(o.y, sqrtf(2.0f * lim * o.cd) * 1.0001f, grid.tile, ly, hy);
            if (lx > hx || ly > hy) {
                keep = false;
            } else {
                tr.x0 = (uint16_t)lx; tr.x1 = (uint16_t)hx; tr.y0 = (uint16_t)ly; tr.y1 = (uint16_t)hy;
                cnt = (uint32_t)(hx - lx + 1) * (uint32_t)(hy - ly + 1);
            }
        }
    }
    if (!keep || cnt == 0u) {
        // reaches no tile of the window: it needs no depth rank, the sort drops it in its first pass
        // (GsxParams.original_index: the keys were pre-filled with exactly this)
        if (!remapped) keys[g] = kEmptyKey;
        return;
    }
    keys[slot] = __float_as_uint(tz);
    if (SHDEG < 0 && !PRE) {
        const float *c = in.colors + 3 * g;
        cr = c[0]; cg = c[1]; cb = c[2];
    }
    Record out;
    // (record, rectangle and side slot in ROW order, whatever `slot` is: neighbours in the arrays write neighbouring lines --
    // filed under scattered original indices, the 60 bytes of a strip's survivor were three partial-line stores, and those,
    // not the reads, were what the windowed kernel spent its time on: 63 -> 34 us on a 1/8 strip of 5M Gaussians)
    pack_record(semantics, o.x, o.y, o.q00, o.q01, o.q10, o.q11, op, cr, cg, cb, o.depth, out, bbox ? bbox + g : nullptr);
    rec[g] = out;
    if (bbox && semantics == GSX_SEM_REF_CUDA) bbox[g] = make_float4(o.min_x, o.max_x, o.min_y, o.max_y);
    rect[g] = tr;
}

// One thread per Gaussian, ORIGINAL order (coalesced reads of the parameter arrays, coalesced
// writes): depth key for the sort, compositing record and tile rectangle.  A Gaussian that reaches no
// tile of the window (culled, off screen, or -- on a rank that owns a strip of the frame -- in another
// rank's strip) only gets its key written: no record, no rectangle, and its opacity and colour are
// never read (40 B in + 4 B out instead of 56 B in + 60 B out; 7/8 of the Gaussians on an 8-GPU rank).
// SHDEG >= 0 (GsxParams.sh, build extension): the colour is evaluated here from spherical harmonics of that
// degree (gsx_sh_device.h: the workgroup streams its 256 Gaussians' coefficients through LDS) with the camera
// centre of the GsxCamera the kernel reads -- no colour array, no colour launch, and a captured frame follows
// a moving camera with SH colours too.  SHDEG = -1: in.colors holds RGB, as in the reference.
template <bool DEVICE_CAMERA, int SHDEG>
__global__ void __launch_bounds__(kBlock)
    project_pack_kernel(GsxCamera cam_arg, const GsxCamera *__restrict__ cam_dev, GaussiansIn in, int64_t n,
                        TileGrid grid, int semantics, bool tight, int vis,
                        uint32_t *__restrict__ keys, Record *__restrict__ rec,
                        TileRect *__restrict__ rect, uint32_t *__restrict__ counters, float4 *__restrict__ bbox,
                        bool sh_vec, SchedJob sched_job, SampleJob sample_job) {
    constexpr int DEG = SHDEG >= 0 ? SHDEG : 0;
    using L = sh::Layout<DEG>;
    // (the spare workgroups' sample lives where the others stage spherical harmonics: kSortSamples + 4 words at least)
    constexpr int kShWords = SHDEG >= 0 ? L::kRows * L::STRIDE : 1;
    __shared__ float sh_lds[kShWords > kSortSamples + 4 ? kShWords : kSortSamples + 4];
    // the launch's spare workgroups -- the first blocks, dispatched FIRST and long done when the Gaussians' blocks are:
    //   eight put the compositing schedule of this frame together, one XCD's share each, from the list lengths the previous
    //   frame left (GsxParams.hints, gsx_schedule_device.h);
    //   on a frame without splitters 128 rank the depth sort's sample (presample_depths)
    const uint32_t nsched = sched_job.sched ? kSchedXcds : 0u;
    if (blockIdx.x < nsched) {
        schedule_from_lengths(sched_job, blockIdx.x);
        return;
    }
    const uint32_t nspare = nsched + (sample_job.splitters ? sample_job.wgs : 0u);
    if (blockIdx.x < nspare) {
        presample_depths(DEVICE_CAMERA ? *cam_dev : cam_arg, in, n, semantics, sample_job, blockIdx.x - nsched, reinterpret_cast<uint32_t *>(sh_lds));
        return;
    }
    const int64_t blk = (int64_t)blockIdx.x - (int64_t)nspare;
    int64_t g = blk * kBlock + threadIdx.x;
    if (g < 4) counters[g] = 0u;   // the depth sort's culled / kept counts start from zero (no memset node)
    // GsxParams.original_index: row g of the inputs is the Gaussian of ORIGINAL index remap[g]; its key, record and rectangle
    // are filed under that index (the keys arrive pre-filled with kEmptyKey: only culled and kept Gaussians write theirs)
    const int32_t *__restrict__ remap = in.original_index;
    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    if (SHDEG >= 0) {
        // colour of every Gaussian of the workgroup first (all threads take part in the staging rounds)
        const float *cc = DEVICE_CAMERA ? cam_dev->camera_center : cam_arg.camera_center;
        const float c0 = cc[0], c1 = cc[1], c2 = cc[2];
        // a part's coefficient block starts 16-byte aligned only if the whole array is and kRows * W * 4 is a multiple of 16
        const bool vec = sh_vec && (L::kRows * L::W) % 4 == 0;
#pragma unroll
        for (int part = 0; part < L::kParts; ++part) {
            if (part) __syncthreads();
            sh::stage<DEG, L::kRows>(in.colors, n, blk * kBlock + part * L::kRows, sh_lds, vec);
            const int row = (int)threadIdx.x - part * L::kRows;
            if (row >= 0 && row < L::kRows && g < n) {
                const float *pp = in.means3d + 3 * g;
                sh::eval<DEG>(sh_lds + row * L::STRIDE, pp[0] - c0, pp[1] - c1, pp[2] - c2, cr, cg, cb);
            }
        }
    }
    // GsxParams.camera_device: the constants as they are in device memory now (uniform scalar loads)
    const GsxCamera &cam = DEVICE_CAMERA ? *cam_dev : cam_arg;
    if (g >= n) return;

    // ---- the exact projection (project_one: shared with the windowed kernel)
    const float *p = in.means3d + 3 * g, *sc = in.scales + 3 * g, *q = in.quats + 4 * g;
    project_one<SHDEG, true, false>(cam, in, g, remap ? (int64_t)remap[g] : g, remap != nullptr, n, grid, semantics, tight, vis, cr, cg,
                                    cb, 0.0f, p[0], p[1], p[2], sc[0], sc[1], sc[2], q[0], q[1], q[2], q[3], keys, rec, rect, bbox);
}

// GsxParams.block_bounds: may the windowed kernel drop the block (lo.xyz .. hi.xyz = box of its means, lo.w = its largest
// |scale|) unread?  Only if every row of it would fail the row test of phase 1 WITHOUT being culled: all eight corners
// in front of the cull plane with a margin (view depth is affine in the mean: every row's lies between the corners'), and
// the hull of the corners' pixel positions -- the image of a box under a projective map with w > 0 is the convex hull of
// its corners' images --, grown by the largest conservative radius a row can have (largest scale at the smallest depth,
// the row test's own formula) plus a pixel for the rounding of these few operations, misses the tile window.
__device__ __forceinline__ bool block_misses_window(const GsxCamera &cam, const TileGrid &grid, float4 lo, float4 hi) {
    const float *V = cam.world2view, *F = cam.full_proj;
    // (a box that is not one -- a NaN or infinite mean in the block -- says nothing: row by row)
    if (!(fabsf(lo.x) < 1e30f && fabsf(lo.y) < 1e30f && fabsf(lo.z) < 1e30f && fabsf(hi.x) < 1e30f && fabsf(hi.y) < 1e30f &&
          fabsf(hi.z) < 1e30f && lo.w >= 0.0f && lo.w < 1e30f))
        return false;
    float tz_min = __builtin_inff(), x_min = __builtin_inff(), x_max = -__builtin_inff(), y_min = __builtin_inff(), y_max = -__builtin_inff();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float p0 = (c & 1) ? hi.x : lo.x, p1 = (c & 2) ? hi.y : lo.y, p2 = (c & 4) ? hi.z : lo.z;
        const float tz = p0 * V[2] + p1 * V[6] + p2 * V[10] + V[14];
        const float cw = row4(p0, p1, p2, F, 3);
        tz_min = fminf(tz_min, fminf(tz, cw));
        const float icw = 1.0f / cw;
        const float xe = (row4(p0, p1, p2, F, 0) * icw + 1.0f) * ((float)cam.width - 1.0f) * 0.5f;
        const float ye = (row4(p0, p1, p2, F, 1) * icw + 1.0f) * ((float)cam.height - 1.0f) * 0.5f;
        x_min = fminf(x_min, xe); x_max = fmaxf(x_max, xe);
        y_min = fminf(y_min, ye); y_max = fmaxf(y_max, ye);
    }
    if (!(tz_min >= 0.21f)) return false;           // (a corner near or behind the plane, or a NaN: row by row)
    const float lx = 1.3f * cam.tan_fovx, ly = 1.3f * cam.tan_fovy;
    const float jn = (cam.fx * cam.fx) * (1.0f + lx * lx) + (cam.fy * cam.fy) * (1.0f + ly * ly);
    const float itz = 1.0f / (tz_min * 0.999f);
    const float rb = 3.0f * sqrtf(jn * (lo.w * lo.w) * (itz * itz) + 0.32f) * 1.03f + 3.0f;
    const float T = (float)grid.tile;
    // (every comparison false on a NaN: the block is then kept)
    return x_min - rb > (float)grid.wx1 * T || x_max + rb < (float)grid.wx0 * T || y_min - rb > (float)grid.wy1 * T ||
           y_max + rb < (float)grid.wy0 * T;
}

// The projection of a call that renders a strict PART of the frame (a multi-GPU rank's strip, a tile window): most
// Gaussians miss the window.  A workgroup takes kWinRows consecutive rows (= one block of GsxParams.block_bounds).
//   dropped  reordered rows with block boxes: prepare_reordered_kernel has marked the blocks that miss the window -- such a
//            workgroup reads one byte and is done;
//   phase 1  every row: cull plane, then a cheap conservative window test (means and scales requested together).  On a rank
//            that owns 1/8 of the frame 7/8 of the Gaussians end here, having cost 24 B of reads;
//   compact  the survivors' row numbers, in LDS, in row order;
//   phase 2  the survivors on dense waves: means, scales, quaternion, opacity and colour of a survivor are requested together,
//            then the exact projection (project_one: the whole-frame kernel's operations, bit for bit).
// SHDEG >= 0: the survivors' coefficient rows go through LDS, kRows at a time, and the colour is evaluated before phase 2.
// (kWinPer rows per thread: 4 -- fewer, fatter workgroups with all loads in flight -- measured SLOWER, 81 -> 94 us on a 1/8
// strip of 5M Gaussians, 73 VGPRs: the kernel never waited for its loads but for its partial-line stores)
constexpr int kWinPer = 1, kWinRows = kBlock * kWinPer;
static_assert(kWinRows == GSX_BOUNDS_ROWS, "a workgroup of the windowed kernel takes one block of GsxParams.block_bounds");
template <bool DEVICE_CAMERA, int SHDEG>
__global__ void __launch_bounds__(kBlock)
    project_window_kernel(GsxCamera cam_arg, const GsxCamera *__restrict__ cam_dev, GaussiansIn in, int64_t n,
                          TileGrid grid, int semantics, bool tight, int vis,
                          uint32_t *__restrict__ keys, Record *__restrict__ rec,
                          TileRect *__restrict__ rect, uint32_t *__restrict__ counters, float4 *__restrict__ bbox,
                          bool sh_vec, SchedJob sched_job, const uint8_t *__restrict__ block_dropped, SampleJob sample_job) {
    constexpr int DEG = SHDEG >= 0 ? SHDEG : 0;
    using L = sh::Layout<DEG>;
    constexpr int kShWords = SHDEG >= 0 ? L::kRows * L::STRIDE : 1;
    __shared__ float sh_lds[kShWords > kSortSamples + 4 ? kShWords : kSortSamples + 4];
    const uint32_t nsched = sched_job.sched ? kSchedXcds : 0u;       // (the spare workgroups of the launch: see project_pack_kernel)
    if (blockIdx.x < nsched) {
        schedule_from_lengths(sched_job, blockIdx.x);
        return;
    }
    const uint32_t nspare = nsched + (sample_job.splitters ? sample_job.wgs : 0u);
    if (blockIdx.x < nspare) {
        presample_depths(DEVICE_CAMERA ? *cam_dev : cam_arg, in, n, semantics, sample_job, blockIdx.x - nsched, reinterpret_cast<uint32_t *>(sh_lds));
        return;
    }
    const int64_t blk = (int64_t)blockIdx.x - (int64_t)nspare;
    __shared__ uint16_t s_list[kWinRows];
    __shared__ uint32_t s_wcnt[kWinPer][kBlock / 64];
    const int64_t base = blk * kWinRows;
    if (blk == 0 && threadIdx.x < 4) counters[threadIdx.x] = 0u;   // the depth sort's culled / kept counts start from zero
    const int32_t *__restrict__ remap = in.original_index;          // (see project_pack_kernel)
    const GsxCamera &cam = DEVICE_CAMERA ? *cam_dev : cam_arg;
    const bool std3dgs = semantics == GSX_SEM_STD_3DGS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // GsxParams.block_bounds: the whole block lies in front of the cull plane and off the window (prepare_reordered_kernel
    // found that out: block_misses_window) -- nothing to read, nothing to write (the keys are pre-filled with kEmptyKey).
    // (The test itself in here, uniform but on the vector ALU of every wave, cost what it saved: 150 instructions a wave.)
    if (block_dropped && block_dropped[blk]) return;

    // ---- phase 1
    float px[kWinPer], py[kWinPer], pz[kWinPer], smax[kWinPer];
#pragma unroll
    for (int k = 0; k < kWinPer; ++k) {
        const int64_t g = base + k * kBlock + threadIdx.x;
        px[k] = py[k] = pz[k] = smax[k] = 0.0f;
        if (g < n) {
            const float *p = in.means3d + 3 * g;
            px[k] = p[0]; py[k] = p[1]; pz[k] = p[2];
            if (!std3dgs) {
                const float *s = in.scales + 3 * g;
                smax[k] = fmaxf(fmaxf(fabsf(s[0]), fabsf(s[1])), fabsf(s[2]));
            }
        }
    }
    unsigned long long mask[kWinPer];
    bool survives[kWinPer];
#pragma unroll
    for (int k = 0; k < kWinPer; ++k) {
        const int64_t g = base + k * kBlock + threadIdx.x;
        survives[k] = false;
        if (g < n) {
            const float p0 = px[k], p1 = py[k], p2 = pz[k];
            const float tz = row4_view(p0, p1, p2, cam.world2view, 2, std3dgs ? (int)kRowsMany : rows_class(n));
            if (std3dgs ? !(tz > 0.2f) : !(tz >= 0.2f)) {               // utils.py:293-310 (all n rows at once)
                keys[remap ? (int64_t)remap[g] : g] = kCulledKey;
            } else {
                survives[k] = true;
                if (!std3dgs) {
                    // (approximate reciprocal / square root: the margins below absorb their last bits)
                    // The exact radius is ceil(3 sqrt(lam)) with lam <= trace(Sigma2D) + sqrt(0.1), trace(Sigma2D) <=
                    // (|J row 0|^2 + |J row 1|^2) max(scale)^2 and |J row 0|^2 = (fx / z)^2 (1 + clamp(x/z)^2) <=
                    // (fx / z)^2 (1 + (1.3 tan_x)^2) -- 2 % and two pixels are added for the rounding of everything
                    // involved.  A Gaussian is dropped only if that generous box misses the window; whatever passes
                    // goes through the exact test of phase 2 (NaNs pass: comparisons are false).
                    const float *F = cam.full_proj;
                    const float icw = __builtin_amdgcn_rcpf(row4(p0, p1, p2, F, 3)), itz = __builtin_amdgcn_rcpf(tz);
                    const float xe = (row4(p0, p1, p2, F, 0) * icw + 1.0f) * ((float)cam.width - 1.0f) * 0.5f;
                    const float ye = (row4(p0, p1, p2, F, 1) * icw + 1.0f) * ((float)cam.height - 1.0f) * 0.5f;
                    const float lx = 1.3f * cam.tan_fovx, ly = 1.3f * cam.tan_fovy;
                    const float jn = (cam.fx * cam.fx) * (1.0f + lx * lx) + (cam.fy * cam.fy) * (1.0f + ly * ly);
                    const float rb = 3.0f * __builtin_amdgcn_sqrtf(jn * (smax[k] * smax[k]) * (itz * itz) + 0.32f) * 1.02f + 2.0f;
                    const float T = (float)grid.tile;
                    if (xe - rb > (float)grid.wx1 * T || xe + rb < (float)grid.wx0 * T || ye - rb > (float)grid.wy1 * T ||
                        ye + rb < (float)grid.wy0 * T) {
                        if (!remap) keys[g] = kEmptyKey;        // (GsxParams.original_index: the keys were pre-filled with this)
                        survives[k] = false;
                    }
                }
            }
        }
        mask[k] = __ballot(survives[k]);
        if (lane == 0) s_wcnt[k][w] = (uint32_t)__popcll(mask[k]);
    }
    __syncthreads();
    // ---- the survivors' rows, in row order (k major, then wave, then lane)
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < kWinPer; ++k) {
        uint32_t before = total;
#pragma unroll
        for (int v = 0; v < kBlock / 64; ++v) {
            before += v < w ? s_wcnt[k][v] : 0u;
            total += s_wcnt[k][v];
        }
        if (survives[k]) s_list[before + (uint32_t)__popcll(mask[k] & ((1ull << lane) - 1ull))] = (uint16_t)(k * kBlock + threadIdx.x);
    }
    if (total == 0u) return;          // (uniform over the workgroup)
    __syncthreads();

    // ---- phase 2
    for (uint32_t b = 0; b < total; b += (uint32_t)kBlock) {
        const uint32_t idx = b + threadIdx.x;
        const bool active = idx < total;
        const int64_t g = base + (active ? (int64_t)s_list[idx] : 0);
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, lg = 0.f;
        float cr = 0.0f, cg = 0.0f, cb = 0.0f;
        int64_t slot = g;
        if (active) {           // everything a survivor needs, requested at once
            const float *p = in.means3d + 3 * g, *s = in.scales + 3 * g, *q = in.quats + 4 * g;
            p0 = p[0]; p1 = p[1]; p2 = p[2];
            s0 = s[0]; s1 = s[1]; s2 = s[2];
            q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
            lg = in.opacity_logit[g];
            if (SHDEG < 0) {
                const float *c = in.colors + 3 * g;
                cr = c[0]; cg = c[1]; cb = c[2];
            }
            if (remap) slot = (int64_t)remap[g];
        }
        if (SHDEG >= 0) {
            // the survivors' coefficient rows through LDS (a row per Gaussian, read by consecutive lanes), kRows at a time
            const float *cc = DEVICE_CAMERA ? cam_dev->camera_center : cam_arg.camera_center;
            const float c0 = cc[0], c1 = cc[1], c2 = cc[2];
            const uint32_t stop = min(total, b + (uint32_t)kBlock);
            for (uint32_t first = b; first < stop; first += (uint32_t)L::kRows) {
                __syncthreads();        // (the rows staged before have been read)
                sh::stage_rows<DEG, L::kRows>(in.colors, base, s_list, (int)stop, (int)first, sh_lds, sh_vec);
                const int row = (int)idx - (int)first;
                if (row >= 0 && row < L::kRows && active)
                    sh::eval<DEG>(sh_lds + row * L::STRIDE, p0 - c0, p1 - c1, p2 - c2, cr, cg, cb);
            }
        }
        if (active)
            project_one<SHDEG, false, true>(cam, in, g, slot, remap != nullptr, n, grid, semantics, tight, vis, cr, cg, cb, lg, p0,
                                            p1, p2, s0, s1, s2, q0, q1, q2, q3, keys, rec, rect, bbox);
    }
}

// GsxParams.original_index, before the projection: the keys start out as kEmptyKey ("reaches no tile": the projection then
// stores only the keys of culled and kept Gaussians, under their original index), and -- GsxParams.block_bounds, a call that
// renders part of the frame -- one byte per block of GSX_BOUNDS_ROWS rows says whether the windowed kernel may drop the block
// unread (block_misses_window).  One launch for both.
template <bool DEVICE_CAMERA>
__global__ void __launch_bounds__(kBlock)
    prepare_reordered_kernel(GsxCamera cam_arg, const GsxCamera *__restrict__ cam_dev, TileGrid grid, uint32_t *__restrict__ keys,
                             int64_t n, const float4 *__restrict__ bounds, int64_t nblocks, uint8_t *__restrict__ dropped) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t i = t * 4;
    if (i + 4 <= n) {
        *reinterpret_cast<uint4 *>(keys + i) = make_uint4(kEmptyKey, kEmptyKey, kEmptyKey, kEmptyKey);
    } else {
        for (int64_t k = i; k < n; ++k) keys[k] = kEmptyKey;
    }
    if (dropped && t < nblocks) {
        const GsxCamera &cam = DEVICE_CAMERA ? *cam_dev : cam_arg;
        dropped[t] = block_misses_window(cam, grid, bounds[2 * t], bounds[2 * t + 1]) ? 1 : 0;
    }
}

// gsx_preprocess, last kernel: all PreprocessedScene fields (the reference's stage-1 API surface) in depth order, one
// thread per rank: order[r] = Gaussian of rank r (the compacting depth sort's output), *m_dev = number of ranks =
// visible Gaussians.  ONE 48-byte gather per rank (project_stage_kernel's slot); inverse covariance, radius and
// bounding box are derived here by the same operations as everywhere else (finish_projection).
__global__ void __launch_bounds__(kBlock)
    project_full_kernel(const Record *__restrict__ stage, const uint32_t *__restrict__ order,
                        const uint32_t *__restrict__ m_dev, int64_t n, StageOneOut out) {
    int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n || r >= (int64_t)*m_dev) return;
    const uint32_t g = order[r];
    const Record q = stage[g];
    Projected o;
    o.x = q.a.x; o.y = q.a.y;
    o.ca = q.b.w; o.cb = q.c.x; o.cc = q.c.y; o.cd = q.c.z;
    finish_projection(q.b.y, o);
    out.points_xy[2 * r] = o.x;
    out.points_xy[2 * r + 1] = o.y;
    out.colors[3 * r] = q.a.z;
    out.colors[3 * r + 1] = q.a.w;
    out.colors[3 * r + 2] = q.b.x;
    out.cov2d[4 * r] = o.ca; out.cov2d[4 * r + 1] = o.cb; out.cov2d[4 * r + 2] = o.cc; out.cov2d[4 * r + 3] = o.cd;
    out.depths[r] = o.depth;
    out.inv_cov[4 * r] = o.q00; out.inv_cov[4 * r + 1] = o.q01; out.inv_cov[4 * r + 2] = o.q10; out.inv_cov[4 * r + 3] = o.q11;
    out.radius[r] = o.radius;
    out.min_x[r] = o.min_x; out.max_x[r] = o.max_x; out.min_y[r] = o.min_y; out.max_y[r] = o.max_y;
    out.sig_op[r] = q.b.z;
    if (out.order) out.order[r] = (int32_t)g;
}

// Stage-1 arrays handed in by the caller (the reference's native argument list) -> records.
__global__ void __launch_bounds__(kBlock)
    pack_preprocessed_kernel(PreprocessedIn in, int64_t n, TileGrid grid, int semantics, Record *__restrict__ rec,
                             TileRect *__restrict__ rect, float4 *__restrict__ bbox) {
    int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    float op = in.opacity[r];
    if (semantics == GSX_SEM_REF_CPU) op = sigmoidf(op);
    float mnx = in.min_x[r], mxx = in.max_x[r], mny = in.min_y[r], mxy = in.max_y[r];
    Record out;
    pack_record(semantics, in.means[2 * r], in.means[2 * r + 1], in.inv_cov[4 * r], in.inv_cov[4 * r + 1],
                in.inv_cov[4 * r + 2], in.inv_cov[4 * r + 3], op, in.colors[3 * r], in.colors[3 * r + 1],
                in.colors[3 * r + 2], 0.0f, out, bbox ? bbox + r : nullptr);
    rec[r] = out;
    if (bbox && semantics == GSX_SEM_REF_CUDA) bbox[r] = make_float4(mnx, mxx, mny, mxy);
    TileRect tr;
    tile_rect(mnx, mxx, mny, mxy, grid, semantics, tr);
    rect[r] = tr;
}

__global__ void __launch_bounds__(kBlock)
    project_points_kernel(GsxCamera cam, const float *__restrict__ means3d, int64_t n, float *__restrict__ pts,
                          uint8_t *__restrict__ in_view) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float p0 = means3d[3 * i], p1 = means3d[3 * i + 1], p2 = means3d[3 * i + 2];
    float tz = row4_view(p0, p1, p2, cam.world2view, 2, rows_class(n));      // in_view_frustum: all n rows at once
    const float *F = cam.full_proj;
    float cw = row4(p0, p1, p2, F, 3);
    float nx = row4(p0, p1, p2, F, 0) / cw, ny = row4(p0, p1, p2, F, 1) / cw, nz = row4(p0, p1, p2, F, 2) / cw;
    pts[3 * i] = (nx + 1.0f) * ((float)cam.width - 1.0f) * 0.5f;
    pts[3 * i + 1] = (ny + 1.0f) * ((float)cam.height - 1.0f) * 0.5f;
    pts[3 * i + 2] = nz;
    in_view[i] = tz >= 0.2f ? 1 : 0;
}

__global__ void __launch_bounds__(kBlock)
    covariance3d_kernel(const float *__restrict__ scales, const float *__restrict__ quats, int64_t n,
                        float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float S[3][3];
    covariance3d(scales[3 * i], scales[3 * i + 1], scales[3 * i + 2], quats[4 * i], quats[4 * i + 1], quats[4 * i + 2],
                 quats[4 * i + 3], S);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) out[9 * i + 3 * a + b] = S[a][b];
}

// GaussianScene.get_2d_covariance (splat/gaussian_scene.py:53-68 -> splat/utils.py:320-354) on caller-given
// points and 3D covariances: no cull, every row is projected.
__global__ void __launch_bounds__(kBlock)
    covariance2d_kernel(GsxCamera cam, const float *__restrict__ points, const float *__restrict__ cov3d, int64_t n,
                        float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float p0 = points[3 * i], p1 = points[3 * i + 1], p2 = points[3 * i + 2];
    const float tz = row4_view(p0, p1, p2, cam.world2view, 2, rows_class(n));       // the caller's n rows are the batch
    float S[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) S[a][b] = cov3d[9 * i + 3 * a + b];
    Projected o;
    ewa_covariance(cam, cam.fx, cam.fy, p0, p1, p2, tz, S, o, rows_class(n));
    out[4 * i] = o.ca; out[4 * i + 1] = o.cb; out[4 * i + 2] = o.cc; out[4 * i + 3] = o.cd;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace

hipError_t launch_covariance2d(const GsxCamera &cam, const float *points, const float *cov3d, int64_t n, float *out,
                               hipStream_t s) {
    if (n == 0) return hipSuccess;
    covariance2d_kernel<<<blocks_for(n), kBlock, 0, s>>>(cam, points, cov3d, n, out);
    return hipGetLastError();
}

hipError_t launch_covariance3d(const float *scales, const float *quats, int64_t n, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    covariance3d_kernel<<<blocks_for(n), kBlock, 0, s>>>(scales, quats, n, out);
    return hipGetLastError();
}

hipError_t launch_project_stage(const GsxCamera &cam, const GaussiansIn &in, int64_t n, uint32_t *keys, Record *stage,
                                uint32_t *counters, int visible_rows, hipStream_t s) {
    if (n == 0) return hipSuccess;
    project_stage_kernel<<<blocks_for(n), kBlock, 0, s>>>(cam, in, n, keys, stage, counters, visible_rows ? visible_rows : rows_class(n));
    return hipGetLastError();
}

template <int SHDEG>
static void launch_project_pack_deg(const GsxCamera &cam, const GsxCamera *cam_device, const GaussiansIn &in, int64_t n,
                                    const TileGrid &grid, int semantics, bool tight_rects, int visible_rows, uint32_t *keys,
                                    Record *rec, TileRect *rect, uint32_t *counters, float4 *bbox, const ScheduleHint &sh,
                                    uint8_t *block_scratch, hipStream_t s, const SampleHint &sample) {
    const SchedJob job{sh.lens, sh.sched, sh.header, sh.ntiles, sh.nwy, sched_cap(sh.ntiles, sh.nwy)};
    const SampleJob sjob{sample.splitters, sample.chunk_sums, sample.nsums, sample.ns,
                         sample.splitters ? sample_rank_workgroups(sample.ns) : 0u, sample.row_of};
    const unsigned spare = (sh.sched ? kSchedXcds : 0u) + sjob.wgs;
    const bool vec = (reinterpret_cast<uintptr_t>(in.colors) & 15u) == 0;
    // a strict part of the frame (a rank's strip, a tile window): most Gaussians miss it -> two-phase kernel
    const bool windowed = grid.wx0 > 0 || grid.wy0 > 0 || grid.wx1 < grid.ntx || grid.wy1 < grid.nty;
    // GsxParams.original_index: pre-filled keys; GsxParams.block_bounds on a windowed call: which blocks to drop unread
    uint8_t *dropped = nullptr;
    if (in.original_index) {
        const int64_t nblocks = (n + kWinRows - 1) / kWinRows;
        if (windowed && in.block_bounds && semantics != GSX_SEM_STD_3DGS && block_scratch) dropped = block_scratch;
        const unsigned blocks = blocks_for((n + 3) / 4 > nblocks ? (n + 3) / 4 : nblocks);
        if (cam_device)
            prepare_reordered_kernel<true><<<blocks, kBlock, 0, s>>>(cam, cam_device, grid, keys, n, in.block_bounds, nblocks, dropped);
        else
            prepare_reordered_kernel<false><<<blocks, kBlock, 0, s>>>(cam, cam_device, grid, keys, n, in.block_bounds, nblocks, dropped);
    }
#define GSX_LAUNCH_PP(DC)                                                                                             \
    do {                                                                                                              \
        if (windowed)                                                                                                 \
            project_window_kernel<DC, SHDEG><<<(unsigned)((n + kWinRows - 1) / kWinRows) + spare, kBlock, 0, s>>>(     \
                cam, cam_device, in, n, grid, semantics, tight_rects, visible_rows ? visible_rows : rows_class(n), keys, rec, rect, counters, bbox, vec, job, dropped, sjob); \
        else                                                                                                          \
            project_pack_kernel<DC, SHDEG><<<blocks_for(n) + spare, kBlock, 0, s>>>(                                   \
                cam, cam_device, in, n, grid, semantics, tight_rects, visible_rows ? visible_rows : rows_class(n), keys, rec, rect, counters, bbox, vec, job, sjob); \
    } while (0)
    if (cam_device) GSX_LAUNCH_PP(true); else GSX_LAUNCH_PP(false);
#undef GSX_LAUNCH_PP
}

// sh_degree < 0: in.colors is (n,3) RGB; 0..3: in.colors is (n, (degree+1)^2, 3) spherical harmonics.
hipError_t launch_project_pack(const GsxCamera &cam, const GsxCamera *cam_device, const GaussiansIn &in, int64_t n,
                               const TileGrid &grid, int semantics, bool tight_rects, int visible_rows, int sh_degree,
                               uint32_t *keys, Record *rec, TileRect *rect, uint32_t *counters, float4 *bbox,
                               const ScheduleHint &sched, uint8_t *block_scratch, hipStream_t s, const SampleHint &sample) {
    if (n == 0) return hipSuccess;
    switch (sh_degree) {
        case 0: launch_project_pack_deg<0>(cam, cam_device, in, n, grid, semantics, tight_rects, visible_rows, keys, rec, rect, counters, bbox, sched, block_scratch, s, sample); break;
        case 1: launch_project_pack_deg<1>(cam, cam_device, in, n, grid, semantics, tight_rects, visible_rows, keys, rec, rect, counters, bbox, sched, block_scratch, s, sample); break;
        case 2: launch_project_pack_deg<2>(cam, cam_device, in, n, grid, semantics, tight_rects, visible_rows, keys, rec, rect, counters, bbox, sched, block_scratch, s, sample); break;
        case 3: launch_project_pack_deg<3>(cam, cam_device, in, n, grid, semantics, tight_rects, visible_rows, keys, rec, rect, counters, bbox, sched, block_scratch, s, sample); break;
        default: launch_project_pack_deg<-1>(cam, cam_device, in, n, grid, semantics, tight_rects, visible_rows, keys, rec, rect, counters, bbox, sched, block_scratch, s, sample); break;
    }
    return hipGetLastError();
}

hipError_t launch_project_full(const Record *stage, const uint32_t *order, const uint32_t *m_dev, int64_t n,
                               const StageOneOut &out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    project_full_kernel<<<blocks_for(n), kBlock, 0, s>>>(stage, order, m_dev, n, out);
    return hipGetLastError();
}

hipError_t launch_pack_preprocessed(const PreprocessedIn &in, int64_t n, const TileGrid &grid, int semantics,
                                    Record *rec, TileRect *rect, float4 *bbox, hipStream_t s) {
    if (n == 0) return hipSuccess;
    pack_preprocessed_kernel<<<blocks_for(n), kBlock, 0, s>>>(in, n, grid, semantics, rec, rect, bbox);
    return hipGetLastError();
}

hipError_t launch_project_points(const GsxCamera &cam, const float *means3d, int64_t n, float *points_out,
                                 uint8_t *in_view, hipStream_t s) {
    if (n == 0) return hipSuccess;
    project_points_kernel<<<blocks_for(n), kBlock, 0, s>>>(cam, means3d, n, points_out, in_view);
    return hipGetLastError();
}

}  // namespace gsx
