/*
 * C restatement of the reference's CPU forward rasteriser (float32, multi-threaded).
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/liboracle.so and loaded with
 * ctypes from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- as the
 * checker / reported CPU baseline, never as part of the product path.
 *
 * Parity status: PINNED through tests/test_oracle_golden.py (golden vectors produced by the
 * reference itself, oracle/capture_golden.py).
 *
 * Same arithmetic, same explicit operation order as oracle/cpu_ref.py and as the HIP projection
 * kernel; compile with -ffp-contract=off (see Makefile) so that ONLY the fmaf calls written here
 * fuse and depths / radii / bounding boxes are bit-identical across the three.
 *
 * Which products fuse is what torch EXECUTES for the reference's expressions, determined by
 * oracle/probe_torch_order.py (torch 2.10 + MKL, any N >= 4, 1 or 8 threads):
 *   (N,4) @ (4,4) and (N,3,3) @ (3,3)   folded into one sgemm: a sequential FMA chain over k
 *   (N,3,3) @ (N,3,3) (batched)         ATen's own loop: products and sums rounded one by one
 *   (1,2) @ (2,2) then (1,2) @ (2,1)    the first fuses (sgemm), the second does not (dot)
 *   ... @ W.T with N_vis <= 3           another MKL kernel: (k0 + k2) + k1, nothing fused (W.T is world2view[:3,:3],
 *                                       a column-major view -- the layout decides, see the probe)
 * and pinned against the reference itself at N = 1e5 and 1e6 (tests/test_oracle_golden.py).
 *
 * Reference lines restated (paths relative to /root/reference):
 *   orc_preprocess   splat/gaussian_scene.py:70-144, splat/gaussians.py:54-69,
 *                    splat/utils.py:132-155, 293-317, 320-354, 368-393, 409-423
 *   orc_render       splat/gaussian_scene.py:146-171 (render_pixel), :173-198 (render_tile),
 *                    :200-238 (render_image tile loop / binning), splat/utils.py:357-365
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float V[16];      /* world2view, row-vector form (splat/image.py:51-53)   */
    float F[16];      /* full_proj_transform (splat/image.py:61-65)            */
    float tan_fovx, tan_fovy, fx, fy;
    int32_t width, height;
} OrcCamera;

/* column `col` of [p,1] @ M: what torch's (N,4) @ (4,4) executes (gaussian_scene.py:79-90, utils.py:305-307, 333) */
static inline float row4(const float *p, const float *M, int col) {
    float acc = p[0] * M[0 * 4 + col];
    acc = fmaf(p[1], M[1 * 4 + col], acc);
    acc = fmaf(p[2], M[2 * 4 + col], acc);
    return acc + M[3 * 4 + col];                       /* fma(1, M3, acc) */
}

/* ... when M is world2view -- a TRANSPOSED view in the reference (splat/image.py:51-53), which its BLAS is told about --
 * and the product has at most three rows, other MKL kernels run (probe_torch_order.py, "few rows"):
 *   one row          ((p0 M0 + p1 M1 fused) + M3) + p2 M2, the last product rounded on its own
 *   two, three rows  (p0 M0 + p2 M2) + (p1 M1 + M3), nothing fused
 * full_proj_transform is a contiguous bmm result: always the chain above. */
static inline float row4_view(const float *p, const float *M, int col, int64_t rows) {
    if (rows == 1) return (fmaf(p[1], M[1 * 4 + col], p[0] * M[0 * 4 + col]) + M[3 * 4 + col]) + p[2] * M[2 * 4 + col];
    if (rows <= 3) return (p[0] * M[0 * 4 + col] + p[2] * M[2 * 4 + col]) + (p[1] * M[1 * 4 + col] + M[3 * 4 + col]);
    return row4(p, M, col);
}

/* batched (N,3,3) @ (N,3,3): every product and sum rounded on its own */
static void mm3(const float *A, const float *B, float *C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j]) + A[i * 3 + 2] * B[2 * 3 + j];
}

/* (N,3,3) @ one (3,3): torch folds it into a (3N,3) @ (3,3) sgemm, a sequential FMA chain over k */
static void mm3_fma(const float *A, const float *B, float *C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = fmaf(A[i * 3 + 2], B[2 * 3 + j], fmaf(A[i * 3 + 1], B[1 * 3 + j], A[i * 3 + 0] * B[0 * 3 + j]));
}

/* ... except X @ W.T (W.T = world2view[:3,:3], a column-major view of the camera matrix) with at most 3 matrices in the
 * batch, which MKL evaluates as (k0 + k2) + k1 with nothing fused (probe_torch_order.py, "small batch") */
static void mm3_small(const float *A, const float *B, float *C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j]) + A[i * 3 + 1] * B[1 * 3 + j];
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* torch.sigmoid on a contiguous float32 array (gaussian_scene.py:143), as torch 2.10 executes it on an AVX-512 host
 * (probed: 0 differing bits in 1e6 values except as stated): 1 / (1 + e) with e = the SIMD exponential of its vector
 * library (Sleef's 1.0-ulp expf: round-to-nearest reduction by ln2 in two FMA steps, degree-5 polynomial, FMA
 * throughout) on whole groups of 32 elements -- and libm's expf on what is left at the END OF EVERY THREAD'S CHUNK:
 * the array is cut into ceil(n / threads')-element chunks, threads' = min(threads, ceil(n / 32768)).  So the value
 * depends on the element's POSITION and on the thread count of the reference run (8 where the fixtures were made). */
static inline float pow2if_(int q) { uint32_t u = (uint32_t)(q + 0x7f) << 23; float f; memcpy(&f, &u, 4); return f; }
static float vexpf_(float d) {
    int q = (int)rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    float s = fmaf((float)q, -0.693145751953125f, d), u;
    s = fmaf((float)q, -1.428606765330187045e-06f, s);
    u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    u = u * pow2if_(q >> 1) * pow2if_(q - (q >> 1));
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = INFINITY;
    return u;
}
static float sigmoid_at(float x, int64_t i, int64_t n, int threads) {
    int64_t parts = (n + 32767) / 32768;
    if (parts > threads) parts = threads;
    if (parts < 1) parts = 1;
    int64_t chunk = (n + parts - 1) / parts, begin = (i / chunk) * chunk, len = n - begin < chunk ? n - begin : chunk;
    if (i - begin >= len - len % 32) return sigmoidf_(x);          /* the chunk's scalar tail */
    return 1.0f / (1.0f + vexpf_(0.0f - x));
}
static int g_ref_threads = 8;
void orc_set_reference_threads(int t) { g_ref_threads = t > 0 ? t : 1; }

/* (((J W) Sigma) W^T) J^T with the view-space point clamped to 1.3 tan(fov/2) (utils.py:320-354).
 * batch = number of rows the reference multiplies at once (its N_vis): selects MKL's kernel for ... @ W.T. */
static void ewa2d(const OrcCamera *cam, float fx, float fy, const float *p, float tz, const float *S, float *D,
                  int64_t batch) {
    const float *V = cam->V;
    float tx = row4_view(p, V, 0, batch), ty = row4_view(p, V, 1, batch);
    float limx = 1.3f * cam->tan_fovx, limy = 1.3f * cam->tan_fovy;
    float cx = fminf(fmaxf(tx / tz, -limx), limx) * tz;
    float cy = fminf(fmaxf(ty / tz, -limy), limy) * tz;
    float J[9] = {0}, Wm[9], Wt[9], Jt[9], A[9], B[9], C[9];
    J[0] = fx / tz;
    J[2] = -(fx * cx) / (tz * tz);
    J[4] = fy / tz;
    J[5] = -(fy * cy) / (tz * tz);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Wm[i * 3 + j] = V[j * 4 + i];            /* V[:3,:3]^T */
            Wt[i * 3 + j] = V[i * 4 + j];
            Jt[i * 3 + j] = J[j * 3 + i];
        }
    mm3_fma(J, Wm, A); mm3(A, S, B);
    if (batch <= 3) mm3_small(B, Wt, C); else mm3_fma(B, Wt, C);
    mm3(C, Jt, D);
}

/* One Gaussian of stage 1.  Returns 0 when culled (z_view < 0.2). */
static int project_one(const OrcCamera *cam, const float *p, const float *s, const float *q, int64_t n_all, int64_t batch,
                       float *xy, float *c2, float *depth, float *inv, float *radius, float *bbox) {
    const float *V = cam->V, *F = cam->F;
    /* the cull multiplies ALL n points at once (utils.py:305-307), everything after it the visible ones (:79-85) */
    if (!(row4_view(p, V, 2, n_all) >= 0.2f)) return 0;                       /* utils.py:293-310 */
    float tz = row4_view(p, V, 2, batch);
    /* Sigma3D: F.normalize then build_rotation's own normalisation (gaussians.py:59-69) */
    float n1 = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
    n1 = fmaxf(n1, 1e-12f);
    float a0 = q[0] / n1, a1 = q[1] / n1, a2 = q[2] / n1, a3 = q[3] / n1;
    float n2 = sqrtf(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3);
    float w = a0 / n2, x = a1 / n2, y = a2 / n2, z = a3 / n2;
    float R[9];
    R[0] = 1.0f - 2.0f * (y * y + z * z);
    R[1] = 2.0f * (x * y - w * z);
    R[2] = 2.0f * (x * z + w * y);
    R[3] = 2.0f * (x * y + w * z);
    R[4] = 1.0f - 2.0f * (x * x + z * z);
    R[5] = 2.0f * (y * z - w * x);
    R[6] = 2.0f * (x * z - w * y);
    R[7] = 2.0f * (y * z + w * x);
    R[8] = 1.0f - 2.0f * (x * x + y * y);
    float M[9], S[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i * 3 + j] = R[i * 3 + j] * s[j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            S[i * 3 + j] = (M[i * 3 + 0] * M[j * 3 + 0] + M[i * 3 + 1] * M[j * 3 + 1]) + M[i * 3 + 2] * M[j * 3 + 2];

    /* pixel position (gaussian_scene.py:87-97, utils.py:313-317) */
    float cw = row4(p, F, 3);
    float ndcx = row4(p, F, 0) / cw, ndcy = row4(p, F, 1) / cw;
    float xp = (ndcx + 1.0f) * ((float)cam->width - 1.0f) * 0.5f;
    float yp = (ndcy + 1.0f) * ((float)cam->height - 1.0f) * 0.5f;

    /* EWA 2D covariance (utils.py:320-354) */
    float D[9];
    ewa2d(cam, cam->fx, cam->fy, p, tz, S, D, batch);
    float ca = D[0], cb = D[1], cc = D[3], cd = D[4];

    /* inverse (utils.py:368-393) */
    float det = ca * cd - cb * cc;
    det = fmaxf(det, 1e-3f);
    inv[0] = cd / det; inv[1] = -cb / det; inv[2] = -cc / det; inv[3] = ca / det;

    /* radius (utils.py:409-423) */
    float mid = 0.5f * (ca + cd);
    float det2 = ca * cd - cb * cb;
    float m = fmaxf(mid * mid - det2, 0.1f);
    float root = sqrtf(m);
    float lam = fmaxf(mid + root, mid - root);
    float r = ceilf(3.0f * sqrtf(lam));

    xy[0] = xp; xy[1] = yp;
    c2[0] = ca; c2[1] = cb; c2[2] = cc; c2[3] = cd;
    *depth = tz; *radius = r;
    bbox[0] = floorf(xp - r); bbox[1] = ceilf(xp + r);   /* min_x, max_x */
    bbox[2] = floorf(yp - r); bbox[3] = ceilf(yp + r);   /* min_y, max_y */
    return 1;
}

/* stable LSD radix sort of (key,val) pairs by 32-bit key */
static void radix_sort_pairs(uint32_t *key, int64_t *val, int64_t n) {
    uint32_t *k2 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    int64_t *v2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int pass = 0; pass < 4; ++pass) {
        int64_t hist[257] = {0};
        int sh = pass * 8;
        for (int64_t i = 0; i < n; ++i) hist[((key[i] >> sh) & 255u) + 1]++;
        for (int b = 0; b < 256; ++b) hist[b + 1] += hist[b];
        for (int64_t i = 0; i < n; ++i) {
            int64_t d = hist[(key[i] >> sh) & 255u]++;
            k2[d] = key[i]; v2[d] = val[i];
        }
        uint32_t *tk = key; key = k2; k2 = tk;
        int64_t *tv = val; val = v2; v2 = tv;
    }
    free(k2); free(v2);   /* 4 passes: data is back in the caller's arrays */
}

/*
 * Stage 1 for n Gaussians.  Outputs are depth-sorted (ties by original index) and compacted to
 * *n_vis rows; every output array must hold n rows.  order[i] = original index of sorted row i.
 */
int orc_preprocess(const OrcCamera *cam, const float *points, const float *colors, const float *scales,
                   const float *quats, const float *opacity, int64_t n,
                   float *xy, float *colors_out, float *cov2d, float *depths, float *inv_cov, float *radius,
                   float *min_x, float *max_x, float *min_y, float *max_y, float *sig_op,
                   int64_t *order, int64_t *n_vis) {
    size_t cap = (size_t)(n > 0 ? n : 1);
    float *t_xy = malloc(cap * 2 * 4), *t_c2 = malloc(cap * 4 * 4), *t_inv = malloc(cap * 4 * 4);
    float *t_r = malloc(cap * 4), *t_bb = malloc(cap * 4 * 4);
    uint32_t *key = malloc(cap * 4);
    int64_t *val = malloc(cap * 8);
    int64_t m = 0, batch = 0;
    for (int64_t i = 0; i < n; ++i) batch += row4_view(points + 3 * i, cam->V, 2, n) >= 0.2f;   /* the reference's N_vis */
    for (int64_t i = 0; i < n; ++i) {
        float d;
        if (project_one(cam, points + 3 * i, scales + 3 * i, quats + 4 * i, n, batch, t_xy + 2 * m, t_c2 + 4 * m, &d,
                        t_inv + 4 * m, t_r + m, t_bb + 4 * m)) {
            memcpy(&key[m], &d, 4);       /* d >= 0.2 > 0: IEEE bits are monotone in d */
            val[m] = m;                   /* position in the compacted arrays */
            order[m] = i;
            ++m;
        }
    }
    int64_t *orig = malloc(cap * 8);
    memcpy(orig, order, (size_t)m * 8);
    radix_sort_pairs(key, val, m);
    for (int64_t i = 0; i < m; ++i) {
        int64_t j = val[i], g = orig[j];
        order[i] = g;
        memcpy(&depths[i], &key[i], 4);
        xy[2 * i] = t_xy[2 * j]; xy[2 * i + 1] = t_xy[2 * j + 1];
        for (int c = 0; c < 4; ++c) { cov2d[4 * i + c] = t_c2[4 * j + c]; inv_cov[4 * i + c] = t_inv[4 * j + c]; }
        for (int c = 0; c < 3; ++c) colors_out[3 * i + c] = colors[3 * g + c];
        radius[i] = t_r[j];
        min_x[i] = t_bb[4 * j]; max_x[i] = t_bb[4 * j + 1]; min_y[i] = t_bb[4 * j + 2]; max_y[i] = t_bb[4 * j + 3];
        sig_op[i] = sigmoid_at(opacity[g], i, m, g_ref_threads);     /* gaussian_scene.py:143: on the SORTED array */
    }
    *n_vis = m;
    free(t_xy); free(t_c2); free(t_inv); free(t_r); free(t_bb); free(key); free(val); free(orig);
    return 0;
}

/* ------------------------------------------------------------------------------ stage 2 */

typedef struct {
    int W, H, tile, ntx, nty;
    const float *means, *colors, *inv, *op2;
    const int64_t *tile_start;   /* ntx*nty+1 */
    const int32_t *tile_items;
    float *image;                /* (W,H,3) indexed [x,y] */
    int wx0, wx1, wy0, wy1;      /* tile-index window */
    int64_t pairs;
    pthread_mutex_t mu;
    /* the work counter is written by every thread: keep it off the cache lines of the read-only
       fields above, or each fetch-add evicts the pointers the inner loops dereference */
    char pad0[128];
    int next;                    /* work counter (tile id) */
    char pad1[128];
} RenderJob;

/* orc_set_exact(1): the same rule set evaluated in float64 from the same float32 stage-1 arrays -- not the
 * reference's arithmetic (that is float32, below) but its exact-arithmetic limit.  Used to tell WHICH side is
 * off when the float32 restatement and the kernel disagree on ill-conditioned footprints (a 200:1 ellipse seen
 * 70 px along its ridge: the three products of e Q e^T are ~5e4 each and cancel to ~3; float32 loses 1e-3). */
static int g_exact = 0;
void orc_set_exact(int on) { g_exact = on; }

static void render_one_tile_exact(RenderJob *jb, int tix, int tiy) {
    int T = jb->tile, x0 = tix * T, y0 = tiy * T, H = jb->H;
    int64_t b = jb->tile_start[tix * jb->nty + tiy], e = jb->tile_start[tix * jb->nty + tiy + 1];
    if (b == e) return;
    for (int px = x0; px < x0 + T; ++px)
        for (int py = y0; py < y0 + T; ++py) {
            double Tw = 1.0, C0 = 0.0, C1 = 0.0, C2 = 0.0;
            for (int64_t k = b; k < e; ++k) {
                int32_t g = jb->tile_items[k];
                const float *Q = jb->inv + 4 * (int64_t)g;
                double e0 = (double)jb->means[2 * (int64_t)g] - px, e1 = (double)jb->means[2 * (int64_t)g + 1] - py;
                double d0 = -0.5 * e0, d1 = -0.5 * e1;
                double t0 = d0 * Q[0] + d1 * Q[2], t1 = d0 * Q[1] + d1 * Q[3];
                double sig = 1.0 / (1.0 + exp(-(double)jb->op2[g]));     /* op2 holds sigmoid_opacity in this mode */
                double alpha = exp(t0 * e0 + t1 * e1) * sig;
                double test = Tw * (1.0 - alpha);
                if (test < 0.000001) break;
                double ta = Tw * alpha;
                const float *c = jb->colors + 3 * (int64_t)g;
                C0 += ta * c[0]; C1 += ta * c[1]; C2 += ta * c[2];
                Tw = test;
            }
            float *o = jb->image + ((int64_t)px * H + py) * 3;
            o[0] = (float)C0; o[1] = (float)C1; o[2] = (float)C2;
        }
}

static void render_one_tile(RenderJob *jb, int tix, int tiy) {
    if (g_exact) { render_one_tile_exact(jb, tix, tiy); return; }
    int T = jb->tile, x0 = tix * T, y0 = tiy * T, H = jb->H;
    int64_t b = jb->tile_start[tix * jb->nty + tiy], e = jb->tile_start[tix * jb->nty + tiy + 1];
    if (b == e) return;                                    /* gaussian_scene.py:219-220 */
    const int32_t *restrict items = jb->tile_items;
    const float *restrict means = jb->means, *restrict inv = jb->inv, *restrict op2 = jb->op2,
                *restrict colors = jb->colors;
    for (int px = x0; px < x0 + T; ++px)
        for (int py = y0; py < y0 + T; ++py) {
            float Tw = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
            float fx_ = (float)px, fy_ = (float)py;
            for (int64_t k = b; k < e; ++k) {              /* render_pixel, :146-171 */
                int32_t g = items[k];
                const float *Q = inv + 4 * (int64_t)g;
                float e0 = means[2 * (int64_t)g] - fx_, e1 = means[2 * (int64_t)g + 1] - fy_;
                /* utils.py:363-364 as torch executes it (probe_torch_order.py): (1,2) @ (2,2) is an sgemm,
                 * one FMA per output; (1,2) @ (2,1) is a dot whose two products are rounded before the sum */
                float d0 = -0.5f * e0, d1 = -0.5f * e1;
                float t0 = fmaf(d1, Q[2], d0 * Q[0]);
                float t1 = fmaf(d1, Q[3], d0 * Q[1]);
                float w = expf(t0 * e0 + t1 * e1);
                float alpha = w * op2[g];                   /* second sigmoid, :164 */
                float test = Tw * (1.0f - alpha);
                if (test < 0.000001f) break;                /* return before accumulating, :166 */
                float ta = Tw * alpha;
                const float *c = colors + 3 * (int64_t)g;
                C0 += ta * c[0]; C1 += ta * c[1]; C2 += ta * c[2];
                Tw = test;
            }
            float *o = jb->image + ((int64_t)px * H + py) * 3;
            o[0] = C0; o[1] = C1; o[2] = C2;
        }
}

static void *render_worker(void *arg) {
    RenderJob *jb = (RenderJob *)arg;
    int nwx = jb->wx1 - jb->wx0, nwy = jb->wy1 - jb->wy0;
    int64_t local_pairs = 0;
    for (;;) {
        int id = __atomic_fetch_add(&jb->next, 1, __ATOMIC_RELAXED);
        if (id >= nwx * nwy) break;
        int tix = jb->wx0 + id / nwy, tiy = jb->wy0 + id % nwy;
        int64_t len = jb->tile_start[tix * jb->nty + tiy + 1] - jb->tile_start[tix * jb->nty + tiy];
        local_pairs += len * jb->tile * jb->tile;
        render_one_tile(jb, tix, tiy);
    }
    pthread_mutex_lock(&jb->mu);
    jb->pairs += local_pairs;
    pthread_mutex_unlock(&jb->mu);
    return NULL;
}

/* NaN compares false in the reference's masks: map it to an empty range (lo > hi). */
static inline int clamp_lo(double v, int lo, int hi) { return !(v == v) ? hi + 1 : (v < lo ? lo : (v > hi ? hi : (int)v)); }
static inline int clamp_hi(double v, int lo, int hi) { return !(v == v) ? lo - 1 : (v < lo ? lo : (v > hi ? hi : (int)v)); }

/*
 * Stage 2 on depth-sorted stage-1 arrays (the argument list of the reference's native boundary,
 * splat/c/render.cu:90-101, with CPU semantics).  image: (W,H,3) float32 indexed [x,y], zeroed
 * here.  window = {tx0,tx1,ty0,ty1} in tile indices or NULL for the whole frame.
 * pairs_out: (list length x tile^2) summed over rendered tiles.  instances_out: total list length
 * over ALL tiles (the D of SURVEY.md section 8).
 */
int orc_render(int H, int W, int tile, const float *means, const float *colors, const float *inv_cov,
               const float *min_x, const float *max_x, const float *min_y, const float *max_y,
               const float *sig_op, int64_t n, float *image, int nthreads, const int32_t *window,
               int64_t *pairs_out, int64_t *instances_out) {
    if (tile <= 0 || W <= 0 || H <= 0) return -1;
    memset(image, 0, (size_t)W * H * 3 * sizeof(float));
    /* range(0, W - tile, tile) has ceil((W - tile)/tile) entries (gaussian_scene.py:208,214) */
    int ntx = W > tile ? (W - tile + tile - 1) / tile : 0;
    int nty = H > tile ? (H - tile + tile - 1) / tile : 0;
    if (pairs_out) *pairs_out = 0;
    if (instances_out) *instances_out = 0;
    if (ntx == 0 || nty == 0 || n == 0) return 0;
    int64_t ntiles = (int64_t)ntx * nty;
    int64_t *start = calloc((size_t)ntiles + 1, 8);
    int32_t *rect = malloc((size_t)n * 4 * 4);
    float *op2 = malloc((size_t)n * 4);
    /* tile test (gaussian_scene.py:209-217): min <= x0 + T and max >= x0, x0 = t*T */
    for (int64_t i = 0; i < n; ++i) {
        double T = tile;
        int lx = clamp_lo(ceil(((double)min_x[i] - T) / T), 0, ntx), hx = clamp_hi(floor((double)max_x[i] / T), -1, ntx - 1);
        int ly = clamp_lo(ceil(((double)min_y[i] - T) / T), 0, nty), hy = clamp_hi(floor((double)max_y[i] / T), -1, nty - 1);
        if (lx > hx || ly > hy) { lx = 0; hx = -1; ly = 0; hy = -1; }
        rect[4 * i] = lx; rect[4 * i + 1] = hx; rect[4 * i + 2] = ly; rect[4 * i + 3] = hy;
        for (int a = lx; a <= hx; ++a)
            for (int b = ly; b <= hy; ++b) start[(int64_t)a * nty + b + 1]++;
        op2[i] = g_exact ? sig_op[i] : sigmoidf_(sig_op[i]);
    }
    for (int64_t t = 0; t < ntiles; ++t) start[t + 1] += start[t];
    int64_t D = start[ntiles];
    int32_t *items = malloc((size_t)(D > 0 ? D : 1) * 4);
    int64_t *fill = malloc((size_t)ntiles * 8);
    memcpy(fill, start, (size_t)ntiles * 8);
    for (int64_t i = 0; i < n; ++i)          /* ascending i == depth order is preserved per tile */
        for (int a = rect[4 * i]; a <= rect[4 * i + 1]; ++a)
            for (int b = rect[4 * i + 2]; b <= rect[4 * i + 3]; ++b) items[fill[(int64_t)a * nty + b]++] = (int32_t)i;

    RenderJob jb;
    memset(&jb, 0, sizeof jb);
    jb.W = W; jb.H = H; jb.tile = tile; jb.ntx = ntx; jb.nty = nty;
    jb.means = means; jb.colors = colors; jb.inv = inv_cov; jb.op2 = op2;
    jb.tile_start = start; jb.tile_items = items; jb.image = image;
    jb.wx0 = 0; jb.wx1 = ntx; jb.wy0 = 0; jb.wy1 = nty;
    if (window) {
        jb.wx0 = window[0] < 0 ? 0 : window[0]; jb.wx1 = window[1] > ntx ? ntx : window[1];
        jb.wy0 = window[2] < 0 ? 0 : window[2]; jb.wy1 = window[3] > nty ? nty : window[3];
    }
    pthread_mutex_init(&jb.mu, NULL);
    if (jb.wx1 > jb.wx0 && jb.wy1 > jb.wy0) {
        if (nthreads < 1) nthreads = 1;
        if (nthreads > 256) nthreads = 256;
        pthread_t th[256];
        for (int t = 1; t < nthreads; ++t) pthread_create(&th[t], NULL, render_worker, &jb);
        render_worker(&jb);
        for (int t = 1; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    pthread_mutex_destroy(&jb.mu);
    if (pairs_out) *pairs_out = jb.pairs;
    if (instances_out) *instances_out = D;
    free(start); free(rect); free(op2); free(items); free(fill);
    return 0;
}

/*
 * The reference's CUDA-kernel semantics, restated on the CPU (splat/c/render.cu:21-87, device
 * function :5-19).  PARITY UNPINNED: the reference kernel needs nvcc + a CUDA GPU, so no golden
 * vector of it can be produced here; this restatement follows the source line by line:
 * every pixel of the frame (partial edge tiles included, :41-44), loop over ALL rows in order,
 * inclusive per-pixel bounding-box test (:55-60), mean truncated to int by the device function's
 * int parameters (:8-9), power = dx a dx + 2 dx dy b + dy dy c with a,b,c = inv[4i], inv[4i+1],
 * inv[4i+3] (:61-68), alpha = min(.99, opacity * strength) (:70-71), break when
 * T (1 - alpha) < 0.001 (:72-76).  image: (H,W,3) indexed [y][x] (:83-85).  `opacity` is used as
 * given (the caller passes sigmoid_opacity, gaussian_scene.py:281).
 */
typedef struct {
    int W, H;
    const float *means, *colors, *inv, *min_x, *max_x, *min_y, *max_y, *opacity;
    int64_t n;
    float *image;
    int next;
} CudaSemJob;

static void *cuda_sem_worker(void *arg) {
    CudaSemJob *jb = (CudaSemJob *)arg;
    for (;;) {
        int py = __atomic_fetch_add(&jb->next, 1, __ATOMIC_RELAXED);
        if (py >= jb->H) break;
        /* rows whose bbox spans this scanline, in order: an exact prefilter of the y test */
        int32_t *cand = malloc((size_t)(jb->n > 0 ? jb->n : 1) * 4);
        int64_t nc = 0;
        for (int64_t i = 0; i < jb->n; ++i)
            if ((float)py >= jb->min_y[i] && (float)py <= jb->max_y[i]) cand[nc++] = (int32_t)i;
        for (int px = 0; px < jb->W; ++px) {
            float T = 1.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
            for (int64_t k = 0; k < nc; ++k) {
                int64_t i = cand[k];
                if (!((float)px >= jb->min_x[i] && (float)px <= jb->max_x[i])) continue;
                int ix = (int)jb->means[2 * i], iy = (int)jb->means[2 * i + 1];
                float dx = (float)(px - ix), dy = (float)(py - iy);
                float a = jb->inv[4 * i], b = jb->inv[4 * i + 1], c = jb->inv[4 * i + 3];
                float power = dx * a * dx + 2 * dx * dy * b + dy * dy * c;
                float strength = expf(-0.5f * power);
                float alpha = fminf(.99f, jb->opacity[i] * strength);
                float test = T * (1 - alpha);
                if (test < 0.001f) break;
                c0 += T * alpha * jb->colors[3 * i];
                c1 += T * alpha * jb->colors[3 * i + 1];
                c2 += T * alpha * jb->colors[3 * i + 2];
                T = test;
            }
            float *o = jb->image + ((int64_t)py * jb->W + px) * 3;
            o[0] = c0; o[1] = c1; o[2] = c2;
        }
        free(cand);
    }
    return NULL;
}

int orc_render_cuda_semantics(int H, int W, const float *means, const float *colors, const float *inv_cov,
                              const float *min_x, const float *max_x, const float *min_y, const float *max_y,
                              const float *opacity, int64_t n, float *image, int nthreads) {
    if (W <= 0 || H <= 0) return -1;
    CudaSemJob jb = {W, H, means, colors, inv_cov, min_x, max_x, min_y, max_y, opacity, n, image, 0};
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    for (int t = 1; t < nthreads; ++t) pthread_create(&th[t], NULL, cuda_sem_worker, &jb);
    cuda_sem_worker(&jb);
    for (int t = 1; t < nthreads; ++t) pthread_join(th[t], NULL);
    return 0;
}

/* ---------------------------------------------------------------- GSX_SEM_STD_3DGS (extension)
 *
 * CPU restatement of the forward pass of the published 3D Gaussian Splatting rasteriser (Kerbl,
 * Kopanas, Leimkuehler, Drettakis, SIGGRAPH 2023; its CUDA rasteriser "diff-gaussian-rasterization",
 * forward pass: preprocess -> duplicate with keys -> radix sort -> per-tile ranges -> render).
 * PARITY UNPINNED: that rasteriser is NOT part of /root/reference (SURVEY.md 8(f) rank 3 lists the
 * mode as a build extension) and cannot be built here (CUDA); this follows the published
 * algorithm step by step:
 *   stage 1  z_view <= 0.2 culled; quaternion normalised once; Sigma = (R S)(R S)^T; p_w = 1/(w+1e-7);
 *            pixel = ((ndc + 1) extent - 1)/2; focal = extent/(2 tan(fov/2)); EWA with the 1.3 clamp;
 *            cov00 += 0.3, cov11 += 0.3; det == 0 dropped; conic = (c, -b, a)/det;
 *            lambda = mid +- sqrt(max(0.1, mid^2 - det)); r = ceil(3 sqrt(max lambda));
 *            tiles [(int)((p - r)/T), (int)((p + r + T - 1)/T)) clamped to the grid; empty dropped;
 *            opacity = sigmoid(logit).
 *   order    per tile by view depth, ties by Gaussian index (stable radix sort of tile|depth keys).
 *   stage 2  pixel (px,py) at integer coordinates; d = mean - pixel; power = -0.5 (A dx^2 + C dy^2)
 *            - B dx dy; power > 0 skipped; alpha = min(0.99, opacity exp(power)); alpha < 1/255
 *            skipped; T(1 - alpha) < 1e-4 stops the pixel; C += c alpha T; out = C + T bg.
 * image: (H,W,3) float32 indexed [y][x].  window = {tx0,tx1,ty0,ty1} (tile indices) or NULL.
 * stage1 (optional, n x 8): x, y, conic A, B, C, radius, depth, opacity per Gaussian in INPUT order
 * (NaN row when culled) -- for bit-level comparison of stage 1.
 */
typedef struct {
    int W, H, tile, ntx, nty;
    const float *g;              /* n x 8 packed stage-1 rows */
    const float *colors;
    const int64_t *tile_start;
    const int32_t *tile_items;
    float bg[3];
    float *image;
    int wx0, wx1, wy0, wy1;
    char pad0[128];
    int next;
    char pad1[128];
} StdJob;

static void *std_worker(void *arg) {
    StdJob *jb = (StdJob *)arg;
    int nwx = jb->wx1 - jb->wx0, nwy = jb->wy1 - jb->wy0, T_ = jb->tile;
    const float *restrict G = jb->g, *restrict colors = jb->colors;
    const int32_t *restrict items = jb->tile_items;
    for (;;) {
        int id = __atomic_fetch_add(&jb->next, 1, __ATOMIC_RELAXED);
        if (id >= nwx * nwy) break;
        int tix = jb->wx0 + id / nwy, tiy = jb->wy0 + id % nwy;
        int64_t b = jb->tile_start[(int64_t)tix * jb->nty + tiy], e = jb->tile_start[(int64_t)tix * jb->nty + tiy + 1];
        for (int py = tiy * T_; py < tiy * T_ + T_ && py < jb->H; ++py)
            for (int px = tix * T_; px < tix * T_ + T_ && px < jb->W; ++px) {
                float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
                float fx_ = (float)px, fy_ = (float)py;
                for (int64_t k = b; k < e; ++k) {
                    const float *r = G + 8 * (int64_t)items[k];
                    float dx = r[0] - fx_, dy = r[1] - fy_;
                    float power = -0.5f * (r[2] * dx * dx + r[4] * dy * dy) - r[3] * dx * dy;
                    if (power > 0.0f) continue;
                    float alpha = fminf(0.99f, r[7] * expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    float test_T = T * (1.0f - alpha);
                    if (test_T < 0.0001f) break;
                    const float *c = colors + 3 * (int64_t)items[k];
                    C0 += c[0] * alpha * T; C1 += c[1] * alpha * T; C2 += c[2] * alpha * T;
                    T = test_T;
                }
                float *o = jb->image + ((int64_t)py * jb->W + px) * 3;
                o[0] = C0 + T * jb->bg[0]; o[1] = C1 + T * jb->bg[1]; o[2] = C2 + T * jb->bg[2];
            }
    }
    return NULL;
}

static inline int std_tile_index(float v, int nt) {
    if (!(v == v)) return -1;
    const float big = 1073741824.0f;
    int i = (int)fminf(fmaxf(v, -big), big);
    return i < 0 ? 0 : (i > nt ? nt : i);
}

int orc_render_std3dgs(const OrcCamera *cam, const float *points, const float *colors, const float *scales,
                       const float *quats, const float *opacity_logit, int64_t n, int tile, const float *bg,
                       float *image, int nthreads, const int32_t *window, int64_t *n_vis_out,
                       int64_t *instances_out, float *stage1) {
    int W = cam->width, H = cam->height;
    if (tile <= 0 || W <= 0 || H <= 0) return -1;
    memset(image, 0, (size_t)W * H * 3 * sizeof(float));
    int ntx = (W + tile - 1) / tile, nty = (H + tile - 1) / tile;
    size_t cap = (size_t)(n > 0 ? n : 1);
    float *G = malloc(cap * 8 * 4);
    int32_t *rect = malloc(cap * 4 * 4);
    uint32_t *key = malloc(cap * 4);
    int64_t *val = malloc(cap * 8);
    int64_t m = 0;
    const float *V = cam->V, *F = cam->F;
    float fx = (float)W / (2.0f * cam->tan_fovx), fy = (float)H / (2.0f * cam->tan_fovy);
    for (int64_t i = 0; i < n; ++i) {
        const float *p = points + 3 * i, *s = scales + 3 * i, *q = quats + 4 * i;
        float *g = G + 8 * i;
        for (int c = 0; c < 8; ++c) g[c] = NAN;
        rect[4 * i] = 1; rect[4 * i + 1] = 0; rect[4 * i + 2] = 1; rect[4 * i + 3] = 0;
        float tz = row4(p, V, 2);
        if (!(tz > 0.2f)) continue;
        memcpy(&key[m], &tz, 4);
        val[m++] = i;
        float n1 = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
        n1 = fmaxf(n1, 1e-12f);
        float w = q[0] / n1, x = q[1] / n1, y = q[2] / n1, z = q[3] / n1;
        float R[9], M[9], S[9], D[9];
        R[0] = 1.0f - 2.0f * (y * y + z * z); R[1] = 2.0f * (x * y - w * z); R[2] = 2.0f * (x * z + w * y);
        R[3] = 2.0f * (x * y + w * z); R[4] = 1.0f - 2.0f * (x * x + z * z); R[5] = 2.0f * (y * z - w * x);
        R[6] = 2.0f * (x * z - w * y); R[7] = 2.0f * (y * z + w * x); R[8] = 1.0f - 2.0f * (x * x + y * y);
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) M[a * 3 + b] = R[a * 3 + b] * s[b];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                S[a * 3 + b] = (M[a * 3 + 0] * M[b * 3 + 0] + M[a * 3 + 1] * M[b * 3 + 1]) + M[a * 3 + 2] * M[b * 3 + 2];
        float pw = 1.0f / (row4(p, F, 3) + 0.0000001f);
        float ndcx = row4(p, F, 0) * pw, ndcy = row4(p, F, 1) * pw;
        float xp = ((ndcx + 1.0f) * (float)W - 1.0f) * 0.5f, yp = ((ndcy + 1.0f) * (float)H - 1.0f) * 0.5f;
        ewa2d(cam, fx, fy, p, tz, S, D, INT64_MAX);
        float ca = D[0] + 0.3f, cb = D[1], cd = D[4] + 0.3f;
        float det = ca * cd - cb * cb;
        if (det == 0.0f) continue;
        float det_inv = 1.0f / det;
        float mid = 0.5f * (ca + cd);
        float root = sqrtf(fmaxf(0.1f, mid * mid - det));
        float lam = fmaxf(mid + root, mid - root);
        float r = ceilf(3.0f * sqrtf(lam));
        g[0] = xp; g[1] = yp; g[2] = cd * det_inv; g[3] = -cb * det_inv; g[4] = ca * det_inv;
        g[5] = r; g[6] = tz; g[7] = sigmoidf_(opacity_logit[i]);
        float T = (float)tile;
        int lx = std_tile_index((xp - r) / T, ntx), hx = std_tile_index(((xp + r + T) - 1.0f) / T, ntx);
        int ly = std_tile_index((yp - r) / T, nty), hy = std_tile_index(((yp + r + T) - 1.0f) / T, nty);
        if (lx < 0 || hx < 0 || ly < 0 || hy < 0 || (hx - lx) * (hy - ly) <= 0) continue;
        rect[4 * i] = lx; rect[4 * i + 1] = hx - 1; rect[4 * i + 2] = ly; rect[4 * i + 3] = hy - 1;
    }
    if (stage1) memcpy(stage1, G, (size_t)n * 8 * 4);
    radix_sort_pairs(key, val, m);                 /* stable: ties keep ascending Gaussian index */
    int64_t ntiles = (int64_t)ntx * nty;
    int64_t *start = calloc((size_t)ntiles + 1, 8);
    for (int64_t k = 0; k < m; ++k) {
        int64_t i = val[k];
        for (int a = rect[4 * i]; a <= rect[4 * i + 1]; ++a)
            for (int b = rect[4 * i + 2]; b <= rect[4 * i + 3]; ++b) start[(int64_t)a * nty + b + 1]++;
    }
    for (int64_t t = 0; t < ntiles; ++t) start[t + 1] += start[t];
    int64_t Dn = start[ntiles];
    int32_t *items = malloc((size_t)(Dn > 0 ? Dn : 1) * 4);
    int64_t *fill = malloc((size_t)(ntiles > 0 ? ntiles : 1) * 8);
    memcpy(fill, start, (size_t)ntiles * 8);
    for (int64_t k = 0; k < m; ++k) {
        int64_t i = val[k];
        for (int a = rect[4 * i]; a <= rect[4 * i + 1]; ++a)
            for (int b = rect[4 * i + 2]; b <= rect[4 * i + 3]; ++b) items[fill[(int64_t)a * nty + b]++] = (int32_t)i;
    }
    StdJob jb;
    memset(&jb, 0, sizeof jb);
    jb.W = W; jb.H = H; jb.tile = tile; jb.ntx = ntx; jb.nty = nty;
    jb.g = G; jb.colors = colors; jb.tile_start = start; jb.tile_items = items; jb.image = image;
    jb.bg[0] = bg ? bg[0] : 0.0f; jb.bg[1] = bg ? bg[1] : 0.0f; jb.bg[2] = bg ? bg[2] : 0.0f;
    jb.wx0 = 0; jb.wx1 = ntx; jb.wy0 = 0; jb.wy1 = nty;
    if (window) {
        jb.wx0 = window[0] < 0 ? 0 : window[0]; jb.wx1 = window[1] > ntx || window[1] <= 0 ? ntx : window[1];
        jb.wy0 = window[2] < 0 ? 0 : window[2]; jb.wy1 = window[3] > nty || window[3] <= 0 ? nty : window[3];
    }
    int64_t inst = 0;
    for (int a = jb.wx0; a < jb.wx1; ++a)
        for (int b = jb.wy0; b < jb.wy1; ++b) inst += start[(int64_t)a * nty + b + 1] - start[(int64_t)a * nty + b];
    if (jb.wx1 > jb.wx0 && jb.wy1 > jb.wy0) {
        if (nthreads < 1) nthreads = 1;
        if (nthreads > 256) nthreads = 256;
        pthread_t th[256];
        for (int t = 1; t < nthreads; ++t) pthread_create(&th[t], NULL, std_worker, &jb);
        std_worker(&jb);
        for (int t = 1; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    if (n_vis_out) *n_vis_out = m;
    if (instances_out) *instances_out = inst;
    free(G); free(rect); free(key); free(val); free(start); free(items); free(fill);
    return 0;
}

int orc_version(void) { return 3; }
