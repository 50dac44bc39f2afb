// Stage 2 on gfx950: front-to-back alpha compositing of each tile's depth-ordered list.
//
// Reference behaviour restated (paths relative to the reference repository):
//   splat/gaussian_scene.py:146-171  render_pixel: T=1, C=0; alpha = w * sigmoid(opacity);
//                                    test = T(1-alpha); if test < 1e-6 return C (before adding);
//                                    C += T alpha c; T = test
//   splat/utils.py:357-365           w = exp(-1/2 d Q d^T), d = mean - pixel, full 2x2 Q
//   splat/gaussian_scene.py:173-198  render_tile: every listed Gaussian at every pixel of the tile,
//                                    pixel centres at integer coordinates
//   splat/gaussian_scene.py:200-238  render_image: image[x][y][c]
//
// CDNA4 mapping.  Under these semantics nothing is culled per pixel, so a tile costs
// |list| x 256 weight evaluations (~11 VALU ops + one v_exp_f32 each) against 48 B of record
// traffic per list entry: the kernel is VALU-bound, not HBM-bound.  Hence:
//   - one 64-lane wavefront per 16x16 tile, 4 pixels per lane along the memory-contiguous image
//     axis, so each record fetched from LDS is amortised over 4 evaluations, the terms of the
//     quadratic form that depend only on the shared coordinate are computed once per lane, and
//     the per-lane output is 48 contiguous bytes;
//   - a single-wave workgroup needs no cross-wave barrier and no LDS flag: "every pixel of the
//     tile is saturated" is one __ballot over the wave;
//   - each lane gathers one 48-B record per batch of 64 (3 x dwordx4) into LDS; the k-loop then
//     reads record k with uniform-address (broadcast) ds_reads;
//   - the saturation test of the reference (stop when T(1-alpha) < 1e-6, before accumulating)
//     is kept exact but off the common path: one v_min3 + v_min + v_cmp per record decides, for
//     the whole wave, whether any of its 256 pixels stops at this record; only then are the
//     per-pixel selects executed;
//   - blockIdx -> tile mapping gives each XCD a contiguous stripe of tiles, so neighbouring
//     tiles -- which share most of their Gaussians -- hit the same 4 MiB L2.
//
// Arithmetic.  This translation unit is compiled with -ffp-contract=off and every fused
// operation is an explicit fmaf, so the result does not depend on how the compiler would have
// contracted each template instance: both output layouts give bit-identical pixels.  With
// Q'' = -1/2 log2(e) Q (packed by the projection kernels), e = mean - pixel:
//     alpha = exp2(e0^2 Q''00 + e0 e1 (Q''01 + Q''10) + e1^2 Q''11 + log2 op)    (one v_exp_f32)
// which is the reference's exp(-1/2 e Q e^T) * op with the four products re-associated; the
// difference is a few ulp of the largest product, like the reference's own float32 rounding.
// T(1 - alpha) is evaluated as T - T alpha (1 ulp).
#include "gsx_internal.h"

namespace gsx {
namespace {

constexpr float kStopRefCpu = 0.000001f;  // gaussian_scene.py:153

// Contiguous-chunk remap: hardware places block b on XCD b % 8; give XCD x the x-th eighth of
// the tile list.  Bijective for every n_tiles.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    uint32_t xcd = b & 7u, i = b >> 3, q = n >> 3, r = n & 7u;
    uint32_t start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return start + i;
}

struct Splat {  // one record, unpacked (wave-uniform values); lop = log2(opacity factor)
    float mx, my, q00, qs, q11, lop, cr, cg, cb;
};

__device__ __forceinline__ Splat read_splat(const float4 (*sh)[64], uint32_t k) {
    const float4 A = sh[0][k], B = sh[1][k];
    const float cb = sh[2][k].x;
    return Splat{A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w, cb};
}

// NPX pixels of one lane against one Gaussian.  The lane's pixels share x (e_s = e_x) and differ in
// y (e_p[j] = e_y); callers always pass (q_ss, q_pp) = (Q''00, Q''11), so the association is fixed:
//   exponent = e_x^2 Q''00 + e_x e_y qs + e_y^2 Q''11 = fma(e_y, fma(e_y, Q''11, e_x qs), e_x^2 Q''00)
template <int NPX>
__device__ __forceinline__ void composite(float e_s, const float (&e_p)[NPX], float q_ss, float qs, float q_pp,
                                          const Splat &g, float (&T)[NPX], float (&c0)[NPX], float (&c1)[NPX],
                                          float (&c2)[NPX]) {
    const float a0 = __builtin_fmaf(e_s * e_s, q_ss, g.lop);  // opacity rides in the exponent
    const float b0 = e_s * qs;
    float ta[NPX], test[NPX];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const float pw = __builtin_fmaf(e_p[j], __builtin_fmaf(e_p[j], q_pp, b0), a0);
        const float alpha = __builtin_amdgcn_exp2f(pw);
        ta[j] = T[j] * alpha;
        test[j] = T[j] - ta[j];
    }
    float m = test[0];
#pragma unroll
    for (int j = 1; j < NPX; ++j) m = fminf(m, test[j]);
    if (__builtin_expect(__any(m < kStopRefCpu), 0)) {
        // some pixel of the wave saturates here (or already has, T = 0): it must not receive
        // this Gaussian and stays at T = 0, like the reference's early return.
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const bool stop = test[j] < kStopRefCpu;
            ta[j] = stop ? 0.0f : ta[j];
            test[j] = stop ? 0.0f : test[j];
        }
    }
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        c0[j] = __builtin_fmaf(ta[j], g.cr, c0[j]);
        c1[j] = __builtin_fmaf(ta[j], g.cg, c1[j]);
        c2[j] = __builtin_fmaf(ta[j], g.cb, c2[j]);
        T[j] = test[j];
    }
}

// Fast path, tile = 16: one wave per tile, 4 pixels per lane.  A lane owns pixels
// (x, y..y+3): x is the coordinate its pixels share, so the x-only terms of the exponent are
// computed once per record.  The assignment (and therefore every bit of the result) is the same
// for both output layouts; only the store addressing differs:
//   GSX_LAYOUT_WH3  out[x][y][c]: the lane's 4 pixels are 48 contiguous bytes (3 x dwordx4);
//   GSX_LAYOUT_HW3  out[y][x][c]: 4 stores of 12 B; the 16 lanes that share a y write 192
//                   contiguous bytes per store instruction.
__global__ void __launch_bounds__(64)
    blend_tile16_kernel(const Record *__restrict__ rec, const uint32_t *__restrict__ vals,
                        const uint2 *__restrict__ ranges, TileGrid g, OutDesc out) {
    __shared__ float4 sh[3][64];
    const int lane = threadIdx.x;
    const uint32_t t = xcd_remap(blockIdx.x, (uint32_t)g.count());
    const int tx = g.wx0 + (int)(t / (uint32_t)g.nwy()), ty = g.wy0 + (int)(t % (uint32_t)g.nwy());
    // WH3: lanes 4q..4q+3 cover one x (contiguous 192 B); HW3: lanes 16q..16q+15 cover one y-quad
    const bool y_contig = out.stride_y < out.stride_x;
    const int px = tx * 16 + (y_contig ? (lane >> 2) : (lane & 15));
    const int py0 = ty * 16 + 4 * (y_contig ? (lane & 3) : (lane >> 4));
    const float cx = (float)px;  // pixel coordinates as floats (exact)
    float cy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cy[j] = (float)(py0 + j);

    float T[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    float c0[4] = {0, 0, 0, 0}, c1[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0};

    const uint2 rg = ranges[t];
    for (uint32_t base = rg.x; base < rg.y; base += 64) {
        const uint32_t nb = min(64u, rg.y - base);
        if ((uint32_t)lane < nb) {
            const Record *p = rec + vals[base + lane];
            sh[0][lane] = p->a;
            sh[1][lane] = p->b;
            sh[2][lane] = p->c;
        }
        __syncthreads();
        for (uint32_t k = 0; k < nb; ++k) {
            const Splat s = read_splat(sh, k);
            // e = mean - pixel, the pixel coordinate formed first, as the reference does
            const float e_x = s.mx - cx;
            float e_y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e_y[j] = s.my - cy[j];
            composite<4>(e_x, e_y, s.q00, s.qs, s.q11, s, T, c0, c1, c2);
        }
        __syncthreads();
        const bool live = (T[0] > 0.0f) | (T[1] > 0.0f) | (T[2] > 0.0f) | (T[3] > 0.0f);
        if (__ballot(live) == 0ull) break;
    }

    float *o = out.ptr + (int64_t)(px - out.x0) * out.stride_x + (int64_t)(py0 - out.y0) * out.stride_y;
    if (y_contig && (reinterpret_cast<uintptr_t>(o) & 15u) == 0) {
        float4 *o4 = reinterpret_cast<float4 *>(o);
        o4[0] = make_float4(c0[0], c1[0], c2[0], c0[1]);
        o4[1] = make_float4(c1[1], c2[1], c0[2], c1[2]);
        o4[2] = make_float4(c2[2], c0[3], c1[3], c2[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float *oj = o + (int64_t)j * out.stride_y;
            oj[0] = c0[j];
            oj[1] = c1[j];
            oj[2] = c2[j];
        }
    }
}

// Any tile size: one wave per tile, one pixel per lane, tile*tile/64 sweeps over the list.
// Same arithmetic as the fast path; exists so that tile_size is a run-time argument as in the
// reference (its notebooks use 16 and 2).
__global__ void __launch_bounds__(64)
    blend_generic_kernel(const Record *__restrict__ rec, const uint32_t *__restrict__ vals,
                         const uint2 *__restrict__ ranges, TileGrid g, OutDesc out) {
    __shared__ float4 sh[3][64];
    const int lane = threadIdx.x;
    const uint32_t t = xcd_remap(blockIdx.x, (uint32_t)g.count());
    const int tx = g.wx0 + (int)(t / (uint32_t)g.nwy()), ty = g.wy0 + (int)(t % (uint32_t)g.nwy());
    const int Ts = g.tile, npx = Ts * Ts;
    const uint2 rg = ranges[t];
    const bool fast_y = out.stride_y < out.stride_x;
    for (int chunk = 0; chunk < npx; chunk += 64) {
        const int p = chunk + lane;
        const bool valid = p < npx;
        const int pf = p % Ts, ps = p / Ts;
        const int px = tx * Ts + (fast_y ? ps : pf), py = ty * Ts + (fast_y ? pf : ps);
        const float fx = (float)px, fy = (float)py;
        float T[1] = {1.0f}, c0[1] = {0.0f}, c1[1] = {0.0f}, c2[1] = {0.0f};
        for (uint32_t base = rg.x; base < rg.y; base += 64) {
            const uint32_t nb = min(64u, rg.y - base);
            if ((uint32_t)lane < nb) {
                const Record *q = rec + vals[base + lane];
                sh[0][lane] = q->a;
                sh[1][lane] = q->b;
                sh[2][lane] = q->c;
            }
            __syncthreads();
            for (uint32_t k = 0; k < nb; ++k) {
                const Splat s = read_splat(sh, k);
                const float e_p[1] = {s.my - fy};
                composite<1>(s.mx - fx, e_p, s.q00, s.qs, s.q11, s, T, c0, c1, c2);
            }
            __syncthreads();
            if (__ballot(valid && T[0] > 0.0f) == 0ull) break;
        }
        if (valid) {
            float *o = out.ptr + (int64_t)(px - out.x0) * out.stride_x + (int64_t)(py - out.y0) * out.stride_y;
            o[0] = c0[0];
            o[1] = c1[0];
            o[2] = c2[0];
        }
    }
}

}  // namespace

hipError_t launch_blend(const Record *rec, const uint32_t *sorted_vals, const uint2 *ranges, const TileGrid &grid,
                        const OutDesc &out, int semantics, hipStream_t s) {
    if (semantics != GSX_SEM_REF_CPU) return hipErrorNotSupported;
    const int64_t nt = grid.count();
    if (nt <= 0) return hipSuccess;
    if (grid.tile == 16) {
        blend_tile16_kernel<<<(unsigned)nt, 64, 0, s>>>(rec, sorted_vals, ranges, grid, out);
    } else {
        blend_generic_kernel<<<(unsigned)nt, 64, 0, s>>>(rec, sorted_vals, ranges, grid, out);
    }
    return hipGetLastError();
}

}  // namespace gsx
