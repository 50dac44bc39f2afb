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
// |list| x 256 weight evaluations (~20 VALU ops + one v_exp_f32 each) against 48 B of record
// traffic per list entry: the kernel is VALU-bound, not HBM-bound.  Hence:
//   - one 64-lane wavefront per 16x16 tile, 4 pixels per lane along the memory-contiguous image
//     axis, so each record fetched from LDS is amortised over 4 evaluations and the per-lane
//     output is 48 contiguous bytes;
//   - a single-wave workgroup needs no cross-wave barrier and no LDS flag: "every pixel of the
//     tile is saturated" is one __ballot over the wave;
//   - each lane gathers one 48-B record per batch of 64 (3 x dwordx4) into LDS; the k-loop then
//     reads record k with three uniform-address (broadcast) ds_read_b128;
//   - blockIdx -> tile mapping gives each XCD a contiguous stripe of tiles, so neighbouring
//     tiles -- which share most of their Gaussians -- hit the same 4 MiB L2.
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

// One Gaussian against one pixel; state (T, C) updated in place.  A saturated pixel is
// represented by T = 0: it adds 0 and stays at 0, exactly like the reference's early return.
__device__ __forceinline__ void composite(float e0, float e1, const float4 &A, const float4 &B, const float4 &C,
                                          float &T, float &c0, float &c1, float &c2) {
    float d0 = -0.5f * e0, d1 = -0.5f * e1;
    float t0 = d0 * A.z + d1 * B.x;  // (d @ Q)[0] = d0 Q00 + d1 Q10
    float t1 = d0 * A.w + d1 * B.y;  // (d @ Q)[1] = d0 Q01 + d1 Q11
    float w = __expf(t0 * e0 + t1 * e1);
    float alpha = w * B.z;
    float test = T * (1.0f - alpha);
    bool stop = test < kStopRefCpu;
    float ta = stop ? 0.0f : T * alpha;
    c0 += ta * C.x;
    c1 += ta * C.y;
    c2 += ta * C.z;
    T = stop ? 0.0f : test;
}

// Fast path, tile = 16: one wave per tile, 4 pixels per lane.
// FAST_Y: the 4 pixels of a lane are consecutive in y (GSX_LAYOUT_WH3) or in x (GSX_LAYOUT_HW3).
template <bool FAST_Y>
__global__ void __launch_bounds__(64)
    blend_tile16_kernel(const Record *__restrict__ rec, const uint32_t *__restrict__ vals,
                        const uint2 *__restrict__ ranges, TileGrid g, OutDesc out) {
    __shared__ float4 sh[3][64];
    const int lane = threadIdx.x;
    const uint32_t t = xcd_remap(blockIdx.x, (uint32_t)g.count());
    const int tx = g.wx0 + (int)(t / (uint32_t)g.nwy()), ty = g.wy0 + (int)(t % (uint32_t)g.nwy());
    const int slow = lane >> 2, fast0 = (lane & 3) * 4;
    const int px0 = tx * 16 + (FAST_Y ? slow : fast0);
    const int py0 = ty * 16 + (FAST_Y ? fast0 : slow);
    const float fx = (float)px0, fy = (float)py0;

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
            const float4 A = sh[0][k], B = sh[1][k], C = sh[2][k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // d = mean - pixel with the pixel coordinate formed first (exact for integers), as
                // the reference does; both layouts therefore produce identical bits per pixel.
                const float e0 = A.x - (FAST_Y ? fx : fx + (float)j);
                const float e1 = A.y - (FAST_Y ? fy + (float)j : fy);
                composite(e0, e1, A, B, C, T[j], c0[j], c1[j], c2[j]);
            }
        }
        __syncthreads();
        const bool live = (T[0] > 0.0f) | (T[1] > 0.0f) | (T[2] > 0.0f) | (T[3] > 0.0f);
        if (__ballot(live) == 0ull) break;
    }

    const int lx = px0 - out.x0, ly = py0 - out.y0;
    float *o = out.ptr + (int64_t)lx * out.stride_x + (int64_t)ly * out.stride_y;
    if ((reinterpret_cast<uintptr_t>(o) & 15u) == 0) {
        float4 *o4 = reinterpret_cast<float4 *>(o);
        o4[0] = make_float4(c0[0], c1[0], c2[0], c0[1]);
        o4[1] = make_float4(c1[1], c2[1], c0[2], c1[2]);
        o4[2] = make_float4(c2[2], c0[3], c1[3], c2[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[3 * j] = c0[j];
            o[3 * j + 1] = c1[j];
            o[3 * j + 2] = c2[j];
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
        float T = 1.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
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
                const float4 A = sh[0][k], B = sh[1][k], C = sh[2][k];
                composite(A.x - fx, A.y - fy, A, B, C, T, c0, c1, c2);
            }
            __syncthreads();
            if (__ballot(valid && T > 0.0f) == 0ull) break;
        }
        if (valid) {
            float *o = out.ptr + (int64_t)(px - out.x0) * out.stride_x + (int64_t)(py - out.y0) * out.stride_y;
            o[0] = c0;
            o[1] = c1;
            o[2] = c2;
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
        if (out.stride_y < out.stride_x)
            blend_tile16_kernel<true><<<(unsigned)nt, 64, 0, s>>>(rec, sorted_vals, ranges, grid, out);
        else
            blend_tile16_kernel<false><<<(unsigned)nt, 64, 0, s>>>(rec, sorted_vals, ranges, grid, out);
    } else {
        blend_generic_kernel<<<(unsigned)nt, 64, 0, s>>>(rec, sorted_vals, ranges, grid, out);
    }
    return hipGetLastError();
}

}  // namespace gsx
