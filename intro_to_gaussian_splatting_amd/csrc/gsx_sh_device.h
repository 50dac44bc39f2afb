// Real spherical harmonics (degree 0..3) -> RGB: the device code shared by the standalone kernel
// (gsx_sh.hip, gsx_sh_to_rgb) and the projection kernel that evaluates the colour inline
// (gsx_project.hip, GsxParams.sh).  BUILD EXTENSION, parity unpinned: see gsx_sh.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsx {
namespace sh {

constexpr int kBlock = 256;
constexpr float C0 = 0.28209479177387814f;
constexpr float C1 = 0.4886025119029199f;
constexpr float C2a = 1.0925484305920792f, C2b = -1.0925484305920792f, C2c = 0.31539156525252005f,
                C2d = -1.0925484305920792f, C2e = 0.5462742152960396f;
constexpr float C3a = -0.5900435899266435f, C3b = 2.890611442640554f, C3c = -0.4570457994644658f,
                C3d = 0.3731763325901154f, C3e = -0.4570457994644658f, C3f = 1.445305721320277f,
                C3g = -0.5900435899266435f;

template <int DEG>
struct Layout {
    static constexpr int K = (DEG + 1) * (DEG + 1), W = 3 * K, STRIDE = W + 1;
    static constexpr int kLdsFloats = kBlock * STRIDE;
    // the projection kernel stages its 256 Gaussians in kParts rounds of kRows, so that the LDS footprint stays
    // below 26 KiB (6 workgroups per CU) at every degree
    static constexpr int kParts = DEG >= 3 ? 2 : 1, kRows = kBlock / kParts;
};

// sh: (n, K, 3) row-major, K = (degree + 1)^2 -- 12 K bytes per Gaussian (192 B at degree 3).  A thread
// that walked its own Gaussian's coefficients would touch 64 different cache lines per load
// instruction, so a workgroup of 256 threads first streams the contiguous block of its 256 Gaussians
// (3 K x 256 floats, up to 48 KiB) into LDS with 16-byte loads that are coalesced across the wave, then
// every thread reads its coefficients back from LDS at a padded stride (3 K + 1 words: conflict free).
// Every thread of the workgroup must call this (it ends with a barrier).  ROWS: Gaussians staged (the block
// size, or a part of it: then g0 is the first Gaussian of the part).
template <int DEG, int ROWS = kBlock>
__device__ __forceinline__ void stage(const float *__restrict__ sh, int64_t n, int64_t g0, float *lds, bool vec) {
    constexpr int W = Layout<DEG>::W, STRIDE = Layout<DEG>::STRIDE;
    const int64_t left = n - g0 > 0 ? n - g0 : 0;
    const int64_t block_floats = (left < ROWS ? left : ROWS) * W;   // multiple of 3, maybe not of 4
    const float *src = sh + (size_t)g0 * W;               // 16-B aligned when sh is (vec): 256 W floats per block
    for (int64_t v = threadIdx.x; v * 4 < block_floats; v += kBlock) {
        const int64_t e = v * 4;
        float q[4];
        if (vec && e + 4 <= block_floats) {
            const float4 f = *reinterpret_cast<const float4 *>(src + e);
            q[0] = f.x; q[1] = f.y; q[2] = f.z; q[3] = f.w;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) q[t] = e + t < block_floats ? src[e + t] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = (int)e + t;
            if (idx < ROWS * W) lds[(idx / W) * STRIDE + idx % W] = q[t];
        }
    }
    __syncthreads();
}

// The same for a LIST of Gaussians (the windowed projection: the survivors of its window test, a small share of the
// workgroup's 256): rows[first .. first + ROWS) of the list -- indices relative to g0 -- are staged, each row's 3 K
// floats read by consecutive lanes; nothing is read for a Gaussian that is not on the list.  count: list length.
// Every thread of the workgroup must call this (it ends with a barrier).
template <int DEG, int ROWS>
__device__ __forceinline__ void stage_rows(const float *__restrict__ sh, int64_t g0, const uint16_t *rows, int count,
                                           int first, float *lds, bool vec) {
    constexpr int W = Layout<DEG>::W, STRIDE = Layout<DEG>::STRIDE;
    const int have = count - first < ROWS ? count - first : ROWS;
    if (W % 4 == 0 && vec) {
        constexpr int Q = W % 4 == 0 ? W / 4 : 1;           // 16-byte words per row
        for (int v = threadIdx.x; v < have * Q; v += kBlock) {
            const int r = v / Q, q = v % Q;
            const float4 f = *reinterpret_cast<const float4 *>(sh + (size_t)(g0 + rows[first + r]) * W + 4 * q);
            float *d = lds + r * STRIDE + 4 * q;
            d[0] = f.x; d[1] = f.y; d[2] = f.z; d[3] = f.w;
        }
    } else {
        for (int v = threadIdx.x; v < have * W; v += kBlock) {
            const int r = v / W, c = v % W;
            lds[r * STRIDE + c] = sh[(size_t)(g0 + rows[first + r]) * W + c];
        }
    }
    __syncthreads();
}

// colour = max(0, 0.5 + sum_k Y_k(d) sh[k]), d = normalize(mean - camera centre); c = this thread's LDS row.
template <int DEG>
__device__ __forceinline__ void eval(const float *c, float dx, float dy, float dz, float &r_out, float &g_out, float &b_out) {
    constexpr int K = Layout<DEG>::K;
    float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx * inv, y = dy * inv, z = dz * inv;
    float basis[K];
    basis[0] = C0;
    if (DEG > 0) {
        basis[1 % K] = -C1 * y;
        basis[2 % K] = C1 * z;
        basis[3 % K] = -C1 * x;
    }
    if (DEG > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        basis[4 % K] = C2a * xy;
        basis[5 % K] = C2b * yz;
        basis[6 % K] = C2c * (2.0f * zz - xx - yy);
        basis[7 % K] = C2d * xz;
        basis[8 % K] = C2e * (xx - yy);
        if (DEG > 2) {
            basis[9 % K] = C3a * y * (3.0f * xx - yy);
            basis[10 % K] = C3b * xy * z;
            basis[11 % K] = C3c * y * (4.0f * zz - xx - yy);
            basis[12 % K] = C3d * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            basis[13 % K] = C3e * x * (4.0f * zz - xx - yy);
            basis[14 % K] = C3f * z * (xx - yy);
            basis[15 % K] = C3g * x * (xx - 3.0f * yy);
        }
    }
    float r = 0.0f, g = 0.0f, b = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        r += basis[k] * c[3 * k];
        g += basis[k] * c[3 * k + 1];
        b += basis[k] * c[3 * k + 2];
    }
    r_out = fmaxf(r + 0.5f, 0.0f);
    g_out = fmaxf(g + 0.5f, 0.0f);
    b_out = fmaxf(b + 0.5f, 0.0f);
}

}  // namespace sh
}  // namespace gsx
