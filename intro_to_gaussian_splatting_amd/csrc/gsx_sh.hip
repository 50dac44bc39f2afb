// View-dependent colour: real spherical harmonics (degree 0..3) -> RGB, one thread per Gaussian.
//
// BUILD EXTENSION -- the reference has no spherical harmonics at all (its colour is the stored
// rgb/256, splat/gaussians.py:20-22), so parity
// of this kernel is UNPINNED by the reference.  It follows the published 3D Gaussian Splatting
// convention (Kerbl et al. 2023, `eval_sh` of graphdeco-inria/gaussian-splatting, not part of
// /root/reference): colour = max(0, 0.5 + sum_k Y_k(d) * sh[k]), d = normalize(mean - camera
// centre), with the constants below.  Degree 0 with sh0 = (rgb - 0.5) / 0.28209479 reproduces the
// reference's RGB path, which is how it is tested against the pinned pipeline.
//
// HBM-bound elementwise op: 12 B (mean) + 12 (deg+1)^2 B (coefficients) read, 12 B written.
#include "gsx_internal.h"

namespace gsx {
namespace {

constexpr int kBlock = 256;
constexpr float C0 = 0.28209479177387814f;
constexpr float C1 = 0.4886025119029199f;
constexpr float C2a = 1.0925484305920792f, C2b = -1.0925484305920792f, C2c = 0.31539156525252005f,
                C2d = -1.0925484305920792f, C2e = 0.5462742152960396f;
constexpr float C3a = -0.5900435899266435f, C3b = 2.890611442640554f, C3c = -0.4570457994644658f,
                C3d = 0.3731763325901154f, C3e = -0.4570457994644658f, C3f = 1.445305721320277f,
                C3g = -0.5900435899266435f;

// sh: (n, K, 3) row-major, K = (degree + 1)^2 -- 12 K bytes per Gaussian (192 B at degree 3).  A thread
// that walked its own Gaussian's coefficients would touch 64 different cache lines per load
// instruction, so a workgroup of 256 threads first streams the contiguous block of its 256 Gaussians
// (3 K x 256 floats, up to 48 KiB) into LDS with 16-byte loads that are coalesced across the wave, then
// every thread reads its coefficients back from LDS at a padded stride (3 K + 1 words: conflict free).
template <int DEG>
__global__ void __launch_bounds__(kBlock)
    sh_to_rgb_kernel(const float *__restrict__ means3d, const float *__restrict__ sh, int64_t n, float cx, float cy,
                     float cz, float *__restrict__ colors, bool vec) {
    constexpr int K = (DEG + 1) * (DEG + 1), W = 3 * K, STRIDE = W + 1;
    __shared__ float lds[kBlock * STRIDE];
    const int64_t g0 = (int64_t)blockIdx.x * kBlock;
    const int64_t block_floats = ((n - g0 < kBlock) ? (n - g0) : kBlock) * W;   // multiple of 3, maybe not of 4
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
            if (idx < kBlock * W) lds[(idx / W) * STRIDE + idx % W] = q[t];
        }
    }
    __syncthreads();
    const int64_t i = g0 + threadIdx.x;
    if (i >= n) return;
    const float *c = lds + threadIdx.x * STRIDE;
    float dx = means3d[3 * i] - cx, dy = means3d[3 * i + 1] - cy, dz = means3d[3 * i + 2] - cz;
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
    colors[3 * i] = fmaxf(r + 0.5f, 0.0f);
    colors[3 * i + 1] = fmaxf(g + 0.5f, 0.0f);
    colors[3 * i + 2] = fmaxf(b + 0.5f, 0.0f);
}

}  // namespace

hipError_t launch_sh_to_rgb(const float *means3d, const float *sh, int degree, int64_t n, const float *center,
                            float *colors, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned nb = (unsigned)((n + kBlock - 1) / kBlock);
    const float cx = center[0], cy = center[1], cz = center[2];
    const bool vec = (reinterpret_cast<uintptr_t>(sh) & 15u) == 0;
    switch (degree) {
        case 0: sh_to_rgb_kernel<0><<<nb, kBlock, 0, s>>>(means3d, sh, n, cx, cy, cz, colors, vec); break;
        case 1: sh_to_rgb_kernel<1><<<nb, kBlock, 0, s>>>(means3d, sh, n, cx, cy, cz, colors, vec); break;
        case 2: sh_to_rgb_kernel<2><<<nb, kBlock, 0, s>>>(means3d, sh, n, cx, cy, cz, colors, vec); break;
        case 3: sh_to_rgb_kernel<3><<<nb, kBlock, 0, s>>>(means3d, sh, n, cx, cy, cz, colors, vec); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gsx
