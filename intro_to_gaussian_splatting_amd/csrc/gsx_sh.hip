// View-dependent colour: real spherical harmonics (degree 0..3) -> RGB, one thread per Gaussian.
//
// BUILD EXTENSION -- the reference has no spherical harmonics at all (its colour is the stored
// rgb/256, splat/gaussians.py:20-22), so parity
// of this kernel is UNPINNED by the reference.  It follows the published 3D Gaussian Splatting
// convention (Kerbl et al. 2023, `eval_sh` of graphdeco-inria/gaussian-splatting, not part of
// /root/reference): colour = max(0, 0.5 + sum_k Y_k(d) * sh[k]), d = normalize(mean - camera
// centre), with the constants of gsx_sh_device.h.  Degree 0 with sh0 = (rgb - 0.5) / 0.28209479
// reproduces the reference's RGB path, which is how it is tested against the pinned pipeline.
//
// This file is the standalone entry (gsx_sh_to_rgb: colours for the stage-1 API).  The whole-path render
// evaluates the same code inside the projection kernel (GsxParams.sh), so a frame has no colour launch
// and no colour array at all.
//
// HBM-bound elementwise op: 12 B (mean) + 12 (deg+1)^2 B (coefficients) read, 12 B written.
#include "gsx_internal.h"
#include "gsx_sh_device.h"

namespace gsx {
namespace {

template <int DEG>
__global__ void __launch_bounds__(sh::kBlock)
    sh_to_rgb_kernel(const float *__restrict__ means3d, const float *__restrict__ coeffs, int64_t n, float cx, float cy,
                     float cz, float *__restrict__ colors, bool vec) {
    __shared__ float lds[sh::Layout<DEG>::kLdsFloats];
    const int64_t g0 = (int64_t)blockIdx.x * sh::kBlock;
    sh::stage<DEG>(coeffs, n, g0, lds, vec);
    const int64_t i = g0 + threadIdx.x;
    if (i >= n) return;
    float r, g, b;
    sh::eval<DEG>(lds + threadIdx.x * sh::Layout<DEG>::STRIDE, means3d[3 * i] - cx, means3d[3 * i + 1] - cy,
                  means3d[3 * i + 2] - cz, r, g, b);
    colors[3 * i] = r;
    colors[3 * i + 1] = g;
    colors[3 * i + 2] = b;
}

}  // namespace

hipError_t launch_sh_to_rgb(const float *means3d, const float *sh, int degree, int64_t n, const float *center,
                            float *colors, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned nb = (unsigned)((n + gsx::sh::kBlock - 1) / gsx::sh::kBlock);
    const float cx = center[0], cy = center[1], cz = center[2];
    const bool vec = (reinterpret_cast<uintptr_t>(sh) & 15u) == 0;
    switch (degree) {
        case 0: sh_to_rgb_kernel<0><<<nb, gsx::sh::kBlock, 0, s>>>(means3d, sh, n, cx, cy, cz, colors, vec); break;
        case 1: sh_to_rgb_kernel<1><<<nb, gsx::sh::kBlock, 0, s>>>(means3d, sh, n, cx, cy, cz, colors, vec); break;
        case 2: sh_to_rgb_kernel<2><<<nb, gsx::sh::kBlock, 0, s>>>(means3d, sh, n, cx, cy, cz, colors, vec); break;
        case 3: sh_to_rgb_kernel<3><<<nb, gsx::sh::kBlock, 0, s>>>(means3d, sh, n, cx, cy, cz, colors, vec); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gsx
