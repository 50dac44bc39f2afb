// Is v_mfma_f32_4x4x1_16b_f32 usable as the colour accumulation of the compositing loop?  Per lane: a = one colour
// channel of the record (channel = lane & 3), b = T alpha of one of the lane's pixels, acc = that pixel's (r, g, b, -)
// accumulators.  Checks (1) the operand / result layout: lane 4k + j receives A[i] * B[j] + C in acc[i], A[i] = the a of
// lane 4k + i; (2) that every element is fmaf(a, b, c) bit for bit, denormals and large values included; and times
// (3) a loop of 4 such MFMAs per record against the 12 v_fma_f32 they replace, beside a stream of VALU work.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_mfma_colour tools/microbench_mfma_colour.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float *a, const float *b, const float *c, float *o) {
    const int l = threadIdx.x;
    v4f acc = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    o[4 * l] = acc.x; o[4 * l + 1] = acc.y; o[4 * l + 2] = acc.z; o[4 * l + 3] = acc.w;
}

template <int MODE>
__global__ void __launch_bounds__(64) loop_kernel(const float *in, float *out, int iters) {
    const int l = threadIdx.x;
    float ta[4], col[3], x = in[l], y = in[64 + l];
    for (int j = 0; j < 4; ++j) ta[j] = in[128 + 4 * l + j];
    for (int j = 0; j < 3; ++j) col[j] = in[512 + j];
    const float chan = in[512 + (l & 3) % 3];
    float c[12] = {0};
    v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
        // a stand-in for the rest of the trip: ~8 dependent-ish VALU ops per pixel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float w = fmaf(-x, (float)j, y);
            float e = fmaf(-w, w, ta[j]);
            float al = __builtin_amdgcn_exp2f(e);
            float t = ta[j] * al;
            ta[j] = ta[j] - t * 1e-7f;
            if (MODE == 0) {
                c[3 * j] = fmaf(t, col[0], c[3 * j]);
                c[3 * j + 1] = fmaf(t, col[1], c[3 * j + 1]);
                c[3 * j + 2] = fmaf(t, col[2], c[3 * j + 2]);
            } else {
                acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(chan, t, acc[j], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 12; ++j) s += c[j];
    for (int j = 0; j < 4; ++j) s += acc[j].x + acc[j].y + acc[j].z + ta[j];
    out[blockIdx.x * 64 + l] = s;
}

int main() {
    float ha[64], hb[64], hc[256], ho[256];
    srand(1);
    int bad_layout = 0, bad_bits = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        for (int i = 0; i < 64; ++i) {
            auto rnd = [&](int mode) { float v = (float)rand() / RAND_MAX * 2 - 1; if (mode == 1) v *= 1e-38f; if (mode == 2) v *= 1e30f; if (mode == 3) v *= 1e-20f; return v; };
            const int m = trial % 4;
            ha[i] = rnd(m == 1 ? 3 : 0); hb[i] = rnd(m); 
            for (int k = 0; k < 4; ++k) hc[4 * i + k] = rnd(m == 2 ? 2 : (m == 1 ? 1 : 0));
        }
        float *da, *db, *dc, *dout;
        hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dc, 1024); hipMalloc(&dout, 1024);
        hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice); hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice);
        layout_kernel<<<1, 64>>>(da, db, dc, dout);
        hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const float want = fmaf(ha[(l & ~3) + i], hb[l], hc[4 * l + i]);
                uint32_t x, y; memcpy(&x, &want, 4); memcpy(&y, &ho[4 * l + i], 4);
                if (x != y) { ++bad_bits; if (fabsf(want - ho[4 * l + i]) > 1e-3f * fabsf(want) + 1e-30f) ++bad_layout; if (bad_bits < 6) printf("trial %d lane %d i %d: want %a got %a (a %a b %a c %a)\n", trial, l, i, want, ho[4 * l + i], ha[(l & ~3) + i], hb[l], hc[4 * l + i]); }
            }
        hipFree(da); hipFree(db); hipFree(dc); hipFree(dout);
    }
    printf("layout mismatches %d, bit mismatches vs fmaf %d (of %d)\n", bad_layout, bad_bits, 2000 * 256);
    float *din, *dout;
    hipMalloc(&din, 4096); hipMalloc(&dout, 8192 * 64 * 4);
    float hin[1024]; for (int i = 0; i < 1024; ++i) hin[i] = (float)rand() / RAND_MAX;
    hipMemcpy(din, hin, 4096, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) loop_kernel<0><<<8192, 64>>>(din, dout, 20000); else loop_kernel<1><<<8192, 64>>>(din, dout, 20000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.3f ms (8192 waves x 20000 records x 4 pixels)\n", mode == 0 ? "12 v_fma_f32 per record " : "4 v_mfma_4x4x1 per record", ms);
        }
    return 0;
}
