// Calibration of the FETCH_SIZE counter for the compositing kernel's access pattern (round-3 verdict, weak #5): every
// lane gathers ONE 48-byte record (3 x global_load_dwordx4) by an index list, like stage_batch.  The index lists are
// random and their footprint is known exactly: the program prints, per configuration, the bytes of the distinct 64-byte
// and 128-byte pieces of the table the launch touches.  Run under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- ./microbench_gather
// and divide (tools/calibrate_fetch.py): FETCH_SIZE x 1024 x correction = bytes fetched.  Configurations: a table that
// fits the 256 MiB Infinity Cache (re-read warm and, after a 1 GiB sweep, cold) and one that does not.
//   hipcc --offload-arch=gfx950 -O3 -o microbench_gather tools/microbench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unordered_set>
#include <vector>

struct __attribute__((aligned(16))) Rec { float4 a, b, c; };

__global__ void __launch_bounds__(64) gather_kernel(const Rec *__restrict__ rec, const uint32_t *__restrict__ idx, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const Rec *q = rec + idx[i];
    const float4 a = q->a, b = q->b, c = q->c;
    const float s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w;
    if (s == 12345.678f) out[0] = s;
}
// the wide streaming read the correction was calibrated on in rounds 1-3 (project_pack_kernel: 56 B per item)
__global__ void __launch_bounds__(256) stream_kernel(const float4 *__restrict__ p, size_t n16, float *out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) sweep_kernel(float4 *p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const size_t big = (size_t)12 << 20, small = (size_t)1 << 20;       // records: 576 MiB and 48 MiB tables
    Rec *rec;
    float *out;
    float4 *junk;
    const size_t junk16 = ((size_t)1 << 30) / 16;
    (void)hipMalloc(&rec, big * sizeof(Rec));
    (void)hipMalloc(&out, 64);
    (void)hipMalloc(&junk, junk16 * 16);
    sweep_kernel<<<4096, 256>>>((float4 *)rec, big * 3);
    uint32_t seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 4; };
    struct Cfg { const char *name; size_t table; uint32_t n; bool dup; };
    // dup: every index appears ~4 times in the list (a record is listed for ~4 tiles), shuffled -- the compositing pattern
    const Cfg cfgs[] = {{"gather_48B_table48MiB_1M_unique", small, 1u << 20, false}, {"gather_48B_table576MiB_4M_unique", big, 4u << 20, false},
                        {"gather_48B_table48MiB_4M_each_x4", small, 4u << 20, true}, {"gather_48B_table576MiB_8M_each_x4", big, 8u << 20, true}};
    for (const Cfg &c : cfgs) {
        std::vector<uint32_t> idx(c.n);
        const uint32_t distinct = c.dup ? c.n / 4 : c.n;
        for (uint32_t i = 0; i < distinct; ++i) idx[i] = (uint32_t)(rnd() % c.table);
        for (uint32_t i = distinct; i < c.n; ++i) idx[i] = idx[rnd() % distinct];
        for (uint32_t i = c.n - 1; i > 0; --i) std::swap(idx[i], idx[rnd() % (i + 1)]);
        std::unordered_set<uint64_t> s64, s128;
        s64.reserve(c.n * 2);
        s128.reserve(c.n * 2);
        for (uint32_t v : idx) {
            const uint64_t b0 = (uint64_t)v * 48, b1 = b0 + 47;
            for (uint64_t s = b0 / 64; s <= b1 / 64; ++s) s64.insert(s);
            for (uint64_t s = b0 / 128; s <= b1 / 128; ++s) s128.insert(s);
        }
        uint32_t *d_idx;
        (void)hipMalloc(&d_idx, (size_t)c.n * 4);
        (void)hipMemcpy(d_idx, idx.data(), (size_t)c.n * 4, hipMemcpyHostToDevice);
        sweep_kernel<<<4096, 256>>>(junk, junk16);          // push the table out of the Infinity Cache: a cold launch
        gather_kernel<<<(c.n + 63) / 64, 64>>>(rec, d_idx, c.n, out);
        gather_kernel<<<(c.n + 63) / 64, 64>>>(rec, d_idx, c.n, out);      // and the same again, warm
        (void)hipDeviceSynchronize();
        printf("CFG %s launches 2 (cold, warm) records %u list_bytes %zu bytes64 %zu bytes128 %zu record_bytes %zu\n", c.name, c.n, (size_t)c.n * 4,
               s64.size() * 64, s128.size() * 128, (size_t)(c.dup ? c.n / 4 : c.n) * 48);
        (void)hipFree(d_idx);
    }
    sweep_kernel<<<4096, 256>>>(junk, junk16);
    stream_kernel<<<8192, 256>>>((const float4 *)rec, big * 3, out);
    (void)hipDeviceSynchronize();
    printf("CFG stream_576MiB launches 1 bytes %zu\n", big * 48);
    return 0;
}
