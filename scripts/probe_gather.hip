// probe_gather.hip -- what bounds the exact re-scoring passes of a batched search?  They fetch ~110 / ~300
// random 3-KB rows per query out of a 30-GB corpus and run at 2.5 / 3.8 TB/s, where the guide quotes 5.7 TB/s for whole-row
// gathers.  This probe reads LIST rows of ROWBYTES out of a buffer of N rows, one wave per row with four rows in flight per
// wave (the re-scoring kernel's shape), 4 waves per block:
//   order "random"  : the list as drawn (what a per-query candidate list looks like)
//   order "sorted"  : the same rows in ascending address order (consecutive waves touch neighbouring pages)
// and over buffers of different sizes (address-translation reach).  Every variant reads the same number of bytes.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/bin/probe_gather scripts/probe_gather.hip
// run:   scripts/bin/probe_gather [list=340000] [rowbytes=3072]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define HIP_OK(x)                                                                                \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

__global__ void __launch_bounds__(256) gather_kernel(const unsigned char* rows, uint32_t pitch, const uint32_t* list, uint32_t m, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    const uint32_t V = pitch / 16;
    uint32_t acc = 0;
    for (uint32_t c0 = wave * 4u; c0 < m; c0 += nwaves * 4u) {
        const unsigned char* rp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) rp[u] = rows + (size_t)list[min(c0 + u, m - 1)] * pitch;
        for (uint32_t v = lane; v < V; v += 64) {
            u32x4 x[4];
#pragma unroll
            for (int u = 0; u < 4; u++) x[u] = *reinterpret_cast<const u32x4*>(rp[u] + (size_t)v * 16);
#pragma unroll
            for (int u = 0; u < 4; u++) acc += x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;  // keep the loads
}

int main(int argc, char** argv) {
    const uint32_t m = argc > 1 ? (uint32_t)atoi(argv[1]) : 340000u;
    const uint32_t pitch = argc > 2 ? (uint32_t)atoi(argv[2]) : 3072u;
    uint32_t* out;
    HIP_OK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    printf("%u rows of %u B per launch (%.1f MB), one wave per row, 4 rows in flight per wave\n", m, pitch, (double)m * pitch / 1e6);
    for (uint64_t gb : {1ull, 4ull, 16ull, 30ull, 120ull}) {
        const uint64_t n = gb * 1000000000ull / pitch;
        unsigned char* rows;
        if (hipMalloc(&rows, n * pitch) != hipSuccess) {
            printf("%llu GB: allocation failed\n", (unsigned long long)gb);
            (void)hipGetLastError();
            continue;
        }
        HIP_OK(hipMemset(rows, 1, n * pitch));
        std::vector<uint32_t> list(m);
        uint64_t s = 88172645463325252ull;
        for (auto& r : list) {
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            r = (uint32_t)(s % n);
        }
        uint32_t* dl;
        HIP_OK(hipMalloc(&dl, (size_t)m * 4));
        for (int order = 0; order < 2; order++) {
            if (order == 1) std::sort(list.begin(), list.end());
            HIP_OK(hipMemcpy(dl, list.data(), (size_t)m * 4, hipMemcpyHostToDevice));
            for (uint32_t blocks : {2048u, 8192u}) {
                float best = 1e9f;
                for (int rep = 0; rep < 6; rep++) {
                    HIP_OK(hipEventRecord(e0));
                    hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, 0, rows, pitch, dl, m, out);
                    HIP_OK(hipEventRecord(e1));
                    HIP_OK(hipEventSynchronize(e1));
                    float ms = 0;
                    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms);
                }
                printf("buffer %4llu GB  %-6s  %5u blocks: %7.1f us  %5.2f TB/s\n", (unsigned long long)gb, order ? "sorted" : "random", blocks,
                       best * 1e3, (double)m * pitch / (best * 1e-3) / 1e12);
            }
        }
        HIP_OK(hipFree(dl));
        HIP_OK(hipFree(rows));
    }
    return 0;
}
