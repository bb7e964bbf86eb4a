// Read bandwidth of the access shape a register-streamed 16x16 MFMA B operand would use: a wave takes 16 consecutive
// rows of `pitch` bytes; load instruction j reads bytes [64 j, 64 j + 64) of each (lane l: row l % 16, 16 B at
// 64 j + 16 (l / 16)); all pitch / 64 loads of a 16-row group are issued back to back.  Compared with the streaming
// kernel's shape (16 lanes x 16 B = 256 contiguous bytes per row, 4 rows per instruction).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/bin/probe_access_shape scripts/probe_access_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int KT, int SHAPE>
__global__ void __launch_bounds__(256) read_kernel(const unsigned char* rows, uint32_t n, uint32_t pitch, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    uint32_t acc = 0;
    const uint32_t ngroups = n / 16;
    for (uint32_t g = wave; g < ngroups; g += nwaves) {
        u32x4 x[KT];
        if (SHAPE == 0) {  // MFMA B fragment shape: 16 rows x 64 B per instruction
            const unsigned char* rp = rows + (size_t)(g * 16 + (lane & 15)) * pitch + (lane >> 4) * 16;
#pragma unroll
            for (int j = 0; j < KT; j++) x[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rp + j * 64));
        } else if (SHAPE == 2) {  // 4 rows x 256 B per instruction, COLUMN-BLOCK-major over 16 rows (scan_mfma16_sb.hip's stages)
#pragma unroll
            for (int j = 0; j < KT; j++) {
                const int seg = j / 4, rr = (j % 4) * 4 + (lane >> 4);
                x[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rows + (size_t)(g * 16 + rr) * pitch + seg * 256 + (lane & 15) * 16));
            }
        } else {           // 4 rows x 256 B per instruction (K1's G = 16 shape), KT / 4 instructions per 4 rows
#pragma unroll
            for (int j = 0; j < KT; j++) {
                const int rr = (j / (KT / 4)) * 4 + (lane >> 4), seg = j % (KT / 4);
                x[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rows + (size_t)(g * 16 + rr) * pitch + seg * 256 + (lane & 15) * 16));
            }
        }
#pragma unroll
        for (int j = 0; j < KT; j++) acc += x[j][0] ^ x[j][1] ^ x[j][2] ^ x[j][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int KT>
void run(const unsigned char* d, uint32_t n, uint32_t pitch, uint32_t* o, hipEvent_t e0, hipEvent_t e1) {
    for (int shape = 0; shape < 3; shape++) {
        float best = 1e9f;
        for (int it = 0; it < 6; it++) {
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL((read_kernel<KT, 0>), dim3(1024), dim3(256), 0, 0, d, n, pitch, o);
            else if (shape == 1) hipLaunchKernelGGL((read_kernel<KT, 1>), dim3(1024), dim3(256), 0, 0, d, n, pitch, o);
            else hipLaunchKernelGGL((read_kernel<KT, 2>), dim3(1024), dim3(256), 0, 0, d, n, pitch, o);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("pitch %4u rows %8u shape %d (%s): %.3f ms  %.2f TB/s\n", pitch, n, shape,
               shape == 0 ? "16 rows x 64 B" : shape == 1 ? "4 rows x 256 B, row-major" : "4 rows x 256 B, column-block-major over 16 rows", best,
               (double)n * (KT * 64) / best / 1e9);
    }
}

int main() {
    unsigned char* d;
    uint32_t* o;
    const size_t bytes = (size_t)12500000 * 1040;
    hipMalloc(&d, bytes);
    hipMalloc(&o, 4);
    hipMemset(d, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    run<12>(d, 10000000, 768, o, e0, e1);
    run<16>(d, 12500000, 1024, o, e0, e1);
    run<16>(d, 12500000, 1040, o, e0, e1);   // the same row bytes, rows 16 bytes further apart
    run<8>(d, 20000000, 512, o, e0, e1);
    return 0;
}
