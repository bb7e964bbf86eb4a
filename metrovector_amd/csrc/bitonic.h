// bitonic.h — block-wide bitonic sort of u64 composites in LDS (ascending).
// Pairs of one step are disjoint, so 4 are loaded before any is stored: the
// LDS round trips overlap instead of serialising on possible aliasing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvf {

template <int NT>
__device__ __forceinline__ void bitonic_sort_u64(uint64_t* buf, uint32_t P, int tid) {
    const uint32_t half = P >> 1;
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t0 = tid; t0 < half; t0 += 4 * NT) {
                uint64_t a[4], b[4];
                uint32_t ii[4];
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const uint32_t t = t0 + x * NT;
                    if (t < half) {
                        ii[x] = 2 * t - (t & (stride - 1));
                        a[x] = buf[ii[x]];
                        b[x] = buf[ii[x] + stride];
                    }
                }
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const uint32_t t = t0 + x * NT;
                    if (t < half) {
                        const bool up = (ii[x] & size) == 0;
                        if ((a[x] > b[x]) == up) {
                            buf[ii[x]] = b[x];
                            buf[ii[x] + stride] = a[x];
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace mvf
