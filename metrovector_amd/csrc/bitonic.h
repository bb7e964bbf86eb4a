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

// The same network with the entries in REGISTERS (E per thread: entry i = tid + x * NT): a step whose partners sit in the same
// wave (stride < 64) is a lane exchange -- no LDS round trip, no barrier -- and only the strides of 64 and more go through LDS
// (store, barrier, read the partner, barrier).  A 512-entry list: 6 barrier steps instead of 45 (round 5: the per-phase
// compaction of a 256-query search 29 -> ~9 us).  buf[0 .. P) filled (padding included) and a barrier passed on entry; P a
// power of two, 2 <= P <= E * NT; sorted ascending in buf, barrier passed, on return.
template <int NT, int E>
__device__ __forceinline__ void bitonic_sort_u64_reg(uint64_t* buf, uint32_t P, int tid) {
    uint64_t v[E];
#pragma unroll
    for (int x = 0; x < E; x++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)(x * NT);
        v[x] = i < P ? buf[i] : ~0ull;
    }
    for (uint32_t size = 2; size <= P; size <<= 1) {
        uint32_t stride = size >> 1;
        for (; stride >= 64; stride >>= 1) {
            __syncthreads();  // every read of the step before is done
#pragma unroll
            for (int x = 0; x < E; x++) {
                const uint32_t i = (uint32_t)tid + (uint32_t)(x * NT);
                if (i < P) buf[i] = v[x];
            }
            __syncthreads();
#pragma unroll
            for (int x = 0; x < E; x++) {
                const uint32_t i = (uint32_t)tid + (uint32_t)(x * NT);
                if (i < P) {
                    const uint64_t w = buf[i ^ stride];
                    const bool keep_min = ((i & stride) == 0) == ((i & size) == 0);
                    v[x] = (w < v[x]) == keep_min ? w : v[x];
                }
            }
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            if ((uint32_t)s <= stride) {  // (block-uniform)
#pragma unroll
                for (int x = 0; x < E; x++) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)(x * NT);
                    if ((i & ~63u) < P) {  // (wave-uniform: whole waves behind the list sit the exchanges out)
                        const uint64_t w = __shfl_xor((unsigned long long)v[x], s, 64);
                        const bool keep_min = ((i & (uint32_t)s) == 0) == ((i & size) == 0);
                        v[x] = (w < v[x]) == keep_min ? w : v[x];
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < E; x++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)(x * NT);
        if (i < P) buf[i] = v[x];
    }
    __syncthreads();
}

// Merge of a few new composites into a short sorted list, by counting instead of sorting: buf[0 .. kp) ascending (kp <= NT),
// buf[kp .. kp + c) in any order (c <= NT), all composites distinct; afterwards buf[0 .. min(kp + c, k)) holds the smallest
// of them ascending (whatever ranks at k or behind is dropped).  An element's place = the number of elements in front of it:
// an old one keeps its index plus the new ones below it, a new one counts the new ones below it plus the old ones (counted
// one by one while the list is short, by bisection beyond that).  Two barriers and c (+ kp) broadcast reads, where the sort
// network on next_pow2(kp + c) entries takes log^2 steps with a barrier each -- the usual merge of a scan's piece brings a
// dozen survivors to a list of k.  One old and one new element per thread: a handful of registers (the callers' row loops
// set their kernels' register counts; this must not).  The caller has synchronised after the last append; the list is
// complete on return.
template <int NT>
__device__ __forceinline__ void merge_ranked_u64(uint64_t* buf, uint32_t kp, uint32_t c, uint32_t k, int tid) {
    const bool has_old = (uint32_t)tid < kp, has_new = (uint32_t)tid < c;
    const uint64_t ov = has_old ? buf[tid] : 0ull;  // 0: nothing ranks below it
    const uint64_t nv = has_new ? buf[kp + (uint32_t)tid] : 0ull;
    uint32_t orank = (uint32_t)tid, nrank = 0;
    const uint64_t* nb = buf + kp;
    // (eight broadcast reads in flight: one dependent LDS read per step made a first piece's 64 survivors a 2-us merge -- a
    // quarter of a 10k-row scan, and four of them in a row on the four-query pass)
#pragma unroll 8
    for (uint32_t n = 0; n < c; n++) {
        const uint64_t e = nb[n];
        orank += e < ov ? 1u : 0u;
        nrank += e < nv ? 1u : 0u;
    }
    if ((uint32_t)(tid & ~63) < c) {  // the waves that hold new elements
        if (kp <= 128u) {
#pragma unroll 8
            for (uint32_t j = 0; j < kp; j++) nrank += buf[j] < nv ? 1u : 0u;
        } else {
            uint32_t lo = 0, hi = kp;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (buf[mid] < nv) lo = mid + 1;
                else hi = mid;
            }
            nrank += lo;
        }
    }
    __syncthreads();
    if (has_old && orank != (uint32_t)tid && orank < k) buf[orank] = ov;
    if (has_new && nrank < k) buf[nrank] = nv;
    __syncthreads();
}

// The same merge by ONE wave (the four-query pass: wave w merges query w's survivors while the other three waves merge theirs --
// one barrier for the four lists instead of eight in a row).  buf[0 .. kp) ascending (kp <= 256), buf[kp .. kp + c) in any
// order (c <= 256); each lane carries four old and four new elements.  No barrier inside: a wave's LDS operations execute in
// order, and every read of the list comes before the first write.
__device__ __forceinline__ void merge_ranked_u64_wave(uint64_t* buf, uint32_t kp, uint32_t c, uint32_t k, uint32_t lane) {
    uint64_t ov[4], nv[4];
    uint32_t orank[4], nrank[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = lane + 64u * (uint32_t)j;
        ov[j] = i < kp ? buf[i] : 0ull;  // 0: nothing ranks below it
        nv[j] = i < c ? buf[kp + i] : 0ull;
        orank[j] = i;
        nrank[j] = 0;
    }
    const uint64_t* nb = buf + kp;
#pragma unroll 4
    for (uint32_t n = 0; n < c; n++) {
        const uint64_t e = nb[n];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            orank[j] += e < ov[j] ? 1u : 0u;
            nrank[j] += e < nv[j] ? 1u : 0u;
        }
    }
    if (kp <= 128u) {
#pragma unroll 4
        for (uint32_t i = 0; i < kp; i++) {
            const uint64_t e = buf[i];
#pragma unroll
            for (int j = 0; j < 4; j++) nrank[j] += e < nv[j] ? 1u : 0u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t lo = 0, hi = kp;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (buf[mid] < nv[j]) lo = mid + 1;
                else hi = mid;
            }
            nrank[j] += lo;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = lane + 64u * (uint32_t)j;
        if (i < kp && orank[j] != i && orank[j] < k) buf[orank[j]] = ov[j];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = lane + 64u * (uint32_t)j;
        if (i < c && nrank[j] < k) buf[nrank[j]] = nv[j];
    }
}

}  // namespace mvf
