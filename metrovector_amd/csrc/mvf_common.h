// mvf_common.h — shared host/device helpers of libmvf_gpu.so.
//
// Order keys: every (metric, dtype) score is mapped to a u32 whose ascending
// order is "best first", so the whole top-k machinery is "k smallest u64
// composites (key << 32 | local row)".  DESIGN.md §3 defines the semantics;
// the reference pins only L2 over f32 (examples/similarity_search.rs:152-157).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvf_status.h"

#define MVF_HD __host__ __device__ __forceinline__

namespace mvf {

constexpr uint64_t kPadComposite = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t kNanKey = 0xFFFFFFFFu;

MVF_HD uint32_t elem_size(uint8_t dtype) {
    // reference src/vectors/vector_space.rs:122-127
    return dtype == MVF_DTYPE_FLOAT32 ? 4u : dtype == MVF_DTYPE_FLOAT16 ? 2u
         : (dtype == MVF_DTYPE_INT8 || dtype == MVF_DTYPE_UINT8) ? 1u : 0u;
}

MVF_HD bool is_int_dtype(uint8_t dtype) { return dtype == MVF_DTYPE_INT8 || dtype == MVF_DTYPE_UINT8; }

// Does selection run on the exact integer (L2 / InnerProduct on Int8/UInt8)?
MVF_HD bool key_is_raw(uint8_t dtype, uint8_t metric) {
    return is_int_dtype(dtype) && metric != MVF_METRIC_COSINE;
}

MVF_HD uint32_t f32_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
#endif
}

MVF_HD float bits_f32(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}

// ascending-float order as ascending-u32; NaN last; -0.0 == +0.0
MVF_HD uint32_t ord_f32(float x) {
    if (x != x) return kNanKey;
    x = x + 0.0f;
    uint32_t b = f32_bits(x);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

MVF_HD float unord_f32(uint32_t k) {
    if (k == kNanKey) return bits_f32(0x7FC00000u);
    uint32_t b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return bits_f32(b);
}

MVF_HD uint32_t key_from_score(float s, uint8_t metric) {
    return metric == MVF_METRIC_L2 ? ord_f32(s) : ord_f32(-s);
}

MVF_HD float score_from_key(uint32_t k, uint8_t metric) {
    float s = unord_f32(k);
    return metric == MVF_METRIC_L2 ? s : (s != s ? s : -s + 0.0f);
}

MVF_HD uint32_t key_from_raw(int32_t raw, uint8_t metric) {
    uint32_t u = (uint32_t)raw ^ 0x80000000u;
    return metric == MVF_METRIC_L2 ? u : ~u;
}

MVF_HD int32_t raw_from_key(uint32_t k, uint8_t metric) {
    uint32_t u = metric == MVF_METRIC_L2 ? k : ~k;
    return (int32_t)(u ^ 0x80000000u);
}

// Entries of the device-wide sorts (sort_topk.hip): (position << 32 | key'), sorted on the KEY half only -- the radix sort is
// stable and the entries are produced in ascending position, so equal keys keep the position order: the order of the
// composites (key << 32 | position) at half the digit passes.  (Not bits 32..63 of the composite itself: rocPRIM's small-input
// path builds its bit-range mask as (1 << (begin + bits)) - 1, which is undefined for begin + bits = 64 and compares the LOW
// word there.)  key' tells what the composite's low word did: a dead entry (deleted row, padding) is 0xFFFFFFFF and sorts last,
// a live NaN score -- kNanKey = 0xFFFFFFFF in a composite -- becomes 0xFFFFFFFE, a value no score and no exact integer maps to
// (float keys end at 0xFF800000 = +inf; |raw| < 2^31 - 2^16 by MVFGPU_MAX_INT_DIM).
MVF_HD uint64_t rank_entry(uint32_t key, uint32_t pos, bool dead) {
    const uint32_t kq = dead ? 0xFFFFFFFFu : (key == kNanKey ? 0xFFFFFFFEu : key);
    return ((uint64_t)pos << 32) | kq;
}

MVF_HD uint64_t composite_of_rank_entry(uint64_t e) {
    const uint32_t kq = (uint32_t)e;
    if (kq == 0xFFFFFFFFu) return kPadComposite;
    return ((uint64_t)(kq == 0xFFFFFFFEu ? kNanKey : kq) << 32) | (uint32_t)(e >> 32);
}

MVF_HD float pad_score(uint8_t metric) {
    return metric == MVF_METRIC_L2 ? bits_f32(0x7F800000u) : bits_f32(0xFF800000u);
}

// splitmix64 output function; the synthetic generator is
//   u = mix64(mix64(seed) + row*dim + col)        
MVF_HD uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

MVF_HD float synth_f32(uint64_t u) { return (float)(uint32_t)(u >> 40) * 0x1p-23f - 1.0f; }

MVF_HD uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace mvf
