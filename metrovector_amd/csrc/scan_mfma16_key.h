// scan_mfma16_key.h -- the order key of one (query, row) pair of K2's i32-accumulator flavours (Int8 / UInt8 rows, the
// int8 shadow of a float corpus) from its exact integer sum.
//
// Round 3: the LDS-DMA kernel hands RAW records {sum, row, query, 1} to its per-block candidate regions and the
// scatter pass (scan_mfma.hip) turns them into keys -- thousands of threads beside nothing, instead of ~50 instructions
// per candidate in front of an idle matrix pipe.  The arithmetic is epilogue16's second stage, operation for
// operation, so the candidate lists are those of round 2's kernels.
#pragma once

#include "mvf_common.h"
#include "scan_mfma.h"

namespace mvf {

struct RawKeyArgs {
    const float* qaux0;      // int8 shadow: s_q; Int8 / UInt8 rows: bit pattern of i32 sum q^2
    const float* qaux1;      // int8 shadow: |q|; UInt8 rows: bit pattern of c_q = 128 Sq_s + 16384 d
    const uint32_t* tau;     // [nq_pad] thresholds of the launch
    const float* xnorm_f;    // int8 shadow: |x| of the stored rows
    const float* xx2;        // int8 shadow: sum x^2 of the stored rows
    const float* xscale;     // int8 shadow: s_r
    const int32_t* xnorm_i;  // Int8 / UInt8 rows: sum x^2 (UInt8: of the shifted values)
    const int32_t* xbias_i;  // UInt8 rows: 128 sum (x - 128)
    const uint32_t* tomb;    // deletion bitmap or NULL
    uint32_t dim, nq;
    uint8_t metric, dtype, xs, pad;
};

MVF_HD RawKeyArgs raw_key_args(const Batch16Params& p, int metric, int dtype) {
    RawKeyArgs a{};
    a.qaux0 = p.qaux0;
    a.qaux1 = p.qaux1;
    a.tau = p.tau;
    a.xnorm_f = p.xnorm_f;
    a.xx2 = p.xx2;
    a.xscale = p.xscale;
    a.xnorm_i = p.xnorm_i;
    a.xbias_i = p.xbias_i;
    a.tomb = p.tomb;
    a.dim = p.dim;
    a.nq = p.nq;
    a.metric = (uint8_t)metric;
    a.dtype = (uint8_t)dtype;
    a.xs = p.xscale != nullptr && dtype == MVF_DTYPE_INT8;
    return a;
}

// true: (q, r) is a candidate of this launch (a real query, key within its threshold, row not deleted); key returned
__device__ __forceinline__ bool raw_record_key(const RawKeyArgs& a, int32_t av, uint32_t r, uint32_t q, uint32_t& key) {
    if (q >= a.nq) return false;
    if (a.tomb && ((a.tomb[r >> 5] >> (r & 31)) & 1u)) return false;
    const uint8_t metric = a.metric;
    if (a.xs) {  // float scores from the int8 shadow: dot ~ sum * s_q * s_r
        float sc_ = (float)av * a.qaux0[q] * a.xscale[r];
        if (metric == MVF_METRIC_COSINE) {
            const float den = a.qaux1[q] * a.xnorm_f[r];
            sc_ = den > 0.0f ? sc_ / den : 0.0f;
        }
        if (metric == MVF_METRIC_L2) {
            const float qb = a.qaux1[q];
            sc_ = qb * qb + a.xx2[r] - 2.0f * sc_;  // GEMM-form s2
        }
        key = key_from_score(sc_, metric);
    } else {
        const bool u8 = a.dtype == MVF_DTYPE_UINT8;
        const int32_t qq = __float_as_int(a.qaux0[q]);
        const int32_t cqq = __float_as_int(a.qaux1[q]);
        const int32_t xxi = metric != MVF_METRIC_INNER_PRODUCT ? a.xnorm_i[r] : 0;
        const int32_t bx = (u8 && metric != MVF_METRIC_L2) ? a.xbias_i[r] : 0;
        const int32_t dot = u8 ? av + bx + cqq : av;  // dot in the space's own domain
        if (metric == MVF_METRIC_L2) {
            key = key_from_raw(qq + xxi - 2 * av, metric);  // shift invariant
        } else if (metric == MVF_METRIC_INNER_PRODUCT) {
            key = key_from_raw(dot, metric);
        } else {
            const int32_t qqn = u8 ? qq + 2 * cqq - 16384 * (int32_t)a.dim : qq;
            const int32_t xxn = u8 ? xxi + 2 * bx + 16384 * (int32_t)a.dim : xxi;
            const float den = sqrtf((float)qqn) * sqrtf((float)xxn);
            key = key_from_score(den > 0.0f ? (float)dot / den : 0.0f, metric);
        }
    }
    return key <= a.tau[q];
}

}  // namespace mvf
