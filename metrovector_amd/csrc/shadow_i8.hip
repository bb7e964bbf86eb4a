// shadow_i8.hip — the INT8 SHADOW of a Float32 / Float16 corpus and its queries (selection only).
//
// Under the power limit that bounds the MFMA loops on this part the int8 MFMA delivers ~1.9 POP/s where
// the f16 one holds ~1.05 PFLOP/s, and only ~k rows per query need more than a coarse score.  So the batched path may
// SELECT on an int8 copy of the rows -- per-row scale s_r = max|x| / 127, x8 = rint(x / s_r) -- with int8 queries
// (s_q = max|q| / 127), keep every candidate inside a PROVEN error bound of the k-th best, and re-score the kept rows
// exactly from the stored rows and the caller's f32 query (compact_margin_kernel / rescore_kernel, as for the f16
// shadow): results are those of the exact paths.
//
// The bound.  With x = s_r (x8 + ex), q = s_q (q8 + eq), |ex|, |eq| <= 1/2 per element (rounding; nothing clamps):
//     q.x = s_r s_q [ x8.q8  +  x8.eq  +  ex.q8  +  ex.eq ]
// and by Cauchy-Schwarz  |q.x - s_r s_q x8.q8| <= s_r s_q [ (|x8| + |ex|) |eq| + |ex| |q8| ]  (2-norms; |ex| and |eq|
// are MEASURED, ~0.29 sqrt(dim), not bounded by sqrt(dim)/2).  Per row the kernel below stores nothing but x8 and
// s_r; what the margins need is the corpus-wide maximum of s_r (|x8| + |ex|) and of s_r |ex| -- plain (InnerProduct,
// L2 = qq + xx - 2 q.x) and divided by |x| (Cosine) -- four floats, accumulated with atomicMax while the shadow is
// built.  Per query: delta = s_q (|eq| A + |q8| B) (+ the f32 evaluation of the approximate score), see
// prep_queries_i8s_kernel.  On the benchmark's uniform rows that is 0.23 sigma of the score distribution at dim 768
// (0.27 at 1024): 5-8 x k candidates per query reach the re-scoring (the f16 shadow: a handful beyond k).
// A row holding Inf / NaN (or whose sum x^2 overflows) makes the maxima +inf: every query is then answered by K1's
// repair launches, exactly as with the f16 shadow.

#include "scan_mfma.h"

#include "mvf_common.h"

#include <hip/hip_fp16.h>

namespace mvf {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void atomic_max_nonneg(float* dst, float v) {  // v >= 0 or +inf; NaN -> +inf
    if (!(v == v)) v = __uint_as_float(0x7F800000u);
    atomicMax(reinterpret_cast<unsigned int*>(dst), __float_as_uint(v));
}

// One wave per row.  SRC = MVF_DTYPE_FLOAT32 / MVF_DTYPE_FLOAT16 rows at `pitch`; out: int8 rows at pitch8 (dim rounded
// up to 16, zero padded), xscale8[r] = s_r, stats[0..3] = max s_r(|x8|+|ex|), max s_r|ex|, the same two over |x|.
template <int SRC>
__global__ void __launch_bounds__(256) shadow_i8_kernel(const unsigned char* rows, uint32_t n, uint32_t pitch, uint32_t dim,
                                                         unsigned char* rows8, uint32_t pitch8, float* xscale8, float* stats) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    const uint32_t V8 = pitch8 / 16;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    auto elem = [&](const unsigned char* rp, uint32_t c) __attribute__((always_inline)) -> float {
        if (c >= dim) return 0.f;
        if (SRC == MVF_DTYPE_FLOAT32) return reinterpret_cast<const float*>(rp)[c];
        return __half2float(reinterpret_cast<const __half*>(rp)[c]);
    };
    for (uint32_t r = wave; r < n; r += nwaves) {
        const unsigned char* rp = rows + (size_t)r * pitch;
        float mx = 0.f, ss = 0.f;
        bool bad = false;
        for (uint32_t c = lane; c < dim; c += 64) {
            const float v = elem(rp, c);
            bad |= !(fabsf(v) < 3.0e38f);
            mx = fmaxf(mx, fabsf(v));
            ss = fmaf(v, v, ss);
        }
        for (int off = 32; off > 0; off >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            ss += __shfl_xor(ss, off, 64);
        }
        bad = __builtin_amdgcn_ballot_w64(bad) != 0 || !(ss < 3.0e38f);
        const float sr = bad ? 0.f : mx / 127.0f;
        float x2 = 0.f, e2 = 0.f;
        for (uint32_t v8 = lane; v8 < V8; v8 += 64) {
            uint32_t w[4] = {0, 0, 0, 0};
            float xs[16];
            if (v8 * 16 + 16 <= dim) {  // whole vectors: 16-byte loads (the row pitch is a multiple of 16)
                if (SRC == MVF_DTYPE_FLOAT32) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const u32x4 x = *reinterpret_cast<const u32x4*>(rp + (size_t)v8 * 64 + k * 16);
#pragma unroll
                        for (int i = 0; i < 4; i++) xs[4 * k + i] = __uint_as_float(x[i]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const u32x4 x = *reinterpret_cast<const u32x4*>(rp + (size_t)v8 * 32 + k * 16);
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            xs[8 * k + 2 * i] = __half2float(__ushort_as_half((unsigned short)(x[i] & 0xFFFFu)));
                            xs[8 * k + 2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(x[i] >> 16)));
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) xs[i] = elem(rp, v8 * 16 + i);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float x = xs[i];
                float t = sr > 0.f ? x / sr : 0.f;     // IEEE division: x = sr (t / (1 + eps)), |eps| <= 2^-24
                float q = rintf(t);
                q = fminf(fmaxf(q, -127.f), 127.f);     // |t| <= 127 (1 + 2^-23): the clamp never bites beyond rounding
                const float e = t - q;
                x2 = fmaf(q, q, x2);
                e2 = fmaf(e, e, e2);
                w[i >> 2] |= (uint32_t)(uint8_t)(int8_t)(int)q << (8 * (i & 3));
            }
            *reinterpret_cast<u32x4*>(rows8 + (size_t)r * pitch8 + (size_t)v8 * 16) = u32x4{w[0], w[1], w[2], w[3]};
        }
        for (int off = 32; off > 0; off >>= 1) {
            x2 += __shfl_xor(x2, off, 64);
            e2 += __shfl_xor(e2, off, 64);
        }
        if (lane == 0) {
            xscale8[r] = sr;
            const float inf = __uint_as_float(0x7F800000u);
            // measured norms, inflated for the f32 sums above and the (1 + eps) of the division
            const float ex = sqrtf(e2) * 1.0005f + 1e-3f, xa = sqrtf(x2) * 1.0005f + ex;
            const float a = bad ? inf : sr * xa, b = bad ? inf : sr * ex;
            const float xn = sqrtf(ss);
            m0 = fmaxf(m0, a);
            m1 = fmaxf(m1, b);
            if (bad) m2 = m3 = inf;
            else if (xn > 0.f) {
                m2 = fmaxf(m2, a / xn * 1.000001f);
                m3 = fmaxf(m3, b / xn * 1.000001f);
            }
        }
    }
    if (lane == 0) {
        atomic_max_nonneg(stats + 0, m0);
        atomic_max_nonneg(stats + 1, m1);
        atomic_max_nonneg(stats + 2, m2);
        atomic_max_nonneg(stats + 3, m3);
    }
}

// Queries for the int8 shadow: q8 = rint(q / s_q), s_q = max|q| / 127, zero padded to KPB bytes per row.
//   qaux0 = s_q (the scale the epilogue undoes), qaux1 = |q| (f32 norm of the ORIGINAL query),
//   delta[q] = the proven bound of |approximate score - exact score| for this query over ALL rows of the corpus:
//     InnerProduct  s_q (|eq| A + |q8| B) + 4e-7 |q| max|x|
//     Cosine        s_q (|eq| Ac + |q8| Bc) / |q| + 4e-7
//     L2 (on the GEMM-form squared distance qq + xx - 2 q.x)   2 x the InnerProduct bound + 4e-7 (qq + max xx)
//   (A, B, Ac, Bc = stats[0..3]; the 4e-7 terms cover the f32 evaluation of acc * s_r * s_q and of the norms.)
// A non-finite query gets delta = +inf: every row is kept, the query overflows its budget and K1 repairs it.
__global__ void __launch_bounds__(256) prep_queries_i8s_kernel(const float* q, uint32_t nq, uint32_t dim, uint32_t KPB, int metric,
                                                                const float* stats, const float* xxmax, unsigned char* qprep,
                                                                float* qaux0, float* qaux1, float* delta) {
    const uint32_t row = blockIdx.x;
    __shared__ float red[12];
    float mx = 0.f, ss = 0.f;
    bool bad = false;
    if (row < nq)
        for (uint32_t c = threadIdx.x; c < dim; c += 256) {
            const float v = q[(size_t)row * dim + c];
            bad |= !(fabsf(v) < 3.0e38f);
            mx = fmaxf(mx, fabsf(v));
            ss = fmaf(v, v, ss);
        }
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        ss += __shfl_xor(ss, off, 64);
    }
    const bool wbad = __builtin_amdgcn_ballot_w64(bad) != 0;
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = mx;
        red[4 + (threadIdx.x >> 6)] = ss;
        red[8 + (threadIdx.x >> 6)] = wbad ? 1.f : 0.f;
    }
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    ss = red[4] + red[5] + red[6] + red[7];
    bad = (red[8] + red[9] + red[10] + red[11]) > 0.f || !(ss < 3.0e38f);
    __syncthreads();
    const float sq = (bad || !(mx > 0.f)) ? 0.f : mx / 127.0f;
    float q2 = 0.f, e2 = 0.f;
    for (uint32_t c = threadIdx.x; c < KPB; c += 256) {
        const float v = (row < nq && c < dim) ? q[(size_t)row * dim + c] : 0.f;
        const float t = sq > 0.f ? v / sq : 0.f;
        float r8 = rintf(t);
        r8 = fminf(fmaxf(r8, -127.f), 127.f);
        const float e = t - r8;
        q2 = fmaf(r8, r8, q2);
        e2 = fmaf(e, e, e2);
        reinterpret_cast<int8_t*>(qprep)[(size_t)row * KPB + c] = (int8_t)(int)r8;
    }
    for (int off = 32; off > 0; off >>= 1) {
        q2 += __shfl_xor(q2, off, 64);
        e2 += __shfl_xor(e2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = q2;
        red[4 + (threadIdx.x >> 6)] = e2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        q2 = red[0] + red[1] + red[2] + red[3];
        e2 = red[4] + red[5] + red[6] + red[7];
        const float qn = sqrtf(ss), inf = __uint_as_float(0x7F800000u);
        const float eq = sqrtf(e2) * 1.0005f + 1e-3f, q8n = sqrtf(q2) * 1.0005f;
        float d;
        if (row >= nq) d = 0.f;
        else if (bad) d = inf;
        else if (metric == MVF_METRIC_COSINE) d = qn > 0.f ? sq * (eq * stats[2] + q8n * stats[3]) / qn * 1.0001f + 4e-7f : 0.f;
        else {
            const float xm = sqrtf(xxmax[0]);
            d = sq * (eq * stats[0] + q8n * stats[1]) * 1.0001f + 4e-7f * qn * xm;
            if (metric == MVF_METRIC_L2) d = 2.0f * d + 4e-7f * (ss + xxmax[0]);
        }
        qaux0[row] = sq;
        qaux1[row] = qn;
        delta[row] = d;
    }
}

}  // namespace

hipError_t launch_shadow_i8(const unsigned char* rows, int src_dtype, uint32_t n, uint32_t pitch, uint32_t dim, unsigned char* rows8,
                            uint32_t pitch8, float* xscale8, float* stats, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 3) / 4, 256u * 8u);
    if (src_dtype == MVF_DTYPE_FLOAT32)
        hipLaunchKernelGGL(shadow_i8_kernel<MVF_DTYPE_FLOAT32>, dim3(blocks), dim3(256), 0, s, rows, n, pitch, dim, rows8, pitch8, xscale8, stats);
    else
        hipLaunchKernelGGL(shadow_i8_kernel<MVF_DTYPE_FLOAT16>, dim3(blocks), dim3(256), 0, s, rows, n, pitch, dim, rows8, pitch8, xscale8, stats);
    return hipGetLastError();
}

hipError_t launch_prep_queries_i8s(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t KPB, int metric,
                                   const float* stats, const float* xxmax, unsigned char* qprep, float* qaux0, float* qaux1,
                                   float* delta, hipStream_t s) {
    hipLaunchKernelGGL(prep_queries_i8s_kernel, dim3(nq_pad), dim3(256), 0, s, q, nq, dim, KPB, metric, stats, xxmax, qprep, qaux0,
                       qaux1, delta);
    return hipGetLastError();
}

}  // namespace mvf
