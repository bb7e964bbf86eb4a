/*
 * mvf_oracle.h — CPU ORACLE for the MVF brute-force similarity-search path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under metrovector_amd/ (the product) may
 * include, link, import or execute anything under oracle/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / the timed CPU baseline.
 *
 * PARITY STATUS: "parity unpinned".  The reference (thegenem0/metrovector) is
 * Rust; no rustc/cargo/flatc exists in this image, so the reference cannot be
 * built or run, and its own tests hold no golden vector for
 * this path.  This file is a line-by-line restatement of
 *   examples/similarity_search.rs:140-176  (find_top_k_similar, R1/R2/R3)
 *   src/vectors/vector_space.rs:101-142    (get_vector row addressing, R4)
 *   src/vectors/vector.rs:71-92            (as_f32 decode, R5)
 *   src/builder.rs:175-193                 (f32 / f16 encode, R8)
 * pinned only by hand-derived known answers (tests/golden/).
 * Third-party arithmetic restated: crate `half` 2.6.0 (Cargo.lock)
 * f16::from_f32 (IEEE-754 binary16 round-to-nearest-even) and f16::to_f32
 * (exact widening).
 *
 * Everything the reference leaves undefined (cosine, dot, Int8/UInt8, batched
 * queries, tie order, NaN) is DEFINED here; DESIGN.md §3 is the prose twin.
 */
#ifndef MVF_ORACLE_H
#define MVF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* schema/types.fbs:3-8 */
enum { MVFO_F32 = 0, MVFO_F16 = 1, MVFO_I8 = 2, MVFO_U8 = 3 };
/* schema/types.fbs:20-25 */
enum { MVFO_L2 = 0, MVFO_IP = 1, MVFO_COS = 2 };

enum {
    MVFO_OK = 0,
    MVFO_ERR_INDEX = 1,      /* MvfError::IndexOutOfBounds  src/errors.rs:21 */
    MVFO_ERR_DIM = 2,        /* MvfError::DimensionMismatch src/errors.rs:24 */
    MVFO_ERR_CORRUPT = 4,    /* MvfError::CorruptedData     src/errors.rs:32 */
    MVFO_ERR_BUILD = 5,      /* MvfError::Build             src/errors.rs:38 */
    MVFO_ERR_ARG = 7
};

/* crate half 2.6.0: f16::from_f32 (RNE, overflow -> inf) / f16::to_f32 (exact). */
uint16_t mvfo_f32_to_f16(float f);
float mvfo_f16_to_f32(uint16_t h);

/* Element size per src/vectors/vector_space.rs:122-127; 0 for unsupported. */
uint32_t mvfo_elem_size(uint8_t dtype);

/*
 * Order key: a u32 whose ascending order is "best first" for `metric`.
 * Float scores: NaN -> 0xFFFFFFFF (last); -0.0 canonicalised to +0.0;
 * L2 ascending score, IP/COS descending score.
 */
uint32_t mvfo_key_from_score(float score, uint8_t metric);
/* Integer spaces, L2 and IP: key is taken from the exact i32, not the f32. */
uint32_t mvfo_key_from_raw(int32_t raw, uint8_t metric);

/*
 * Score one query against n rows (strict left-to-right f32 arithmetic, no FMA,
 * no reassociation — examples/similarity_search.rs:152-157).
 *   rows   : n rows, `stride` bytes apart, each dim*elem_size(dtype) bytes, LE.
 *   query  : f32[dim] for F32/F16 spaces; i8[dim]/u8[dim] for I8/U8 spaces.
 *   out_scores[n], out_keys[n] required; out_raw[n] optional (integer spaces:
 *   exact i32 of L2 (sum of squared differences) / IP (dot); 0 otherwise).
 * Rows are scored in parallel (OpenMP) — per-row arithmetic is unchanged.
 */
int mvfo_scores(const void* rows, uint64_t n, uint32_t dim, uint8_t dtype,
                uint64_t stride, uint8_t metric, const void* query,
                float* out_scores, uint32_t* out_keys, int32_t* out_raw);

/*
 * k best of n by (key asc, index asc).  Writes min(k,n) indices sorted
 * best-first and pads the rest with UINT64_MAX.
 */
int mvfo_topk_from_keys(const uint32_t* keys, uint64_t n, uint32_t k,
                        uint64_t* out_idx);

/*
 * Full search: nq queries (row-major, contiguous) -> out_scores[nq*k],
 * out_idx[nq*k] (+ index_base added), optional out_raw[nq*k].
 * Padding when k > n: idx UINT64_MAX, score +inf (L2) / -inf (IP, COS), raw 0.
 */
int mvfo_search(const void* rows, uint64_t n, uint32_t dim, uint8_t dtype,
                uint64_t stride, uint8_t metric, const void* queries,
                uint32_t nq, uint32_t k, uint64_t index_base,
                float* out_scores, uint64_t* out_idx, int32_t* out_raw);

/*
 * Merge `nlists` per-shard results (each [nq][k], sorted best-first, padded
 * with UINT64_MAX) into the global [nq][k] by (key, global index).
 * merge(top-k per shard) == top-k(global).
 */
int mvfo_merge_topk(const float* scores, const uint64_t* idx, const int32_t* raw,
                    uint32_t nlists, uint32_t nq, uint32_t k, uint8_t metric,
                    uint8_t dtype, float* out_scores, uint64_t* out_idx,
                    int32_t* out_raw);

/*
 * FAITHFUL restatement of find_top_k_similar (examples/similarity_search.rs:
 * 140-176) including its per-row cost structure: get_vector address math,
 * as_f32 malloc+decode, strict serial L2, BinaryHeap push / pop-when->k, final
 * ascending sort.  `block` is the vector block as the mmap holds it (n rows of
 * dim*es bytes).  Only F32/F16 decode (vector.rs:90 errors otherwise ->
 * MVFO_ERR_BUILD).  Single-threaded, like the reference.
 *   farthest=1 : AS WRITTEN (Ord reversed + max-heap pop keeps the k LARGEST
 *                distances — SURVEY.md F5);
 *   farthest=0 : INTENDED (k nearest), the semantics the product implements.
 * Returns the number of results (<= k) in *out_count.
 */
int mvfo_find_top_k_similar_faithful(const uint8_t* block, uint64_t block_len,
                                     uint64_t total_vectors, uint32_t dim,
                                     uint8_t dtype, const float* query,
                                     uint32_t query_len, uint32_t k,
                                     int farthest, uint64_t* out_idx,
                                     float* out_scores, uint32_t* out_count);

/*
 * Counter-based synthetic generator; the HIP generator in
 * metrovector_amd/csrc produces identical bytes.
 *   u = splitmix64_mix(seed ^ (row*dim + col))
 *   f32: (float)(u>>40) * 2^-23 - 1.0f   in [-1, 1)
 *   f16: RNE(f32 value);  i8: (int8)(u>>56);  u8: (uint8)(u>>56)
 * Writes nrows*dim elements, rows contiguous, starting at row `row0`.
 */
void mvfo_synth_rows(uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim,
                     uint8_t dtype, void* out);

/*
 * NOT THE CHECKER (mvf_cpu_best_effort.c): the host's best effort on one Float32 query -- `threads` OpenMP threads over
 * rows, no per-row allocation, 16 partial sums per row so the fold vectorises (hence not bit-exact with the reference's
 * strict left fold), one bounded heap per thread.  bench.py times it as `cpu_baseline_best_effort`; nothing is compared
 * against its results.
 */
int mvfo_search_best_effort_f32(const float* rows, uint64_t n, uint32_t dim, uint8_t metric, const float* query,
                                uint32_t k, uint64_t index_base, int threads, float* out_scores, uint64_t* out_idx);

#ifdef __cplusplus
}
#endif
#endif
