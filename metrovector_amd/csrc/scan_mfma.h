// scan_mfma.h — launch interface of K2 (MFMA batched scan) and its helpers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

namespace mvf {

struct BatchParams {
    const float* qmat;          // [nq_pad][KP] f32 queries, zero padded (nq_pad multiple of 128)
    const float* qnorm;         // [nq_pad] sqrt(sum q^2)
    const unsigned char* rows;  // device rows
    const float* xnorm;         // [n] sqrt(sum x^2) (cosine)
    const float* xx2;           // [n] sum x^2 (batched L2)
    const float* xxmax;         // [1] max over rows of sum x^2 (batched L2 error margin)
    const uint32_t* tomb;       // deletion bitmap over local rows, or NULL
    const uint32_t* tau;        // [nq_pad] order-key thresholds (0xFFFFFFFF = none yet)
    uint64_t* cand;             // [nq_pad][cap] composites
    uint32_t* cnt;              // [nq_pad] entries appended (may exceed cap: overflow)
    uint32_t pitch, V;          // row pitch in bytes, pitch/16
    uint32_t KP, KT;            // padded k (floats), k-tiles of 32
    uint32_t nq;
    uint32_t row_begin, row_end;  // rows of this launch (phase)
    uint32_t ntiles, mtiles;      // ceil((row_end-row_begin)/128), nq_pad/128
    uint32_t cap;
    uint32_t direct;              // phase 0 (row_end - row_begin <= cap): candidate slot = row - row_begin, no counter
};

// K2 for 16-byte-operand MFMAs (Float16 / Int8 rows), scan_mfma16.hip
struct Batch16Params {
    const unsigned char* qprep;  // [nq_pad][KPB]: f16: one plane f16(q 2^e); i8: int8; zero padded
    const float* qaux0;          // [nq_pad] f16: 2^-e (undo of the query scale); i8: bit pattern of i32 sum q^2
    const float* qaux1;          // [nq_pad] f16: |q| (f32)
    const unsigned char* rows;
    const float* xnorm_f;        // [n] f16 rows: sqrt(sum x^2)
    const float* xx2;            // [n] f16 rows: sum x^2 (batched L2)
    const float* xxmax;          // [1] max over rows of sum x^2
    const float* xscale;         // [n] or NULL: rows are the scaled-f16 shadow of a Float32 corpus, row r times xscale[r]
    const unsigned char* zeros;  // >= 16 zero bytes (LDS-DMA kernel: source of k beyond a row's pitch)
    const int32_t* xnorm_i;      // [n] int rows: sum x^2 (UInt8: of the shifted values x-128)
    const int32_t* xbias_i;      // [n] UInt8 rows: 128 * sum (x-128)
    uint32_t dim;
    const uint32_t* tomb;        // deletion bitmap over local rows, or NULL
    const uint32_t* tau;
    uint64_t* cand;
    uint32_t* cnt;
    uint32_t pitch, V;
    uint32_t KPB, KT;            // padded row bytes of qprep, k-tiles (128 bytes; 64 for the LDS-DMA kernel)
    uint32_t nq, nq_pad;
    uint32_t row_begin, row_end;
    uint32_t ntiles, mtiles;     // ceil(rows/256), nq_pad / queries-per-block
    uint32_t cap;
    uint32_t direct;             // phase 0: candidate slot = row - row_begin, no counter
    // Candidate hand-off without global atomics (persistent grids of <= kBlkMaxBlocks blocks): a block appends
    // {key, row, query, 0} records to ITS OWN region blk_cand[blockIdx.x][blk_cap], the slot from a counter in LDS
    // (a returning GLOBAL atomic here would stall the appending wave for microseconds and, behind the barriers, the
    // whole block); the final count goes to blk_cnt[blockIdx.x] and scatter_cand_kernel files the records into the
    // per-query lists cand[] / cnt[] before the compaction.  NULL: the epilogue appends to cand[] / cnt[] directly.
    uint4* blk_cand;
    uint32_t* blk_cnt;
    uint32_t blk_cap;
    // The LDS-DMA kernel's i32-accumulator flavours (scan_mfma16_bias.inc) split a block's region into one slice per wave
    // (8 x blk_cap / 8 records, counts in blk_cnt[block * 8 + wave]) and write RAW records {exact sum, row, query, 1}; the
    // scatter pass computes their keys (scan_mfma16_key.h).  Set by the caller for launches that
    // scan_mfma16_dma_wave_regions() allows and whose grid fits the regions; the scatter pass is then given
    // 8 x the regions at 1/8 of the capacity.
    uint32_t wave_regions;
};

constexpr uint32_t kBlkMaxBlocks = 512;  // grids this size or smaller use per-block candidate regions ...
constexpr uint32_t kBlkCap = 16384;      // ... of at least this many 16-byte records each (128 MiB in all; more for large batches)
constexpr uint32_t kBlkWaves = 8;        // waves per block of the kernels that split their region per wave

struct CompactParams {
    uint64_t* cand;
    uint32_t* cnt;
    uint32_t* tau;
    uint32_t* overflow;
    uint32_t cap, k;
    uint32_t direct_cnt;  // != 0 after a direct phase: every query holds exactly this many candidates (cnt[] untouched)
    // approximate selection (float L2; every metric on Float16 rows): keys carry a score whose error is bounded by
    // eps * (qq + xxmax) [L2], eps [cosine], eps * |q| * sqrt(xxmax) [inner product]; the compaction keeps everything
    // within 2x that of the k-th value (compact_margin_kernel)
    const float* qnorm;   // [nq] |q|
    const float* xxmax;   // [1] max over rows of sum x^2
    float eps;
    // streaming over the f16 shadow (K1 keys): L2 keys are DISTANCES, | |q - x~| - |q - x| | <= |x~ - x| <= eps max|x|,
    // plus the accumulation, eps_acc (|q| + max|x|); and the list was
    // cut at `truncated_at` candidates by the selection before -- if every one of them is inside the margin, rows
    // beyond the cut may be too: the query is flagged and redone exactly
    uint8_t l2_is_distance;
    float eps_acc;
    uint32_t truncated_at;
    // int8-shadow selection (shadow_i8.hip): the proven bound of |approximate - exact score| per query, replacing the
    // eps formulas above (the L2 entry bounds the squared GEMM-form distance)
    const float* delta;   // [nq] or NULL
    // threshold refinement (int8-shadow selection): the kept list is written with its `ntop[q]` best approximate entries
    // (key <= the k-th best key) FIRST, for refine (below) to score them exactly; NULL = any order, nothing reported
    uint32_t* ntop;       // [nq] or NULL
    // final stage only
    uint8_t metric, dtype;
    uint64_t index_base;
    const uint64_t* ids;  // vector ids per local row, or NULL
    float* out_scores;
    uint64_t* out_indices;
    int32_t* out_raw;
};

constexpr uint32_t kBatchCap = 4096;  // candidate slots per query between compactions (32 KiB of LDS to sort)

size_t scan_mfma_lds_bytes();
hipError_t launch_scan_mfma_f32(const BatchParams& p, int metric, int num_cus, int force_persistent /* -1: the default grid */, hipStream_t s);
hipError_t launch_prep_queries(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t KP, float* qmat,
                               float* qnorm, hipStream_t s);
hipError_t launch_row_norms_f32(const unsigned char* rows, uint32_t n, uint32_t pitch, float* xnorm, float* xx2,
                                float* xxmax, hipStream_t s);
hipError_t launch_compact(const CompactParams& p, uint32_t nq, bool final_stage, hipStream_t s);

// approximate selection: margin-aware compaction + exact re-scoring of the kept candidates
struct RescoreParams {
    uint64_t* cand;             // [nq][cap] approximate composites, first cnt[q] valid; keys replaced by the exact ones
    uint32_t* cnt;              // re-armed to 0
    uint32_t* tau;              // re-armed to "none"
    uint32_t cap, k;
    const float* queries;       // device [nq][dim] f32 (the caller's queries)
    const unsigned char* rows;
    uint32_t pitch, dim;
    uint8_t dtype;              // Float32 or Float16 rows
    uint64_t index_base;
    const uint64_t* ids;        // vector ids per local row, or NULL
    float* out_scores;
    uint64_t* out_indices;
    int32_t* out_raw;
    // The exact scores of the head of the FINAL lists are computed once (round 5): the refinement pass over the final lists
    // (write_head != 0) leaves the exact composites of the rows it scores -- the k best by approximate key and their ties, the
    // final score's form -- in the upper half of the list, where the final pass writes its own; the final pass (head_done = the
    // refinement's ntop[]) neither fetches nor writes those entries again.  A third fewer random row fetches in the final pass.
    uint32_t write_head;
    const uint32_t* head_done;
};
hipError_t launch_compact_margin(const CompactParams& p, uint32_t nq, hipStream_t s);
hipError_t launch_rescore(const RescoreParams& p, int metric, uint32_t nq, hipStream_t s);
// Threshold refinement between two phases of an int8-shadow selection: the ntop[q] >= k best approximate candidates of
// every query are scored EXACTLY (rescore's arithmetic; L2 as the squared distance the selection works on); the worst of
// those exact scores, L, is a lower bound of the final k-th best score, so a row of the final top-k has an approximate
// score >= L - delta: tau[q] tightens from ord(v_k - 2 delta) to ord(L - delta) -- 2.7 x fewer scores get through the
// next phase's pre-filter, the candidate lists and the final re-scoring.  lkey: [nq] scratch, zero on entry and exit.
hipError_t launch_refine_tau(const RescoreParams& p, int metric, uint32_t nq, const uint32_t* ntop, uint32_t* lkey,
                             const float* delta, hipStream_t s);

uint32_t scan_mfma16_queries_per_block(int dtype);
hipError_t launch_scan_mfma16(const Batch16Params& p, int dtype, int metric, int num_cus, int force_persistent /* -1: per type */, hipStream_t s);
// small batches (one tile of <= 64 queries): streaming kernel with MFMA dots (scan_mfma16_sb.hip)
bool scan_mfma16_sb_usable(uint32_t nq_pad, uint32_t KT, uint32_t nq);
hipError_t launch_scan_mfma16_sb(const Batch16Params& p, int dtype, int metric, int num_cus, hipStream_t s);
uint32_t scan_mfma16_dma_queries_per_block(uint32_t nq, int forced_tile = 0 /* MVF_K2_TILE */);
uint32_t scan_mfma16_dma_tile_rows(uint32_t bmq);
bool scan_mfma16_dma_wave_regions(int dtype, uint32_t bmq, bool direct, bool has_regions, uint32_t dim);
hipError_t launch_scan_mfma16_dma(const Batch16Params& p, int dtype, int metric, int num_cus, uint32_t bmq, bool persistent,
                                  hipStream_t s);
// ping-pong schedule of the 256-query tile (scan_mfma16_pp.hip); same parameters as the LDS-DMA kernel with bmq = 256
bool scan_mfma16_pp_usable(uint32_t mtiles, int num_cus, uint32_t KT);
hipError_t launch_scan_mfma16_pp(const Batch16Params& p, int dtype, int metric, int num_cus, hipStream_t s);
// files the per-block candidate records of one K2 launch into the per-query lists (p.cand / p.cnt); RAW records (the
// exact integer sum instead of a key: scan_mfma16_key.h) get their keys here, with `metric` and the rows' `dtype`
// (re-arms blk_cnt[] to zero for the next phase, unless more than `scatter_rearm_max_queries()` queries take the simple form)
hipError_t launch_scatter_cand(const Batch16Params& p, uint32_t nblocks, int metric, int dtype, hipStream_t s);
uint32_t scatter_rearm_max_queries();
hipError_t launch_prep_queries16(const void* q, int dtype, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t KPB,
                                 unsigned char* qprep, float* qaux0, float* qaux1, hipStream_t s);
hipError_t launch_shadow_f16(const unsigned char* rows32, uint32_t n, uint32_t pitch32, uint32_t dim, unsigned char* rows16,
                             uint32_t pitch16, float* xscale, hipStream_t s);
// int8 shadow of a Float32 / Float16 corpus + its queries (shadow_i8.hip); stats: 4 floats, zeroed by the caller
hipError_t launch_shadow_i8(const unsigned char* rows, int src_dtype, uint32_t n, uint32_t pitch, uint32_t dim, unsigned char* rows8,
                            uint32_t pitch8, float* xscale8, float* stats, hipStream_t s);
hipError_t launch_prep_queries_i8s(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t KPB, int metric,
                                   const float* stats, const float* xxmax, unsigned char* qprep, float* qaux0, float* qaux1,
                                   float* delta, hipStream_t s);
hipError_t launch_row_norms16(const unsigned char* rows, int dtype, uint32_t n, uint32_t pitch, uint32_t dim, void* out,
                              float* xx2, float* xxmax, hipStream_t s);  // UInt8: xx2 receives the int32 bias array

}  // namespace mvf
