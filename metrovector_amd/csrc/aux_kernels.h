// aux_kernels.h — launch interface of K3 / cross-shard merge / generators.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvf {

struct SelectParams {
    const uint64_t* lists;  // [nq][nlists][kcap] composites, each list sorted ascending, ~0-padded
    uint32_t nlists;
    uint32_t kcap;
    uint32_t heads;         // entries taken from each list for the threshold: ceil(k / nlists), >= 1
    uint32_t P;             // LDS capacity in entries (power of two, >= max(2*k, nlists*heads))
    uint32_t k;
    uint8_t metric, dtype;
    uint64_t index_base;
    const uint64_t* ids;    // vector ids per local row (schema/core.fbs:54), or NULL: the reported index is index_base + row
    float* out_scores;      // [nq][k]
    uint64_t* out_indices;
    int32_t* out_raw;       // nullable
    // alternative output (streaming over the f16 shadow): the k best COMPOSITES, unformatted, as the candidate list
    // of the margin compaction / re-scoring kernels: out_cand[q * cand_cap + i], out_cnt[q] = how many
    uint64_t* out_cand;     // nullable
    uint32_t* out_cnt;
    uint32_t cand_cap;
    // margin mode (streaming over the int8 shadow; out_cand set): the lists are each block's k best APPROXIMATE scores;
    // out_cand receives EVERY entry within 2 delta[q] of the k-th best of them all (at most keep_cap), out_tau[q] the
    // key of that bound; out_overflow[q] is raised when more than keep_cap entries are inside it or when a block's
    // list was cut inside it (rows beyond the cut may be inside too): the query is then redone exactly
    const float* delta;     // nullable; [nq]
    uint32_t* out_tau;
    uint32_t* out_overflow;
    uint32_t keep_cap;
    uint32_t margin_rank;   // the caller's k (<= k, the length the lists were cut at)
    // repair launches: block b serves query redo_list[redo_base + b] (its lists are the b-th of the launch) and exits
    // at once when *redo_cnt <= redo_base + b
    const uint32_t* redo_list;  // nullable
    const uint32_t* redo_cnt;
    uint32_t redo_base;
    // one pass of a k > MVFGPU_K_PER_PASS search (api.hip: search_large_k): the result row of a query is out_stride entries
    // long (0: k) and this pass fills [out_offset, out_offset + k); out_floor1[query] receives the last composite written
    // + 1 -- the next pass's floor (ScanParams::floor1) -- or ~0 when fewer than k rows were left
    uint32_t out_stride, out_offset;
    uint64_t* out_floor1;  // nullable
    // the blocking host call waits on a word of pinned host memory instead of the stream (api.hip: search_host): the block that
    // finishes LAST (a ticket in *done_ticket, which it leaves at 0; one block: no ticket) stores done_seq to *done_flag behind
    // its results, at system scope.  NULL = none.
    uint32_t* done_flag;
    uint32_t* done_ticket;
    uint32_t done_seq;
    // the payload rows of the results (mvfgpu_search_fetch on small results: the reference's ScoredVector.vector): the block
    // copies its query's k rows rows[local row * pitch .. + row_bytes) to gather_out[(query * stride + offset + rank) *
    // row_bytes ..) behind the results -- no third kernel, and the host waits on the flag.  Padding results give zero rows.
    // NULL = none.
    const unsigned char* gather_rows;
    unsigned char* gather_out;
    uint32_t gather_pitch, gather_row_bytes;
};

// queries flagged by the K2 compactions -> a dense list + its length, flags cleared (one block)
hipError_t launch_flag_compact(uint32_t* overflow, uint32_t nq, uint32_t* redo_list, uint32_t* redo_cnt, uint32_t* host_mirror /* pinned, or NULL */,
                               hipStream_t s);

struct ShardMergeParams {
    const float* scores;      // list l, query q, rank j at [l * ls_scores + q * k + j]
    const uint64_t* indices;  //                           [l * ls_indices + q * k + j]
    const int32_t* raw;       // nullable                  [l * ls_raw + q * k + j]
    size_t ls_scores, ls_indices, ls_raw;  // list strides in ELEMENTS (nq * k each for three separate arrays)
    uint32_t nlists, nq, k, P;
    uint8_t metric, dtype;
    float* out_scores;        // [nq][k]
    uint64_t* out_indices;
    int32_t* out_raw;         // nullable
};

constexpr uint32_t kMergeMaxEntries = 8192;  // nlists * k of one cross-shard merge: 8-byte composites, 64 KiB of LDS

hipError_t launch_select_final(const SelectParams& p, uint32_t nq, hipStream_t s);
hipError_t launch_merge_shards(const ShardMergeParams& p, hipStream_t s);
// nlists * k beyond kMergeMaxEntries: query q's composites to HBM / the first k of the sorted composites gathered (api.hip)
hipError_t launch_merge_build(const ShardMergeParams& p, uint32_t q, uint64_t* comps, hipStream_t s);
hipError_t launch_merge_write(const ShardMergeParams& p, uint32_t q, const uint64_t* sorted, hipStream_t s);
// result row of a query from the ascending composites of ALL its rows (sort_topk.hip); reads metric, dtype, index_base, ids,
// out_*, k of `p`; the row starts at element out_base of the output arrays
hipError_t launch_write_sorted(const SelectParams& p, const uint64_t* sorted, uint32_t n, size_t out_base, hipStream_t s);
// The first min(k, n) of n rank entries (mvf_common.h: position << 32 | key) that arrive in ascending position, in the order of
// their composites (key << 32 | position): radix select of the k-th key + position-ordered compaction, then a sort of the
// survivors -- one block's LDS up to 16384, a stable LSD radix sort of the key half beyond (sort_topk.hip; hand-written, no
// library).  Stream-ordered, no host wait.  tmp == NULL: only *tmp_bytes is written.  The result ends up in `a` or `b` (the
// other is scratch): *sorted says which.
hipError_t sort_composites(void* tmp, size_t* tmp_bytes, uint64_t* a, uint64_t* b, size_t n, size_t k, uint64_t** sorted, hipStream_t s,
                           uint32_t nb = 1, size_t stride = 0);  // nb lists of the same n and k, list l at a + l stride / b + l stride
hipError_t launch_synth_rows(unsigned char* rows, uint64_t n, uint32_t dim, uint32_t pitch, uint8_t dtype,
                             uint64_t seed, uint64_t row0, hipStream_t s);
hipError_t launch_repack_rows(const unsigned char* src, unsigned char* dst, uint64_t n, uint32_t row_bytes,
                              uint64_t src_stride, uint32_t pitch, hipStream_t s);
hipError_t launch_gather_rows(const unsigned char* rows, uint64_t n, uint32_t pitch, uint32_t row_bytes, uint64_t index_base,
                              const uint64_t* d_idx, uint32_t count, unsigned char* d_out, hipStream_t s,
                              uint32_t idx_stride = 0, uint32_t per_list = 0);  // per_list != 0: out row i = d_idx[(i / per_list) * idx_stride + i % per_list]
hipError_t launch_synth_packed(void* out, uint64_t nelem, uint8_t dtype, uint64_t seed, hipStream_t s);

}  // namespace mvf
