// scan_stream.h — launch interface of K1 (scan_stream.inc).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvf {

struct ScanParams {
    const unsigned char* rows;  // device rows, `pitch` bytes apart, 16-B aligned
    const void* queries;        // device [nq_total][dim]: f32, or the space's int type
    const float* xscale;        // dt1x / dt2x: the rows are a scaled shadow (f16 of a Float32 corpus / int8 of a float corpus), row r times xscale[r]
    // dt2x (int8 shadow, float scores): the queries are the int8 rows the query preparation wrote, `qstride` bytes apart,
    // with their scale qaux0[q] and norm qaux1[q] = |q|; xrow[r] = |x_r| (cosine) / sum x_r^2 (L2) of the STORED row
    const float* xrow;
    const float* qaux0;
    const float* qaux1;
    uint32_t qstride;
    const uint32_t* tomb;       // deletion bitmap, bit (r & 31) of word (r >> 5) = local row r is deleted; NULL = none
    uint64_t* cand;             // out: [launch queries][gridDim.x][kcap] sorted composites, ~0-padded
    uint32_t n;                 // rows in the shard
    uint32_t pitch;             // bytes per device row (multiple of 16)
    uint32_t dim;
    uint32_t V;                 // pitch / 16
    uint32_t J;                 // ceil(V / G): 16-B steps per lane per row
    uint32_t q0;                // first query handled by this launch
    uint32_t nq_total;
    uint32_t k;
    uint32_t kcap;              // next_pow2(k): entries per emitted list
    uint32_t pmax;              // next_pow2(k + chunk_safe): LDS buffer entries per query
    uint32_t chunk_rows;        // rows per chunk (blocks take chunks in turn); a multiple of 16*64/G
    uint32_t chunk_safe;        // rows per piece that cannot overflow the buffer (<= chunk_rows)
    uint32_t first_piece;       // rows per piece while a threshold is still unset (<= chunk_safe; a multiple of 16*64/G, >= k when that fits):
                                // EVERY row of such a piece is a survivor, so it is kept short enough for the counting merge
    uint32_t nchunks;
    uint32_t rank_merge_max;    // a piece's survivors up to this many join the list by counting (bitonic.h); 0: always the sort network
    // REPAIR launches (api.hip: queries whose K2 candidate budget overflowed are redone exactly, decided ON THE DEVICE):
    // the queries are redo_list[redo_base + i], i < min(*redo_cnt - redo_base, redo_max); the block walks them in
    // groups of NQ (one pass over the rows per group) and emits list (i, block).  *redo_cnt <= redo_base: exit at once.
    const uint32_t* redo_list;
    const uint32_t* redo_cnt;
    uint32_t redo_base, redo_max;
    // k beyond MVFGPU_K_PER_PASS (api.hip: search_large_k): a search for more results than one pass holds runs as several
    // passes, each returning the best k composites STRICTLY BEHIND the last one the pass before returned.
    // floor1[query] = that composite + 1 (0: no floor; ~0: the rows are exhausted, nothing may pass); NULL = none.
    const uint64_t* floor1;
    // k beyond what passes are worth (api.hip: search_sorted_k): the floor instantiation run as a DUMP -- no threshold, no
    // candidates; the rank entry (mvf_common.h) of every row of query q0 + q goes to dump[q * n + row] (a deleted row marked
    // dead) and a device-wide sort of the n entries ranks the whole shard.  NULL = none.
    uint64_t* dump;
};

// nqv: queries per pass, 1 or 4; p.redo_list != NULL selects the repair variant of the kernel, p.floor1 != NULL the
// floor variant, which p.dump != NULL selects as well (stored rows only: units 0..3)
#define MVF_DECL_SCAN(dt)                                                                          \
    hipError_t scan_stream_launch_dt##dt(const ScanParams& p, int metric, int G, int nqv, dim3 grid, \
                                         size_t lds, hipStream_t s);                               \
    const void* scan_stream_kernel_ptr_dt##dt(int metric, int G, int nqv, bool redo = false, bool floor = false);
MVF_DECL_SCAN(0)
MVF_DECL_SCAN(1)
MVF_DECL_SCAN(2)
MVF_DECL_SCAN(3)
MVF_DECL_SCAN(1x)  // Float16 rows of an f32 corpus' shadow, scaled back by ScanParams::xscale; nqv = 1 only
MVF_DECL_SCAN(2x)  // Int8 shadow rows of a float corpus: exact i32 dot, float keys dot * xscale[r] * qaux0[q]; nqv = 1 or 4
#undef MVF_DECL_SCAN

// Rows per SAFE piece for a lane-group width G (multiple of the 16*64/G rows a block covers per iteration): the candidate
// buffer of a query holds pmax = next_pow2(k + safe) entries, so a piece of `safe` rows cannot overflow it.
inline uint32_t scan_chunk_safe(int G) { return G == 1 ? 1024u : 512u; }

// Rows per chunk.  A wave covers 4 * 64/G rows per iteration, so on short rows a 512-row chunk is 2-8 iterations between
// two barriers, each starting from an empty memory pipeline (64-B Int8 rows ran at 4.1 TB/s, 128-B Float32 rows at 4.8).
// A pass takes chunks of >= 16 row-steps per wave; the kernel scans them as one guarded piece once the thresholds are
// set and falls back to safe pieces if a buffer overflows (scan_stream.inc).  Larger buffers instead cost
// occupancy: 2048-row chunks with a 4096-entry buffer were slower than 1024.  16 row-steps is the measured optimum: 8 is
// equal or 4 % slower, 32 and more LOSE on rows <= 128 B (64-B rows 5.7 -> 4.6 TB/s: the threshold goes stale inside a
// piece and the survivors' sorts grow); rows >= 512 B do not care.  Four queries per pass gain the most (each has its own
// buffer of the same size): 64-B Int8 rows 0.87 -> 1.9 TB/s, 128-B rows 1.2 -> 2.1, 256-B 2.5 -> 3.4.
inline uint32_t scan_chunk_rows(int G, uint32_t J, int /*nqv*/) {
    const uint32_t safe = scan_chunk_safe(G);
    const uint32_t step = 16u * 64u / (uint32_t)G, want = 16u * step / (J ? J : 1u);
    return want <= safe ? safe : (want + safe - 1) / safe * safe;
}

// bytes of dynamic LDS the kernel carves
inline size_t scan_lds_bytes(int dtype, int G, uint32_t J, int nqv, uint32_t pmax) {  // dtype: of the rows the kernel reads
    const uint32_t qb = (dtype == 0) ? 16u : (dtype == 1) ? 32u : 16u;
    size_t q = ((size_t)nqv * J * G * qb + 15u) & ~(size_t)15u;
    return q + (size_t)nqv * pmax * 8u + (size_t)nqv * 32u;
}

}  // namespace mvf
