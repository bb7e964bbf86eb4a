// internal.h — pieces of api.hip the other translation units of libmvf_gpu.so use.
#pragma once

#include "../../include/mvf_status.h"

#include <cstdint>
#include <cstring>
#include <string>

namespace mvf {

// records the calling thread's failure detail (mvfgpu_last_error_message) and returns `status`
int set_fail(int status, const std::string& msg);

// Out-structs of the C ABI start with a caller-set `struct_size` (include/mvf_gpu.h "OUT-STRUCTS GROW"): copy at most
// that many bytes of `full` and report how many were filled.
template <class T>
int copy_out_struct(T* out, T full) {
    const uint32_t have = out->struct_size;
    if (have < 8u) return set_fail(MVF_ERR_INVALID_ARGUMENT, "struct_size not set (MVFGPU_INIT the out-struct before the call)");
    const uint32_t n = have < (uint32_t)sizeof(T) ? have : (uint32_t)sizeof(T);
    full.struct_size = n;
    std::memcpy(out, &full, n);
    return 0;
}

// Tuning switches of the environment (INTEGRATION.md lists them).  Read ONCE per handle, when it is created
// (mvfgpu_corpus_reload_tuning re-reads them for A/B scripts): nothing on the per-search path calls getenv.
// -1 / 0 = "not set: the library's own rule".
struct Tuning {
    int k1_g = 0;               // MVF_K1_G: lanes per row of the streaming kernel (sweeps)
    bool k2_dma = true;         // MVF_K2_DMA=0: the register-staged A/B kernel instead of the LDS-DMA ones
    bool k2_sb = true;          // MVF_K2_SB=0: the 64-query tile shape instead of the streaming MFMA kernel
    int k2_pp = -1;             // MVF_K2_PP=0|1: force the lockstep / ping-pong schedule on 256-query tiles
    uint32_t k2_growth = 4;     // MVF_K2_GROWTH: largest phase-to-phase growth of the batched scan
    uint32_t k2_growth_small = 6;  // MVF_K2_GROWTH_SMALL: ... of batches of up to 128 queries (HBM-bound scans: their records cost them little); follows MVF_K2_GROWTH where only that is set
    int k2_direct64 = 1;        // MVF_K2_DIRECT64: the direct phase of a 256-query-tile search runs in 64-query tiles (more, smaller blocks: it is all latency)
    bool k2_bias = true;        // MVF_K2_BIAS=0: round 2's epilogue instead of the folded pre-filter
    int k2_persistent = -1;     // MVF_K2_PERSISTENT: the f32 MFMA kernel's grid
    int k2_persistent16 = -1;   // MVF_K2_PERSISTENT16: the narrow-type kernels' grid
    int k2_tile = 0;            // MVF_K2_TILE=64|128|256: force a query-tile shape
    bool f16_shadow = true;     // MVF_F16_SHADOW=0
    bool i8_shadow = true;      // MVF_I8_SHADOW=0
    bool i8_shadow_partial = true;  // MVF_I8_SHADOW_PARTIAL=0: no int8 shadow of a prefix of the rows where all rows do not fit
    uint64_t i8_shadow_rows = 0;    // MVF_I8_SHADOW_ROWS=n (tests): as if only the first n rows' shadow fitted
    bool qs_refine = true;      // MVF_QS_REFINE=0
    uint32_t qs_refine_phases = 2;  // MVF_QS_REFINE_PHASES: the threshold is refined in front of this many of the last phases (round 5: 2; each costs ~70 us for 1024 queries and spares the phase behind it two thirds of its records)
    bool debug_repair = false;  // MVF_DEBUG_REPAIR: report repaired queries on stderr (synchronises inside a search)
    uint32_t repair_window = 0; // MVF_REPAIR_WINDOW: queries per repair launch pair (tests: several windows)
    uint64_t region_records = 0;  // MVF_K2_REGION_RECORDS: size of the candidate regions (tests: force overflows)
    bool stream_i8 = false;     // MVF_STREAM_I8=1
    bool stream_shadow = false; // MVF_STREAM_SHADOW=1
    unsigned upload_threads = 0;  // MVF_UPLOAD_THREADS
    uint32_t k1_rank_merge = 256;  // MVF_K1_RANK_MERGE: a piece's survivors up to this many are merged by counting (0: always sorted; <= 256; 128 until the counting loops got eight reads in flight: profiles/r04_k1_merge_ab.txt)
    size_t host_zc_query = 64u << 10;     // MVF_HOST_ZC_QUERY: mvfgpu_search reads queries up to this size in place (pinned host)
    size_t host_zc_results = 256u << 10;  // MVF_HOST_ZC_RESULTS: ... and writes results up to this size in place
    bool host_flag_wait = true; // MVF_HOST_FLAG_WAIT=0: the blocking host call always waits on its stream (not on the flag the final select writes)
    bool k1_first_piece = true; // MVF_K1_FIRST_PIECE=0: pieces of chunk_safe rows while a threshold is unset (as before round 4's short first piece)
    int large_k = 0;            // MVF_LARGE_K=1|2: k beyond one pass always by passes (1; k <= 16384) / always by the whole-shard sort (2); 0: the cheaper one
};
Tuning read_tuning();

}  // namespace mvf
