// api.hip — the C ABI of libmvf_gpu.so (include/mvf_gpu.h).
//
// Host-side orchestration only: validation in the reference's error
// vocabulary (src/errors.rs:8-40), HBM residency of one row-range shard,
// kernel sequencing on the caller's HIP stream.  No CPU compute path exists
// here: without a device every compute entry point fails with MVF_ERR_DEVICE.

#include "../../include/mvf_gpu.h"

#include "aux_kernels.h"
#include "internal.h"
#include "mvf_common.h"
#include "scan_mfma.h"
#include "scan_stream.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace mvf;

namespace {

thread_local std::string g_last_error;

int fail(int status, const std::string& msg) {
    g_last_error = msg;
    return status;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return fail(MVF_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));   \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// Pinned host memory the device reads and writes in place (hipHostMalloc: mapped, coherent): the host-buffer API's
// mirrors for small queries / results.
struct PinBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        need = (need + 4095) & ~(size_t)4095;
        hipError_t e = hipHostMalloc(&p, need, hipHostMallocDefault);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
};

}  // namespace

namespace mvf {
int set_fail(int status, const std::string& msg) { return fail(status, msg); }

Tuning read_tuning() {
    Tuning t;
    auto num = [](const char* name, long dflt) {
        const char* e = getenv(name);
        return e ? atol(e) : dflt;
    };
    auto flag = [](const char* name, bool dflt) {
        const char* e = getenv(name);
        return e ? atoi(e) != 0 : dflt;
    };
    const long g = num("MVF_K1_G", 0);
    t.k1_g = (g == 64 || g == 32 || g == 16 || g == 8 || g == 4 || g == 1) ? (int)g : 0;
    t.k2_dma = flag("MVF_K2_DMA", true);
    t.k2_sb = flag("MVF_K2_SB", true);
    t.k2_pp = getenv("MVF_K2_PP") ? (int)flag("MVF_K2_PP", false) : -1;
    t.k2_growth = (uint32_t)std::max(2l, num("MVF_K2_GROWTH", 4));
    t.k2_growth_small = (uint32_t)std::max(2l, num("MVF_K2_GROWTH_SMALL", getenv("MVF_K2_GROWTH") ? (long)t.k2_growth : 6));
    t.k2_direct64 = (int)num("MVF_K2_DIRECT64", 1);
    t.qs_refine_phases = (uint32_t)std::max(0l, num("MVF_QS_REFINE_PHASES", 2));
    t.k2_bias = flag("MVF_K2_BIAS", true);
    t.k2_persistent = getenv("MVF_K2_PERSISTENT") ? (int)flag("MVF_K2_PERSISTENT", false) : -1;
    t.k2_persistent16 = getenv("MVF_K2_PERSISTENT16") ? (int)flag("MVF_K2_PERSISTENT16", true) : -1;
    const long tile = num("MVF_K2_TILE", 0);
    t.k2_tile = tile == 0 ? 0 : tile == 64 ? 64 : tile == 128 ? 128 : 256;
    t.f16_shadow = flag("MVF_F16_SHADOW", true);
    t.i8_shadow = flag("MVF_I8_SHADOW", true);
    t.i8_shadow_partial = flag("MVF_I8_SHADOW_PARTIAL", true);
    t.i8_shadow_rows = (uint64_t)std::max(0l, num("MVF_I8_SHADOW_ROWS", 0));
    t.qs_refine = flag("MVF_QS_REFINE", true);
    t.debug_repair = getenv("MVF_DEBUG_REPAIR") != nullptr;
    t.repair_window = (uint32_t)std::max(0l, num("MVF_REPAIR_WINDOW", 0));
    if (const char* e = getenv("MVF_K2_REGION_RECORDS")) t.region_records = strtoull(e, nullptr, 10);
    t.stream_i8 = flag("MVF_STREAM_I8", false);
    t.stream_shadow = flag("MVF_STREAM_SHADOW", false);
    t.upload_threads = (unsigned)std::max(0l, num("MVF_UPLOAD_THREADS", 0));
    t.k1_rank_merge = (uint32_t)std::min(256l, std::max(0l, num("MVF_K1_RANK_MERGE", 256)));
    t.host_zc_query = (size_t)std::max(0l, num("MVF_HOST_ZC_QUERY", 64l << 10));
    t.host_zc_results = (size_t)std::max(0l, num("MVF_HOST_ZC_RESULTS", 256l << 10));
    t.large_k = (int)std::min(2l, std::max(0l, num("MVF_LARGE_K", 0)));
    t.k1_first_piece = flag("MVF_K1_FIRST_PIECE", true);
    t.host_flag_wait = flag("MVF_HOST_FLAG_WAIT", true);
    return t;
}
}  // namespace mvf

// search_host's request to the search it is about to enqueue: "store `seq` to `flag` behind your results if your last kernel
// can" (the streaming path's final select); `armed` comes back true if it will.  Handed to search_device (below) explicitly and
// kept in the handle (flag_req) while that call holds the handle's lock.
struct HostFlagReq {
    uint32_t* flag;
    uint32_t* ticket;
    uint32_t seq;
    bool armed;
    uint64_t gen;  // the handle's work_gen of this search
    unsigned char* gather_out;  // mvfgpu_search_fetch: the final select copies the payload rows here too (pinned host memory); NULL = none
    bool gathered;              // ... and did, for every query
};

struct mvfgpu_corpus {
    int device = 0;
    uint64_t n = 0, index_base = 0;
    uint32_t dim = 0, pitch = 0, V = 0, J = 0;
    int G = 64;
    uint8_t dtype = 0;
    unsigned char* d_rows = nullptr;
    size_t rows_bytes = 0;
    int num_cus = 256;

    mutable std::mutex mu;       // guards the scratch + timing state
    mutable std::mutex host_mu;  // serialises the host-buffer API's device mirrors
    mutable DevBuf cand;                  // scratch: per-block candidate lists (K1)
    mutable DevBuf bq, bstate, bcand, xnorm;  // K2: padded queries + norms; tau/cnt/overflow; candidates; row norms
    mutable DevBuf blk;                   // K2 narrow types: per-block candidate regions + their counts (scan_mfma.h)
    mutable const void* blk_armed_cnt = nullptr;  // the region counters at this address are all zero: the last batched search's scatters left them so
    mutable DevBuf repair;                // K2 overflow repair: gathered queries + their results
    mutable std::vector<std::pair<uint64_t, int>> occ_cache;  // (kernel, dynamic LDS) -> blocks per CU (scan_occupancy)
    mutable const unsigned char* bq_zeros = nullptr;  // where the prepared-query buffer's 64 zero bytes were last set ...
    mutable size_t bq_zero_bytes = 0;                 // ... and the buffer's size then
    mutable DevBuf floor1;                // k > MVFGPU_K_PER_PASS: per query, the last composite the pass before returned, + 1
    mutable DevBuf rank_a, rank_b, rank_tmp;  // k > MVFGPU_K_PER_PASS by the whole-shard sort: the composites of every row (x the queries of a pass), twice, + the sort's scratch
    mutable DevBuf shadow, xscale;        // Float32 corpora: scaled-f16 shadow rows (selection only) + 2^-s_r per row
    DevBuf tomb, ids;                     // deletion bitmap (u32 words over local rows) / vector ids (u64 per local row)
    uint64_t deleted = 0;                 // bits set in the bitmap
    std::vector<uint64_t> h_ids;          // host copy of the ids ...
    mutable std::vector<std::pair<uint64_t, uint32_t>> id_index;  // ... and, built by the first gather, (id, row) sorted by id
    mutable int shadow_state = 0;         // 0 not built yet, 1 ready, -1 unavailable (no memory)
    mutable DevBuf shadow8, xscale8, qs_stats;  // Float32 / Float16 corpora: int8 shadow rows, s_r per row, the 4 bound maxima
    mutable int shadow8_state = 0;             // 0 not tried, 1 all rows, 2 a PREFIX of the rows (shadow8_rows; batched path only), -1 no room for all rows, -2 none for a useful prefix either
    mutable uint64_t shadow8_rows = 0;         // rows the int8 shadow covers
    mutable DevBuf split_out;                  // partial shadow: the two row ranges' result lists before their merge
    // feedback for the automatic choice: after a search that selected on the int8 shadow the number of queries the
    // repair launches had to redo is copied to pinned host memory (no wait); a later search that finds it large
    // (the data defeats the int8 bound: near-duplicates everywhere, heavy-tailed rows) switches this corpus back to the
    // f16 selection for good
    // Two slots, used alternately: a search first CONSUMES the count the search two before it posted into its slot
    // (waiting for that copy if need be: it is two searches old), then posts its own -- so which path a search takes
    // depends on the sequence of searches alone, never on how fast the host runs ahead of the device.
    mutable uint32_t* qs_redo_host = nullptr;  // [2] pinned
    mutable hipEvent_t qs_redo_ev[2] = {nullptr, nullptr};
    mutable bool qs_redo_pending[2] = {false, false};
    mutable bool qs_disabled = false;
    mutable uint32_t qs_redo_nq[2] = {0, 0};
    mutable uint32_t qs_slot = 0;  // the slot the next post goes to
    mutable uint32_t qs_seen = 0, qs_redone = 0;  // running totals of int8-selected queries / of those the repair pass redid
    // the same feedback guards the folded pre-filter of the i32-accumulator K2 kernels (scan_mfma16_bias.inc): rows whose
    // norms differ wildly inside a lane's four defeat its per-lane bounds, the wave regions overflow and the queries go
    // to the repair pass -- exact, but 50x slower; such a corpus goes back to round 2's epilogue first (fb_bias: the
    // search the pending count belongs to used the folded pre-filter; fb_qs: it selected on the int8 shadow)
    mutable bool bias_disabled = false, fb_bias[2] = {false, false}, fb_qs[2] = {false, false};
    mutable const uint32_t* last_redo_cnt = nullptr;  // device: the count the newest repair pass produced
    mutable const uint32_t* fb_mirrored = nullptr;    // the pinned slot flag_compact_kernel of the newest repair pass stored that count into itself (no copy then)
    mutable bool xnorm_ready = false;
    mutable uint32_t bstate_slots = 0;    // queries the K2 state arrays are armed for
    mutable DevBuf h_q, h_s, h_i, h_r;    // device mirrors for the host-buffer API
    // Completion of a small blocking search without the stream: the final select stores a sequence number into pinned host
    // memory behind its results and search_host spins on it -- hipStreamSynchronize learns of a finished kernel ~5 us later
    // (end-of-pipe flush, completion signal, wake-up; profiles/r04_flag_wait.txt).
    mutable PinBuf pin_flag;
    mutable DevBuf done_ticket;
    mutable uint32_t flag_seq = 0;
    mutable HostFlagReq* flag_req = nullptr;  // the host-buffer call's request to the search being enqueued (set and read under mu)
    mutable uint64_t work_gen = 0, confirmed_gen = ~0ull;  // searches enqueued on the handle / the newest one seen complete through the flag
    mutable PinBuf pin_q, pin_out, pin_vec;  // ... and its pinned host mirrors (small queries / results / payload rows: no copy engine at all)
    mutable DevBuf h_v;                   // payload rows of mvfgpu_search_fetch too large for that
    mutable hipStream_t own_stream = nullptr;
    hipStream_t up_stream = nullptr;      // upload pipeline: re-pitch / norms / shadow of chunk i beside the copy of chunk i+1
    mutable hipEvent_t ev_done = nullptr;
    mutable hipStream_t last_stream = nullptr;
    mutable bool has_done = false;

    bool profiling = false;
    int scan_path = 0;
    mvf::Tuning tune;  // the environment's tuning switches as they were when the handle was created
    struct ProfSlot {
        hipEvent_t e[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // [0,1] the timed scan launch, [2] end of its select; [3,4] the whole search
        bool scanned = false, whole = false;
    };
    static constexpr int kProfSlots = 64;
    mutable ProfSlot prof[kProfSlots];
    mutable uint64_t prof_next = 0;  // searches profiled since profiling was switched on
    mutable mvfgpu_timing timing{};
};

namespace {

// Lanes per row for K1, from the measured sweep of every data type over 64 <= dim <= 2048 at every width the kernel has
// (profiles/r02_k1_shape_sweep.csv, scripts/sweep_k1_shapes.py) and the full-size check of the benchmark shapes
// (profiles/r02_k1_group_full_size.txt).  V = 16-byte vectors per row.
//   V <= 2 (rows <= 32 B): one lane per row; V <= 8 (<= 128 B): 4 lanes; up to 3 KiB rows: 16 lanes -- the widest group
//   is NOT the fastest there (a 2-KiB Float16 row: 6.7 vs 6.3 TB/s; 768-B Int8 rows: 6.6 vs 3.9 TB/s: with 64 lanes on a
//   short row most of the wave idles in the row's last step); 64 lanes from 3 KiB on (10M x 768 f32: 6.9 vs 6.6 TB/s).
//   Four queries per pass (NQ = 4) carry 16 sums per lane group: 16 lanes stay ahead up to 8 KiB rows.
// round 1 picked "the widest group within 80 % of the best lane utilisation": up to 46 % off the best width on the
// shapes it had not been measured on (dim 100 / 200 f32, 512 f16, 384 / 1024 int8).  MVF_K1_G forces a width (sweeps).
void choose_group(uint32_t V, int nqv, int* G_out, uint32_t* J_out, int forced = 0) {
    // round 3: 5..8 vectors (65..128 B) take 8 lanes -- with 4 a row took two steps and every 128-byte line two separate
    // wave-loads.
    int g = V <= 1 ? 1 : V <= 4 ? 4 : V <= 8 ? 8 : (V < 192 || (nqv == 4 && V < 512)) ? 16 : 64;
    // Rows that are not a multiple of 128 B (V % 8 != 0) start in the middle of a cache line: a 16-lane group's load is four
    // separate unaligned spans and the ragged last step idles most lanes (4.0-4.5 TB/s for every such V from 17 up,
    // profiles/r03_k1_vectors_per_row.csv).  The widest group whose ONE load covers whole adjacent rows is better there:
    // 64 lanes from 33 vectors on (800-B rows 4.2 -> 5.1 TB/s, 1600-B 4.8 -> 5.6), 32 lanes for 17..31 (two adjacent rows
    // per wave-load); 32-B rows take 4 lanes, half of them idle, instead of one lane per row (3.4 -> 4.6 TB/s).
    if (nqv == 1 && V % 8 != 0 && V > 16 && V < 192) g = V > 32 ? 64 : 32;
    if (forced) g = forced;  // MVF_K1_G (sweeps)
    *G_out = g;
    *J_out = (V + g - 1) / g;
}

int init_common(mvfgpu_corpus* c) {
    c->tune = mvf::read_tuning();
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    return MVF_OK;
}

int validate_shape(uint64_t n, uint32_t dim, uint8_t dtype) {
    if (elem_size(dtype) == 0)
        return fail(MVF_ERR_BUILD, "Unsupported vector data type");  // reference src/vectors/vector_space.rs:126
    if (dim == 0) return fail(MVF_ERR_INVALID_ARGUMENT, "dimension must be > 0");
    if (is_int_dtype(dtype) && dim > MVFGPU_MAX_INT_DIM)
        return fail(MVF_ERR_BUILD, "Int8/UInt8 dimension exceeds the exact-i32 bound (33025)");
    if ((uint64_t)dim * elem_size(dtype) > (1ull << 30)) return fail(MVF_ERR_INVALID_ARGUMENT, "a row holds at most 1 GiB");
    // local row numbers are u32, and the streaming kernel rounds the last chunk up (by < 2^16 rows) before it masks
    if (n > 0xFFFF0000ull) return fail(MVF_ERR_INVALID_ARGUMENT, "a shard holds at most 2^32-65536 rows");
    return MVF_OK;
}

int alloc_rows(mvfgpu_corpus* c) {
    c->pitch = (c->dim * elem_size(c->dtype) + 15u) & ~15u;
    c->V = c->pitch / 16;
    choose_group(c->V, 1, &c->G, &c->J, c->tune.k1_g);
    c->rows_bytes = (size_t)c->n * c->pitch;
    if (c->rows_bytes) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_rows), c->rows_bytes));
    return MVF_OK;
}

hipError_t scan_launch(uint8_t dtype, const ScanParams& p, int metric, int G, int nqv, dim3 grid, size_t lds,
                       hipStream_t s) {
    switch (dtype) {
    case MVF_DTYPE_FLOAT32: return scan_stream_launch_dt0(p, metric, G, nqv, grid, lds, s);
    case MVF_DTYPE_FLOAT16: return scan_stream_launch_dt1(p, metric, G, nqv, grid, lds, s);
    case MVF_DTYPE_INT8: return scan_stream_launch_dt2(p, metric, G, nqv, grid, lds, s);
    default: return scan_stream_launch_dt3(p, metric, G, nqv, grid, lds, s);
    }
}

const void* scan_kernel(uint8_t dtype, int metric, int G, int nqv, bool redo = false, bool floor = false) {
    switch (dtype) {
    case MVF_DTYPE_FLOAT32: return scan_stream_kernel_ptr_dt0(metric, G, nqv, redo, floor);
    case MVF_DTYPE_FLOAT16: return scan_stream_kernel_ptr_dt1(metric, G, nqv, redo, floor);
    case MVF_DTYPE_INT8: return scan_stream_kernel_ptr_dt2(metric, G, nqv, redo, floor);
    default: return scan_stream_kernel_ptr_dt3(metric, G, nqv, redo, floor);
    }
}

// Streaming over the scaled-f16 shadow of a Float32 corpus (scan path 4): K1 reads the shadow rows instead of the stored
// ones and hands the k best COMPOSITES per query to the margin compaction instead of formatting results.
// Buffers of the whole-shard sort (search_sorted_k): a / b hold `nqv` x n composites each, tmp the sort's scratch.
struct RankAll {
    uint64_t *a, *b;
    void* tmp;
    size_t tmp_bytes;
    int nqv;         // queries per dump pass the buffers hold: 4 or 1
    uint32_t k_out;  // the caller's k
};

struct ShadowStream {
    const unsigned char* rows;
    const float* xscale;
    // int8 shadow (dt2x unit) only: prepared int8 queries `qstride` bytes apart, their scale / norm, the rows' norm array
    bool i8;
    const unsigned char* qprep;
    const float *qaux0, *qaux1, *xrow;
    uint32_t qstride;
    // ... and select_final's margin mode instead of the k best: every row within 2 delta[q] of the rank_k-th best
    const float* delta;
    uint32_t *tau, *overflow;
    uint32_t rank_k;
    uint32_t pitch, V, J;
    int G;
    uint64_t* cand;   // [nq][cand_cap]
    uint32_t* cnt;    // [nq]
    uint32_t cand_cap;
};

// Blocks per CU of a streaming-kernel instantiation at a dynamic-LDS size: asked once per (kernel, LDS bytes) and handle -- the
// runtime's answer costs 1-2 us and sits in front of the first launch of every search (a 10k-row search is 25 us in all).
int scan_occupancy(const mvfgpu_corpus* c, const void* kfn, size_t lds, int* occ_out) {
    const uint64_t key = (uint64_t)(reinterpret_cast<uintptr_t>(kfn)) * 0x9E3779B97F4A7C15ull ^ (uint64_t)lds;
    for (const auto& e : c->occ_cache)
        if (e.first == key) {
            *occ_out = e.second;
            return MVF_OK;
        }
    // the attribute is only ever RAISED: to the part's ceiling, once per kernel and handle (set to this miss's size, a later hit at a
    // larger size the cache already knew would launch above it -- a runtime that enforces the attribute would refuse the launch)
    if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, 256, lds));
    if (occ < 1) occ = 1;
    if (c->occ_cache.size() < 64) c->occ_cache.emplace_back(key, occ);
    *occ_out = occ;
    return MVF_OK;
}


int search_stream_path(const mvfgpu_corpus* c, uint8_t metric, const void* d_queries, uint32_t nq, uint32_t k,
                       float* d_scores, uint64_t* d_indices, int32_t* d_raw, hipStream_t s, bool profile = true,
                       const ShadowStream* alt = nullptr, const uint64_t* floor1 = nullptr, uint64_t* out_floor1 = nullptr,
                       uint32_t out_stride = 0, uint32_t out_offset = 0, const RankAll* rank = nullptr) {
    // floor1 / out_floor1 / out_stride / out_offset: one pass of a k > MVFGPU_K_PER_PASS search (search_large_k)
    // rank: the whole-shard sort (search_sorted_k) -- the scan DUMPS every row's composite, the selection kernel is replaced
    // by a device-wide sort of each query's n composites + the formatting of its first k_out; `k` is then only the (small)
    // list length the kernel's LDS layout is sized for
    const uint32_t kcap = next_pow2(k);
    const uint32_t ostride = out_stride ? out_stride : k;
    const bool alt8 = alt && alt->i8;
    const uint8_t kdtype = alt8 ? (uint8_t)MVF_DTYPE_INT8 : alt ? (uint8_t)MVF_DTYPE_FLOAT16 : c->dtype;
    const uint32_t kV = alt ? alt->V : c->V;

    mvfgpu_timing tm{};
    tm.scan_kernel = alt8 ? 7u : alt ? 5u : 1u;
    bool first = true;
    mvfgpu_corpus::ProfSlot* ps = nullptr;
    if (c->profiling && profile) {
        ps = &c->prof[c->prof_next % mvfgpu_corpus::kProfSlots];
        for (auto& e : ps->e)
            if (!e) HIP_TRY(hipEventCreate(&e));
        ps->scanned = false;
    }

    for (uint32_t q0 = 0; q0 < nq;) {
        int nqv = (nq - q0) >= 2 && (!alt || alt8) && !(rank && rank->nqv == 1) ? 4 : 1;
        int G;
        uint32_t J;
        choose_group(kV, nqv, &G, &J, c->tune.k1_g);  // the lane-group width depends on the queries per pass
        uint32_t chunk_rows = scan_chunk_rows(G, J, nqv), pmax = next_pow2(k + scan_chunk_safe(G));
        size_t lds = scan_lds_bytes(kdtype, G, J, nqv, pmax);
        if (nqv == 4 && lds > 150 * 1024) {
            nqv = 1;
            choose_group(kV, nqv, &G, &J, c->tune.k1_g);
            chunk_rows = scan_chunk_rows(G, J, nqv);
            pmax = next_pow2(k + scan_chunk_safe(G));
            lds = scan_lds_bytes(kdtype, G, J, nqv, pmax);
        }
        uint32_t nchunks = (uint32_t)((c->n + chunk_rows - 1) / chunk_rows);
        if (lds > 160 * 1024) return fail(MVF_ERR_BUILD, "dimension too large for the streaming kernel's LDS query tile");
        uint32_t nq_here = std::min<uint32_t>(nqv, nq - q0);
        uint32_t npass = 1;  // passes of nqv queries in this launch (grid.y)

        uint32_t nblocks = 0;
        if (nchunks > 0) {
            const void* kfn = alt8  ? scan_stream_kernel_ptr_dt2x(metric, G, nqv)
                              : alt ? scan_stream_kernel_ptr_dt1x(metric, G, nqv)
                                    : scan_kernel(c->dtype, metric, G, nqv, /*redo=*/false, floor1 != nullptr || rank != nullptr);
            int occ = 1;
            {
                const int orc = scan_occupancy(c, kfn, lds, &occ);
                if (orc != MVF_OK) return orc;
            }
            {  // A corpus of fewer 512-row chunks than the GPU holds blocks (n < ~650K rows) would leave most of them idle:
               // smaller chunks, one per resident block -- n / blocks rounded up to the kernel's step (100K x 128 f32: 196
               // blocks -> 782, 52 -> 31 us per search).  Up to four chunks per block the last round is uneven (700K x 256:
               // 1368 chunks on 1280 blocks, i.e. two rounds for 7 % of the blocks): P = ceil(chunks / blocks) rounds of
               // equal, smaller chunks instead (181 -> 144 us; 1M..3M rows: 1-5 %).  Larger corpora keep the 512-row chunks
               // (the headline's 496-row "balanced" chunks measured 1.4 % slower than 512); the long chunks of short rows
               // (scan_chunk_rows) are few per block and stay balanced up to 16 rounds.
                const uint32_t slots = (uint32_t)occ * (uint32_t)c->num_cus, step = 16u * 64u / (uint32_t)G;
                if (nchunks >= slots && nchunks < (chunk_rows > scan_chunk_safe(G) ? 16u : 4u) * slots) {
                    const uint32_t P = (nchunks + slots - 1) / slots;
                    const uint64_t per = (c->n + (uint64_t)P * slots - 1) / ((uint64_t)P * slots);
                    const uint32_t cr = (uint32_t)((per + step - 1) / step * step);
                    if (cr >= step && cr < chunk_rows) {
                        chunk_rows = cr;
                        nchunks = (uint32_t)((c->n + chunk_rows - 1) / chunk_rows);
                    }
                }
                if (nchunks < slots) {
                    const uint64_t per = (c->n + slots - 1) / slots;
                    const uint32_t cr = (uint32_t)((per + step - 1) / step * step);
                    if (cr >= step && cr < chunk_rows) {
                        chunk_rows = cr;
                        nchunks = (uint32_t)((c->n + chunk_rows - 1) / chunk_rows);
                    }
                }
            }
            nblocks = std::min<uint32_t>(nchunks, (uint32_t)occ * (uint32_t)c->num_cus);
            // A corpus that leaves most of the GPU idle (one small chunk per block and blocks to spare) takes ALL its four-query
            // passes in one launch, pass = blockIdx.y: the passes of a small batch were launch pairs in a row, ~40 us each
            // (10k x 128 f32, 16 queries: 165 -> 50 us; profiles/r04_host_api_latency.txt).  Up to 8 passes = every batch the
            // small-corpus rule of use_batched_path leaves to this kernel.
            if (nqv == 4 && !alt && !floor1 && !rank && nblocks == nchunks && 2u * nblocks <= (uint32_t)occ * (uint32_t)c->num_cus) {
                npass = std::min<uint32_t>(8u, (nq - q0 + 3u) / 4u);
                nq_here = std::min<uint32_t>(npass * 4u, nq - q0);
            }
            HIP_TRY(c->cand.reserve((size_t)npass * nqv * nblocks * kcap * 8));

            ScanParams sp{};
            sp.rows = alt ? alt->rows : c->d_rows;
            sp.xscale = alt ? alt->xscale : nullptr;
            sp.queries = alt8 ? static_cast<const void*>(alt->qprep) : d_queries;
            if (alt8) {
                sp.qaux0 = alt->qaux0;
                sp.qaux1 = alt->qaux1;
                sp.xrow = alt->xrow;
                sp.qstride = alt->qstride;
            }
            sp.tomb = static_cast<const uint32_t*>(c->tomb.p);
            sp.cand = static_cast<uint64_t*>(c->cand.p);
            sp.n = (uint32_t)c->n;
            sp.pitch = alt ? alt->pitch : c->pitch;
            sp.dim = c->dim;
            sp.V = alt ? alt->V : c->V;
            sp.J = J;
            sp.q0 = q0;
            sp.nq_total = nq;
            sp.k = k;
            sp.kcap = kcap;
            sp.pmax = pmax;
            sp.chunk_rows = chunk_rows;
            sp.chunk_safe = std::min(scan_chunk_safe(G), chunk_rows);
            {
                const uint32_t step = 16u * 64u / (uint32_t)G, want = std::max(k, 64u);
                // (not under the long chunks of short rows: a threshold from 128 rows lets too many of the next 4000 through --
                // 4 GB of <= 128-byte rows at k = 100 lost 5-8 %, profiles/r04_k1_first_piece_ab.txt)
                sp.first_piece = (c->tune.k1_first_piece && chunk_rows <= scan_chunk_safe(G)) ? std::min(sp.chunk_safe, (want + step - 1) / step * step)
                                                                                             : sp.chunk_safe;
            }
            sp.nchunks = nchunks;
            sp.rank_merge_max = c->tune.k1_rank_merge;
            sp.floor1 = floor1;
            sp.dump = rank ? rank->a : nullptr;
            if (ps && first) HIP_TRY(hipEventRecord(ps->e[0], s));
            if (alt8) HIP_TRY(scan_stream_launch_dt2x(sp, metric, G, nqv, dim3(nblocks), lds, s));
            else if (alt) HIP_TRY(scan_stream_launch_dt1x(sp, metric, G, nqv, dim3(nblocks), lds, s));
            else HIP_TRY(scan_launch(c->dtype, sp, metric, G, nqv, dim3(nblocks, npass), lds, s));
            if (ps && first) {
                HIP_TRY(hipEventRecord(ps->e[1], s));
                ps->scanned = true;
                tm.scan_bytes = (uint64_t)c->n * c->dim * elem_size(kdtype);
                tm.scan_flops = 2ull * nq_here * c->n * c->dim;
                if (npass > 1) tm.scan_bytes *= npass;
            }
            tm.scan_launches++;
        }
        if (rank) {  // every row of these queries is ranked: sort each query's n composites, format the first k_out
            SelectParams fp{};
            fp.k = rank->k_out;
            fp.metric = metric;
            fp.dtype = c->dtype;
            fp.index_base = c->index_base;
            fp.ids = static_cast<const uint64_t*>(c->ids.p);
            fp.out_scores = d_scores;
            fp.out_indices = d_indices;
            fp.out_raw = d_raw;
            uint64_t* sorted = rank->a;
            if (c->n > 0) {  // the pass's queries in one set of launches
                size_t tb = rank->tmp_bytes;
                HIP_TRY(sort_composites(rank->tmp, &tb, rank->a, rank->b, (size_t)c->n, (size_t)rank->k_out, &sorted, s, nq_here, (size_t)c->n));
            }
            for (uint32_t q = 0; q < nq_here; q++)
                HIP_TRY(launch_write_sorted(fp, sorted + (size_t)q * c->n, (uint32_t)c->n, (size_t)(q0 + q) * rank->k_out, s));
        } else {
            SelectParams fp{};
            fp.lists = static_cast<const uint64_t*>(c->cand.p);
            fp.nlists = nblocks;
            fp.kcap = kcap;
            fp.heads = nblocks ? (k + nblocks - 1) / nblocks : 1;
            fp.P = 4096;  // 32 KiB of LDS: >= k + kcap (fold path) and >= nlists*heads (nlists <= 2048)
            fp.k = k;
            fp.metric = metric;
            fp.dtype = c->dtype;
            fp.index_base = c->index_base;
            fp.ids = static_cast<const uint64_t*>(c->ids.p);
            if (alt) {
                fp.out_cand = alt->cand + (size_t)q0 * alt->cand_cap;
                fp.out_cnt = alt->cnt + q0;
                fp.cand_cap = alt->cand_cap;
                if (alt8) {
                    fp.delta = alt->delta + q0;
                    fp.out_tau = alt->tau + q0;
                    fp.out_overflow = alt->overflow + q0;
                    fp.keep_cap = alt->cand_cap / 2;
                    fp.margin_rank = alt->rank_k;
                }
            } else {
                fp.out_scores = d_scores + (size_t)q0 * ostride;
                fp.out_indices = d_indices + (size_t)q0 * ostride;
                fp.out_raw = d_raw ? d_raw + (size_t)q0 * ostride : nullptr;
                fp.out_stride = out_stride;
                fp.out_offset = out_offset;
                fp.out_floor1 = out_floor1 ? out_floor1 + q0 : nullptr;
                if (c->flag_req && !out_floor1 && c->flag_req->gather_out && !c->ids.p) {  // payload rows behind the results, by the same block
                    fp.gather_rows = c->d_rows;
                    fp.gather_out = c->flag_req->gather_out + (size_t)q0 * ostride * (c->dim * elem_size(c->dtype));
                    fp.gather_pitch = c->pitch;
                    fp.gather_row_bytes = c->dim * elem_size(c->dtype);
                    if (q0 + nq_here == nq) c->flag_req->gathered = true;
                }
                if (c->flag_req && !out_floor1 && q0 + nq_here == nq) {  // the search's last kernel
                    fp.done_flag = c->flag_req->flag;
                    fp.done_ticket = c->flag_req->ticket;
                    fp.done_seq = c->flag_req->seq;
                    c->flag_req->armed = true;
                }
            }
            HIP_TRY(launch_select_final(fp, nq_here, s));
        }
        if (ps && first) HIP_TRY(hipEventRecord(ps->e[2], s));
        first = false;
        q0 += nq_here;
    }
    if (ps) {
        c->timing = tm;  // event times are read back lazily by mvfgpu_last_timing
        c->prof_next++;
    }
    return MVF_OK;
}


// The f16/int8 MFMA kernel: scan_mfma16_dma.hip (LDS-DMA ring, 16x16 MFMA shape) by default; MVF_K2_DMA=0 selects the
// register-staged scan_mfma16.hip (kept as the A/B reference: same results, ~7 % slower).
bool k2_dma_enabled(const mvfgpu_corpus* c) { return c->tune.k2_dma; }

// One tile of at most 64 queries: the streaming MFMA kernel (scan_mfma16_sb.hip) instead of the 64-query shape of the
// LDS-DMA tile kernel; MVF_K2_SB=0 goes back (A/B runs).
bool k2_sb_enabled(const mvfgpu_corpus* c) { return c->tune.k2_sb; }

// 256-query tile: the ping-pong schedule (scan_mfma16_pp.hip) or the lockstep LDS-DMA kernel.  Measured (MI355X,
// profiles/r02_k2_ab.txt): Float16 rows / the f16 shadow 5 % faster on the ping-pong kernel once a block walks several
// tiles (cfg5 last phase 20.7 -> 19.7 ms), short phases and Int8 rows a few percent slower (its longer prologue; cfg4
// 6.94 vs 7.02 ms).  MVF_K2_PP=0|1 forces one of them (A/B runs).
bool k2_pp_wanted(const mvfgpu_corpus* c, uint8_t kdtype, uint32_t ntiles, uint32_t mtiles, int num_cus) {
    if (c->tune.k2_pp >= 0) return c->tune.k2_pp != 0;
    return kdtype == MVF_DTYPE_FLOAT16 && (uint64_t)ntiles * mtiles >= 8ull * (uint64_t)num_cus;
}

// Largest phase-to-phase growth of the K2 scan (MVF_K2_GROWTH overrides, for A/B runs).  A phase lets through
// ~k (g - 1) candidates per query (its threshold is the k-th best of 1/g of the rows it sees); with 1024 queries that is
// several per 256 x 256 tile at g = 8, and every candidate sends its wave through the epilogue's second stage.  g = 4
// costs one or two more (small) launches and measured 2-3 % faster on cfg3 / cfg5 / cfg4 (profiles/r02_k2_ab.txt); g = 16
// and 32 were 4 % slower in round 1.
// Batches of up to 128 queries (round 5, one process, the switch toggled between rounds -- profiles/r05_k2_walk_and_phase_costs.txt 8d): their
// scans are HBM-bound and their small kernels pure latency, so a phase less is worth more than its records cost: g = 6 is 2-4 % faster than
// 4 on 1M-3M rows, 0.6-1.5 % on 10M (g = 8 the same or a little more; 6 keeps a phase's records near half of the list capacity); 128 queries
// -4 % on 3M and 10M rows; 256 queries -2.4 % / -0.9 % / +1.4 % on three shapes, 384 and more: 4 wins.
// Only where a phase's records stay near half of the list capacity: with the int8 selection's margin a phase files about
// 7 k (g - 1) records per query (measured: 1800 at k = 100, g = 4), so g <= 1 + cap / (14 k) -- k <= 117 for g = 6 on 8192 slots.
uint32_t k2_growth_for(const Tuning& t, uint32_t nq, uint32_t k, uint32_t cap) {
    uint32_t g = std::min(t.k2_growth, std::max(2u, cap / (2u * k)));
    if (nq <= 128u) g = std::max(g, std::min(t.k2_growth_small, 1u + cap / (14u * k)));
    return g;
}
uint32_t k2_growth_for(const mvfgpu_corpus* c, uint32_t nq, uint32_t k, uint32_t cap) { return k2_growth_for(c->tune, nq, k, cap); }

// The phase boundaries R_1 .. R_{P+1} = nr of a batched search over nr rows (a pure function: mvfgpu_selftest_schedule pins it
// without a GPU).  Phase 0 passes everything (no threshold yet: R_1 <= cap rows, stored by row offset), later phases grow by g:
// expected survivors per query k (g - 1) + k carried <= cap / 2.  The boundaries are laid out BACKWARDS from the range's end --
// R_j = nr / g^(P - j), P the fewest steps that bring R_1 under cap -- so every phase, the last one included, scans (g - 1)
// times what its threshold has seen, and the last phase is always the largest ((1 - 1/g) of the rows: the launch bench.py times
// and prices).  Round 1 grew forwards from cap rows, which left a small odd phase at the end (and the one before it carrying
// most of the corpus).
std::vector<uint64_t> k2_phase_bounds(uint64_t nr, uint32_t cap, uint32_t g) {
    std::vector<uint64_t> bounds;
    uint32_t P = 0;
    for (uint64_t f = nr; f > cap; f = (f + g - 1) / g) P++;
    for (uint32_t j = 0; j <= P; j++) {
        uint64_t div = 1;
        for (uint32_t i = j; i < P; i++) div *= g;
        uint64_t e = (nr + div - 1) / div;
        e = std::min<uint64_t>(nr, (e + 255) / 256 * 256);
        if (j == 0) e = std::min<uint64_t>(e, cap);  // the direct phase's slots are row offsets
        if (bounds.empty() || e > bounds.back()) bounds.push_back(e);
    }
    if (bounds.empty() || bounds.back() < nr) bounds.push_back(nr);
    return bounds;
}

// The folded pre-filter of the LDS-DMA kernel's i32-accumulator flavours (scan_mfma16_bias.inc); MVF_K2_BIAS=0 keeps
// round 2's epilogue (A/B runs).
bool k2_bias_enabled(const mvfgpu_corpus* c) { return c->tune.k2_bias; }

bool k2_dma_persistent(const mvfgpu_corpus* c) {  // measured: int8 15 % and f16 5 % faster with one persistent block per CU
    return c->tune.k2_persistent16 < 0 ? true : c->tune.k2_persistent16 != 0;
}

// ---- scaled-f16 shadow of a Float32 corpus (selection only) -----------------------------------------------------
uint32_t shadow_pitch(uint32_t dim) { return (dim * 2u + 15u) & ~15u; }

bool shadow_enabled(const mvfgpu_corpus* c) {  // MVF_F16_SHADOW=0 keeps Float32 corpora on the exact f32 MFMA kernel (scan path 3 overrides)
    return c->tune.f16_shadow;
}

// Built on the first search that wants it (like the row norms): +50 % of the corpus' HBM.  Unless the caller insists
// (scan paths 3 and 4) it is skipped when that would leave less than 2 GiB free on the device.
hipError_t ensure_shadow(const mvfgpu_corpus* c, hipStream_t s, bool insist) {
    if (c->shadow_state == -1 && insist) c->shadow_state = 0;  // skipped automatically earlier: try now
    if (c->shadow_state != 0) return hipSuccess;
    const size_t need = (size_t)std::max<uint64_t>(c->n, 1) * shadow_pitch(c->dim);
    if (!insist) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + ((size_t)2 << 30)) {
            c->shadow_state = -1;
            return hipSuccess;
        }
    }
    if (c->shadow.reserve(need) != hipSuccess || c->xscale.reserve(((size_t)std::max<uint64_t>(c->n, 1) + 256) * 4) != hipSuccess) {
        (void)hipGetLastError();
        c->shadow.release();
        c->xscale.release();
        c->shadow_state = -1;
        return hipSuccess;
    }
    hipError_t e = launch_shadow_f16(c->d_rows, (uint32_t)c->n, c->pitch, c->dim, static_cast<unsigned char*>(c->shadow.p),
                                     shadow_pitch(c->dim), static_cast<float*>(c->xscale.p), s);
    if (e == hipSuccess) c->shadow_state = 1;
    return e;
}

// ---- int8 shadow of a Float32 / Float16 corpus (selection only; shadow_i8.hip) --------------------------------------
uint32_t shadow8_pitch(uint32_t dim) { return (dim + 15u) & ~15u; }
constexpr uint32_t kBatchCapQS = 8192;  // candidate slots per query with int8 selection (5-8 x k rows ride in the margin)

// Batched searches on Float32 / Float16 rows select on the int8 shadow by default (measured on cfg3 / cfg5 and on 2..512
// queries: 1.3-1.6 x the f16 selection at every batch size, profiles/r02_k2_ab.txt, r02_small_batches_*.txt): twice the
// MFMA rate of the f16 kernel under the same power limit, half (a quarter) of the bytes of Float16 (Float32) rows, a 15 x
// wider proven margin.  MVF_I8_SHADOW=0, scan path 3 (f16 selection) and scan path 2 (stored rows) opt out; scan path 5
// insists; a corpus whose data defeats the int8 bound switches itself back (qs_disabled).
// The int8 bound lets ~7-10 x k rows per query through to the re-scoring (0.23 sigma of the score distribution on the
// benchmark's rows), and a query whose margin holds more than half its candidate slots is redone by K1: beyond these k
// the selection is left to the f16 shadow / the stored rows, whose margins hold a handful of rows beyond k (k = 1000 on
// 10M x 768, 16 queries: 42 ms with every query repaired, 3 ms without).
constexpr uint32_t kQsMaxK = kBatchCapQS / 2 / 10;        // batched: 4096 kept candidates per query
constexpr uint32_t kQsStreamMaxK = kBatchCap / 2 / 10;    // streamed (select_final's margin mode): 2048

// everything but the shadow's own state
bool qs_possible(const mvfgpu_corpus* c, uint32_t k) {
    if (is_int_dtype(c->dtype) || c->n == 0) return false;
    if (k > kQsMaxK) return false;
    if (!k2_dma_enabled(c)) return false;  // the register-staged A/B kernel (MVF_K2_DMA=0) has no int8-shadow flavour
    if ((size_t)((c->dim + 7u) & ~7u) * 4 + kBatchCapQS * 4 > 64 * 1024) return false;  // re-scoring: query + candidates in LDS
    if (c->scan_path == 5 || c->scan_path == 6) return true;
    if (c->scan_path != 0 && c->scan_path != 4) return false;
    return !c->qs_disabled && c->tune.i8_shadow;
}

bool qs_wanted(const mvfgpu_corpus* c, uint32_t k = 0) {
    if (!qs_possible(c, k)) return false;
    return c->scan_path == 5 || c->scan_path == 6 || c->shadow8_state >= 0;
}

// The batched path alone can live with a shadow of a PREFIX of the rows (search_batched_path): where all rows did not fit
// (state -1) it still asks, once, for what does.
bool qs_wanted_batched(const mvfgpu_corpus* c, uint32_t k) {
    if (!qs_possible(c, k)) return false;
    return c->scan_path == 5 || c->scan_path == 6 || c->shadow8_state >= 0 || (c->shadow8_state == -1 && c->tune.i8_shadow_partial);
}

// The decision itself, a pure function of the sequence of samples (mvfgpu_selftest_feedback runs it without a GPU).
// A sample = (queries of the search, queries its repair pass redid, whether it ran with the folded pre-filter, whether it
// selected on the int8 shadow).  Running totals (a single streamed query says little by itself), halved now and then so
// that old history fades.  Too many repairs blame the folded pre-filter first (its per-lane bounds), the int8 selection
// second.  A sample of a search that still RAN WITH the pre-filter, consumed after the pre-filter has been switched off
// (two searches are in flight), is dropped: its repairs are the suspect's, and counted against the fresh totals they
// would switch the int8 selection off one search later, for good.
void feedback_consume(uint32_t& seen, uint32_t& redone, bool& bias_disabled, bool& qs_disabled, uint32_t nq, uint32_t redo,
                      bool used_bias, bool used_qs) {
    if (used_bias && bias_disabled) return;
    seen += nq;
    redone += std::min(redo, nq);
    if (redone >= 4 && (uint64_t)redone * 8 > seen) {
        if (used_bias && !bias_disabled) {
            bias_disabled = true;
            seen = redone = 0;
        } else if (used_qs) {
            qs_disabled = true;
        }
    } else if (seen >= 8192) {
        seen /= 2;
        redone /= 2;
    }
}

// What the search before the previous one had to repair (its count was copied to pinned memory behind it; waited for
// here -- by now it is two searches old -- so the decision does not depend on timing).
void qs_feedback_poll(const mvfgpu_corpus* c) {
    const uint32_t sl = c->qs_slot;
    if (!c->qs_redo_pending[sl]) return;
    if (hipEventSynchronize(c->qs_redo_ev[sl]) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    c->qs_redo_pending[sl] = false;
    const bool bias_was = c->bias_disabled, qs_was = c->qs_disabled;
    feedback_consume(c->qs_seen, c->qs_redone, c->bias_disabled, c->qs_disabled, c->qs_redo_nq[sl], c->qs_redo_host[sl],
                     c->fb_bias[sl], c->fb_qs[sl]);
    if (c->tune.debug_repair && (bias_was != c->bias_disabled || qs_was != c->qs_disabled))
        fprintf(stderr, "[mvfgpu] %s switched off for this corpus: too many queries needed the repair path\n",
                bias_was != c->bias_disabled ? "folded pre-filter" : "int8-shadow selection");
}

int feedback_slots(const mvfgpu_corpus* c) {
    if (c->qs_redo_host) return MVF_OK;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->qs_redo_host), 64, hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&c->qs_redo_ev[0], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->qs_redo_ev[1], hipEventDisableTiming));
    return MVF_OK;
}

// The pinned slot the next post will use, for the repair pass's flag_compact_kernel to store its count into directly (round 5:
// the 4-byte copy behind every batched search was a launch of its own, 4.1 us) -- NULL while that slot still holds a sample
// nobody has consumed (the post will skip it too).
uint32_t* feedback_mirror(const mvfgpu_corpus* c) {
    c->fb_mirrored = nullptr;
    if (feedback_slots(c) != MVF_OK || c->qs_redo_pending[c->qs_slot]) return nullptr;
    c->fb_mirrored = c->qs_redo_host + c->qs_slot;
    return c->qs_redo_host + c->qs_slot;
}

// ... and the request for it: the repair count of the search just enqueued, in pinned memory behind an event.
int qs_feedback_post(const mvfgpu_corpus* c, uint32_t nq, hipStream_t s, bool used_bias = false, bool used_qs = true) {
    const uint32_t sl = c->qs_slot;
    if (c->qs_redo_pending[sl] || !c->repair.p) return MVF_OK;  // (pending: this search did not poll -- it took another path first)
    {
        int rc = feedback_slots(c);
        if (rc != MVF_OK) return rc;
    }
    c->fb_bias[sl] = used_bias;
    c->fb_qs[sl] = used_qs;
    if (c->fb_mirrored != c->qs_redo_host + sl)  // (the repair pass could not store it there itself)
        HIP_TRY(hipMemcpyAsync(c->qs_redo_host + sl, c->last_redo_cnt, 4, hipMemcpyDeviceToHost, s));
    c->fb_mirrored = nullptr;
    HIP_TRY(hipEventRecord(c->qs_redo_ev[sl], s));
    c->qs_redo_pending[sl] = true;
    c->qs_redo_nq[sl] = nq;
    c->qs_slot = sl ^ 1u;
    return MVF_OK;
}

// allow_partial (the batched path): where all rows do not fit beside the corpus, shadow the longest PREFIX of rows that does
// (a 204.8-GB Float16 corpus has no room for its 102.4-GB shadow, but for 90+ % of it on this part): the search then runs
// as two row ranges -- int8 selection over the prefix, the f16 kernels over the rest -- whose lists are merged like two
// shards' (search_batched_path).  At least a quarter of the rows and a million, 4 GiB left free for the scratch buffers.
hipError_t ensure_shadow8(const mvfgpu_corpus* c, hipStream_t s, bool insist, bool allow_partial = false) {
    if (c->shadow8_state == -1 && insist) c->shadow8_state = 0;
    if (c->shadow8_state > 0 || c->shadow8_state == -2 || (c->shadow8_state == -1 && !allow_partial)) return hipSuccess;
    const size_t pitch8 = shadow8_pitch(c->dim);
    uint64_t rows = std::max<uint64_t>(c->n, 1);
    const uint64_t forced = c->tune.i8_shadow_rows && c->tune.i8_shadow_rows < c->n ? c->tune.i8_shadow_rows : 0;  // tests
    if (c->shadow8_state == 0 && !insist) {
        size_t free_b = 0, total_b = 0;
        if (forced || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < rows * pitch8 + ((size_t)2 << 30)) c->shadow8_state = -1;
    }
    if (c->shadow8_state == -1) {
        if (!allow_partial || !c->tune.i8_shadow_partial) return hipSuccess;
        size_t free_b = 0, total_b = 0;
        uint64_t fit = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > ((size_t)4 << 30)) fit = ((free_b - ((size_t)4 << 30)) / (pitch8 + 4)) & ~(uint64_t)65535;
        if (fit >= c->n) fit = c->n;  // (memory was freed meanwhile)
        if (forced) fit = std::min(fit, forced);
        else if (fit < c->n / 4 || fit < (1u << 20)) fit = 0;
        if (fit == 0) {
            c->shadow8_state = -2;
            return hipSuccess;
        }
        rows = fit;
    }
    if (c->shadow8.reserve(rows * pitch8) != hipSuccess || c->xscale8.reserve(((size_t)rows + 256) * 4) != hipSuccess ||
        c->qs_stats.reserve(16) != hipSuccess) {
        (void)hipGetLastError();
        c->shadow8.release();
        c->xscale8.release();
        c->shadow8_state = c->shadow8_state == -1 ? -2 : -1;
        return hipSuccess;
    }
    hipError_t e = hipMemsetAsync(c->qs_stats.p, 0, 16, s);
    if (e == hipSuccess)
        e = launch_shadow_i8(c->d_rows, c->dtype, (uint32_t)rows, c->pitch, c->dim, static_cast<unsigned char*>(c->shadow8.p),
                             (uint32_t)pitch8, static_cast<float*>(c->xscale8.p), static_cast<float*>(c->qs_stats.p), s);
    if (e == hipSuccess) {
        c->shadow8_rows = rows;
        c->shadow8_state = rows == std::max<uint64_t>(c->n, 1) ? 1 : 2;
    }
    return e;
}

// Threshold refinement of the int8 selection (scan_mfma.h: launch_refine_tau) in front of the LAST phase -- (g - 1) / g of
// the corpus -- when that is at least this many rows (in front of the second largest phase too it cost what it saved);
// MVF_QS_REFINE=0 switches it off (A/B runs).
constexpr uint64_t kRefineMinRows = 200000;
// ... and in front of a phase only where the phase is worth it: a refinement is ~12-16 us of latency whatever the batch, what it
// spares the phase behind it grows with queries x rows.  One process, the switch toggled between rounds (scripts/probe_tune_ab.py,
// profiles/r05_k2_walk_and_phase_costs.txt 8c): 16 / 64 queries on 1M x 768 -4 % without, 32 on 3M -1.3 %, 16 on 10M -1 %;
// 64 on 10M +1 % without, 128 and more: with.
constexpr uint64_t kRefineMinPairs = 256ull << 20;
// is the threshold refined behind phase bi (rows [.., bounds[bi])), in front of the phase [bounds[bi], bounds[bi + 1])?
bool k2_refine_before(const Tuning& t, const std::vector<uint64_t>& bounds, size_t bi, uint32_t nq) {
    if (!t.qs_refine_phases || bi + 1 >= bounds.size()) return false;
    const uint64_t next_rows = bounds[bi + 1] - bounds[bi];
    return bi + t.qs_refine_phases >= bounds.size() - 1 && next_rows >= kRefineMinRows && (uint64_t)nq * next_rows >= kRefineMinPairs;
}
bool qs_refine_enabled(const mvfgpu_corpus* c) { return c->tune.qs_refine; }

// K2 per-query state (threshold key, candidate count, overflow flag; for the refinement the number of best candidates
// at the head of the list and the worst exact key among them), armed once and re-armed by the kernels that end a search.
int ensure_bstate(const mvfgpu_corpus* c, uint32_t nq_pad, hipStream_t s) {
    if (c->bstate_slots >= nq_pad) return MVF_OK;
    HIP_TRY(c->bstate.reserve((size_t)nq_pad * 20));
    HIP_TRY(hipMemsetAsync(c->bstate.p, 0xFF, (size_t)nq_pad * 4, s));                                                     // tau
    HIP_TRY(hipMemsetAsync(static_cast<unsigned char*>(c->bstate.p) + (size_t)nq_pad * 4, 0, (size_t)nq_pad * 16, s));  // cnt, overflow, ntop, lkey
    c->bstate_slots = nq_pad;
    return MVF_OK;
}

// K4: per-row norms of the STORED rows, once per resident corpus (a shadow only feeds the dot products).
// float rows: |x| [n], sum x^2 [n], max sum x^2 [1]; Int8 rows: sum x^2 (i32) [n]; UInt8 rows: sum (x-128)^2 [n],
// 128 * sum (x-128) [n]
// Layout: two arrays of norm_stride(n) entries (a multiple of 256, so both are 1-KB aligned), 256 entries of padding
// (the ping-pong K2 kernel DMAs the 256 entries of a row tile at once, also for the corpus' last, partial tile), then
// the one-element maximum.
size_t norm_stride(uint64_t n) { return ((size_t)std::max<uint64_t>(n, 1) + 255u) & ~(size_t)255u; }
size_t norm_max_at(uint64_t n) { return 2 * norm_stride(n) + 256; }

int ensure_norms(const mvfgpu_corpus* c, hipStream_t s) {
    if (c->xnorm_ready) return MVF_OK;
    const uint32_t n = (uint32_t)c->n;
    const size_t nn = norm_stride(n);
    HIP_TRY(c->xnorm.reserve((norm_max_at(n) + 1) * 4));
    float* xn = static_cast<float*>(c->xnorm.p);
    float* xmax = xn + norm_max_at(n);
    if (!is_int_dtype(c->dtype)) HIP_TRY(hipMemsetAsync(xmax, 0, 4, s));
    if (c->dtype == MVF_DTYPE_FLOAT32) HIP_TRY(launch_row_norms_f32(c->d_rows, n, c->pitch, xn, xn + nn, xmax, s));
    else HIP_TRY(launch_row_norms16(c->d_rows, c->dtype, n, c->pitch, c->dim, c->xnorm.p, xn + nn, xmax, s));
    c->xnorm_ready = true;
    return MVF_OK;
}

// Queries whose candidate budget overflowed (or whose margin reached past a truncated list) are redone exactly by K1 on
// the stored rows -- DECIDED ON THE DEVICE, so the search stays asynchronous (round 1 read the flags back and
// synchronised the stream): flag_compact_kernel turns the flags into a dense list + count; then ceil(nq / R) pairs of
// REPAIR launches follow unconditionally -- K1's repair variant (every block walks its window of the list in groups of
// four queries, one pass over the rows per group) and select_final's -- each of which reads the count and exits at once
// when its window is empty.  The common case (nothing flagged) costs a few empty launches (~2.5 us each on the device,
// hidden behind the scan on the host); an adversarial corpus is still answered exactly.  R (queries per pair) is what
// 256 MiB of per-block lists hold.
int repair_flagged_queries(const mvfgpu_corpus* c, uint8_t metric, const void* d_queries, uint32_t nq, uint32_t nq_pad,
                           uint32_t k, uint32_t* overflow, float* d_scores, uint64_t* d_indices, int32_t* d_raw,
                           hipStream_t s) {
    (void)nq_pad;
    if (c->n == 0) return MVF_OK;
    const uint32_t kcap = next_pow2(k);
    int nqv = 4, G;
    uint32_t J;
    choose_group(c->V, nqv, &G, &J, c->tune.k1_g);
    uint32_t chunk_rows = scan_chunk_rows(G, J, nqv), pmax = next_pow2(k + scan_chunk_safe(G));
    size_t lds = scan_lds_bytes(c->dtype, G, J, nqv, pmax);
    if (lds > 150 * 1024) {
        nqv = 1;
        choose_group(c->V, nqv, &G, &J, c->tune.k1_g);
        chunk_rows = scan_chunk_rows(G, J, nqv);
        pmax = next_pow2(k + scan_chunk_safe(G));
        lds = scan_lds_bytes(c->dtype, G, J, nqv, pmax);
    }
    const uint32_t nchunks = (uint32_t)((c->n + chunk_rows - 1) / chunk_rows);
    if (lds > 160 * 1024) return fail(MVF_ERR_BUILD, "dimension too large for the streaming kernel's LDS query tile");
    const void* kfn = scan_kernel(c->dtype, metric, G, nqv, /*redo=*/true);
    int occ = 1;
    {
        const int orc = scan_occupancy(c, kfn, lds, &occ);
        if (orc != MVF_OK) return orc;
    }
    // Two blocks per CU at most: the repair is rare, and every block's list costs scratch for EVERY query a launch pair may
    // serve -- with fewer lists a pair serves more queries (up to 4096 within 256 MiB of lists: 512 at k = 100, all of a
    // 10,000-query batch in three pairs at k = 10), and a 1024-query search enqueues 2 (empty) pairs instead of 16
    // (0.14 ms of launches on a 10-ms search; a 10,000-query search on 1M x 128 spent 0.36 of its 5.6 ms in 40 empty pairs
    // while the cap was 256).
    const uint32_t nblocks = std::min<uint32_t>(nchunks, (uint32_t)std::min(occ, 2) * (uint32_t)c->num_cus);
    const size_t per_query = (size_t)nblocks * kcap * 8;
    uint32_t R = (uint32_t)std::min<size_t>(4096, std::max<size_t>(4, ((size_t)256 << 20) / per_query));
    if (c->tune.repair_window) R = std::min<uint32_t>(R, std::max<uint32_t>(4, c->tune.repair_window));  // tests: several windows on small batches
    HIP_TRY(c->repair.reserve((size_t)R * per_query + (size_t)nq * 4 + 16));
    uint64_t* lists = static_cast<uint64_t*>(c->repair.p);
    uint32_t* redo_cnt = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(c->repair.p) + (size_t)R * per_query);
    uint32_t* redo_list = redo_cnt + 4;
    HIP_TRY(launch_flag_compact(overflow, nq, redo_list, redo_cnt, feedback_mirror(c), s));
    c->last_redo_cnt = redo_cnt;
    for (uint32_t base = 0; base < nq; base += R) {
        ScanParams sp{};
        sp.rows = c->d_rows;
        sp.queries = d_queries;
        sp.tomb = static_cast<const uint32_t*>(c->tomb.p);
        sp.cand = lists;
        sp.n = (uint32_t)c->n;
        sp.pitch = c->pitch;
        sp.dim = c->dim;
        sp.V = c->V;
        sp.J = J;
        sp.q0 = 0;
        sp.nq_total = nq;
        sp.k = k;
        sp.kcap = kcap;
        sp.pmax = pmax;
        sp.chunk_rows = chunk_rows;
        sp.chunk_safe = std::min(scan_chunk_safe(G), chunk_rows);
        {
            const uint32_t step = 16u * 64u / (uint32_t)G, want = std::max(k, 64u);
            sp.first_piece = (c->tune.k1_first_piece && chunk_rows <= scan_chunk_safe(G)) ? std::min(sp.chunk_safe, (want + step - 1) / step * step)
                                                                                         : sp.chunk_safe;
        }
        sp.nchunks = nchunks;
        sp.rank_merge_max = c->tune.k1_rank_merge;
        sp.redo_list = redo_list;
        sp.redo_cnt = redo_cnt;
        sp.redo_base = base;
        sp.redo_max = R;
        HIP_TRY(scan_launch(c->dtype, sp, metric, G, nqv, dim3(nblocks), lds, s));
        SelectParams fp{};
        fp.lists = lists;
        fp.nlists = nblocks;
        fp.kcap = kcap;
        fp.heads = (k + nblocks - 1) / nblocks;
        fp.P = 4096;
        fp.k = k;
        fp.metric = metric;
        fp.dtype = c->dtype;
        fp.index_base = c->index_base;
        fp.ids = static_cast<const uint64_t*>(c->ids.p);
        fp.out_scores = d_scores;
        fp.out_indices = d_indices;
        fp.out_raw = d_raw;
        fp.redo_list = redo_list;
        fp.redo_cnt = redo_cnt;
        fp.redo_base = base;
        HIP_TRY(launch_select_final(fp, std::min(R, nq - base), s));
    }
    if (c->tune.debug_repair) {  // diagnostics only: how many queries took the repair path (synchronises)
        uint32_t n = 0;
        HIP_TRY(hipMemcpyAsync(&n, redo_cnt, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (n) fprintf(stderr, "[mvfgpu] overflow repair: %u of %u queries redone by K1\n", n, nq);
    }
    return MVF_OK;
}

// K2 path: MFMA batched scan in geometric phases with per-query candidate
// compaction between them (scan_mfma.hip for Float32 rows, scan_mfma16.hip for
// Float16 / Int8 rows).  Asynchronous: queries whose candidate budget overflowed are
// redone exactly by K1 in repair launches that decide on the device whether to run.
// One ROW RANGE [lo, hi) of the corpus (the whole of it, or one of the two ranges of a corpus whose int8 shadow covers a prefix
// of the rows: search_batched_path below).  allow_qs: the range may select on the int8 shadow (it lies inside it).  defer: the
// caller merges this range's lists with the other's first and runs the repair of flagged queries / the feedback itself --
// what it needs for that comes back in *dr.  profile: this range's last phase is the one the handle's timing reports.
struct BatchedDeferred {
    uint32_t* overflow = nullptr;
    uint32_t nq_pad = 0;
    bool used_bias = false, used_qs = false;
};
int search_batched_range(const mvfgpu_corpus* c, uint8_t metric, const void* d_queries, uint32_t nq, uint32_t k,
                         float* d_scores, uint64_t* d_indices, int32_t* d_raw, hipStream_t s, uint64_t lo, uint64_t hi, bool allow_qs,
                         bool defer, bool profile, BatchedDeferred* dr) {
    // Float32 rows: either the exact f32 MFMA kernel on the rows themselves, or -- 4x faster -- the f16 kernel on a
    // scaled-f16 SHADOW copy that only selects candidates (error bound below); the kept rows are re-scored from the
    // f32 rows and the f32 query either way, so results do not depend on which one ran.
    bool use_shadow = false, use_qs = false;
    const bool rescore_fits = (size_t)((c->dim + 7u) & ~7u) * 4 + kBatchCap * 4 <= 64 * 1024;  // query + candidates in LDS
    if (allow_qs && qs_wanted(c, k)) {  // int8 shadow: selection at the int8 MFMA rate (Float32 and Float16 corpora)
        int rc = ensure_norms(c, s);
        if (rc != MVF_OK) return rc;
        use_qs = c->shadow8_state == 1 || (c->shadow8_state == 2 && hi <= c->shadow8_rows);  // built by the caller
    }
    if (!use_qs && c->dtype == MVF_DTYPE_FLOAT32 && c->scan_path != 2 && (c->scan_path == 3 || shadow_enabled(c)) && rescore_fits) {
        HIP_TRY(ensure_shadow(c, s, c->scan_path == 3));
        use_shadow = c->shadow_state == 1;
    }
    const bool wide = c->dtype == MVF_DTYPE_FLOAT32 && !use_shadow && !use_qs;  // f32 rows: 128x128x32-float tiles
    const uint8_t kdtype = use_qs ? (uint8_t)MVF_DTYPE_INT8 : use_shadow ? (uint8_t)MVF_DTYPE_FLOAT16 : c->dtype;  // what the scan kernel reads
    const unsigned char* krows = use_qs       ? static_cast<const unsigned char*>(c->shadow8.p)
                                 : use_shadow ? static_cast<const unsigned char*>(c->shadow.p)
                                              : c->d_rows;
    const uint32_t kpitch = use_qs ? shadow8_pitch(c->dim) : use_shadow ? shadow_pitch(c->dim) : c->pitch;
    const bool dma = !wide && k2_dma_enabled(c);            // LDS-DMA kernel (default) or the register-staged one
    const uint32_t qpb = wide ? 128u : dma ? scan_mfma16_dma_queries_per_block(nq, c->tune.k2_tile) : scan_mfma16_queries_per_block(kdtype);
    const uint32_t tile_rows = wide ? 128u : dma ? scan_mfma16_dma_tile_rows(qpb) : 256u;
    const uint32_t nq_pad = (nq + qpb - 1u) / qpb * qpb;
    const uint32_t ktb = dma ? 64u : 128u;                  // k-tile bytes of the f16/int8 kernel in use
    const uint32_t KT = wide ? (c->dim + 31u) / 32u : (c->dim * elem_size(kdtype) + ktb - 1u) / ktb;
    const uint32_t KPB = KT * ktb;                         // prepared query row, bytes
    const uint32_t planes = 1u;
    const uint32_t cap = use_qs ? kBatchCapQS : kBatchCap;
    const uint32_t n = (uint32_t)c->n;          // the corpus: layout of the norm arrays
    const uint64_t nr = hi - lo;                // the range: what the phases cover

    HIP_TRY(c->bq.reserve((size_t)planes * nq_pad * KPB + (size_t)nq_pad * 12 + 64));
    unsigned char* qprep = static_cast<unsigned char*>(c->bq.p);
    float* qaux0 = reinterpret_cast<float*>(qprep + (size_t)planes * nq_pad * KPB);
    float* qaux1 = qaux0 + nq_pad;
    unsigned char* zeros = reinterpret_cast<unsigned char*>(qaux1 + nq_pad);  // 64 zero bytes
    float* qdelta = reinterpret_cast<float*>(zeros + 64);                       // [nq_pad] int8-shadow selection: bound per query
    if (c->bq_zeros != zeros || c->bq_zero_bytes != c->bq.bytes) {  // nothing in this path writes them: set once per allocation and layout, not
        HIP_TRY(hipMemsetAsync(zeros, 0, 64, s));                      // once per search (a fill kernel is 4.8 us); the streaming paths, which lay
        c->bq_zeros = zeros;                                            // the buffer out differently, forget the mark
        c->bq_zero_bytes = c->bq.bytes;
    }
    {
        int rc = ensure_bstate(c, nq_pad, s);
        if (rc != MVF_OK) return rc;
    }
    uint32_t* tau = static_cast<uint32_t*>(c->bstate.p);
    uint32_t* cnt = tau + c->bstate_slots;
    uint32_t* overflow = cnt + c->bstate_slots;
    HIP_TRY(c->bcand.reserve((size_t)nq_pad * cap * 8));
    const bool is_float = !is_int_dtype(c->dtype);
    // approximate selection + exact re-scoring: float L2 (GEMM-form distances) and every metric on Float16 rows
    // (single f16 query plane); the other combinations carry final keys through the phases
    const bool approx = is_float && (metric == MVF_METRIC_L2 || kdtype == MVF_DTYPE_FLOAT16 || use_qs);
    const bool need_norms = approx || metric != MVF_METRIC_INNER_PRODUCT || c->dtype == MVF_DTYPE_UINT8;
    const size_t nn = norm_stride(n);
    if (need_norms) {
        int rc = ensure_norms(c, s);
        if (rc != MVF_OK) return rc;
    }
    const float* xx2 = is_float && c->xnorm.p ? static_cast<const float*>(c->xnorm.p) + nn : nullptr;
    const float* xxmax = is_float && c->xnorm.p ? static_cast<const float*>(c->xnorm.p) + norm_max_at(n) : nullptr;
    if (use_qs)
        HIP_TRY(launch_prep_queries_i8s(static_cast<const float*>(d_queries), nq, nq_pad, c->dim, KPB, metric,
                                        static_cast<const float*>(c->qs_stats.p), xxmax, qprep, qaux0, qaux1, qdelta, s));
    else if (wide)
        HIP_TRY(launch_prep_queries(static_cast<const float*>(d_queries), nq, nq_pad, c->dim, KPB / 4,
                                    reinterpret_cast<float*>(qprep), qaux0, s));
    else
        HIP_TRY(launch_prep_queries16(d_queries, kdtype, nq, nq_pad, c->dim, KPB, qprep, qaux0, qaux1, s));

    BatchParams bp{};
    bp.qmat = reinterpret_cast<const float*>(qprep);
    bp.qnorm = qaux0;
    bp.rows = c->d_rows;
    bp.xnorm = static_cast<const float*>(c->xnorm.p);
    bp.xx2 = xx2;
    bp.xxmax = xxmax;
    bp.tomb = static_cast<const uint32_t*>(c->tomb.p);
    bp.tau = tau;
    bp.cand = static_cast<uint64_t*>(c->bcand.p);
    bp.cnt = cnt;
    bp.pitch = c->pitch;
    bp.V = c->V;
    bp.KP = KPB / 4;
    bp.KT = KT;
    bp.nq = nq;
    bp.mtiles = nq_pad / qpb;
    bp.cap = cap;

    Batch16Params hp{};
    hp.qprep = qprep;
    hp.qaux0 = qaux0;
    hp.qaux1 = qaux1;
    hp.rows = krows;
    hp.xscale = use_qs ? static_cast<const float*>(c->xscale8.p) : use_shadow ? static_cast<const float*>(c->xscale.p) : nullptr;
    hp.zeros = zeros;
    hp.xnorm_f = static_cast<const float*>(c->xnorm.p);
    hp.xnorm_i = static_cast<const int32_t*>(c->xnorm.p);
    hp.xbias_i = c->xnorm.p ? static_cast<const int32_t*>(c->xnorm.p) + nn : nullptr;
    hp.dim = c->dim;
    hp.xx2 = xx2;
    hp.xxmax = xxmax;
    hp.tomb = bp.tomb;
    hp.tau = tau;
    hp.cand = bp.cand;
    hp.cnt = cnt;
    hp.pitch = kpitch;
    hp.V = kpitch / 16;
    hp.KPB = KPB;
    hp.KT = KT;
    hp.nq = nq;
    hp.nq_pad = nq_pad;
    hp.mtiles = nq_pad / qpb;
    hp.cap = cap;
    // candidates leave the narrow-type kernels through per-block regions (no global atomics in the epilogue): T records in
    // all -- twice what the per-query lists can hold, at least the 128 MiB of round 2 -- split evenly over the blocks of
    // the launch; the LDS-DMA kernel's i32-accumulator flavours split a block's share once more per wave (raw records)
    uint64_t blk_records = 0;
    if (dma) {
        blk_records = std::min<uint64_t>(std::max<uint64_t>((uint64_t)kBlkMaxBlocks * kBlkCap, 2ull * nq_pad * cap), 32ull << 20);
        if (c->tune.region_records)  // MVF_K2_REGION_RECORDS (tests: force the regions to overflow)
            blk_records = std::max<uint64_t>((uint64_t)kBlkMaxBlocks * kBlkWaves, c->tune.region_records);
        blk_records -= blk_records % ((uint64_t)kBlkMaxBlocks * kBlkWaves);
        {
            const void* before = c->blk.p;
            HIP_TRY(c->blk.reserve((size_t)blk_records * 16 + (size_t)kBlkMaxBlocks * kBlkWaves * 4));
            if (c->blk.p != before) c->blk_armed_cnt = nullptr;  // fresh memory
        }
        hp.blk_cand = static_cast<uint4*>(c->blk.p);
        hp.blk_cnt = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(c->blk.p) + (size_t)blk_records * 16);
        hp.blk_cap = (uint32_t)(blk_records / kBlkMaxBlocks);
    }

    CompactParams cp{};
    cp.cand = bp.cand;
    cp.cnt = cnt;
    cp.tau = tau;
    cp.overflow = overflow;
    cp.cap = cap;
    cp.k = k;
    cp.metric = metric;
    cp.dtype = c->dtype;
    cp.index_base = c->index_base;
    cp.ids = static_cast<const uint64_t*>(c->ids.p);
    cp.out_scores = d_scores;
    cp.out_indices = d_indices;
    cp.out_raw = d_raw;
    cp.qnorm = wide ? qaux0 : qaux1;  // |q| (f32 path: qnorm; f16 path: second aux array)
    cp.xxmax = xxmax;
    // bound of the approximate score's error in units of (qq + xx) [L2], 1 [cosine], |q||x| [inner product]:
    // f32 accumulation of `dim` terms plus the norms, and on Float16 rows the query's rounding to f16 (2^-11 per
    // element, relative; elements that land in the f16 subnormals add < 2^-39 of it)
    cp.eps = (float)(std::max<uint32_t>(c->dim, 64) + 16) * 1.1920929e-7f;
    if (kdtype == MVF_DTYPE_FLOAT16) cp.eps += 4.8828125e-4f * 1.001f;
    if (use_shadow) cp.eps += 4.8828125e-4f * 1.001f;  // the shadow rows' own rounding (same bound, per element of x)
    cp.delta = use_qs ? qdelta : nullptr;               // int8 selection: the per-query bound from the query preparation
    // int8 selection: between the large phases the threshold is refined with exact scores of the k best (scan_mfma.h)
    uint32_t* ntop = overflow + c->bstate_slots;
    uint32_t* lkey = ntop + c->bstate_slots;
    const bool refine = use_qs && qs_refine_enabled(c);
    cp.ntop = refine ? ntop : nullptr;

    mvfgpu_timing tm{};
    tm.scan_kernel = wide ? 2u : use_qs ? 6u : use_shadow ? 4u : 3u;
    mvfgpu_corpus::ProfSlot* ps = nullptr;
    if (c->profiling && profile) {
        ps = &c->prof[c->prof_next % mvfgpu_corpus::kProfSlots];
        for (auto& e : ps->e)
            if (!e) HIP_TRY(hipEventCreate(&e));
        ps->scanned = false;
    }

    // Phase p scans rows [R_p, R_{p+1}) of the range (k2_phase_bounds above)
    const uint32_t g = k2_growth_for(c, nq, k, cap);
    const std::vector<uint64_t> bounds = k2_phase_bounds(nr, cap, g);  // R_1 .. R_{P+1} = nr
    size_t bi = 0;
    uint64_t begin = 0, end = bounds[0];
    // The region counters are zero when a search starts if the search before it on this handle left them so (every scatter re-arms
    // what it has read) and they still sit at the same address: no fill kernel per search then (4.7 us: 1.6 % of a 16-query search
    // on 1M rows).  The mark is taken down while this search runs: a search that fails half way leaves it down.
    bool regions_armed = hp.blk_cnt != nullptr && c->blk_armed_cnt == hp.blk_cnt, used_bias = false;
    c->blk_armed_cnt = nullptr;
    for (;;) {
        const bool last = end >= nr;
        if (end > begin) {
            bp.row_begin = hp.row_begin = (uint32_t)(lo + begin);
            bp.row_end = hp.row_end = (uint32_t)(lo + end);
            bp.ntiles = hp.ntiles = (uint32_t)((end - begin + tile_rows - 1) / tile_rows);
            bp.direct = hp.direct = (begin == 0 && end - begin <= cap) ? 1u : 0u;
            const bool regions = hp.blk_cand && !hp.direct;
            // which kernel takes the phase
            const bool use_pp = !wide && dma && qpb == 256u && k2_pp_wanted(c, kdtype, hp.ntiles, hp.mtiles, c->num_cus) &&
                                scan_mfma16_pp_usable(hp.mtiles, c->num_cus, KT);
            const bool use_sb = !wide && !use_pp && dma && qpb == 64u && k2_sb_enabled(c) && scan_mfma16_sb_usable(nq_pad, KT, nq);
            const bool persistent = k2_dma_persistent(c);
            // per-wave regions of raw records (scan_mfma16_bias.inc): the persistent LDS-DMA kernel on i32 accumulators
            // (per-wave regions: num_cus blocks x kBlkWaves counters in blk_cnt, which holds kBlkMaxBlocks * kBlkWaves -- a part
            // with more CUs than kBlkMaxBlocks keeps round 2's per-block regions instead of reading counters past the array)
            const bool wave_regions = !wide && dma && !use_pp && !use_sb && persistent && !c->bias_disabled && k2_bias_enabled(c) &&
                                      (uint32_t)c->num_cus <= kBlkMaxBlocks &&
                                      scan_mfma16_dma_wave_regions(kdtype, qpb, hp.direct != 0, regions, c->dim);
            hp.wave_regions = wave_regions ? 1u : 0u;
            used_bias |= wave_regions;
            const uint32_t region_blocks = wave_regions ? (uint32_t)c->num_cus : kBlkMaxBlocks;
            if (regions) hp.blk_cap = (uint32_t)(blk_records / region_blocks) & ~(kBlkWaves - 1u);
            // the region counters: zeroed once, re-armed by every scatter after it has read them
            if (regions && (!regions_armed || nq_pad > scatter_rearm_max_queries())) {
                HIP_TRY(hipMemsetAsync(hp.blk_cnt, 0, (size_t)kBlkMaxBlocks * kBlkWaves * 4, s));
                regions_armed = true;
            }
            if (ps && last) HIP_TRY(hipEventRecord(ps->e[0], s));
            if (wide) HIP_TRY(launch_scan_mfma_f32(bp, metric, c->num_cus, c->tune.k2_persistent, s));
            else if (use_pp) HIP_TRY(launch_scan_mfma16_pp(hp, kdtype, metric, c->num_cus, s));
            else if (use_sb) HIP_TRY(launch_scan_mfma16_sb(hp, kdtype, metric, c->num_cus, s));
            else if (dma && hp.direct && qpb >= 128u && c->tune.k2_direct64) {
                // The direct phase is a few thousand rows: 10 row tiles x 4 query tiles of 256 x 256 leave 216 CUs idle while 40 blocks
                // multiply, key and store 65536 pairs each.  In 64-query tiles (64 x 512) the same pairs spread over twice the blocks
                // at half the work each (the prepared queries and the slots by row offset do not depend on the tile shape).
                Batch16Params dp = hp;
                const uint32_t tr = scan_mfma16_dma_tile_rows(64u);
                dp.mtiles = nq_pad / 64u;
                dp.ntiles = (uint32_t)((end - begin + tr - 1) / tr);
                HIP_TRY(launch_scan_mfma16_dma(dp, kdtype, metric, c->num_cus, 64u, persistent, s));
            } else if (dma) HIP_TRY(launch_scan_mfma16_dma(hp, kdtype, metric, c->num_cus, qpb, persistent, s));
            else HIP_TRY(launch_scan_mfma16(hp, kdtype, metric, c->num_cus, c->tune.k2_persistent16, s));
            if (ps && last) {
                HIP_TRY(hipEventRecord(ps->e[1], s));
                ps->scanned = true;
                tm.scan_bytes = (end - begin) * c->dim * elem_size(kdtype);
                tm.scan_flops = 2ull * nq * (end - begin) * c->dim;
            }
            tm.scan_launches++;
            if (regions) {
                Batch16Params sp = hp;  // the scatter pass sees a wave's slice as a region of its own
                if (wave_regions) sp.blk_cap = hp.blk_cap / kBlkWaves;
                HIP_TRY(launch_scatter_cand(sp, wave_regions ? region_blocks * kBlkWaves : region_blocks, metric, kdtype, s));
            }
        }
        cp.direct_cnt = (begin == 0 && end > begin && end - begin <= cap) ? (uint32_t)(end - begin) : 0u;
        if (approx) HIP_TRY(launch_compact_margin(cp, nq, s));
        else HIP_TRY(launch_compact(cp, nq, last, s));
        if (last) break;
        if (refine && k2_refine_before(c->tune, bounds, bi, nq)) {  // worth its ~0.07 ms in front of the last (largest) phases
            RescoreParams rp{};
            rp.cand = bp.cand;
            rp.cnt = cnt;
            rp.tau = tau;
            rp.cap = cap;
            rp.k = k;
            rp.queries = static_cast<const float*>(d_queries);
            rp.rows = c->d_rows;
            rp.pitch = c->pitch;
            rp.dim = c->dim;
            rp.dtype = c->dtype;
            HIP_TRY(launch_refine_tau(rp, metric, nq, ntop, lkey, qdelta, s));
        }
        begin = end;
        end = bounds[++bi];
    }
    if (hp.blk_cnt && nq_pad <= scatter_rearm_max_queries()) c->blk_armed_cnt = hp.blk_cnt;  // every phase's scatter has been enqueued
    if (approx) {  // exact scores of the kept candidates from the caller's f32 queries, final top-k
        if (refine && nr >= kRefineMinRows) {  // once more on the final lists: the re-scoring skips what falls outside
            RescoreParams fp{};
            fp.cand = bp.cand;
            fp.cnt = cnt;
            fp.tau = tau;
            fp.cap = cap;
            fp.k = k;
            fp.queries = static_cast<const float*>(d_queries);
            fp.rows = c->d_rows;
            fp.pitch = c->pitch;
            fp.dim = c->dim;
            fp.dtype = c->dtype;
            fp.write_head = 1u;  // the exact composites of the head stay where the final pass would write them
            HIP_TRY(launch_refine_tau(fp, metric, nq, ntop, lkey, qdelta, s));
        }
        RescoreParams rp{};
        rp.head_done = (refine && nr >= kRefineMinRows) ? ntop : nullptr;
        rp.cand = bp.cand;
        rp.cnt = cnt;
        rp.tau = tau;
        rp.cap = cap;
        rp.k = k;
        rp.queries = static_cast<const float*>(d_queries);
        rp.rows = c->d_rows;
        rp.pitch = c->pitch;
        rp.dim = c->dim;
        rp.dtype = c->dtype;
        rp.index_base = c->index_base;
        rp.ids = static_cast<const uint64_t*>(c->ids.p);
        rp.out_scores = d_scores;
        rp.out_indices = d_indices;
        rp.out_raw = d_raw;
        HIP_TRY(launch_rescore(rp, metric, nq, s));
    }
    if (ps) {
        HIP_TRY(hipEventRecord(ps->e[2], s));
        c->timing = tm;
        c->prof_next++;
    }

    if (dr) {
        dr->overflow = overflow;
        dr->nq_pad = std::max(dr->nq_pad, nq_pad);
        dr->used_bias |= used_bias;
        dr->used_qs |= use_qs && c->scan_path != 5 && c->scan_path != 6;
    }
    if (defer) return MVF_OK;
    int rc = repair_flagged_queries(c, metric, d_queries, nq, nq_pad, k, overflow, d_scores, d_indices, d_raw, s);
    if (rc == MVF_OK && ((use_qs && c->scan_path != 5 && c->scan_path != 6) || used_bias))
        rc = qs_feedback_post(c, nq, s, used_bias, use_qs && c->scan_path != 5 && c->scan_path != 6);
    return rc;
}

int merge_topk_device_impl(const float* d_scores, const uint64_t* d_indices, const int32_t* d_raw, size_t ls_scores, size_t ls_indices,
                           size_t ls_raw, uint32_t nlists, uint32_t nq, uint32_t k, uint8_t metric, uint8_t data_type, float* d_out_scores,
                           uint64_t* d_out_indices, int32_t* d_out_raw, int device, void* hip_stream);

// The batched search of a corpus.  Usually ONE range, all rows.  A Float32 / Float16 corpus whose int8 selection shadow does
// not fit beside it as a whole (the 100M x 1024 Float16 corpus of BASELINE configs[4] on one GPU: 204.8 GB of rows, 102.4 GB of
// shadow) gets a shadow of the longest prefix of rows that does (ensure_shadow8) and is searched as TWO row ranges -- the
// prefix by the int8 selection, the rest by the f16 kernels on the stored rows -- whose exact top-k lists are merged like two
// shards' (the same device merge: ties by ascending position); the repair of flagged queries runs once, behind the merge, over
// the whole corpus.  100M x 1024 f16, 1024 queries: 181-184 ms -> 95.5 (profiles/r05_partial_shadow.json).
int search_batched_path(const mvfgpu_corpus* c, uint8_t metric, const void* d_queries, uint32_t nq, uint32_t k,
                        float* d_scores, uint64_t* d_indices, int32_t* d_raw, hipStream_t s) {
    qs_feedback_poll(c);
    if (qs_wanted_batched(c, k)) {
        int rc = ensure_norms(c, s);
        if (rc != MVF_OK) return rc;
        HIP_TRY(ensure_shadow8(c, s, c->scan_path == 5 || c->scan_path == 6, /*allow_partial=*/true));
    }
    if (!(c->shadow8_state == 2 && qs_wanted(c, k) && c->shadow8_rows < c->n))
        return search_batched_range(c, metric, d_queries, nq, k, d_scores, d_indices, d_raw, s, 0, c->n, true, false, true, nullptr);
    {   // the per-query state (thresholds, counts, overflow flags) must not be re-allocated between the two ranges -- the second
        // would lose the first's overflow flags: sized here for the widest query tile either range may take
        const int brc = ensure_bstate(c, (nq + 255u) & ~255u, s);
        if (brc != MVF_OK) return brc;
    }
    const size_t ls = (size_t)nq * k;
    HIP_TRY(c->split_out.reserve(2 * ls * 16));
    unsigned char* t = static_cast<unsigned char*>(c->split_out.p);
    uint64_t* ti = reinterpret_cast<uint64_t*>(t);                 // [2][nq k]
    float* ts = reinterpret_cast<float*>(t + 2 * ls * 8);          // [2][nq k]
    int32_t* tr = reinterpret_cast<int32_t*>(t + 2 * ls * 12);     // [2][nq k]
    BatchedDeferred dr{};
    int rc = search_batched_range(c, metric, d_queries, nq, k, ts, ti, tr, s, 0, c->shadow8_rows, true, true, true, &dr);
    if (rc != MVF_OK) return rc;
    rc = search_batched_range(c, metric, d_queries, nq, k, ts + ls, ti + ls, tr + ls, s, c->shadow8_rows, c->n, false, true, false, &dr);
    if (rc != MVF_OK) return rc;
    rc = merge_topk_device_impl(ts, ti, tr, ls, ls, ls, 2, nq, k, metric, c->dtype, d_scores, d_indices, d_raw, c->device, s);
    if (rc != MVF_OK) return rc;
    rc = repair_flagged_queries(c, metric, d_queries, nq, dr.nq_pad, k, dr.overflow, d_scores, d_indices, d_raw, s);
    if (rc == MVF_OK && (dr.used_qs || dr.used_bias)) rc = qs_feedback_post(c, nq, s, dr.used_bias, dr.used_qs);
    return rc;
}

// Scan path 4 (opt-in): one or two queries on a Float32 corpus STREAM ITS SCALED-F16 SHADOW -- half the bytes of the
// stored rows, so half the time of the HBM-bound K1 -- with the batched path's select-with-a-margin / re-score-exactly
// scheme: K1 (dt1x unit: f16 rows times xscale[r]) keeps the k' > k best approximate scores; every candidate within
// twice the error bound of the k-th is re-scored from the stored f32 rows and the f32 query; if ALL k' are inside the
// margin (rows beyond the cut may be too) the query is flagged and redone by the exact K1 (on-device conditional
// repair launches, like the batched path: asynchronous).
//   x~ = x (1 + e), |e| <= 2^-11 per element:  |q.x~ - q.x| <= 2^-11 |q||x|;  | |q - x~| - |q - x| | <= 2^-11 |x|;
//   cosine (numerator and denominator both from x~) <= 2 * 2^-11; plus the f32 accumulation, (dim + 16) 2^-23.
int search_stream_shadow_path(const mvfgpu_corpus* c, uint8_t metric, const void* d_queries, uint32_t nq, uint32_t k,
                              float* d_scores, uint64_t* d_indices, int32_t* d_raw, hipStream_t s) {
    const uint32_t cap = kBatchCap;
    const uint32_t nq_pad = (nq + 255u) & ~255u;
    const uint32_t n = (uint32_t)c->n;
    int rc = ensure_bstate(c, nq_pad, s);
    if (rc == MVF_OK) rc = ensure_norms(c, s);
    if (rc != MVF_OK) return rc;
    uint32_t* tau = static_cast<uint32_t*>(c->bstate.p);
    uint32_t* cnt = tau + c->bstate_slots;
    uint32_t* overflow = cnt + c->bstate_slots;
    HIP_TRY(c->bcand.reserve((size_t)nq_pad * cap * 8));
    // |q| per query for the margins: the f16 query preparation computes it (its planes are not used here)
    const uint32_t KPB = ((c->dim * 2u + 63u) / 64u) * 64u;
    HIP_TRY(c->bq.reserve((size_t)nq_pad * KPB + (size_t)nq_pad * 8 + 64));
    c->bq_zeros = nullptr;  // another layout of the buffer: the batched path's zero bytes may be overwritten
    unsigned char* qprep = static_cast<unsigned char*>(c->bq.p);
    float* qaux0 = reinterpret_cast<float*>(qprep + (size_t)nq_pad * KPB);
    float* qaux1 = qaux0 + nq_pad;
    HIP_TRY(launch_prep_queries16(d_queries, MVF_DTYPE_FLOAT16, nq, nq_pad, c->dim, KPB, qprep, qaux0, qaux1, s));

    const uint32_t ksel = std::min<uint32_t>(MVFGPU_K_PER_PASS, std::max(2u * k, k + 64u));  // k' candidates per query
    ShadowStream alt{};
    alt.rows = static_cast<const unsigned char*>(c->shadow.p);
    alt.xscale = static_cast<const float*>(c->xscale.p);
    alt.pitch = shadow_pitch(c->dim);
    alt.V = alt.pitch / 16;
    choose_group(alt.V, 1, &alt.G, &alt.J, c->tune.k1_g);
    alt.cand = static_cast<uint64_t*>(c->bcand.p);
    alt.cnt = cnt;
    alt.cand_cap = cap;
    rc = search_stream_path(c, metric, d_queries, nq, ksel, nullptr, nullptr, nullptr, s, /*profile=*/true, &alt);
    if (rc != MVF_OK) return rc;

    CompactParams cp{};
    cp.cand = alt.cand;
    cp.cnt = cnt;
    cp.tau = tau;
    cp.overflow = overflow;
    cp.cap = cap;
    cp.k = k;
    cp.metric = metric;
    cp.dtype = c->dtype;
    cp.index_base = c->index_base;
    cp.ids = static_cast<const uint64_t*>(c->ids.p);
    cp.qnorm = qaux1;
    cp.xxmax = static_cast<const float*>(c->xnorm.p) + norm_max_at(n);
    // per-element relative rounding of the shadow rows (2^-11) and f32 accumulation of `dim` terms ((dim + 16) 2^-23
    // of the sum; half of that, relative, on a square root)
    const float e11 = 4.8828125e-4f * 1.001f, eacc = (float)(std::max<uint32_t>(c->dim, 64) + 16) * 1.1920929e-7f;
    cp.eps = metric == MVF_METRIC_COSINE ? 2.0f * e11 + eacc : metric == MVF_METRIC_INNER_PRODUCT ? e11 + eacc : e11;
    cp.eps_acc = 0.5f * eacc;
    cp.l2_is_distance = 1;
    cp.truncated_at = n > ksel ? ksel : 0u;
    HIP_TRY(launch_compact_margin(cp, nq, s));

    RescoreParams rp{};
    rp.cand = alt.cand;
    rp.cnt = cnt;
    rp.tau = tau;
    rp.cap = cap;
    rp.k = k;
    rp.queries = static_cast<const float*>(d_queries);
    rp.rows = c->d_rows;
    rp.pitch = c->pitch;
    rp.dim = c->dim;
    rp.dtype = c->dtype;
    rp.index_base = c->index_base;
    rp.ids = static_cast<const uint64_t*>(c->ids.p);
    rp.out_scores = d_scores;
    rp.out_indices = d_indices;
    rp.out_raw = d_raw;
    HIP_TRY(launch_rescore(rp, metric, nq, s));
    return repair_flagged_queries(c, metric, d_queries, nq, nq_pad, k, overflow, d_scores, d_indices, d_raw, s);
}

// One to four queries on a Float32 / Float16 corpus STREAM ITS INT8 SHADOW (dim bytes per row: a quarter / half of the
// stored rows) through K1's dt2x unit -- exact i32 dots, float keys -- with the int8 selection's proven per-query bound
// (shadow_i8.hip).  The bound is ~0.25 sigma of the score distribution -- 6-10 x k rows on the benchmark's data, more
// than K1's selection can carry as "the k' best" -- so the selection is by BLOCK: every K1 block keeps the max(k, 32)
// best approximate scores of ITS rows (n / ~1000 of them), select_final's margin mode finds the k-th best of all lists
// and gathers every entry within 2 delta of it; a list that was cut inside the bound (rows beyond the cut may be inside
// too: clustered rows, tiny corpora) or more than 2048 rows inside it flag the query, which the exact K1 then redoes.
// rescore_kernel re-scores what was gathered from the stored rows and the f32 query.
uint32_t stream_qs_klist(uint32_t k) { return std::max(k, 32u); }

int search_stream_qs_path(const mvfgpu_corpus* c, uint8_t metric, const void* d_queries, uint32_t nq, uint32_t k,
                          float* d_scores, uint64_t* d_indices, int32_t* d_raw, hipStream_t s) {
    const uint32_t cap = kBatchCap;
    const uint32_t nq_pad = (nq + 255u) & ~255u;
    const uint32_t n = (uint32_t)c->n;
    int rc = ensure_bstate(c, nq_pad, s);
    if (rc == MVF_OK) rc = ensure_norms(c, s);
    if (rc != MVF_OK) return rc;
    uint32_t* tau = static_cast<uint32_t*>(c->bstate.p);
    uint32_t* cnt = tau + c->bstate_slots;
    uint32_t* overflow = cnt + c->bstate_slots;
    HIP_TRY(c->bcand.reserve((size_t)nq_pad * cap * 8));
    const uint32_t KPB = shadow8_pitch(c->dim);
    HIP_TRY(c->bq.reserve((size_t)nq_pad * KPB + (size_t)nq_pad * 12 + 64));
    c->bq_zeros = nullptr;  // another layout of the buffer: the batched path's zero bytes may be overwritten
    unsigned char* qprep = static_cast<unsigned char*>(c->bq.p);
    float* qaux0 = reinterpret_cast<float*>(qprep + (size_t)nq_pad * KPB);
    float* qaux1 = qaux0 + nq_pad;
    float* qdelta = qaux1 + nq_pad;
    const size_t nn = norm_stride(n);
    const float* xn = static_cast<const float*>(c->xnorm.p);
    const float* xxmax = xn + norm_max_at(n);
    HIP_TRY(launch_prep_queries_i8s(static_cast<const float*>(d_queries), nq, nq, c->dim, KPB, metric,
                                    static_cast<const float*>(c->qs_stats.p), xxmax, qprep, qaux0, qaux1, qdelta, s));

    ShadowStream alt{};
    alt.i8 = true;
    alt.rows = static_cast<const unsigned char*>(c->shadow8.p);
    alt.xscale = static_cast<const float*>(c->xscale8.p);
    alt.qprep = qprep;
    alt.qaux0 = qaux0;
    alt.qaux1 = qaux1;
    alt.xrow = metric == MVF_METRIC_L2 ? xn + nn : xn;  // sum x^2 | |x| of the stored rows
    alt.qstride = KPB;
    alt.pitch = KPB;
    alt.V = KPB / 16;
    alt.cand = static_cast<uint64_t*>(c->bcand.p);
    alt.cnt = cnt;
    alt.cand_cap = cap;
    alt.delta = qdelta;
    alt.tau = tau;
    alt.overflow = overflow;
    alt.rank_k = k;
    rc = search_stream_path(c, metric, d_queries, nq, stream_qs_klist(k), nullptr, nullptr, nullptr, s, /*profile=*/true, &alt);
    if (rc != MVF_OK) return rc;

    RescoreParams rp{};
    rp.cand = alt.cand;
    rp.cnt = cnt;
    rp.tau = tau;
    rp.cap = cap;
    rp.k = k;
    rp.queries = static_cast<const float*>(d_queries);
    rp.rows = c->d_rows;
    rp.pitch = c->pitch;
    rp.dim = c->dim;
    rp.dtype = c->dtype;
    rp.index_base = c->index_base;
    rp.ids = static_cast<const uint64_t*>(c->ids.p);
    rp.out_scores = d_scores;
    rp.out_indices = d_indices;
    rp.out_raw = d_raw;
    HIP_TRY(launch_rescore(rp, metric, nq, s));
    rc = repair_flagged_queries(c, metric, d_queries, nq, nq_pad, k, overflow, d_scores, d_indices, d_raw, s);
    if (rc == MVF_OK && c->scan_path != 6) rc = qs_feedback_post(c, nq, s);
    return rc;
}

// One to four queries stream the int8 shadow on request (scan path 6; MVF_STREAM_I8=1: one query on scan path 0, once the
// corpus HOLDS an int8 shadow anyway): 1.24 instead of 4.5 ms on 10M x 768, same rows, scores within the tolerance (the
// re-scoring kernel sums in another order than K1).  By default one query reads the stored rows, whatever the handle
// has served before (the bench's headline: no extra memory, no build).  Two to four queries are served as fast by the 64-query MFMA tile on the same
// shadow (1.55-1.65 ms against 1.53-1.72: profiles/r02_stream_int8_shadow_1to4_queries.txt).  MVF_STREAM_I8=0 opts
// out; a corpus whose queries keep needing the repair pass switches itself back (qs_disabled).
bool stream_qs_wanted(const mvfgpu_corpus* c, uint32_t nq, uint32_t k) {
    if (nq < 1 || nq > 4 || k > kQsStreamMaxK || !qs_wanted(c, k)) return false;
    if (c->scan_path == 6) return true;
    // scan path 0: only with MVF_STREAM_I8=1 and once the corpus holds an int8 shadow anyway.  (Round 2 did this by itself:
    // the same query on the same handle then gave scores in a different summation order -- K1's against the re-scoring
    // kernel's -- depending on whether an earlier batched search had built the shadow.  An automatic path must not
    // depend on the handle's history.)
    if (c->scan_path != 0 || nq != 1 || c->shadow8_state != 1) return false;
    return c->tune.stream_i8;
}

// Scan path 4 applies to one or two queries on a Float32 corpus whose shadow exists (or can be built now).
bool stream_shadow_wanted(const mvfgpu_corpus* c, uint32_t nq) {
    if (c->dtype != MVF_DTYPE_FLOAT32 || nq > 2 || c->n == 0) return false;
    if (c->scan_path != 4 && !(c->scan_path == 0 && c->tune.stream_shadow)) return false;
    return (size_t)((c->dim + 7u) & ~7u) * 4 + kBatchCap * 4 <= 64 * 1024;  // the re-scoring kernel keeps the query in LDS
}

bool use_batched_path(const mvfgpu_corpus* c, uint8_t metric, uint32_t nq) {
    if (c->scan_path == 1) return false;
    // Float32 / Float16: every metric (L2 = GEMM-form selection with an error margin + exact re-scoring);
    // Int8 / UInt8: every metric, exact integers (UInt8 rides the signed MFMA shifted by 128).
    bool supported = true;
    if ((metric == MVF_METRIC_L2 || c->dtype == MVF_DTYPE_FLOAT16) && !is_int_dtype(c->dtype) &&
        (size_t)((c->dim + 7u) & ~7u) * 4 + kBatchCap * 4 > 64 * 1024)
        supported = false;  // the re-scoring kernel keeps the query in LDS
    if (!supported) return false;
    if (c->scan_path == 2 || c->scan_path == 3 || (c->scan_path == 5 && !is_int_dtype(c->dtype))) return true;
    if (c->scan_path == 6 && !is_int_dtype(c->dtype)) return nq > 4;
    // K1 takes up to 4 queries per pass over the rows; K2 costs a flat padded-tile time -- since the streaming MFMA kernel
    // (scan_mfma16_sb.hip) about ONE pass over the int8 shadow / the Int8 rows for up to 64 queries, plus ~0.1 ms of phase
    // launches.  Measured crossovers (scripts/probe_small_corpora.py, profiles/r02_small_corpora_crossover.txt): Float32 /
    // Float16 rows with a shadow from 2 queries up at every size measured (1M x 768 f32, 16 queries: 0.25 ms against K1's
    // 2.4 -- until round 2 corpora under 1 GiB kept K1 up to 31 queries), Int8 / UInt8 rows from 5 (from 2 on 2 GiB and more,
    // 1 GiB when the rows are <= 256 B: the four-query pass of K1 holds 4.3-5.3 TB/s on 768-B rows and ~3 TB/s on
    // short ones, the streaming MFMA kernel 6.2 behind ~0.13 ms of fixed cost -- 50M x 768, 2 queries: 7.3 -> 6.2 ms), Float32 rows without
    // a shadow (exact f32 MFMA kernel, 128-query tiles) from 9 on >= 1 GiB and from 32 below.  Corpora under 16 MiB keep K1
    // until the batch is MFMA-sized: their searches take tens of microseconds either way, and the batched path's scratch
    // (128 MiB of candidate regions per handle) would dwarf them.
    const uint64_t bytes = c->n * (uint64_t)c->dim * elem_size(c->dtype);
    const bool shadowed = qs_wanted(c) ||  // int8 selection, else the f16 shadow (runs as Float16)
                          (c->dtype == MVF_DTYPE_FLOAT32 && (c->scan_path == 3 || shadow_enabled(c)) && c->shadow_state >= 0 &&
                           (size_t)((c->dim + 7u) & ~7u) * 4 + kBatchCap * 4 <= 64 * 1024);
    // Mid-size corpora, a pass or two of K1 (profiles/r04_small_corpora_crossover.txt, round 4: K1's four-query pass got its
    // counting merge, batched staging and a short first piece; the batched route costs 70-100 us before its first row): two to
    // four queries stay on K1 up to 512 MB of float rows (100k x 128 f32: 41 against 90 us; 300k x 768 f16: 135 against 167),
    // up to eight on <= 160 MB; Int8 / UInt8 rows up to eight queries on <= 96 MB (<= 256 MB when rows are >= 512 B) and
    // sixteen on <= 32 MB of such rows (30k x 768: 54 against 97 us).
    if (bytes >= (16ull << 20)) {
        const uint32_t row_bytes = c->dim * elem_size(c->dtype);
        if (!is_int_dtype(c->dtype)) {
            if (shadowed && nq <= 4 && bytes <= (512ull << 20)) return false;
            if (shadowed && nq <= 8 && bytes <= (160ull << 20)) return false;
        } else {
            if (nq <= 8 && bytes <= ((row_bytes >= 512u ? 256ull : 96ull) << 20)) return false;
            if (nq <= 16 && bytes <= (32ull << 20) && row_bytes >= 512u) return false;
        }
    }
    // (round 5, the same probe on the final tree: K1's passes over SHORT rows cost by the row, not by the byte -- 100k x 128 int8,
    //  13 MB: 12 / 16 queries 111 / 145 us against 97 / 96 batched; 30k x 128 f16, 7.7 MB, whose rows K1 converts: 16 / 32 queries
    //  79 / 153 us against 72 / 76)
    const uint32_t small = c->n >= 65536u ? 12u : (c->n >= 24576u && c->dtype == MVF_DTYPE_FLOAT16) ? 16u : 0u;
    const uint32_t threshold = bytes < (8ull << 20)                           ? (small ? small : 33u)  // (32 queries = 8 fused passes: 37-39 us against 73-85)
                               : bytes < (16ull << 20)                        ? (small ? small : 32u)
                               : c->dtype == MVF_DTYPE_FLOAT32 && !shadowed ? (bytes < (1ull << 30) ? 32u : 9u)
                               : is_int_dtype(c->dtype)                     ? (bytes < ((c->pitch <= 256u ? 2ull : 4ull) << 29) ? 5u : 2u)
                                                                            : 2u;
    return nq >= threshold;
}

// ---- upload pipeline: the step in front of the path ------------------------------------------------
// The host rows (an mmap'd MVF block: any alignment, stride >= row bytes) reach HBM in chunks on the copy stream while
// a second stream re-pitches, and -- on request -- computes the row norms (K4) and the f16 shadow of the chunk before:
//   copy stream   : H2D chunk i           | H2D chunk i+1            | ...
//   compute stream:   (wait ev_copy[i])  re-pitch i, norms i, shadow i |  ...
// so a corpus that will serve batched searches is ready for the first of them when the last chunk lands.
//   * rows tightly packed and a multiple of 16 B: chunks go straight to their place (no staging, no re-pitch);
//   * otherwise a chunk is copied AS IT LIES (gaps included) into one of two device staging buffers and the re-pitch
//     kernel reads it with the host stride -- only row_bytes of each row are read from the stage, the 16-B padding
//     is written as zeros (K1, the norms and the shadow all consume whole 16-B vectors).  Rows further apart than
//     twice their size go through hipMemcpy2DAsync instead (no point in moving the gaps over PCIe).
//   * source memory: pageable by default -- the runtime's own pinned bounce buffers already move a 1-D copy from
//     pageable / mmap'd memory at ~56 of the 63 GB/s PCIe Gen5 x16 offers; MVFGPU_UPLOAD_PINNED_STAGING instead
//     double-buffers through two pinned host chunks filled by a pool of memcpy threads (A/B: scripts/probe_upload.py).
struct PinnedPair {
    void* p[2] = {nullptr, nullptr};
    ~PinnedPair() {
        for (auto q : p)
            if (q) (void)hipHostFree(q);
    }
};
struct StagePair {
    unsigned char* p[2] = {nullptr, nullptr};
    ~StagePair() {
        for (auto q : p)
            if (q) (void)hipFree(q);
    }
};
struct EventSet {
    hipEvent_t e[6] = {};
    ~EventSet() {
        for (auto q : e)
            if (q) (void)hipEventDestroy(q);
    }
};

void parallel_memcpy(void* dst, const void* src, size_t bytes, unsigned threads) {
    if (threads <= 1 || bytes < ((size_t)8 << 20)) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> pool;
    const size_t part = ((bytes / threads) + 4095) & ~(size_t)4095;
    for (unsigned t = 0; t < threads; t++) {
        const size_t o = (size_t)t * part;
        if (o >= bytes) break;
        const size_t len = std::min(part, bytes - o);
        pool.emplace_back([=] { std::memcpy(static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, len); });
    }
    for (auto& t : pool) t.join();
}

int upload_rows(mvfgpu_corpus* c, const void* rows, uint64_t stride, const mvfgpu_upload_options& o) {
    const uint64_t n = c->n, row_bytes = (uint64_t)c->dim * elem_size(c->dtype);
    const bool direct = stride == row_bytes && row_bytes == c->pitch;
    const bool sparse = !direct && stride > 2 * row_bytes;  // 2-D copies: do not move the gaps
    // pinned double-buffered staging is the default for uploads of 256 MiB and more (measured, 30.72 GB of pageable rows,
    // profiles/r02_upload_pipeline.txt: 50 GB/s against 10-21 GB/s handing the pageable source to the runtime)
    const uint64_t total_span = (n - 1) * stride + row_bytes;
    const bool pinned = !sparse && ((o.flags & MVFGPU_UPLOAD_PINNED_STAGING) != 0 ||
                                    ((o.flags & MVFGPU_UPLOAD_PAGEABLE) == 0 && total_span >= ((uint64_t)256 << 20)));
    const uint64_t chunk_bytes = (uint64_t)(o.chunk_mib ? o.chunk_mib : (pinned ? 64u : 256u)) << 20;
    const uint64_t chunk_rows = std::max<uint64_t>(1, chunk_bytes / stride);
    const unsigned char* src = static_cast<const unsigned char*>(rows);
    HIP_TRY(hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    hipStream_t s_copy = c->own_stream, s_comp = c->up_stream;

    // what the compute stream prepares per chunk
    bool want_norms = (o.flags & (MVFGPU_UPLOAD_EAGER_NORMS | MVFGPU_UPLOAD_EAGER_SHADOW)) != 0;
    // which selection copy the batched searches of this corpus will use: the int8 shadow (Float32 and Float16 rows; the
    // default) or, with MVF_I8_SHADOW=0, the f16 shadow of Float32 rows
    bool want_shadow8 = (o.flags & MVFGPU_UPLOAD_EAGER_SHADOW) != 0 && qs_wanted(c);
    bool want_shadow = (o.flags & MVFGPU_UPLOAD_EAGER_SHADOW) != 0 && !want_shadow8 && c->dtype == MVF_DTYPE_FLOAT32 && shadow_enabled(c);
    float* xn = nullptr;
    const size_t nn = norm_stride(n);
    if (want_norms) {
        HIP_TRY(c->xnorm.reserve((norm_max_at(n) + 1) * 4));
        xn = static_cast<float*>(c->xnorm.p);
        if (!is_int_dtype(c->dtype)) HIP_TRY(hipMemsetAsync(xn + norm_max_at(n), 0, 4, s_comp));
    }
    if (want_shadow) {
        const size_t need = (size_t)n * shadow_pitch(c->dim);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + ((size_t)2 << 30) ||
            c->shadow.reserve(need) != hipSuccess || c->xscale.reserve(((size_t)n + 256) * 4) != hipSuccess) {
            (void)hipGetLastError();
            c->shadow.release();
            c->xscale.release();
            want_shadow = false;  // as ensure_shadow: the searches then use the exact f32 kernel
        }
    }

    if (want_shadow8) {
        const size_t need = (size_t)n * shadow8_pitch(c->dim);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + ((size_t)2 << 30) ||
            c->shadow8.reserve(need) != hipSuccess || c->xscale8.reserve(((size_t)n + 256) * 4) != hipSuccess ||
            c->qs_stats.reserve(16) != hipSuccess) {
            (void)hipGetLastError();
            c->shadow8.release();
            c->xscale8.release();
            want_shadow8 = false;
        } else {
            HIP_TRY(hipMemsetAsync(c->qs_stats.p, 0, 16, s_comp));  // the four bound maxima accumulate over the chunks
        }
    }

    StagePair stage;
    PinnedPair pin;
    EventSet ev;  // [0,1] copy of buffer b landed, [2,3] compute on buffer b done, [4,5] H2D out of pinned buffer b done
    for (auto& e : ev.e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const uint64_t span_max = (std::min(chunk_rows, n) - 1) * stride + row_bytes;  // bytes of one chunk as it lies
    if (!direct && !sparse)
        for (auto& q : stage.p) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q), span_max));
    if (pinned)
        for (auto& q : pin.p) HIP_TRY(hipHostMalloc(&q, span_max, hipHostMallocDefault));
    if (sparse && c->pitch != row_bytes) HIP_TRY(hipMemsetAsync(c->d_rows, 0, c->rows_bytes, s_copy));  // the 16-B padding
    unsigned threads = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (c->tune.upload_threads) threads = c->tune.upload_threads;

    // (Rows handed over straight off an mmap'd .mvf: with the page cache warm the copy threads' minor faults cost nothing
    // measurable -- 52 GB/s, the same as anonymous memory; cold, the file's device sets the rate, and MADV_WILLNEED on the
    // chunks ahead of the copy changed nothing (profiles/r04_upload_from_mmap.txt), so there is no readahead code here.)
    uint64_t i = 0;
    for (uint64_t r0 = 0; r0 < n; r0 += chunk_rows, i++) {
        const int b = (int)(i & 1);
        const uint64_t h = std::min(chunk_rows, n - r0);
        const uint64_t span = (h - 1) * stride + row_bytes;
        unsigned char* place = c->d_rows + r0 * c->pitch;
        if (!direct && !sparse && i >= 2) HIP_TRY(hipEventSynchronize(ev.e[2 + b]));  // stage b: its re-pitch is done
        if (sparse) {
            HIP_TRY(hipMemcpy2DAsync(place, c->pitch, src + r0 * stride, stride, row_bytes, h, hipMemcpyHostToDevice, s_copy));
        } else {
            unsigned char* dst = direct ? place : stage.p[b];
            if (pinned) {
                if (i >= 2) HIP_TRY(hipEventSynchronize(ev.e[4 + b]));  // pinned buffer b: its H2D is done
                parallel_memcpy(pin.p[b], src + r0 * stride, span, threads);
                HIP_TRY(hipMemcpyAsync(dst, pin.p[b], span, hipMemcpyHostToDevice, s_copy));
                HIP_TRY(hipEventRecord(ev.e[4 + b], s_copy));
            } else {
                HIP_TRY(hipMemcpyAsync(dst, src + r0 * stride, span, hipMemcpyHostToDevice, s_copy));
            }
        }
        HIP_TRY(hipEventRecord(ev.e[b], s_copy));
        HIP_TRY(hipStreamWaitEvent(s_comp, ev.e[b], 0));
        if (!direct && !sparse)
            HIP_TRY(launch_repack_rows(stage.p[b], place, h, (uint32_t)row_bytes, stride, c->pitch, s_comp));
        if (want_norms) {
            if (c->dtype == MVF_DTYPE_FLOAT32)
                HIP_TRY(launch_row_norms_f32(place, (uint32_t)h, c->pitch, xn + r0, xn + nn + r0, xn + norm_max_at(n), s_comp));
            else
                HIP_TRY(launch_row_norms16(place, c->dtype, (uint32_t)h, c->pitch, c->dim, xn + r0, xn + nn + r0,
                                           xn + norm_max_at(n), s_comp));
        }
        if (want_shadow8)
            HIP_TRY(launch_shadow_i8(place, c->dtype, (uint32_t)h, c->pitch, c->dim,
                                     static_cast<unsigned char*>(c->shadow8.p) + r0 * shadow8_pitch(c->dim), shadow8_pitch(c->dim),
                                     static_cast<float*>(c->xscale8.p) + r0, static_cast<float*>(c->qs_stats.p), s_comp));
        if (want_shadow)
            HIP_TRY(launch_shadow_f16(place, (uint32_t)h, c->pitch, c->dim,
                                      static_cast<unsigned char*>(c->shadow.p) + r0 * shadow_pitch(c->dim), shadow_pitch(c->dim),
                                      static_cast<float*>(c->xscale.p) + r0, s_comp));
        HIP_TRY(hipEventRecord(ev.e[2 + b], s_comp));
    }
    HIP_TRY(hipStreamSynchronize(s_copy));
    HIP_TRY(hipStreamSynchronize(s_comp));
    if (want_norms) c->xnorm_ready = true;
    c->shadow_state = want_shadow ? 1 : c->shadow_state;
    c->shadow8_state = want_shadow8 ? 1 : c->shadow8_state;
    return MVF_OK;
}

// k beyond one pass (MVFGPU_K_PER_PASS): ceil(k / 1024) passes of the streaming kernel, pass p returning the 1024 best rows
// ranked STRICTLY BEHIND the last row of pass p - 1 -- composites (order key << 32 | row) are distinct and totally ordered,
// so "behind the floor" is exactly the set of rows not yet returned.  The floor travels on the device (select_final writes
// it, the next pass's scan reads it): no host wait.  The reference takes any k: usize (examples/similarity_search.rs:143,
// :166-168); its heap holds k + 1 entries whatever k is.
int search_large_k(const mvfgpu_corpus* c, uint8_t metric, const void* d_queries, uint32_t nq, uint32_t k, float* d_scores,
                   uint64_t* d_indices, int32_t* d_raw, hipStream_t s) {
    HIP_TRY(c->floor1.reserve((size_t)nq * 8));
    uint64_t* fl = static_cast<uint64_t*>(c->floor1.p);
    for (uint32_t off = 0; off < k; off += MVFGPU_K_PER_PASS) {
        const uint32_t kk = std::min<uint32_t>(MVFGPU_K_PER_PASS, k - off);
        int rc = search_stream_path(c, metric, d_queries, nq, kk, d_scores, d_indices, d_raw, s, /*profile=*/off == 0, nullptr,
                                    off ? fl : nullptr, fl, k, off);
        if (rc != MVF_OK) return rc;
    }
    return MVF_OK;
}

// k beyond what passes are worth -- and any k beyond MVFGPU_K_BY_PASSES: ONE pass of the streaming kernel per 1-4 queries
// that writes every row's composite (8 bytes per row) instead of selecting, a device-wide sort of each query's composites
// (sort_topk.hip) and the formatting of the first k.  Exact, no host wait, any k (entries beyond the live rows pad); costs
// the scan + ~130 bytes of sort traffic per row, i.e. less than a second pass whenever rows are longer than that.
// Returns MVF_ERR_DEVICE with *no_room set when the buffers (16 bytes per row and query of a pass + scratch) do not fit.
int search_sorted_k(const mvfgpu_corpus* c, uint8_t metric, const void* d_queries, uint32_t nq, uint32_t k, float* d_scores,
                    uint64_t* d_indices, int32_t* d_raw, hipStream_t s, bool* no_room) {
    *no_room = false;
    size_t tmp_bytes = 0;
    if (c->n > 0) HIP_TRY(sort_composites(nullptr, &tmp_bytes, nullptr, nullptr, (size_t)c->n, (size_t)k, nullptr, s, 4, (size_t)c->n));
    tmp_bytes = std::max<size_t>(tmp_bytes, 256);
    RankAll ra{};
    ra.k_out = k;
    // Up to 1 GiB the buffers stay with the handle; beyond it (50M rows x four queries: 3.2 GB) they come from the stream-ordered
    // allocator for this search only -- memory a later shadow build or upload may need, and no host wait either way.
    constexpr size_t kKeep = 1ull << 30;
    for (int nqv = nq >= 2 ? 4 : 1;; nqv = 1) {
        const size_t bytes = (std::max<size_t>((size_t)nqv * c->n * 8, 256) + 255) & ~(size_t)255;
        if (2 * bytes + tmp_bytes > kKeep) {
            void* scratch = nullptr;
            if (hipMallocAsync(&scratch, 2 * bytes + tmp_bytes, s) == hipSuccess) {
                ra.nqv = nqv;
                ra.a = static_cast<uint64_t*>(scratch);
                ra.b = reinterpret_cast<uint64_t*>(static_cast<unsigned char*>(scratch) + bytes);
                ra.tmp = static_cast<unsigned char*>(scratch) + 2 * bytes;
                ra.tmp_bytes = tmp_bytes;
                const int rc = search_stream_path(c, metric, d_queries, nq, 16, d_scores, d_indices, d_raw, s, /*profile=*/true, nullptr,
                                                  nullptr, nullptr, 0, 0, &ra);
                const hipError_t ef = hipFreeAsync(scratch, s);
                if (rc != MVF_OK) return rc;
                HIP_TRY(ef);
                return MVF_OK;
            }
            (void)hipGetLastError();
        } else if (c->rank_a.reserve(bytes) == hipSuccess && c->rank_b.reserve(bytes) == hipSuccess &&
                   c->rank_tmp.reserve(tmp_bytes) == hipSuccess) {
            ra.nqv = nqv;
            break;
        } else {
            (void)hipGetLastError();
            c->rank_a.release();
            c->rank_b.release();
            c->rank_tmp.release();
        }
        if (nqv == 1) {
            if (c->shadow8_state == 2) {  // a shadow of what fitted took the room: the search the caller asked for comes first
                HIP_TRY(hipStreamSynchronize(s));  // nothing may still read it
                c->shadow8.release();
                c->xscale8.release();
                c->shadow8_state = -2;
                c->shadow8_rows = 0;
                continue;  // once more, one query per pass
            }
            *no_room = true;
            return fail(MVF_ERR_DEVICE, "no device memory for the whole-shard sort of a large-k search (16 bytes per row)");
        }
    }
    ra.a = static_cast<uint64_t*>(c->rank_a.p);
    ra.b = static_cast<uint64_t*>(c->rank_b.p);
    ra.tmp = c->rank_tmp.p;
    ra.tmp_bytes = c->rank_tmp.bytes;
    // the kernel's own list length: nothing is selected, so the smallest the LDS layout takes
    return search_stream_path(c, metric, d_queries, nq, 16, d_scores, d_indices, d_raw, s, /*profile=*/true, nullptr, nullptr, nullptr, 0,
                              0, &ra);
}

// Passes or the select + sort?  Measured in one process on every benchmark shape (profiles/r05_any_k.txt; round 4's library
// sort: profiles/r04_any_k.txt): the sort route wins from the second pass on nearly everywhere (10M x 768 f32, k = 16384:
// 72.5 -> 4.9 ms; four queries: 272 -> 5.9) -- a pass reads the rows again and its 1024-entry lists are heavy (doubly so four
// queries at a time), the select moves 8-byte entries only.  Passes keep what a model of the two costs gives them: one or two
// passes over a SMALL corpus (10k rows, k = 2048: 0.11 against 0.17 ms; four queries: 0.12 against 0.19), where the sort
// route's chain of a dozen short launches is the larger cost.  The model: a pass = the scan + what its lists cost (~75 us a
// full 1024-entry pass on corpora that fill the chip, a tenth of it the last, shorter one); the sort route = one dumping scan
// + ~100 us of launches (+10 per further query of the pass) + 40 bytes of traffic per row and query.
// MVF_LARGE_K forces one of them (A/B, tests).
bool large_k_by_sort(const mvfgpu_corpus* c, uint32_t nq, uint32_t k) {
    if (k > MVFGPU_K_BY_PASSES) return true;
    if (c->tune.large_k) return c->tune.large_k == 2;
    const double full = (double)(k / MVFGPU_K_PER_PASS), partial = (k % MVFGPU_K_PER_PASS) ? 1.0 : 0.0, scans = nq >= 2 ? (double)((nq + 3) / 4) : 1.0;
    const double row_bytes = std::max(128.0, (double)c->dim * elem_size(c->dtype));  // short rows scan no faster than 128-byte ones
    const double pass_us = std::max(40.0, (double)c->n * row_bytes / 5.0e6);         // 5 TB/s = 5e6 bytes per us
    const double lists_us = 75.0 * std::min(1.0, (double)c->n / 500.0e3);
    const double passes_us = scans * (nq >= 2 ? 2.0 : 1.0) * (full * (pass_us + lists_us) + partial * (pass_us + 0.15 * lists_us));
    const double sort_us = scans * (pass_us + 100.0 + 10.0 * (double)(std::min(nq, 4u) - 1)) + (double)nq * (double)c->n / 150.0e3;
    return sort_us < passes_us;
}

int check_query_args(const mvfgpu_corpus* c, uint8_t metric, const void* queries, uint8_t query_dtype,
                     uint32_t query_dim, uint32_t nq, uint32_t k, const void* out_scores, const void* out_indices) {
    if (!c) return fail(MVF_ERR_INVALID_ARGUMENT, "corpus is NULL");
    if (metric != MVF_METRIC_L2 && metric != MVF_METRIC_INNER_PRODUCT && metric != MVF_METRIC_COSINE)
        return fail(MVF_ERR_INVALID_ARGUMENT, "unsupported distance metric code " + std::to_string(metric));
    if (nq == 0) return fail(MVF_ERR_INVALID_ARGUMENT, "nq must be > 0");
    if (k == 0 || k > MVFGPU_MAX_K) return fail(MVF_ERR_INVALID_ARGUMENT, "k must be in 1..2^31");
    if (!queries || !out_scores || !out_indices) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL buffer");
    const uint8_t want = is_int_dtype(c->dtype) ? c->dtype : (uint8_t)MVF_DTYPE_FLOAT32;
    if (query_dtype != want)
        return fail(MVF_ERR_BUILD, "query data type must be Float32 for Float32/Float16 spaces and the space's own type for Int8/UInt8");
    if (query_dim != c->dim) {
        g_last_error = "Dimension mismatch: expected " + std::to_string(c->dim) + ", got " + std::to_string(query_dim);
        return MVF_ERR_DIMENSION_MISMATCH;  // reference src/errors.rs:24
    }
    return MVF_OK;
}

}  // namespace

extern "C" {

int mvfgpu_device_count(int* out_count) {
    if (!out_count) return fail(MVF_ERR_INVALID_ARGUMENT, "out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *out_count = n;
    return MVF_OK;
}

const char* mvfgpu_strerror(int status) {
    switch (status) {
    case MVF_OK: return "ok";
    case MVF_ERR_IO: return "I/O error";
    case MVF_ERR_INVALID_FORMAT: return "Invalid file format";
    case MVF_ERR_UNSUPPORTED_VERSION: return "Unsupported version";
    case MVF_ERR_SPACE_NOT_FOUND: return "Vector space not found";
    case MVF_ERR_INDEX_OUT_OF_BOUNDS: return "Index out of bounds";
    case MVF_ERR_DIMENSION_MISMATCH: return "Dimension mismatch";
    case MVF_ERR_INVALID_VECTOR_TYPE: return "Invalid vector type";
    case MVF_ERR_CORRUPTED_DATA: return "Corrupted data";
    case MVF_ERR_EXTENSION: return "Extension error";
    case MVF_ERR_BUILD: return "Build error";
    case MVF_ERR_DEVICE: return "Device error";
    case MVF_ERR_INVALID_ARGUMENT: return "Invalid argument";
    default: return "unknown status";
    }
}

const char* mvfgpu_last_error_message(void) { return g_last_error.c_str(); }

int mvfgpu_corpus_create_ex(const void* rows, uint64_t n, uint32_t dimension, uint8_t data_type, uint64_t stride_bytes,
                            int device, uint64_t index_base, const mvfgpu_upload_options* opts, mvfgpu_corpus** out) {
    if (!out) return fail(MVF_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    mvfgpu_upload_options o{};
    if (opts) {
        if (opts->struct_size < 8 || opts->struct_size > 4096) return fail(MVF_ERR_INVALID_ARGUMENT, "upload options: bad struct_size");
        std::memcpy(&o, opts, std::min<size_t>(opts->struct_size, sizeof(o)));
    }
    int rc = validate_shape(n, dimension, data_type);
    if (rc != MVF_OK) return rc;
    const uint64_t row_bytes = (uint64_t)dimension * elem_size(data_type);
    if (n > 0 && !rows) return fail(MVF_ERR_INVALID_ARGUMENT, "rows is NULL");
    if (n > 0 && stride_bytes < row_bytes)
        return fail(MVF_ERR_CORRUPTED_DATA, "Invalid stride alignment");  // cf. reference src/vectors/mem.rs:53-59
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return fail(MVF_ERR_DEVICE, "no HIP device available (libmvf_gpu has no CPU fallback)");
    }
    if (device < 0 || device >= ndev) return fail(MVF_ERR_INVALID_ARGUMENT, "device index out of range");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(MVF_ERR_DEVICE, "hipSetDevice failed");

    auto* c = new mvfgpu_corpus();
    c->device = device;
    c->n = n;
    c->dim = dimension;
    c->dtype = data_type;
    c->index_base = index_base;
    rc = init_common(c);
    if (rc == MVF_OK) rc = alloc_rows(c);
    if (rc == MVF_OK && n > 0) rc = upload_rows(c, rows, stride_bytes, o);
    if (rc != MVF_OK) {
        mvfgpu_corpus_destroy(c);
        return rc;
    }
    *out = c;
    return MVF_OK;
}

int mvfgpu_corpus_create(const void* rows, uint64_t n, uint32_t dimension, uint8_t data_type,
                         uint64_t stride_bytes, int device, uint64_t index_base, mvfgpu_corpus** out) {
    return mvfgpu_corpus_create_ex(rows, n, dimension, data_type, stride_bytes, device, index_base, nullptr, out);
}

int mvfgpu_corpus_create_synthetic(uint64_t n, uint32_t dimension, uint8_t data_type, uint64_t seed, uint64_t row0,
                                   int device, mvfgpu_corpus** out) {
    if (!out) return fail(MVF_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int rc = validate_shape(n, dimension, data_type);
    if (rc != MVF_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return fail(MVF_ERR_DEVICE, "no HIP device available (libmvf_gpu has no CPU fallback)");
    }
    if (device < 0 || device >= ndev) return fail(MVF_ERR_INVALID_ARGUMENT, "device index out of range");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(MVF_ERR_DEVICE, "hipSetDevice failed");
    auto* c = new mvfgpu_corpus();
    c->device = device;
    c->n = n;
    c->dim = dimension;
    c->dtype = data_type;
    c->index_base = row0;
    rc = init_common(c);
    if (rc == MVF_OK) rc = alloc_rows(c);
    if (rc == MVF_OK && n > 0) {
        hipError_t e = launch_synth_rows(c->d_rows, n, dimension, c->pitch, data_type, seed, row0, c->own_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->own_stream);
        if (e != hipSuccess) rc = fail(MVF_ERR_DEVICE, std::string("synthetic fill: ") + hipGetErrorString(e));
    }
    if (rc != MVF_OK) {
        mvfgpu_corpus_destroy(c);
        return rc;
    }
    *out = c;
    return MVF_OK;
}

void mvfgpu_corpus_destroy(mvfgpu_corpus* c) {
    if (!c) return;
    {
        DeviceGuard guard(c->device);
        (void)hipDeviceSynchronize();
        if (c->d_rows) (void)hipFree(c->d_rows);
        c->cand.release();
        c->bq.release();
        c->bstate.release();
        c->bcand.release();
        c->blk.release();
        c->blk_armed_cnt = nullptr;
        c->xnorm.release();
        c->repair.release();
        c->floor1.release();
        c->rank_a.release();
        c->rank_b.release();
        c->rank_tmp.release();
        c->shadow.release();
        c->xscale.release();
        c->tomb.release();
        c->ids.release();
        c->split_out.release();
        c->shadow8.release();
        c->xscale8.release();
        c->qs_stats.release();
        c->h_q.release();
        c->h_s.release();
        c->h_i.release();
        c->h_r.release();
        c->pin_q.release();
        c->pin_flag.release();
        c->done_ticket.release();
        c->pin_out.release();
        c->pin_vec.release();
        c->h_v.release();
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
        if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
        for (auto e : c->qs_redo_ev)
            if (e) (void)hipEventDestroy(e);
        if (c->qs_redo_host) (void)hipHostFree(c->qs_redo_host);
        if (c->ev_done) (void)hipEventDestroy(c->ev_done);
        for (auto& ps : c->prof)
            for (auto& e : ps.e)
                if (e) (void)hipEventDestroy(e);
    }
    delete c;
}

int mvfgpu_corpus_get_info(const mvfgpu_corpus* c, mvfgpu_corpus_info* out) {
    if (!c || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    mvfgpu_corpus_info inf{};
    inf.rows = c->n;
    inf.index_base = c->index_base;
    inf.dimension = c->dim;
    inf.pitch_bytes = c->pitch;
    inf.data_type = c->dtype;
    inf.device = c->device;
    inf.deleted_rows = c->deleted;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        inf.has_vector_ids = c->ids.p ? 1 : 0;
        inf.shadows = (uint8_t)((c->shadow8_state == 1 ? 1 : 0) | (c->shadow_state == 1 ? 2 : 0) | (c->shadow8_state == 2 ? 4 : 0));
        inf.selection_state = (uint8_t)((c->qs_disabled ? 1 : 0) | (c->bias_disabled ? 2 : 0));
        inf.device_bytes = c->tomb.bytes + c->ids.bytes + c->rows_bytes + c->cand.bytes + c->bq.bytes + c->bstate.bytes + c->bcand.bytes +
                           c->xnorm.bytes + c->repair.bytes + c->floor1.bytes + c->rank_a.bytes + c->rank_b.bytes + c->rank_tmp.bytes + c->blk.bytes + c->shadow8.bytes + c->xscale8.bytes + c->qs_stats.bytes + c->split_out.bytes +
                           c->shadow.bytes + c->xscale.bytes + c->h_q.bytes + c->h_s.bytes + c->h_i.bytes + c->h_r.bytes + c->h_v.bytes;
    }
    return copy_out_struct(out, inf);
}

int mvfgpu_corpus_read_rows(const mvfgpu_corpus* c, uint64_t first, uint64_t count, void* out_rows) {
    if (!c || (!out_rows && count)) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    if (first + count > c->n || first + count < first) {
        g_last_error = "Index out of bounds: " + std::to_string(first + count) + " >= " + std::to_string(c->n);
        return MVF_ERR_INDEX_OUT_OF_BOUNDS;  // reference src/vectors/vector_space.rs:156-161
    }
    if (count == 0) return MVF_OK;
    DeviceGuard guard(c->device);
    const uint64_t row_bytes = (uint64_t)c->dim * elem_size(c->dtype);
    HIP_TRY(hipMemcpy2D(out_rows, row_bytes, c->d_rows + first * c->pitch, c->pitch, row_bytes, count,
                        hipMemcpyDeviceToHost));
    return MVF_OK;
}

namespace {
// mvfgpu_corpus_gather_rows with c->host_mu held by the caller (mvfgpu_search_fetch fetches behind its own search)
int gather_rows_host_locked(const mvfgpu_corpus* c, const uint64_t* indices, uint64_t count, void* out_rows) {
    std::vector<uint64_t> mapped;  // with vector ids a search reports ids: translate them back to positions
    if (!c->h_ids.empty()) {
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->id_index.empty()) {
            c->id_index.reserve(c->h_ids.size());
            for (size_t r = 0; r < c->h_ids.size(); r++) c->id_index.emplace_back(c->h_ids[r], (uint32_t)r);
            std::sort(c->id_index.begin(), c->id_index.end());
        }
        mapped.resize(count);
        for (uint64_t i = 0; i < count; i++) {
            if (indices[i] == ~0ull) {
                mapped[i] = ~0ull;
                continue;
            }
            auto it = std::lower_bound(c->id_index.begin(), c->id_index.end(), std::make_pair(indices[i], 0u));
            if (it == c->id_index.end() || it->first != indices[i]) {
                g_last_error = "Index out of bounds: vector id " + std::to_string(indices[i]) + " is not in this shard";
                return MVF_ERR_INDEX_OUT_OF_BOUNDS;
            }
            mapped[i] = c->index_base + it->second;  // duplicates: the first position holding the id
        }
        indices = mapped.data();
    }
    for (uint64_t i = 0; i < count; i++) {
        const uint64_t g = indices[i];
        if (g == ~0ull) continue;  // padding of a short result list: a zero row
        if (g < c->index_base || g - c->index_base >= c->n) {
            g_last_error = "Index out of bounds: " + std::to_string(g) + " >= " + std::to_string(c->index_base + c->n);
            return MVF_ERR_INDEX_OUT_OF_BOUNDS;  // reference src/vectors/vector_space.rs:102-107
        }
    }
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(MVF_ERR_DEVICE, "hipSetDevice failed");
    const uint32_t row_bytes = c->dim * elem_size(c->dtype);
    // small fetches (the payload rows of one result list) go through pinned host memory in place, like mvfgpu_search's
    const size_t ibytes = (size_t)count * 8, obytes = (size_t)count * row_bytes;
    const bool zc_i = ibytes <= c->tune.host_zc_query, zc_o = obytes <= c->tune.host_zc_results;
    void *di, *dout;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->has_done) HIP_TRY(hipEventSynchronize(c->ev_done));
        if (zc_i) {
            HIP_TRY(c->pin_q.reserve(ibytes));
            memcpy(c->pin_q.p, indices, ibytes);
            di = c->pin_q.p;
        } else {
            HIP_TRY(c->h_i.reserve(ibytes));
            di = c->h_i.p;
        }
        if (zc_o) {
            HIP_TRY(c->pin_out.reserve(obytes));
            dout = c->pin_out.p;
        } else {
            HIP_TRY(c->h_q.reserve(obytes));
            dout = c->h_q.p;
        }
    }
    if (!zc_i) HIP_TRY(hipMemcpyAsync(di, indices, ibytes, hipMemcpyHostToDevice, c->own_stream));
    HIP_TRY(launch_gather_rows(c->d_rows, c->n, c->pitch, row_bytes, c->index_base, static_cast<const uint64_t*>(di),
                               (uint32_t)count, static_cast<unsigned char*>(dout), c->own_stream));
    if (!zc_o) HIP_TRY(hipMemcpyAsync(out_rows, dout, obytes, hipMemcpyDeviceToHost, c->own_stream));
    HIP_TRY(hipStreamSynchronize(c->own_stream));
    if (zc_o) memcpy(out_rows, dout, obytes);
    return MVF_OK;
}
}  // namespace

int mvfgpu_corpus_gather_rows(const mvfgpu_corpus* c, const uint64_t* indices, uint64_t count, void* out_rows) {
    if (!c || (count && (!indices || !out_rows))) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    if (count == 0) return MVF_OK;
    if (count > 0xFFFFFFFFull) return fail(MVF_ERR_INVALID_ARGUMENT, "too many rows in one gather");
    std::lock_guard<std::mutex> host_lk(c->host_mu);
    return gather_rows_host_locked(c, indices, count, out_rows);
}

int mvfgpu_corpus_set_tombstones(mvfgpu_corpus* c, const uint8_t* bitmap, uint64_t first_bit, uint64_t nbits) {
    if (!c) return fail(MVF_ERR_INVALID_ARGUMENT, "corpus is NULL");
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(MVF_ERR_DEVICE, "hipSetDevice failed");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipDeviceSynchronize());  // searches in flight still read the old bitmap
    if (!bitmap || nbits == 0) {
        c->tomb.release();
        c->deleted = 0;
        return MVF_OK;
    }
    if (first_bit + c->n > nbits || first_bit + c->n < first_bit)
        return fail(MVF_ERR_INVALID_ARGUMENT, "tombstone bitmap covers fewer rows than the shard holds");
    const size_t words = ((size_t)c->n + 31) / 32;
    std::vector<uint32_t> w(words + 1, 0u);
    uint64_t dead = 0;
    for (uint64_t r = 0; r < c->n; r++) {  // re-base to local rows (first_bit need not be a multiple of 8)
        const uint64_t b = first_bit + r;
        if ((bitmap[b >> 3] >> (b & 7)) & 1u) {
            w[r >> 5] |= 1u << (r & 31);
            dead++;
        }
    }
    if (dead == 0) {
        c->tomb.release();
        c->deleted = 0;
        return MVF_OK;
    }
    HIP_TRY(c->tomb.reserve((words + 1) * 4));
    HIP_TRY(hipMemcpy(c->tomb.p, w.data(), (words + 1) * 4, hipMemcpyHostToDevice));
    c->deleted = dead;
    return MVF_OK;
}

int mvfgpu_corpus_set_vector_ids(mvfgpu_corpus* c, const void* ids_le, uint64_t n) {
    if (!c) return fail(MVF_ERR_INVALID_ARGUMENT, "corpus is NULL");
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(MVF_ERR_DEVICE, "hipSetDevice failed");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipDeviceSynchronize());
    c->id_index.clear();
    if (!ids_le || n == 0) {
        c->ids.release();
        c->h_ids.clear();
        return MVF_OK;
    }
    if (n != c->n) return fail(MVF_ERR_INVALID_ARGUMENT, "vector id count differs from the shard's row count");
    c->h_ids.resize(n);
    std::memcpy(c->h_ids.data(), ids_le, (size_t)n * 8);  // the block may sit at any alignment in the mapping
    HIP_TRY(c->ids.reserve((size_t)n * 8));
    HIP_TRY(hipMemcpy(c->ids.p, c->h_ids.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    return MVF_OK;
}

namespace {
int search_device(const mvfgpu_corpus* c, uint8_t metric, const void* d_queries, uint8_t query_dtype, uint32_t query_dim, uint32_t nq,
                  uint32_t k, float* d_scores, uint64_t* d_indices, int32_t* d_raw, void* hip_stream, HostFlagReq* req);
}

int mvfgpu_search_device(const mvfgpu_corpus* c, uint8_t metric, const void* d_queries, uint8_t query_dtype,
                         uint32_t query_dim, uint32_t nq, uint32_t k, float* d_scores, uint64_t* d_indices,
                         int32_t* d_raw, void* hip_stream) {
    return search_device(c, metric, d_queries, query_dtype, query_dim, nq, k, d_scores, d_indices, d_raw, hip_stream, nullptr);
}

namespace {
// mvfgpu_search_device, and -- with `req` -- the enqueue step of the host-buffer calls (search_host)
int search_device(const mvfgpu_corpus* c, uint8_t metric, const void* d_queries, uint8_t query_dtype, uint32_t query_dim, uint32_t nq,
                  uint32_t k, float* d_scores, uint64_t* d_indices, int32_t* d_raw, void* hip_stream, HostFlagReq* req) {
    int rc = check_query_args(c, metric, d_queries, query_dtype, query_dim, nq, k, d_scores, d_indices);
    if (rc != MVF_OK) return rc;
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(MVF_ERR_DEVICE, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    std::lock_guard<std::mutex> lk(c->mu);
    c->work_gen++;
    struct ReqGuard {  // the request belongs to this call only
        const mvfgpu_corpus* c;
        ~ReqGuard() { c->flag_req = nullptr; }
    } req_guard{c};
    c->flag_req = req;
    if (req) req->gen = c->work_gen;
    // the scratch buffers are stream-ordered: a call on another stream waits for the previous one
    if (c->has_done && c->last_stream != s) HIP_TRY(hipStreamWaitEvent(s, c->ev_done, 0));
    mvfgpu_corpus::ProfSlot* wps = nullptr;  // whole-search events: every kernel of this call on the stream
    const uint64_t prof_before = c->prof_next;
    if (c->profiling) {
        wps = &c->prof[c->prof_next % mvfgpu_corpus::kProfSlots];
        for (auto& e : wps->e)
            if (!e) HIP_TRY(hipEventCreate(&e));
        wps->whole = false;
        HIP_TRY(hipEventRecord(wps->e[3], s));
    }
    // Whatever happens below, work may already sit on the stream (norms, a shadow build, scratch): the next call on
    // ANOTHER stream orders itself behind ev_done, so it is recorded on every way out.
    struct DoneGuard {
        const mvfgpu_corpus* c;
        hipStream_t s;
        ~DoneGuard() {
            if (hipEventRecord(c->ev_done, s) == hipSuccess) {
                c->has_done = true;
                c->last_stream = s;
            } else {
                (void)hipGetLastError();
            }
        }
    } done_guard{c, s};
    if (k > MVFGPU_K_PER_PASS) {  // more results than one pass selects: the whole shard ranked by a sort, or passes of the exact streaming kernel
        uint32_t scans = 1;
        bool sorted = false;
        if (large_k_by_sort(c, nq, k)) {
            bool no_room = false;
            rc = search_sorted_k(c, metric, d_queries, nq, k, d_scores, d_indices, d_raw, s, &no_room);
            sorted = rc == MVF_OK;
            if (!sorted && !(no_room && k <= MVFGPU_K_BY_PASSES)) return rc;
        }
        if (!sorted) {
            rc = search_large_k(c, metric, d_queries, nq, k, d_scores, d_indices, d_raw, s);
            if (rc != MVF_OK) return rc;
            scans = (k + MVFGPU_K_PER_PASS - 1) / MVFGPU_K_PER_PASS;
        }
        if (wps && c->prof_next == prof_before + 1) {
            HIP_TRY(hipEventRecord(wps->e[4], s));
            wps->whole = true;
            c->timing.search_flops = 2ull * nq * c->n * c->dim * scans;
            if (sorted) c->timing.scan_kernel = 8u;  // the streaming kernel as a dump + the whole-shard sort
        }
        return MVF_OK;
    }
    bool shadow_stream = false, qs_stream = false;
    if (stream_qs_wanted(c, nq, k)) qs_feedback_poll(c);  // may switch the int8 selection off
    if (stream_qs_wanted(c, nq, k)) {
        rc = ensure_norms(c, s);
        if (rc != MVF_OK) return rc;
        hipError_t e = ensure_shadow8(c, s, c->scan_path == 6);
        if (e != hipSuccess) return fail(MVF_ERR_DEVICE, std::string("int8 shadow build: ") + hipGetErrorString(e));
        qs_stream = c->shadow8_state == 1;
    }
    if (!qs_stream && stream_shadow_wanted(c, nq)) {
        hipError_t e = ensure_shadow(c, s, c->scan_path == 4);
        if (e != hipSuccess) return fail(MVF_ERR_DEVICE, std::string("shadow build: ") + hipGetErrorString(e));
        shadow_stream = c->shadow_state == 1;
    }
    rc = qs_stream                           ? search_stream_qs_path(c, metric, d_queries, nq, k, d_scores, d_indices, d_raw, s)
         : shadow_stream                     ? search_stream_shadow_path(c, metric, d_queries, nq, k, d_scores, d_indices, d_raw, s)
         : use_batched_path(c, metric, nq) ? search_batched_path(c, metric, d_queries, nq, k, d_scores, d_indices, d_raw, s)
                                           : search_stream_path(c, metric, d_queries, nq, k, d_scores, d_indices, d_raw, s);
    if (rc != MVF_OK) return rc;
    if (wps && c->prof_next == prof_before + 1) {  // the path filled this slot
        HIP_TRY(hipEventRecord(wps->e[4], s));
        wps->whole = true;
        c->timing.search_flops = 2ull * nq * c->n * c->dim;
    }
    return MVF_OK;  // ev_done: DoneGuard
}
}  // namespace

namespace {
// mvfgpu_search / mvfgpu_search_fetch.  out_vectors (nullable): [nq][k] rows of the corpus in their stored type.
int search_host(const mvfgpu_corpus* c, uint8_t metric, const void* queries, uint8_t query_dtype, uint32_t query_dim, uint32_t nq,
                uint32_t k, float* out_scores, uint64_t* out_indices, int32_t* out_raw, void* out_vectors) {
    int rc = check_query_args(c, metric, queries, query_dtype, query_dim, nq, k, out_scores, out_indices);
    if (rc != MVF_OK) return rc;
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(MVF_ERR_DEVICE, "hipSetDevice failed");
    const size_t qbytes = (size_t)nq * c->dim * (is_int_dtype(c->dtype) ? 1 : 4);
    const size_t nres = (size_t)nq * k;
    if (out_vectors && nres > 0xFFFFFFFFull) return fail(MVF_ERR_INVALID_ARGUMENT, "too many rows in one fetch");
    // Small queries / results skip the copy engine: the query is copied (by the CPU) into pinned host memory the kernels
    // read in place, and the selection kernels write the results into pinned host memory -- 10k x 128 f32, top-10:
    // 65 -> 30 us per call (profiles/r04_host_api_latency.txt); three staged hipMemcpyAsync of pageable memory cost more
    // than the search.  Larger transfers keep the device mirrors (a kernel reading megabytes over PCIe stalls its blocks).
    const size_t out_bytes = nres * 16;
    // (queries in place only where the search is short: on a large corpus every block of a pass -- a thousand and more -- would
    // stage its query over PCIe and the re-scoring waves re-read it uncached; there the pinned copy is mirrored into HBM first,
    // ~10 us in front of a search of milliseconds)
    const bool zc_q = qbytes <= c->tune.host_zc_query, zc_out = out_bytes <= c->tune.host_zc_results;
    const bool mirror_q = zc_q && c->n * (uint64_t)c->pitch > (256ull << 20);
    // The payload rows are gathered on the device BEHIND the search, on its stream, from the result indices where the
    // selection kernel left them: one submission and one wait for "the k best and their vectors" (the reference's
    // ScoredVector carries the vector, examples/similarity_search.rs:18).  A corpus that reports vector ids translates
    // them back on the host (mvfgpu_corpus_gather_rows' table), after the search: two steps, as before.
    // Only the first kv = min(k, rows) results of a query can name a row (the rest is padding): the payload staging holds
    // [nq][kv] rows -- k = 10^6 on a 10k-row corpus is 10k rows per query, not a million zero rows on the device and the host --
    // and only those rows of out_vectors ([nq][k] rows) are written.
    const uint32_t row_bytes = c->dim * elem_size(c->dtype);
    const uint32_t kv = (uint32_t)std::min<uint64_t>(k, c->n);
    const size_t nvec = (size_t)nq * kv;
    const size_t vec_bytes = nvec * row_bytes;
    const bool zc_vec = vec_bytes <= c->tune.host_zc_results;
    void *dq, *ds, *di, *dr, *dv = nullptr;
    std::lock_guard<std::mutex> host_lk(c->host_mu);
    bool fused_fetch = false;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        // wait for any in-flight user of the mirrors before (re)allocating them (nothing is in flight when the newest work
        // was seen complete through the flag: its event would only be signalled a few microseconds from now)
        if (c->has_done && c->confirmed_gen != c->work_gen) HIP_TRY(hipEventSynchronize(c->ev_done));
        fused_fetch = out_vectors && c->h_ids.empty();
        if (zc_q) {
            HIP_TRY(c->pin_q.reserve(qbytes));
            memcpy(c->pin_q.p, queries, qbytes);
            dq = c->pin_q.p;
            if (mirror_q) {
                HIP_TRY(c->h_q.reserve(qbytes));
                dq = c->h_q.p;
                HIP_TRY(hipMemcpyAsync(dq, c->pin_q.p, qbytes, hipMemcpyHostToDevice, c->own_stream));
            }
        } else {
            HIP_TRY(c->h_q.reserve(qbytes));
            dq = c->h_q.p;
            HIP_TRY(hipMemcpyAsync(dq, queries, qbytes, hipMemcpyHostToDevice, c->own_stream));
        }
        if (zc_out) {
            HIP_TRY(c->pin_out.reserve(nres * 16));
            di = c->pin_out.p;  // u64[nres] | f32[nres] | i32[nres]
            ds = static_cast<unsigned char*>(c->pin_out.p) + nres * 8;
            dr = static_cast<unsigned char*>(c->pin_out.p) + nres * 12;
        } else {
            HIP_TRY(c->h_s.reserve(nres * 4));
            HIP_TRY(c->h_i.reserve(nres * 8));
            HIP_TRY(c->h_r.reserve(nres * 4));
            ds = c->h_s.p;
            di = c->h_i.p;
            dr = c->h_r.p;
        }
        if (fused_fetch) {
            if (zc_vec) {
                HIP_TRY(c->pin_vec.reserve(vec_bytes));
                dv = c->pin_vec.p;
            } else {
                HIP_TRY(c->h_v.reserve(vec_bytes));
                dv = c->h_v.p;
            }
        }
    }
    HostFlagReq req{};
    bool want_flag = false;
    if (zc_out && c->tune.host_flag_wait && (!out_vectors || (fused_fetch && zc_vec && kv == k))) {
        std::lock_guard<std::mutex> lk(c->mu);
        if (!c->pin_flag.p) {
            HIP_TRY(c->pin_flag.reserve(64));
            *static_cast<volatile uint32_t*>(c->pin_flag.p) = 0;
            HIP_TRY(c->done_ticket.reserve(256));
            HIP_TRY(hipMemsetAsync(c->done_ticket.p, 0, 256, c->own_stream));
        }
        req.flag = static_cast<uint32_t*>(c->pin_flag.p);
        req.ticket = static_cast<uint32_t*>(c->done_ticket.p);
        req.seq = ++c->flag_seq;
        if (req.seq == 0) req.seq = ++c->flag_seq;
        req.gather_out = out_vectors && kv == k ? static_cast<unsigned char*>(dv) : nullptr;  // the select copies the payload rows as well
        want_flag = true;
    }
    rc = search_device(c, metric, dq, query_dtype, query_dim, nq, k, static_cast<float*>(ds), static_cast<uint64_t*>(di),
                       static_cast<int32_t*>(dr), c->own_stream, want_flag ? &req : nullptr);
    if (rc != MVF_OK) return rc;
    if (req.armed && (!out_vectors || req.gathered)) {
        // the final select writes req.seq behind its results (and the payload rows it copied): spin on it (bounded: a long search falls back to the stream)
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (uint32_t spins = 0;; spins++) {
            if (__atomic_load_n(req.flag, __ATOMIC_ACQUIRE) == req.seq) {
                seen = true;
                break;
            }
            if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
            __builtin_ia32_pause();
        }
        if (!seen) HIP_TRY(hipStreamSynchronize(c->own_stream));
        {
            std::lock_guard<std::mutex> lk(c->mu);
            c->confirmed_gen = req.gen;  // == work_gen unless another thread has enqueued a search meanwhile
        }
        memcpy(out_scores, ds, nres * 4);
        memcpy(out_indices, di, nres * 8);
        if (out_raw) memcpy(out_raw, dr, nres * 4);
        if (out_vectors) memcpy(out_vectors, dv, vec_bytes);
        return MVF_OK;
    }
    if (fused_fetch)  // padding entries (index UINT64_MAX) among a query's first kv results give zero rows
        HIP_TRY(launch_gather_rows(c->d_rows, c->n, c->pitch, row_bytes, c->index_base, static_cast<const uint64_t*>(di), (uint32_t)nvec,
                                   static_cast<unsigned char*>(dv), c->own_stream, k, kv));
    if (!zc_out) {
        HIP_TRY(hipMemcpyAsync(out_scores, ds, nres * 4, hipMemcpyDeviceToHost, c->own_stream));
        HIP_TRY(hipMemcpyAsync(out_indices, di, nres * 8, hipMemcpyDeviceToHost, c->own_stream));
        if (out_raw) HIP_TRY(hipMemcpyAsync(out_raw, dr, nres * 4, hipMemcpyDeviceToHost, c->own_stream));
    }
    if (fused_fetch && !zc_vec) {
        if (kv == k) HIP_TRY(hipMemcpyAsync(out_vectors, dv, vec_bytes, hipMemcpyDeviceToHost, c->own_stream));
        else HIP_TRY(hipMemcpy2DAsync(out_vectors, (size_t)k * row_bytes, dv, (size_t)kv * row_bytes, (size_t)kv * row_bytes, nq,
                                      hipMemcpyDeviceToHost, c->own_stream));
    }
    HIP_TRY(hipStreamSynchronize(c->own_stream));
    if (zc_out) {
        memcpy(out_scores, ds, nres * 4);
        memcpy(out_indices, di, nres * 8);
        if (out_raw) memcpy(out_raw, dr, nres * 4);
    }
    if (fused_fetch && zc_vec)
        for (uint32_t q = 0; q < nq; q++)
            memcpy(static_cast<unsigned char*>(out_vectors) + (size_t)q * k * row_bytes, static_cast<unsigned char*>(dv) + (size_t)q * kv * row_bytes,
                   (size_t)kv * row_bytes);
    if (out_vectors && !fused_fetch) {  // vector ids: mapped back on the host, query by query (the first kv results each)
        if (kv == k) return gather_rows_host_locked(c, out_indices, nres, out_vectors);
        for (uint32_t q = 0; q < nq; q++) {
            rc = gather_rows_host_locked(c, out_indices + (size_t)q * k, kv, static_cast<unsigned char*>(out_vectors) + (size_t)q * k * row_bytes);
            if (rc != MVF_OK) return rc;
        }
    }
    return MVF_OK;
}
}  // namespace

int mvfgpu_search(const mvfgpu_corpus* c, uint8_t metric, const void* queries, uint8_t query_dtype,
                  uint32_t query_dim, uint32_t nq, uint32_t k, float* out_scores, uint64_t* out_indices,
                  int32_t* out_raw) {
    return search_host(c, metric, queries, query_dtype, query_dim, nq, k, out_scores, out_indices, out_raw, nullptr);
}

int mvfgpu_search_fetch(const mvfgpu_corpus* c, uint8_t metric, const void* queries, uint8_t query_dtype,
                        uint32_t query_dim, uint32_t nq, uint32_t k, float* out_scores, uint64_t* out_indices,
                        int32_t* out_raw, void* out_vectors) {
    if (!out_vectors) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL buffer");
    return search_host(c, metric, queries, query_dtype, query_dim, nq, k, out_scores, out_indices, out_raw, out_vectors);
}

int mvfgpu_merge_topk_host(const float* scores, const uint64_t* indices, const int32_t* raw, uint32_t nlists,
                           uint32_t nq, uint32_t k, uint8_t metric, uint8_t data_type, float* out_scores,
                           uint64_t* out_indices, int32_t* out_raw) {
    if (!scores || !indices || !out_scores || !out_indices) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (metric != MVF_METRIC_L2 && metric != MVF_METRIC_INNER_PRODUCT && metric != MVF_METRIC_COSINE)
        return fail(MVF_ERR_INVALID_ARGUMENT, "unsupported distance metric code");
    if (nlists == 0 || nq == 0 || k == 0) return fail(MVF_ERR_INVALID_ARGUMENT, "nlists, nq and k must be > 0");
    const bool use_raw = key_is_raw(data_type, metric) && raw != nullptr;
    struct Ent {
        uint32_t key;
        uint64_t idx;
        size_t slot;
    };
    std::vector<Ent> ents;
    ents.reserve((size_t)nlists * k);
    for (uint32_t q = 0; q < nq; q++) {
        ents.clear();
        for (uint32_t l = 0; l < nlists; l++)
            for (uint32_t j = 0; j < k; j++) {
                const size_t s = ((size_t)l * nq + q) * k + j;
                if (indices[s] == ~0ull) continue;  // padding
                ents.push_back({use_raw ? key_from_raw(raw[s], metric) : key_from_score(scores[s], metric), indices[s], s});
            }
        const size_t keep = std::min<size_t>(k, ents.size());
        // ties: list order, then rank in the list (= ascending global row position for row-range shards in order)
        std::partial_sort(ents.begin(), ents.begin() + keep, ents.end(), [](const Ent& a, const Ent& b) {
            return a.key < b.key || (a.key == b.key && a.slot < b.slot);
        });
        for (uint32_t j = 0; j < k; j++) {
            const size_t o = (size_t)q * k + j;
            if (j < keep) {
                out_scores[o] = scores[ents[j].slot];
                out_indices[o] = ents[j].idx;
                if (out_raw) out_raw[o] = raw ? raw[ents[j].slot] : 0;
            } else {
                out_scores[o] = pad_score(metric);
                out_indices[o] = ~0ull;
                if (out_raw) out_raw[o] = 0;
            }
        }
    }
    return MVF_OK;
}

namespace {
int merge_topk_device_impl(const float* d_scores, const uint64_t* d_indices, const int32_t* d_raw, size_t ls_scores,
                           size_t ls_indices, size_t ls_raw, uint32_t nlists, uint32_t nq, uint32_t k, uint8_t metric,
                           uint8_t data_type, float* d_out_scores, uint64_t* d_out_indices, int32_t* d_out_raw, int device,
                           void* hip_stream) {
    if (!d_scores || !d_indices || !d_out_scores || !d_out_indices) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (metric != MVF_METRIC_L2 && metric != MVF_METRIC_INNER_PRODUCT && metric != MVF_METRIC_COSINE)
        return fail(MVF_ERR_INVALID_ARGUMENT, "unsupported distance metric code");
    if (nlists == 0 || nq == 0 || k == 0) return fail(MVF_ERR_INVALID_ARGUMENT, "nlists, nq and k must be > 0");
    const uint64_t total = (uint64_t)nlists * k;
    if (total > 0xFFFFFFFFull) return fail(MVF_ERR_INVALID_ARGUMENT, "nlists * k exceeds 2^32 - 1");
    const bool large = total > kMergeMaxEntries;  // beyond one block's LDS: a device-wide sort per query
    const uint32_t P = large ? 0u : next_pow2(std::max(2u, nlists * k));
    DeviceGuard guard(device);
    if (!guard.ok) return fail(MVF_ERR_DEVICE, "hipSetDevice failed");
    ShardMergeParams p{};
    p.scores = d_scores;
    p.indices = d_indices;
    p.raw = d_raw;
    p.ls_scores = ls_scores;
    p.ls_indices = ls_indices;
    p.ls_raw = ls_raw;
    p.nlists = nlists;
    p.nq = nq;
    p.k = k;
    p.P = P;
    p.metric = metric;
    p.dtype = data_type;
    p.out_scores = d_out_scores;
    p.out_indices = d_out_indices;
    p.out_raw = d_out_raw;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (!large) {
        HIP_TRY(launch_merge_shards(p, s));
        return MVF_OK;
    }
    // the sort's buffers live for this call only, in stream order (this entry point has no handle to keep them in)
    size_t tmp_bytes = 0;
    HIP_TRY(sort_composites(nullptr, &tmp_bytes, nullptr, nullptr, (size_t)total, (size_t)p.k, nullptr, s));
    tmp_bytes = std::max<size_t>(tmp_bytes, 256);
    const size_t cb = ((size_t)total * 8 + 255) & ~(size_t)255;
    void* scratch = nullptr;
    HIP_TRY(hipMallocAsync(&scratch, 2 * cb + tmp_bytes, s));
    uint64_t *a = static_cast<uint64_t*>(scratch), *b = reinterpret_cast<uint64_t*>(static_cast<unsigned char*>(scratch) + cb);
    void* tmp = static_cast<unsigned char*>(scratch) + 2 * cb;
    hipError_t e = hipSuccess;
    for (uint32_t q = 0; q < nq && e == hipSuccess; q++) {
        uint64_t* sorted = a;
        size_t tb = tmp_bytes;
        e = launch_merge_build(p, q, a, s);
        if (e == hipSuccess) e = sort_composites(tmp, &tb, a, b, (size_t)total, (size_t)p.k, &sorted, s);
        if (e == hipSuccess) e = launch_merge_write(p, q, sorted, s);
    }
    const hipError_t ef = hipFreeAsync(scratch, s);
    HIP_TRY(e);
    HIP_TRY(ef);
    return MVF_OK;
}
}  // namespace

int mvfgpu_merge_topk_device(const float* d_scores, const uint64_t* d_indices, const int32_t* d_raw, uint32_t nlists,
                             uint32_t nq, uint32_t k, uint8_t metric, uint8_t data_type, float* d_out_scores,
                             uint64_t* d_out_indices, int32_t* d_out_raw, int device, void* hip_stream) {
    const size_t ls = (size_t)nq * k;
    return merge_topk_device_impl(d_scores, d_indices, d_raw, ls, ls, ls, nlists, nq, k, metric, data_type, d_out_scores,
                                  d_out_indices, d_out_raw, device, hip_stream);
}

int mvfgpu_merge_topk_packed_device(const void* d_packed, uint32_t nlists, uint32_t nq, uint32_t k, uint8_t metric,
                                    uint8_t data_type, float* d_out_scores, uint64_t* d_out_indices, int32_t* d_out_raw,
                                    int device, void* hip_stream) {
    if (!d_packed) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL buffer");
    if ((reinterpret_cast<uintptr_t>(d_packed) & 7u) != 0) return fail(MVF_ERR_INVALID_ARGUMENT, "packed lists must be 8-byte aligned");
    // one list = { u64 indices[nq k]; f32 scores[nq k]; i32 raw[nq k] } = 16 nq k bytes (MVFGPU_PACKED_LIST_BYTES)
    const size_t n = (size_t)nq * k;
    const unsigned char* b = static_cast<const unsigned char*>(d_packed);
    return merge_topk_device_impl(reinterpret_cast<const float*>(b + 8 * n), reinterpret_cast<const uint64_t*>(b),
                                  reinterpret_cast<const int32_t*>(b + 12 * n), 4 * n, 2 * n, 4 * n, nlists, nq, k, metric,
                                  data_type, d_out_scores, d_out_indices, d_out_raw, device, hip_stream);
}

int mvfgpu_synth_queries_device(void* d_queries, uint32_t nq, uint32_t dimension, uint8_t data_type, uint64_t seed,
                                int device, void* hip_stream) {
    if (!d_queries) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (elem_size(data_type) == 0) return fail(MVF_ERR_BUILD, "Unsupported vector data type");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(MVF_ERR_DEVICE, "hipSetDevice failed");
    const uint8_t qd = is_int_dtype(data_type) ? data_type : (uint8_t)MVF_DTYPE_FLOAT32;
    HIP_TRY(launch_synth_packed(d_queries, (uint64_t)nq * dimension, qd, seed, static_cast<hipStream_t>(hip_stream)));
    return MVF_OK;
}

int mvfgpu_set_profiling(mvfgpu_corpus* c, int enabled) {
    if (!c) return fail(MVF_ERR_INVALID_ARGUMENT, "corpus is NULL");
    std::lock_guard<std::mutex> lk(c->mu);
    c->profiling = enabled != 0;
    c->prof_next = 0;
    c->timing = mvfgpu_timing{};
    return MVF_OK;
}

int mvfgpu_last_timing(const mvfgpu_corpus* c, mvfgpu_timing* out) {
    if (!c || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lk(c->mu);
    mvfgpu_timing tm = c->timing;
    if (c->last_redo_cnt) {  // the newest batched search's repair count (waits for the handle's last search)
        DeviceGuard guard(c->device);
        if (c->has_done) HIP_TRY(hipEventSynchronize(c->ev_done));
        uint32_t n = 0;
        HIP_TRY(hipMemcpy(&n, c->last_redo_cnt, 4, hipMemcpyDeviceToHost));
        tm.repaired_queries = n;
    }
    if (c->prof_next > 0) {
        DeviceGuard guard(c->device);
        const uint64_t newest = c->prof_next - 1;
        const uint64_t oldest = c->prof_next > mvfgpu_corpus::kProfSlots ? c->prof_next - mvfgpu_corpus::kProfSlots : 0;
        double ssum = 0, lsum = 0, wsum = 0;
        uint32_t cnt = 0, wcnt = 0;
        for (uint64_t i = newest + 1; i-- > oldest;) {
            const auto& ps = c->prof[i % mvfgpu_corpus::kProfSlots];
            HIP_TRY(hipEventSynchronize(ps.whole ? ps.e[4] : ps.e[2]));
            float a = 0, b = 0, w = 0;
            if (ps.scanned) {
                HIP_TRY(hipEventElapsedTime(&a, ps.e[0], ps.e[1]));
                HIP_TRY(hipEventElapsedTime(&b, ps.e[1], ps.e[2]));
            }
            if (ps.whole) {
                HIP_TRY(hipEventElapsedTime(&w, ps.e[3], ps.e[4]));
                wsum += w;
                wcnt++;
            }
            if (i == newest) {
                tm.scan_ms = a;
                tm.select_ms = b;
                tm.total_ms = a + b;
                tm.search_ms = w;
            }
            ssum += a;
            lsum += b;
            cnt++;
        }
        tm.samples = cnt;
        tm.scan_ms_avg = cnt ? (float)(ssum / cnt) : 0.f;
        tm.select_ms_avg = cnt ? (float)(lsum / cnt) : 0.f;
        tm.search_ms_avg = wcnt ? (float)(wsum / wcnt) : 0.f;
    }
    return copy_out_struct(out, tm);
}

int mvfgpu_set_scan_path(mvfgpu_corpus* c, int path) {
    if (!c) return fail(MVF_ERR_INVALID_ARGUMENT, "corpus is NULL");
    if (path < 0 || path > 6) return fail(MVF_ERR_INVALID_ARGUMENT, "path must be 0..6");
    std::lock_guard<std::mutex> lk(c->mu);
    c->scan_path = path;
    return MVF_OK;
}

int mvfgpu_corpus_reload_tuning(mvfgpu_corpus* c) {
    if (!c) return fail(MVF_ERR_INVALID_ARGUMENT, "corpus is NULL");
    // host_mu first (search_host's order): the host-buffer searches read c->tune under it, without c->mu.  Device-pointer
    // searches (mvfgpu_search_device) read the switches unlocked: do not call this beside them (include/mvf_gpu.h).
    std::lock_guard<std::mutex> host_lk(c->host_mu);
    std::lock_guard<std::mutex> lk(c->mu);
    c->tune = mvf::read_tuning();
    choose_group(c->V, 1, &c->G, &c->J, c->tune.k1_g);
    return MVF_OK;
}

uint32_t mvfgpu_abi_version(void) { return MVFGPU_ABI_VERSION; }

int mvfgpu_selftest_feedback(const uint32_t* samples, uint32_t n_samples, uint32_t* out_state) {
    if ((!samples && n_samples) || !out_state) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL buffer");
    uint32_t seen = 0, redone = 0;
    bool bias_off = false, qs_off = false;
    for (uint32_t i = 0; i < n_samples; i++)
        feedback_consume(seen, redone, bias_off, qs_off, samples[4 * i], samples[4 * i + 1], samples[4 * i + 2] != 0, samples[4 * i + 3] != 0);
    out_state[0] = seen;
    out_state[1] = redone;
    out_state[2] = bias_off;
    out_state[3] = qs_off;
    return MVF_OK;
}

int mvfgpu_selftest_schedule(uint64_t rows, uint32_t nq, uint32_t k, int int8_selection, uint64_t* out_bounds, uint32_t max_bounds,
                             uint32_t* out_n_bounds, uint32_t* out_growth, uint32_t* out_refined_mask) {
    if (!out_bounds || !out_n_bounds || !out_growth || !out_refined_mask) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (rows == 0 || nq == 0 || k == 0 || k > MVFGPU_K_PER_PASS) return fail(MVF_ERR_INVALID_ARGUMENT, "empty corpus / batch, or k beyond one pass");
    const Tuning t{};  // the DEFAULT tuning, not the environment
    const uint32_t cap = int8_selection ? kBatchCapQS : kBatchCap;
    const uint32_t g = k2_growth_for(t, nq, k, cap);
    const std::vector<uint64_t> bounds = k2_phase_bounds(rows, cap, g);
    if (bounds.size() > max_bounds) return fail(MVF_ERR_INVALID_ARGUMENT, "more phases than the buffer holds");
    uint32_t mask = 0;
    for (size_t i = 0; i < bounds.size(); i++) {
        out_bounds[i] = bounds[i];
        if (int8_selection && i < 32 && k2_refine_before(t, bounds, i, nq)) mask |= 1u << i;
    }
    *out_n_bounds = (uint32_t)bounds.size();
    *out_growth = g;
    *out_refined_mask = mask;
    return MVF_OK;
}

int mvfgpu_selftest_route(uint64_t rows, uint32_t dimension, uint8_t data_type, uint8_t metric, uint32_t nq, uint32_t k, uint32_t* out_route) {
    if (!out_route) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (elem_size(data_type) == 0 || dimension == 0 || nq == 0 || k == 0 || k > MVFGPU_MAX_K)
        return fail(MVF_ERR_INVALID_ARGUMENT, "unsupported type or empty dimension / batch / k");
    if (metric != MVF_METRIC_L2 && metric != MVF_METRIC_INNER_PRODUCT && metric != MVF_METRIC_COSINE)
        return fail(MVF_ERR_INVALID_ARGUMENT, "unsupported distance metric code");
    // a handle that owns nothing on a device: the route is a function of these fields and the DEFAULT tuning (not the environment)
    std::unique_ptr<mvfgpu_corpus> c(new mvfgpu_corpus());
    c->n = rows;
    c->dim = dimension;
    c->dtype = data_type;
    c->pitch = (dimension * elem_size(data_type) + 15u) & ~15u;
    c->V = c->pitch / 16;
    c->tune = Tuning{};
    if (k > MVFGPU_K_PER_PASS) *out_route = large_k_by_sort(c.get(), nq, k) ? 3u : 2u;
    else *out_route = use_batched_path(c.get(), metric, nq) ? 1u : 0u;
    return MVF_OK;
}

}  // extern "C"
