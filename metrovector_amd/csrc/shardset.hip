// shardset.hip — the multi-GPU form of the search behind the C ABI (include/mvf_gpu.h, mvfgpu_shardset_*).
//
// the corpus is sharded by contiguous row range, one mvfgpu_corpus per GPU; a search runs on every
// shard with global indices and ends in ONE exchange step -- an RCCL all-gather of the per-shard top-k lists over xGMI
// -- followed by the (key, list, rank) merge.  This file is the single-process form of it (a Rust or C host holds all
// GPUs of the node): ncclCommInitAll over the shards' devices, one HIP stream per device, the lists travel PACKED
// ({u64 indices | f32 scores | i32 raw} = 16 bytes per result, MVFGPU_PACKED_LIST_BYTES) so the exchange is one
// collective, not three.  The one-process-per-GPU form (torch.distributed) is metrovector_amd/sharded.py; both call the
// same mvfgpu_search_device / mvfgpu_merge_topk_packed_device.
//
// RCCL is loaded lazily (dlopen) the first time a shard set is created: a process that never shards does not pay for
// the library, and libmvf_gpu.so keeps loading where RCCL is absent.  Shards that SHARE a device (a rehearsal of the
// protocol on fewer GPUs than shards; RCCL refuses duplicate devices) exchange their lists with device-to-device
// copies instead -- same packed layout, same merge.

#include "../../include/mvf_gpu.h"

#include "internal.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

using mvf::set_fail;

namespace {

#define SS_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) return set_fail(MVF_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl* rccl() {  // loaded once; nullptr-lib + reason when unavailable
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy already in the process (torch ships its own librccl.so) is preferred over loading a second one
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (int pass = 0; pass < 2 && !r.lib; pass++)
            for (const char* n : names)
                if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0)))) break;
        if (!r.lib) {
            r.why = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(r.lib, n);
            if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + n;
            return p;
        };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!r.why.empty()) r.lib = nullptr;
    });
    return &r;
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int dev = 0;
    hipError_t reserve(int device, size_t need) {
        if (need <= bytes) return hipSuccess;
        (void)hipSetDevice(device);
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        dev = device;
        hipError_t e = hipMalloc(&p, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release() {
        if (!p) return;
        (void)hipSetDevice(dev);
        (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

}  // namespace

struct mvfgpu_shardset {
    std::vector<mvfgpu_corpus*> shards;  // borrowed, ascending row-range order
    std::vector<int> dev;
    std::vector<hipStream_t> st;
    std::vector<hipEvent_t> ev;          // end of the shard's local search (ordering only)
    std::vector<hipEvent_t> ev_t0, ev_t1;  // timing: before the query upload / after the local search, on the shard's stream
    hipEvent_t ev_merged = nullptr;      // timing: end of the merge on shard 0's stream
    std::vector<ncclComm_t> comms;       // one per shard when RCCL is in use
    bool use_rccl = false;
    uint32_t dim = 0;
    uint8_t dtype = 0;
    uint64_t rows = 0;
    std::mutex mu;                       // one search at a time per set (the gather buffers are per set)
    std::vector<DevBuf> d_q, d_gather;   // per shard: queries; [n_shards] packed lists (its own list at slot s)
    DevBuf d_out;                        // shard 0's device: the merged list, packed like the shards' (u64 | f32 | i32)
    // Small queries / results skip the copy engine, as in mvfgpu_search (api.hip): every shard's kernels read the queries
    // from ONE pinned host buffer (portable: mapped for every device) and the merge writes into pinned host memory.
    void* pin_q = nullptr;
    void* pin_out = nullptr;
    size_t pin_q_bytes = 0, pin_out_bytes = 0, zc_query = 0, zc_results = 0;
    hipError_t reserve_pinned(void** p, size_t* have, size_t need) {
        if (need <= *have) return hipSuccess;
        if (*p) (void)hipHostFree(*p);
        *p = nullptr;
        *have = 0;
        need = (need + 4095) & ~(size_t)4095;
        hipError_t e = hipHostMalloc(p, need, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);  // coherent like api.hip's PinBuf: kernels on every device read / write it in place
        if (e == hipSuccess) *have = need;
        return e;
    }
    mvfgpu_shardset_timing tm{};

    // One PERSISTENT host thread per shard beyond the first (the calling thread drives shard 0): a batched search is
    // some fifty launches, ~0.3 ms of host time per shard -- issued from one thread the eighth GPU of a node would start
    // 2 ms after the first -- and a thread spawned per search costs ~50 us each on a latency-bound exchange.
    struct Job {
        uint8_t metric = 0, query_dtype = 0;
        uint32_t query_dim = 0, nq = 0, k = 0;
        const void* queries = nullptr;
        const void* queries_in_place = nullptr;  // pinned host copy every shard reads directly (small batches), or NULL
        size_t qbytes = 0, list_bytes = 0, nres = 0;
    } job;
    std::vector<std::thread> workers;
    std::mutex wmu;
    std::condition_variable wcv, dcv;
    uint64_t gen = 0;
    int pending = 0;
    bool stop = false;
    std::vector<int> rcs;
    std::vector<std::string> msgs;

    void run_shard(int s);
    void worker(int s);
};

// Shard s's part of a search: query upload + local search, both enqueued on the shard's stream; the packed list lands
// straight in slot s of the shard's own gather buffer (in-place all-gather).
void mvfgpu_shardset::run_shard(int s) {
    rcs[s] = MVF_OK;
    if (hipSetDevice(dev[s]) != hipSuccess) {
        rcs[s] = MVF_ERR_DEVICE;
        msgs[s] = "hipSetDevice failed";
        return;
    }
    unsigned char* slot = static_cast<unsigned char*>(d_gather[s].p) + job.list_bytes * s;
    hipError_t e = hipEventRecord(ev_t0[s], st[s]);
    const void* dq = job.queries_in_place;
    if (!dq) {
        dq = d_q[s].p;
        if (e == hipSuccess) e = hipMemcpyAsync(d_q[s].p, job.queries, job.qbytes, hipMemcpyHostToDevice, st[s]);
    }
    if (e != hipSuccess) {
        rcs[s] = MVF_ERR_DEVICE;
        msgs[s] = std::string("query upload: ") + hipGetErrorString(e);
        return;
    }
    rcs[s] = mvfgpu_search_device(shards[s], job.metric, dq, job.query_dtype, job.query_dim, job.nq, job.k,
                                  reinterpret_cast<float*>(slot + 8 * job.nres), reinterpret_cast<uint64_t*>(slot),
                                  reinterpret_cast<int32_t*>(slot + 12 * job.nres), st[s]);
    if (rcs[s] != MVF_OK) {
        msgs[s] = mvfgpu_last_error_message();
        return;
    }
    e = hipEventRecord(ev_t1[s], st[s]);
    if (e == hipSuccess) e = hipEventRecord(ev[s], st[s]);
    if (e != hipSuccess) {
        rcs[s] = MVF_ERR_DEVICE;
        msgs[s] = std::string("hipEventRecord: ") + hipGetErrorString(e);
    }
}

void mvfgpu_shardset::worker(int s) {
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(wmu);
            wcv.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
        }
        run_shard(s);
        {
            std::lock_guard<std::mutex> lk(wmu);
            if (--pending == 0) dcv.notify_one();
        }
    }
}

extern "C" {

int mvfgpu_shardset_create(mvfgpu_corpus* const* shards, int n_shards, mvfgpu_shardset** out) {
    if (!out) return set_fail(MVF_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!shards || n_shards < 1 || n_shards > 64) return set_fail(MVF_ERR_INVALID_ARGUMENT, "n_shards must be 1..64");
    auto* ss = new mvfgpu_shardset();
    uint64_t prev_end = 0;
    for (int s = 0; s < n_shards; s++) {
        mvfgpu_corpus_info inf;
        MVFGPU_INIT(inf);
        if (!shards[s] || mvfgpu_corpus_get_info(shards[s], &inf) != MVF_OK) {
            delete ss;
            return set_fail(MVF_ERR_INVALID_ARGUMENT, "shard " + std::to_string(s) + " is not a corpus handle");
        }
        if (s == 0) {
            ss->dim = inf.dimension;
            ss->dtype = inf.data_type;
        } else if (inf.dimension != ss->dim || inf.data_type != ss->dtype) {
            delete ss;
            return set_fail(MVF_ERR_DIMENSION_MISMATCH, "Dimension mismatch: expected " + std::to_string(ss->dim) + ", got " +
                                                            std::to_string(inf.dimension) + " (shards of one space share dimension and data type)");
        }
        if (s > 0 && inf.index_base < prev_end) {
            delete ss;
            return set_fail(MVF_ERR_INVALID_ARGUMENT, "shards must come in ascending, non-overlapping row-range order (the merge breaks ties by shard order)");
        }
        prev_end = inf.index_base + inf.rows;
        ss->rows += inf.rows;
        ss->shards.push_back(shards[s]);
        ss->dev.push_back(inf.device);
    }
    ss->st.assign(n_shards, nullptr);
    ss->ev.assign(n_shards, nullptr);
    ss->ev_t0.assign(n_shards, nullptr);
    ss->ev_t1.assign(n_shards, nullptr);
    ss->rcs.assign(n_shards, MVF_OK);
    ss->msgs.assign(n_shards, std::string());
    ss->d_q.resize(n_shards);
    {
        const mvf::Tuning t = mvf::read_tuning();  // MVF_HOST_ZC_QUERY / MVF_HOST_ZC_RESULTS, as a corpus handle reads them
        ss->zc_query = t.host_zc_query;
        ss->zc_results = t.host_zc_results;
    }
    ss->d_gather.resize(n_shards);
    int rc = MVF_OK;
    for (int s = 0; s < n_shards && rc == MVF_OK; s++) {
        if (hipSetDevice(ss->dev[s]) != hipSuccess || hipStreamCreateWithFlags(&ss->st[s], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ss->ev[s], hipEventDisableTiming) != hipSuccess || hipEventCreate(&ss->ev_t0[s]) != hipSuccess ||
            hipEventCreate(&ss->ev_t1[s]) != hipSuccess || (s == 0 && hipEventCreate(&ss->ev_merged) != hipSuccess))
            rc = set_fail(MVF_ERR_DEVICE, "stream / event creation failed on device " + std::to_string(ss->dev[s]));
    }
    // RCCL over the shards' devices -- also for a single shard (a 1-rank communicator: the exchange step is then the
    // same code path a node runs).  Shards sharing a device cannot form a communicator: device-to-device copies instead.
    const bool distinct = std::set<int>(ss->dev.begin(), ss->dev.end()).size() == ss->dev.size();
    const char* off = getenv("MVF_SHARDSET_NO_RCCL");
    if (rc == MVF_OK && distinct && !(off && atoi(off) != 0)) {
        Rccl* r = rccl();
        if (!r->lib) {
            rc = set_fail(MVF_ERR_DEVICE, "RCCL is required for a shard set over distinct devices: " + r->why);
        } else {
            ss->comms.assign(n_shards, nullptr);
            ncclResult_t e = r->CommInitAll(ss->comms.data(), n_shards, ss->dev.data());
            if (e != ncclSuccess) {
                ss->comms.clear();
                rc = set_fail(MVF_ERR_DEVICE, std::string("ncclCommInitAll: ") + r->GetErrorString(e));
            } else {
                ss->use_rccl = true;
            }
        }
    }
    if (rc != MVF_OK) {
        mvfgpu_shardset_destroy(ss);
        return rc;
    }
    for (int s = 1; s < n_shards; s++) ss->workers.emplace_back(&mvfgpu_shardset::worker, ss, s);
    *out = ss;
    return MVF_OK;
}

void mvfgpu_shardset_destroy(mvfgpu_shardset* ss) {
    if (!ss) return;
    {
        std::lock_guard<std::mutex> lk(ss->wmu);
        ss->stop = true;
    }
    ss->wcv.notify_all();
    for (auto& t : ss->workers) t.join();
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t s = 0; s < ss->shards.size(); s++) {
        (void)hipSetDevice(ss->dev[s]);
        if (ss->st[s]) (void)hipStreamSynchronize(ss->st[s]);
    }
    if (ss->use_rccl)
        for (auto c : ss->comms)
            if (c) (void)rccl()->CommDestroy(c);
    for (size_t s = 0; s < ss->shards.size(); s++) {
        (void)hipSetDevice(ss->dev[s]);
        if (ss->st[s]) (void)hipStreamDestroy(ss->st[s]);
        if (ss->ev[s]) (void)hipEventDestroy(ss->ev[s]);
        if (ss->ev_t0[s]) (void)hipEventDestroy(ss->ev_t0[s]);
        if (ss->ev_t1[s]) (void)hipEventDestroy(ss->ev_t1[s]);
        if (s == 0 && ss->ev_merged) (void)hipEventDestroy(ss->ev_merged);
        ss->d_q[s].release();
        ss->d_gather[s].release();
    }
    ss->d_out.release();
    if (ss->pin_q) (void)hipHostFree(ss->pin_q);
    if (ss->pin_out) (void)hipHostFree(ss->pin_out);
    if (prev >= 0) (void)hipSetDevice(prev);
    delete ss;
}

int mvfgpu_shardset_get_info(const mvfgpu_shardset* ss, mvfgpu_shardset_info* out) {
    if (!ss || !out) return set_fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    mvfgpu_shardset_info inf{};
    inf.n_shards = (uint32_t)ss->shards.size();
    inf.rccl_ranks = ss->use_rccl ? (uint32_t)ss->comms.size() : 0u;
    inf.dimension = ss->dim;
    inf.data_type = ss->dtype;
    inf.rows = ss->rows;
    return mvf::copy_out_struct(out, inf);
}

int mvfgpu_shardset_last_timing(const mvfgpu_shardset* ss, mvfgpu_shardset_timing* out) {
    if (!ss || !out) return set_fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    return mvf::copy_out_struct(out, ss->tm);
}

int mvfgpu_shardset_search(mvfgpu_shardset* ss, uint8_t metric, const void* queries, uint8_t query_dtype, uint32_t query_dim,
                           uint32_t nq, uint32_t k, float* out_scores, uint64_t* out_indices, int32_t* out_raw) {
    // every argument is checked before anything is reserved, woken or enqueued (the per-shard searches repeat the checks
    // against their own handle)
    if (!ss) return set_fail(MVF_ERR_INVALID_ARGUMENT, "shard set is NULL");
    if (!queries || !out_scores || !out_indices) return set_fail(MVF_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (nq == 0 || k == 0 || k > MVFGPU_MAX_K) return set_fail(MVF_ERR_INVALID_ARGUMENT, "nq must be > 0 and k in 1..2^31");
    if (metric != MVF_METRIC_L2 && metric != MVF_METRIC_INNER_PRODUCT && metric != MVF_METRIC_COSINE)
        return set_fail(MVF_ERR_INVALID_ARGUMENT, "unsupported distance metric code");
    const bool int_space = ss->dtype == MVF_DTYPE_INT8 || ss->dtype == MVF_DTYPE_UINT8;
    const uint8_t want_q = int_space ? ss->dtype : (uint8_t)MVF_DTYPE_FLOAT32;
    if (query_dtype != want_q)
        return set_fail(MVF_ERR_BUILD, "Unsupported query data type for this space (Float32 queries for Float32/Float16 spaces, the space's own type for Int8/UInt8)");
    if (query_dim != ss->dim)
        return set_fail(MVF_ERR_DIMENSION_MISMATCH, "Dimension mismatch: expected " + std::to_string(ss->dim) + ", got " + std::to_string(query_dim));
    const int S = (int)ss->shards.size();
    if ((uint64_t)S * k > 0xFFFFFFFFull) return set_fail(MVF_ERR_INVALID_ARGUMENT, "n_shards * k exceeds 2^32 - 1 (the cross-shard merge's capacity)");
    std::lock_guard<std::mutex> lk(ss->mu);
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Restore {
        int d;
        ~Restore() {
            if (d >= 0) (void)hipSetDevice(d);
        }
    } restore{prev};

    const size_t nres = (size_t)nq * k, list_bytes = MVFGPU_PACKED_LIST_BYTES(nq, k);
    const size_t qbytes = (size_t)nq * query_dim * (int_space ? 1u : 4u);
    const bool zc_q = qbytes <= ss->zc_query, zc_out = nres * 16 <= ss->zc_results;
    for (int s = 0; s < S; s++) {
        if (!zc_q) SS_HIP(ss->d_q[s].reserve(ss->dev[s], qbytes));
        SS_HIP(ss->d_gather[s].reserve(ss->dev[s], list_bytes * S));
    }
    if (zc_q) {
        SS_HIP(ss->reserve_pinned(&ss->pin_q, &ss->pin_q_bytes, qbytes));
        memcpy(ss->pin_q, queries, qbytes);
    }
    if (zc_out) SS_HIP(ss->reserve_pinned(&ss->pin_out, &ss->pin_out_bytes, nres * 16));
    else SS_HIP(ss->d_out.reserve(ss->dev[0], list_bytes));

    // ---- per-shard searches, concurrently: the calling thread drives shard 0, a persistent worker each of the others
    const auto t0 = std::chrono::steady_clock::now();
    ss->job.metric = metric;
    ss->job.query_dtype = query_dtype;
    ss->job.query_dim = query_dim;
    ss->job.nq = nq;
    ss->job.k = k;
    ss->job.queries = queries;
    ss->job.queries_in_place = zc_q ? ss->pin_q : nullptr;
    ss->job.qbytes = qbytes;
    ss->job.list_bytes = list_bytes;
    ss->job.nres = nres;
    if (S > 1) {
        {
            std::lock_guard<std::mutex> wl(ss->wmu);
            ss->pending = S - 1;
            ss->gen++;
        }
        ss->wcv.notify_all();
    }
    ss->run_shard(0);
    if (S > 1) {
        std::unique_lock<std::mutex> wl(ss->wmu);
        ss->dcv.wait(wl, [&] { return ss->pending == 0; });
    }
    int failed = -1;
    for (int s = 0; s < S && failed < 0; s++)
        if (ss->rcs[s] != MVF_OK) failed = s;
    if (failed >= 0) {
        for (int s = 0; s < S; s++) {  // what the other shards enqueued still reads the caller's queries
            (void)hipSetDevice(ss->dev[s]);
            (void)hipStreamSynchronize(ss->st[s]);
        }
        return set_fail(ss->rcs[failed], "shard " + std::to_string(failed) + ": " + ss->msgs[failed]);
    }

    // ---- the exchange step: ONE grouped all-gather of the packed lists (stream-ordered behind each shard's search)
    if (ss->use_rccl) {
        Rccl* r = rccl();
        ncclResult_t e = r->GroupStart();
        for (int s = 0; s < S && e == ncclSuccess; s++) {
            unsigned char* g = static_cast<unsigned char*>(ss->d_gather[s].p);
            e = r->AllGather(g + list_bytes * s, g, list_bytes, ncclInt8, ss->comms[s], ss->st[s]);
        }
        ncclResult_t e2 = r->GroupEnd();
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) return set_fail(MVF_ERR_DEVICE, std::string("ncclAllGather: ") + r->GetErrorString(e));
    } else {
        // shards share a device (rehearsal): gather to shard 0's buffer with device-to-device copies
        (void)hipSetDevice(ss->dev[0]);
        for (int s = 1; s < S; s++) {
            SS_HIP(hipStreamWaitEvent(ss->st[0], ss->ev[s], 0));
            SS_HIP(hipMemcpyPeerAsync(static_cast<unsigned char*>(ss->d_gather[0].p) + list_bytes * s, ss->dev[0],
                                      static_cast<unsigned char*>(ss->d_gather[s].p) + list_bytes * s, ss->dev[s], list_bytes,
                                      ss->st[0]));
        }
    }
    // ---- merge on shard 0's device into one more packed list ({u64 indices | f32 scores | i32 raw}: the u64 array first,
    // so every array is aligned to its element whatever nq * k is), results to the host
    (void)hipSetDevice(ss->dev[0]);
    unsigned char* ob = static_cast<unsigned char*>(zc_out ? ss->pin_out : ss->d_out.p);
    uint64_t* mi = reinterpret_cast<uint64_t*>(ob);
    float* ms = reinterpret_cast<float*>(ob + 8 * nres);
    int32_t* mr = reinterpret_cast<int32_t*>(ob + 12 * nres);
    int rc = mvfgpu_merge_topk_packed_device(ss->d_gather[0].p, (uint32_t)S, nq, k, metric, ss->dtype, ms, mi, mr, ss->dev[0], ss->st[0]);
    if (rc != MVF_OK) return rc;
    SS_HIP(hipEventRecord(ss->ev_merged, ss->st[0]));
    const auto t1 = std::chrono::steady_clock::now();  // everything is enqueued (a copy to pageable host memory blocks)
    if (!zc_out) {
        SS_HIP(hipMemcpyAsync(out_scores, ms, nres * 4, hipMemcpyDeviceToHost, ss->st[0]));
        SS_HIP(hipMemcpyAsync(out_indices, mi, nres * 8, hipMemcpyDeviceToHost, ss->st[0]));
        if (out_raw) SS_HIP(hipMemcpyAsync(out_raw, mr, nres * 4, hipMemcpyDeviceToHost, ss->st[0]));
    }
    for (int s = 0; s < S; s++) {  // every rank's part of the collective has to finish before the buffers are reused
        (void)hipSetDevice(ss->dev[s]);
        SS_HIP(hipStreamSynchronize(ss->st[s]));
    }
    if (zc_out) {
        memcpy(out_scores, ms, nres * 4);
        memcpy(out_indices, mi, nres * 8);
        if (out_raw) memcpy(out_raw, mr, nres * 4);
    }
    const auto t2 = std::chrono::steady_clock::now();
    mvfgpu_shardset_timing& tm = ss->tm;
    tm.n_shards = (uint32_t)S;
    tm.total_ms = std::chrono::duration<float, std::milli>(t2 - t0).count();
    tm.enqueue_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    tm.search_ms = 0.f;
    for (int s = 0; s < S; s++) {
        float ms_s = 0.f;
        (void)hipSetDevice(ss->dev[s]);
        if (hipEventElapsedTime(&ms_s, ss->ev_t0[s], ss->ev_t1[s]) != hipSuccess) ms_s = 0.f;
        tm.shard_search_ms[s] = ms_s;
        tm.search_ms = std::max(tm.search_ms, ms_s);
    }
    (void)hipSetDevice(ss->dev[0]);
    if (hipEventElapsedTime(&tm.exchange_merge_ms, ss->ev_t1[0], ss->ev_merged) != hipSuccess) tm.exchange_merge_ms = 0.f;
    (void)hipGetLastError();
    tm.searches++;
    return MVF_OK;
}

}  // extern "C"
