/*
 * mvf_gpu.h — C ABI of libmvf_gpu.so: MI355X (gfx950) brute-force top-k
 * similarity search over one MVF vector space.
 *
 * WHAT IT REPLACES.  The reference has no FFI for this path; the scan is the
 * inline loop `find_top_k_similar` (reference examples/similarity_search.rs:
 * 140-176): per row VectorSpace::get_vector (src/vectors/vector_space.rs:
 * 101-142) -> Vector::as_f32 (src/vectors/vector.rs:71-92) -> scalar distance
 * (similarity_search.rs:152-157) -> BinaryHeap (:159-168) -> sort (:172-173).
 * This library replaces that whole loop.  The hand-off is what
 * VectorSpace::map_vector_range(0, total) (vector_space.rs:155-188) +
 * VectorSlice::as_ptr (src/vectors/mem.rs:75-77) already expose: base
 * pointer, row stride, row count, DataType; plus dimension()/distance_metric().
 * INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add.
 *
 * CONVENTIONS
 *  - every call returns enum mvf_status (mvf_status.h), 0 = OK; a thread-local
 *    detail string is kept for the last failure (mvfgpu_last_error_message).
 *  - dtype / metric arguments use the schema's codes (schema/types.fbs).
 *  - a corpus handle is one ROW-RANGE SHARD resident on ONE GPU; it is
 *    immutable after creation, owned by the library, freed by
 *    mvfgpu_corpus_destroy.  Multi-GPU = one handle per GPU (one process per
 *    GPU under torch.distributed/RCCL, or several handles in one process) and
 *    a merge of the per-shard results (mvfgpu_merge_topk_*).
 *  - searches on one handle may be issued from several threads; they are
 *    serialised on the handle's scratch space internally.
 *  - there is NO CPU fallback: without a gfx950 device every compute entry
 *    point returns MVF_ERR_DEVICE.
 *
 * SEMANTICS (the reference only pins L2 over f32/f16)
 *  - L2: sqrt(sum (q-x)^2), k smallest.  InnerProduct: sum q*x, k largest.
 *    Cosine: dot/(|q||x|), 0 when a norm is 0, k largest.
 *  - Float32/Float16 spaces take f32 queries (f16 widened exactly, as
 *    Vector::as_f32).  Int8/UInt8 spaces take queries of the space's own
 *    dtype; sums are exact i32 (dimension <= 33025), bit-exact vs the CPU.
 *  - results are sorted best-first, ties by ascending row index, NaN last;
 *    when k > rows the tail is padded with index UINT64_MAX and score
 *    +inf (L2) / -inf (InnerProduct, Cosine).
 */
#ifndef MVF_GPU_H
#define MVF_GPU_H

#include <stddef.h>
#include <string.h>
#include <stdint.h>

#include "mvf_status.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvfgpu_corpus mvfgpu_corpus;

#define MVFGPU_MAX_K 0x80000000u  /* largest k a search accepts (the reference takes any usize, similarity_search.rs:143);
                                     entries beyond the shard's live rows are the padding result */
#define MVFGPU_K_PER_PASS 1024u   /* results one pass over the rows selects.  Beyond it a search either runs ceil(k / 1024)
                                     passes of the streaming kernel per 1..4 queries, each returning the rows ranked strictly
                                     behind the last one of the pass before, or -- when that is cheaper, and always beyond
                                     MVFGPU_K_BY_PASSES -- has the streaming kernel write every row's order key (8 bytes per
                                     row) and ranks the WHOLE shard with a device-wide sort (16 bytes of scratch per row and
                                     query of a pass).  Both exact, whatever the batch size, no host wait */
#define MVFGPU_K_BY_PASSES 16384u /* largest k the pass formulation serves (the fallback when the sort's scratch does not fit) */
#define MVFGPU_MAX_INT_DIM 33025u /* d*255^2 < 2^31 */

/*
 * OUT-STRUCTS GROW.  Every struct a getter fills starts with `struct_size`: the CALLER sets it to sizeof(its struct)
 * before the call, the library writes at most that many bytes (fields a newer library knows and the caller does not are
 * dropped; fields a newer caller knows and an older library does not stay as the caller initialised them) and stores
 * the number of bytes it filled back into struct_size.  struct_size < 8 -> MVF_ERR_INVALID_ARGUMENT ("struct_size not
 * set").  MVFGPU_INIT(s) zeroes a struct and sets the field.
 */
#define MVFGPU_INIT(s) (memset(&(s), 0, sizeof(s)), (s).struct_size = (uint32_t)sizeof(s))

typedef struct mvfgpu_corpus_info {
    uint32_t struct_size; /* in: sizeof(mvfgpu_corpus_info); out: bytes filled */
    int32_t device;
    uint64_t rows;        /* rows in this shard */
    uint64_t index_base;  /* global index of the shard's first row */
    uint32_t dimension;
    uint32_t pitch_bytes; /* device row pitch: dimension*elem_size rounded up to 16 */
    uint8_t data_type;    /* enum mvf_data_type */
    uint8_t has_vector_ids; /* 1 when vector ids are attached (searches then report ids, not positions) */
    uint8_t shadows;      /* bit 0: the int8 selection shadow is resident (all rows), bit 1: the scaled-f16 one, bit 2: an int8
                             shadow of a PREFIX of the rows (all rows did not fit: batched searches run as two row ranges) */
    uint8_t selection_state; /* bit 0: the repair feedback has switched the int8-shadow selection off for this corpus,
                                bit 1: it has switched the folded pre-filter of the int8 kernels off */
    uint32_t reserved2;
    uint64_t device_bytes; /* HBM held by the handle: rows, deletion bitmap, ids, norms, every scratch buffer and the
                              selection shadows (int8: +dimension bytes per row; scaled f16: +2*dimension) with their
                              per-row scales and bound statistics, once built */
    uint64_t deleted_rows; /* rows masked by the tombstone bitmap */
} mvfgpu_corpus_info;

typedef struct mvfgpu_timing {
    uint32_t struct_size; /* in: sizeof(mvfgpu_timing); out: bytes filled */
    uint32_t samples;    /* searches averaged */
    /* HIP-event times of searches on the handle, milliseconds.  Events are
     * recorded (never waited for) on the search's own stream while
     * mvfgpu_set_profiling(corpus, 1) is in effect; mvfgpu_last_timing waits
     * for the newest and reads them back. */
    float scan_ms;     /* newest search: the dominant kernel (streaming or MFMA scan) */
    float select_ms;   /* newest search: candidate select / top-k kernels */
    float total_ms;    /* newest search: scan_ms + select_ms */
    float scan_ms_avg;   /* mean over the (up to 64) newest profiled searches */
    float select_ms_avg;
    uint32_t scan_kernel; /* 1 = streaming (K1) on the stored rows, 5 = K1 on the f16 shadow of a Float32
                             corpus (scan path 4); MFMA batched (K2): 2 = f32 kernel on Float32 rows,
                             3 = f16/int8 kernel on the stored rows, 4 = f16 kernel on the f16 shadow,
                             6 = int8 kernel on the int8 shadow of a Float32 / Float16 corpus (scan path 5);
                             7 = K1 on the int8 shadow (scan path 6; one query once the shadow exists);
                             8 = K1 writing every row's order key + the whole-shard sort (k > MVFGPU_K_PER_PASS) */
    uint32_t scan_launches; /* scan launches of one search (timing covers the first) */
    uint64_t scan_bytes; /* algorithmic bytes one scan launch reads */
    uint64_t scan_flops; /* algorithmic flops of one scan launch (2*nq*rows*dim) */
    /* the WHOLE search on the device: first to last kernel of the call on its stream (query preparation, every scan
     * phase, compactions, re-scoring, the repair launches) */
    float search_ms;      /* newest search */
    float search_ms_avg;  /* mean over the profiled searches */
    uint64_t search_flops; /* 2 * nq * rows * dim of the whole search */
    uint32_t repaired_queries; /* newest BATCHED search (whether profiled or not): queries whose candidate budget or region
                                  overflowed and that the streaming kernel re-did exactly (0 on sane data; a corpus that
                                  keeps producing them goes back to the slower selection paths by itself) */
    uint32_t reserved;
} mvfgpu_timing;

/* ---- library / device ---------------------------------------------------- */

/* Number of visible GPUs (0 and MVF_OK when there is none). */
int mvfgpu_device_count(int* out_count);
const char* mvfgpu_strerror(int status);
/* Detail of the calling thread's last failure ("" if none). */
const char* mvfgpu_last_error_message(void);

/* ---- corpus -------------------------------------------------------------- */

/*
 * Upload `n` rows to HBM on `device`.
 *   rows         : host pointer, any alignment (an mmap'd MVF block starts at
 *                  file offset 4 — src/builder.rs:421); BORROWED for the call
 *                  only, the mapping may be dropped afterwards.
 *   stride_bytes : bytes between consecutive rows; must be >= dimension *
 *                  elem_size.  The reference always passes dimension*elem_size
 *                  (VectorSlice stride, vector_space.rs:177,187).
 *   index_base   : added to every returned index (global row of rows[0]); 0
 *                  for an unsharded space.
 * Errors: MVF_ERR_BUILD for a data type other than Float32/Float16/Int8/UInt8
 * ("Unsupported vector data type", vector_space.rs:126); MVF_ERR_INVALID_
 * ARGUMENT for dimension 0, NULL rows with n > 0, n > 2^32-65536 rows per shard;
 * MVF_ERR_DEVICE for HIP failures (incl. out of memory).
 */
int mvfgpu_corpus_create(const void* rows, uint64_t n, uint32_t dimension,
                         uint8_t data_type, uint64_t stride_bytes, int device,
                         uint64_t index_base, mvfgpu_corpus** out);

/*
 * The same upload with options (NULL = the defaults of mvfgpu_corpus_create).  The upload is a two-stream pipeline
 * (DESIGN.md §6): chunks of `chunk_mib` MiB (0 = 256) cross PCIe on one stream while the other re-pitches the chunk
 * before (rows whose size is not a multiple of 16 bytes, or that lie further apart than their size) and, on request,
 * computes what the first BATCHED search would otherwise build -- the row norms, and for Float32 / Float16 spaces the
 * int8 shadow used for candidate selection (+25 % / +50 % device memory; skipped silently when that would leave < 2 GiB
 * free).
 * Checksum validation of the block (the reference leaves it `todo!()`, src/reader.rs:220) lives in libmvf_host
 * (mvf_reader_validate_with_checksum): run it on another thread beside this call.
 */
typedef struct mvfgpu_upload_options {
    uint32_t struct_size; /* sizeof(mvfgpu_upload_options): lets the struct grow */
    uint32_t flags;       /* MVFGPU_UPLOAD_* */
    uint32_t chunk_mib;   /* chunk size in MiB, 0 = default (64 with pinned staging, else 256) */
    uint32_t reserved;
} mvfgpu_upload_options;
#define MVFGPU_UPLOAD_EAGER_NORMS 1u    /* row norms (K4) per chunk, beside the copy of the next */
#define MVFGPU_UPLOAD_EAGER_SHADOW 2u   /* Float32 / Float16 spaces: norms + the selection shadow batched searches use, per
                                           chunk: the INT8 shadow (+dimension bytes per row; the default selection), or
                                           the scaled-f16 one of a Float32 space when MVF_I8_SHADOW=0 */
#define MVFGPU_UPLOAD_PINNED_STAGING 4u /* double-buffer through two pinned host chunks filled by memcpy threads: the default
                                           for uploads of >= 256 MiB (50 GB/s measured against 10-21 GB/s for the
                                           pageable source handed to the runtime); this flag forces it for small ones */
#define MVFGPU_UPLOAD_PAGEABLE 8u       /* never stage: hand the (pageable / mmap'd) source to the runtime as it is */
int mvfgpu_corpus_create_ex(const void* rows, uint64_t n, uint32_t dimension,
                            uint8_t data_type, uint64_t stride_bytes, int device,
                            uint64_t index_base, const mvfgpu_upload_options* options,
                            mvfgpu_corpus** out);

/*
 * Generate rows [row0, row0+n) of the synthetic corpus on the device
 * (counter-based: element (r,c) = f(seed, r*dimension + c), DESIGN.md §6 —
 * the CPU oracle regenerates any row).  index_base = row0.
 */
int mvfgpu_corpus_create_synthetic(uint64_t n, uint32_t dimension,
                                   uint8_t data_type, uint64_t seed,
                                   uint64_t row0, int device,
                                   mvfgpu_corpus** out);

void mvfgpu_corpus_destroy(mvfgpu_corpus* corpus);
int mvfgpu_corpus_get_info(const mvfgpu_corpus* corpus, mvfgpu_corpus_info* out);

/* Copy `count` rows starting at local row `first` back to the host, tightly
 * packed (dimension*elem_size per row) — the device-side get_vector
 * (vector_space.rs:101-142); MVF_ERR_INDEX_OUT_OF_BOUNDS past the end. */
int mvfgpu_corpus_read_rows(const mvfgpu_corpus* corpus, uint64_t first,
                            uint64_t count, void* out_rows);

/* Gather arbitrary rows by GLOBAL index (as returned by a search) from HBM, tightly packed, in the order given:
 * the payload of the reference's ScoredVector.vector (examples/similarity_search.rs:18,:159-163) without touching
 * the file again.  Indices of UINT64_MAX (padding of a short result list) give zero rows; any other index outside
 * [index_base, index_base + rows) -> MVF_ERR_INDEX_OUT_OF_BOUNDS. */
int mvfgpu_corpus_gather_rows(const mvfgpu_corpus* corpus, const uint64_t* indices,
                              uint64_t count, void* out_rows);

/*
 * Deletions and vector ids (schema/core.fbs:35-39 TombstoneInfo, :54 vector_ids_block_index, :56 tombstones).  The
 * reference neither writes nor honours them (src/builder.rs:483-485), so the semantics are this library's:
 *   - tombstones: bit (first_bit + r) of `bitmap` (bit b of a byte array = byte b >> 3, bit b & 7) set = the shard's
 *     LOCAL row r is deleted; `nbits` = bits the array holds, >= first_bit + rows.  A deleted row is never returned: the
 *     streaming kernel tests the bit where a row would enter a candidate list, the MFMA kernels where a candidate is
 *     appended.  libmvf_host's mvf_space_tombstone_bitmap produces the bitmap from either on-disk format; pass the
 *     whole space's bitmap with first_bit = the shard's first row.  NULL / 0 removes the mask.
 *   - vector ids: one u64 (little endian, any alignment) per LOCAL row; searches then report ids[row] instead of
 *     index_base + row (ties still break by row position, then -- across shards -- by shard order), and
 *     mvfgpu_corpus_gather_rows accepts the reported ids.  NULL / 0 removes them.
 * Both calls wait for the device to go idle; like create / destroy they must not race with searches on the handle.
 */
int mvfgpu_corpus_set_tombstones(mvfgpu_corpus* corpus, const uint8_t* bitmap,
                                 uint64_t first_bit, uint64_t nbits);
int mvfgpu_corpus_set_vector_ids(mvfgpu_corpus* corpus, const void* ids_le, uint64_t n);

/* ---- search -------------------------------------------------------------- */

/*
 * Replaces find_top_k_similar (similarity_search.rs:140-176) for a batch.
 *   queries    : host, row-major [nq][dimension], contiguous; query_dtype
 *                must be Float32 for Float32/Float16 spaces and the space's
 *                dtype for Int8/UInt8 spaces (else MVF_ERR_BUILD).
 *   query_dim  : length of each query; != corpus dimension ->
 *                MVF_ERR_DIMENSION_MISMATCH (the reference's zip silently
 *                truncates, similarity_search.rs:154; we refuse).
 *   out_scores : host [nq][k] f32, out_indices: host [nq][k] u64 (the
 *                reference's ScoredVector.index is u64, :16), both
 *                caller-owned.  out_raw (nullable): [nq][k] exact i32 of
 *                L2 (sum of squared differences) / InnerProduct for Int8/UInt8
 *                spaces, 0 otherwise.
 * Blocking: returns after the results are on the host.  Small transfers (queries <= 64 KiB, results <= 256 KiB) use no
 * copy engine: the kernels read the query from and write the results into pinned host memory of the handle, the CPU
 * copies to / from the caller's (pageable) buffers -- one query on 10k x 128 f32: 61 -> 33 us per call
 * (profiles/r04_host_api_latency.txt).  MVF_HOST_ZC_QUERY / MVF_HOST_ZC_RESULTS (bytes; 0 = always copy) move the limits.
 * Such a call (results in place, no payload) does not wait on its stream either: the final select stores a sequence number into
 * pinned host memory behind its results and the call spins on that word (up to 300 us, then the stream) -- the host learns of a
 * finished kernel ~5 us sooner that way (profiles/r04_flag_wait.txt; MVF_HOST_FLAG_WAIT=0 waits on the stream).
 */
int mvfgpu_search(const mvfgpu_corpus* corpus, uint8_t metric,
                  const void* queries, uint8_t query_dtype, uint32_t query_dim,
                  uint32_t nq, uint32_t k, float* out_scores,
                  uint64_t* out_indices, int32_t* out_raw);

/*
 * The same search, returning the payload as well: out_vectors = host [nq][k][dimension] in the space's stored type -- what
 * the reference's ScoredVector.vector holds (examples/similarity_search.rs:18, :159-163).  Only the first min(k, rows of the
 * corpus) rows of each query's k are WRITTEN (a padding result among them -- deleted rows -- gives a zero row); the rows behind
 * them, whose results are always padding, are left untouched: k far beyond the corpus (the reference takes any k and returns
 * min(k, n) items) costs min(k, n) rows per query on the device and in the copies, and with nq = 1 a buffer of min(k, rows)
 * rows is enough.  The rows are gathered on the device behind the search, from the result
 * indices where the selection kernel left them: one submission and one wait instead of mvfgpu_search +
 * mvfgpu_corpus_gather_rows.  Small results (rows and results <= 256 KiB, positions not ids) need no gather kernel at all:
 * the final select copies its query's k rows behind the results and the call waits on the flag it stores last (10k x 128 f32,
 * top-10 with vectors: 47.5 us for the two calls, 29.8 for this one).  A corpus that reports vector ids maps them back on the
 * host after the search (the two steps, inside this call).
 */
int mvfgpu_search_fetch(const mvfgpu_corpus* corpus, uint8_t metric,
                        const void* queries, uint8_t query_dtype, uint32_t query_dim,
                        uint32_t nq, uint32_t k, float* out_scores,
                        uint64_t* out_indices, int32_t* out_raw, void* out_vectors);

/*
 * Same search with queries and outputs RESIDENT ON THE CORPUS' DEVICE,
 * asynchronous on `hip_stream` (a hipStream_t; NULL = the default stream).
 * This is the timed region of bench.py and the producer of the per-shard
 * lists that RCCL all-gathers.  d_raw may be NULL.
 * The call does not wait for the device -- with ONE bound: a handle keeps at most TWO int8-selected batched searches
 * in flight.  Such a search posts how many of its queries the repair pass redid (a 4-byte copy behind an event) and the
 * search two calls later consumes that sample before it picks its path, waiting for it if it has not arrived: which path
 * a search takes then depends on the sequence of searches alone, never on how far the host runs ahead.  A caller that
 * pipelines three or more batched searches on one handle has its third enqueue wait for the first search to finish.
 * k > MVFGPU_K_PER_PASS: ceil(k / 1024) passes of the streaming kernel (the floor travels on the device) or one dump pass +
 * a device-wide sort of the shard's order keys, whichever is cheaper (see MVFGPU_K_PER_PASS); no host wait either way.
 * Small batches run the streaming kernel (below 32 queries on corpora under
 * 1 GiB; on larger ones a single query, below 5 for Int8/UInt8 and below 9 for Float32 without the f16 shadow -- the
 * measured crossovers: the streaming kernel takes up to 4 queries per pass over the rows, the MFMA path uses a
 * 64-query tile up to 128 queries).  Larger batches run the MFMA path in phases; an adversarially ordered corpus can
 * overflow a query's candidate buffer there, and such queries are redone exactly by the streaming kernel in REPAIR
 * launches that follow every batched search and decide ON THE DEVICE whether they have anything to do (round 1 read
 * the flags back and synchronised; now e.g. an RCCL all-gather can be queued right behind the search).
 */
int mvfgpu_search_device(const mvfgpu_corpus* corpus, uint8_t metric,
                         const void* d_queries, uint8_t query_dtype,
                         uint32_t query_dim, uint32_t nq, uint32_t k,
                         float* d_scores, uint64_t* d_indices, int32_t* d_raw,
                         void* hip_stream);

/*
 * Merge `nlists` per-shard results, each [nq][k] sorted best-first with
 * UINT64_MAX padding, laid out [nlists][nq][k], into the global [nq][k]
 * ordered by (score order, list order, rank inside the list).  The lists must
 * come in ASCENDING ROW-RANGE ORDER (the rank order of an all-gather): each is
 * sorted by (score order, row position), so ties come out in ascending global
 * row position -- also when the shards report vector ids instead of positions
 * (mvfgpu_corpus_set_vector_ids).  nlists * k < 2^32: up to 8192 entries per query merge in one block's LDS, more by a
 * device-wide sort of the query's entries (the _device forms take their scratch from the stream-ordered allocator:
 * hipMallocAsync / hipFreeAsync on hip_stream).  data_type tells whether `raw`
 * carries the exact integer score (Int8/UInt8 with L2/InnerProduct).
 * _host: plain host buffers, no GPU needed.  _device: device buffers on
 * `device` (e.g. the output of an RCCL all-gather), async on hip_stream.
 */
int mvfgpu_merge_topk_host(const float* scores, const uint64_t* indices,
                           const int32_t* raw, uint32_t nlists, uint32_t nq,
                           uint32_t k, uint8_t metric, uint8_t data_type,
                           float* out_scores, uint64_t* out_indices,
                           int32_t* out_raw);
int mvfgpu_merge_topk_device(const float* d_scores, const uint64_t* d_indices,
                             const int32_t* d_raw, uint32_t nlists, uint32_t nq,
                             uint32_t k, uint8_t metric, uint8_t data_type,
                             float* d_out_scores, uint64_t* d_out_indices,
                             int32_t* d_out_raw, int device, void* hip_stream);

/*
 * The same merge over PACKED lists, the form one all-gather delivers: shard l's
 * list occupies MVFGPU_PACKED_LIST_BYTES(nq, k) bytes at d_packed + l * that,
 * laid out { uint64 indices[nq*k]; float scores[nq*k]; int32 raw[nq*k] }.
 * A rank points mvfgpu_search_device's three outputs into its own list and
 * exchanges it with ONE collective instead of three (the exchange is
 * latency-bound: 16 bytes per result).  d_packed must be 8-byte aligned.
 */
#define MVFGPU_PACKED_LIST_BYTES(nq, k) ((size_t)16 * (size_t)(nq) * (size_t)(k))
int mvfgpu_merge_topk_packed_device(const void* d_packed, uint32_t nlists,
                                    uint32_t nq, uint32_t k, uint8_t metric,
                                    uint8_t data_type, float* d_out_scores,
                                    uint64_t* d_out_indices, int32_t* d_out_raw,
                                    int device, void* hip_stream);

/* ---- several GPUs in one process ------------------------------------------ */

/*
 * A shard set = the row-range shards of ONE vector space, one corpus handle per GPU, searched as a whole
 * (SURVEY.md §8e): every shard searches its rows with global indices on its own device and stream, the per-shard
 * top-k lists cross xGMI in ONE packed RCCL all-gather (ncclCommInitAll over the shards' devices; single process, no
 * MPI, no torch), and the (score order, shard order, rank) merge runs on the first shard's device.  This is what a
 * Rust host that owns all GPUs of a node binds; one process per GPU (torch.distributed / RCCL) composes
 * mvfgpu_search_device + an all-gather + mvfgpu_merge_topk_packed_device itself (metrovector_amd/sharded.py).
 *   shards : handles on DISTINCT devices, in ascending row-range order (index_base ascending, ranges disjoint), all of
 *            one dimension and data type; BORROWED -- they must outlive the set and are not destroyed with it.
 *            A single shard is allowed (a 1-rank RCCL communicator: the same exchange code path).  Shards that share
 *            a device (rehearsing the protocol on fewer GPUs than shards; RCCL refuses duplicate devices), or
 *            MVF_SHARDSET_NO_RCCL=1, exchange their lists with device-to-device copies instead.
 * RCCL is loaded lazily (dlopen librccl.so) by the first mvfgpu_shardset_create; MVF_ERR_DEVICE if it is needed and
 * cannot be loaded.  Any k the shards take (n_shards * k < 2^32).  One search at a time per set (calls serialise); results as mvfgpu_search.
 */
typedef struct mvfgpu_shardset mvfgpu_shardset;
typedef struct mvfgpu_shardset_info {
    uint32_t struct_size; /* in: sizeof(mvfgpu_shardset_info); out: bytes filled */
    uint32_t n_shards;
    uint32_t rccl_ranks; /* ranks of the RCCL communicator (= n_shards), 0 when the lists travel by device copies */
    uint32_t dimension;
    uint8_t data_type;
    uint8_t reserved[7];
    uint64_t rows;       /* over all shards */
} mvfgpu_shardset_info;
#define MVFGPU_MAX_SHARDS 64
typedef struct mvfgpu_shardset_timing { /* the newest search, milliseconds */
    uint32_t struct_size;    /* in: sizeof(mvfgpu_shardset_timing); out: bytes filled */
    uint32_t n_shards;
    uint64_t searches;       /* searches the set has served */
    float total_ms;          /* HOST wall clock, call to return: query upload, searches, exchange, merge, results on the host */
    float enqueue_ms;        /* HOST wall clock until every shard's search and the exchange step were enqueued */
    float search_ms;         /* DEVICE (HIP events on the shard's own stream): query upload + local search of the SLOWEST
                                shard; the shards run concurrently */
    float exchange_merge_ms; /* DEVICE, first shard's stream: from the end of ITS local search to the end of the merge =
                                waiting for the slowest shard + the all-gather (RCCL, or device copies) + merge_shards */
    float shard_search_ms[MVFGPU_MAX_SHARDS]; /* DEVICE: query upload + local search per shard (row-range order) */
} mvfgpu_shardset_timing;
int mvfgpu_shardset_create(mvfgpu_corpus* const* shards, int n_shards, mvfgpu_shardset** out);
void mvfgpu_shardset_destroy(mvfgpu_shardset* set);
int mvfgpu_shardset_get_info(const mvfgpu_shardset* set, mvfgpu_shardset_info* out);
int mvfgpu_shardset_search(mvfgpu_shardset* set, uint8_t metric, const void* queries,
                           uint8_t query_dtype, uint32_t query_dim, uint32_t nq, uint32_t k,
                           float* out_scores, uint64_t* out_indices, int32_t* out_raw);
int mvfgpu_shardset_last_timing(const mvfgpu_shardset* set, mvfgpu_shardset_timing* out);

/* ---- utilities ----------------------------------------------------------- */

/* Fill a device buffer [nq][dimension] with synthetic queries (query dtype of
 * `data_type`: f32 for Float32/Float16 spaces, else the int type). */
int mvfgpu_synth_queries_device(void* d_queries, uint32_t nq, uint32_t dimension,
                                uint8_t data_type, uint64_t seed, int device,
                                void* hip_stream);

int mvfgpu_set_profiling(mvfgpu_corpus* corpus, int enabled);
int mvfgpu_last_timing(const mvfgpu_corpus* corpus, mvfgpu_timing* out);

/* Force a scan path for A/B measurements and tests: 0 = automatic,
 * 1 = streaming kernel (K1) for every nq, 2 = MFMA batched kernel (K2) on the
 * stored rows, 3 = K2 with the f16 shadow on Float32 corpora (same as 2 on
 * the other types), 4 = as 0, but one or two queries on a Float32 corpus
 * STREAM THE F16 SHADOW instead of the stored rows (half the bytes, so about
 * half the time; same proven-margin selection and exact re-scoring as the
 * batched path: the same rows as path 1, scores within the 1e-5 tolerance --
 * the re-scoring kernel sums in another order than the streaming kernel;
 * also enabled by MVF_STREAM_SHADOW=1 in the environment).  Off by default:
 * the default single-query path reads the stored rows, whatever the handle
 * has served before.
 * 5 = K2 selecting on an INT8 SHADOW of a Float32 / Float16 corpus (per-row
 * scale; built on first use, `dimension` bytes per row; also enabled by
 * MVF_I8_SHADOW=1): the int8 MFMA runs at about twice the f16 kernel's rate
 * under the part's power limit; every row whose approximate score is within a
 * PROVEN bound of the k-th best (5-8 x k rows per query on uniform data; the
 * f16 shadow keeps a handful) is re-scored from the stored rows and the f32
 * query, so the rows are again those of the exact path (up to ties within
 * the tolerance; tests/test_gpu_round3.py compares all 1024 lists of the
 * benchmark batch).  Same as 0 on Int8 /
 * UInt8 corpora, and for k > 409 (path 6's streaming: k > 204): the margin
 * would not fit the candidate budget, such requests take the f16 shadow / the
 * stored rows.
 * 6 = as 5, and ONE TO FOUR queries STREAM THE INT8 SHADOW through K1
 * (`dimension` bytes per row instead of 4x / 2x that; same bound, same exact
 * re-scoring: 1.2-1.3 ms for one query instead of 4.5 on 10M x 768 f32).
 * MVF_STREAM_I8=1 makes path 0 do this for ONE query once the corpus holds an
 * int8 shadow anyway (round 2 did so unasked: the answer to the same query then
 * depended, within the tolerance, on what the handle had served before).  Two
 * to four queries are served as fast by the 64-query MFMA tile.
 *
 * Selection shadows: batched searches on a Float32 / Float16 corpus select
 * candidates on an INT8 copy of the rows (path 5's, the default: built by the
 * first such search or an eager upload; skipped when it would leave < 2 GiB
 * free; MVF_I8_SHADOW=0 opts out) -- or, Float32 corpora on scan path 3 /
 * without the int8 one, on a scaled-f16 copy (+50 %; MVF_F16_SHADOW=0 opts
 * out) -- keep every row whose approximate score is within a proven error bound
 * of the k-th, and re-score the kept rows from the stored rows and the f32
 * query.  Rows and order are those of the exact path; only the selection
 * arithmetic differs.  mvfgpu_corpus_get_info reports which shadows a handle
 * holds (`shadows`) and counts them in `device_bytes`. */
int mvfgpu_set_scan_path(mvfgpu_corpus* corpus, int path);

/*
 * The tuning switches of the environment (MVF_K1_G, MVF_K1_RANK_MERGE, MVF_K2_*, MVF_I8_SHADOW, MVF_F16_SHADOW, MVF_QS_REFINE,
 * MVF_STREAM_*, MVF_REPAIR_WINDOW, MVF_UPLOAD_THREADS, MVF_HOST_ZC_*, MVF_HOST_FLAG_WAIT, MVF_LARGE_K, MVF_DEBUG_REPAIR; INTEGRATION.md lists them) are read ONCE per
 * handle, when it is created: a search never calls getenv.  An A/B script that changes the environment of a live handle
 * calls this to have it read again.  A development aid: it waits for the handle's host-buffer searches, but
 * mvfgpu_search_device reads the switches unlocked -- do not call it beside device-pointer searches of the same handle.
 */
int mvfgpu_corpus_reload_tuning(mvfgpu_corpus* corpus);

/*
 * ABI version of the library: bumped whenever a struct layout or a function signature of this header changes in a way
 * an older caller would misread (2: every out-struct starts with struct_size, round 3; 3: corpus_info.selection_state,
 * reload_tuning, k beyond 1024; the later lift of the k <= 16384 limit changed no layout and no signature).  A binding compares it with the MVFGPU_ABI_VERSION it was built against
 * at load time.
 */
#define MVFGPU_ABI_VERSION 3u
uint32_t mvfgpu_abi_version(void);

/*
 * Self-test of the automatic path choice (no GPU needed).  Batched searches watch how many of their queries the repair
 * pass had to redo and switch a corpus whose data defeats the cheap bounds back to slower selections: first the folded
 * pre-filter of the int8 kernels goes, then the int8-shadow selection.  `samples` holds n_samples records of four u32
 * {queries of the search, queries repaired, ran with the folded pre-filter, selected on the int8 shadow} in the order
 * the searches were consumed; out_state receives {queries counted, repairs counted, pre-filter off, int8 selection off}.
 */
int mvfgpu_selftest_feedback(const uint32_t* samples, uint32_t n_samples, uint32_t* out_state);

/*
 * Self-test of the route a search takes (no GPU needed): which kernels serve `nq` queries for `k` results on a corpus of
 * `rows` x `dimension` of `data_type` under the default tuning, scan path 0 and no history -- a function of these numbers
 * alone (the thresholds come from the measured crossover tables under profiles/).
 * *out_route: 0 = the streaming kernel K1 (one pass per 1..4 queries), 1 = the batched MFMA route; for k >
 * MVFGPU_K_PER_PASS: 2 = passes of K1 behind a floor, 3 = K1 as a dump + the whole-shard sort.
 */
int mvfgpu_selftest_route(uint64_t rows, uint32_t dimension, uint8_t data_type, uint8_t metric, uint32_t nq, uint32_t k,
                          uint32_t* out_route);

/*
 * Self-test of a batched search's phase schedule (no GPU needed): the row boundaries R_1 .. R_last = rows of the geometric phases a
 * batch of `nq` queries for `k` results runs over `rows` rows under the default tuning (int8_selection != 0: the lists of the
 * int8-shadow selection, 8192 slots per query; 0: 4096), the growth factor between them, and in *out_refined_mask bit i set where the
 * threshold is refined with exact scores BEHIND phase i (in front of phase i + 1).  DESIGN.md section 5; the constants come from
 * the in-process A/B runs of profiles/r05_k2_walk_and_phase_costs.txt.  k <= MVFGPU_K_PER_PASS.
 */
int mvfgpu_selftest_schedule(uint64_t rows, uint32_t nq, uint32_t k, int int8_selection, uint64_t* out_bounds, uint32_t max_bounds,
                             uint32_t* out_n_bounds, uint32_t* out_growth, uint32_t* out_refined_mask);

#ifdef __cplusplus
}
#endif
#endif
