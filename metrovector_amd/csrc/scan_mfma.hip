// scan_mfma.hip — K2: the batched-query scan on the matrix cores, plus its
// helpers (K4 row norms, query preparation, candidate compaction).
//
// Replaces, for nq >> 1, the same reference loop as K1
// (examples/similarity_search.rs:147-169) evaluated for a whole batch of
// queries at once:  S[q][r] = sum_d Q[q][d] * X[r][d]  is a GEMM
// (M = queries, N = corpus rows, K = dimension); scores are never
// materialised (1024 x 10M x 4 B = 41 GB) — the epilogue keeps only entries
// that beat each query's current threshold.
//
// f32 path: v_mfma_f32_32x32x2_f32 — exact f32 (a k-ordered fmaf chain), 64
// FLOP/clk/SIMD = 157.3 TFLOP/s peak (MI355X_MICROARCH.md §Matrix cores).
//
// Block = 256 threads = 4 waves (2 x 2), block tile 128 queries x 128 rows x
// 32 k; each wave owns a 64 x 64 output = 2 x 2 MFMA tiles (64 acc VGPRs).
// A (queries) and B (corpus rows) tiles are staged global -> registers -> LDS
// (two LDS stages, one barrier per k-tile; the next tile's global loads are
// issued before the current tile's MFMAs).  LDS rows are padded to 36 floats
// so the ds_read_b128 operand fetches (lane l: row l&31, k = 4*(l>>5)..+3 of
// each group of 8 k) are bank-conflict-free.  Using 4 consecutive k per lane
// half (instead of the instruction's natural k = l>>5) is legal because A and
// B use the same k permutation and a dot product is order-independent.
//
// Grid order is XCD-aware: blocks b and b+8 share an XCD (round-robin
// dispatch), so XCD x walks corpus tiles x, x+8, ... and runs all query tiles
// of one corpus tile back to back — the corpus tile is fetched from HBM once
// and served from that XCD's L2 to the other query tiles.
//
// Algorithmic work per launch: 2 * nq * rows * dim flop; rows*dim*es bytes.

#include "scan_mfma.h"

#include "bitonic.h"
#include "mvf_common.h"
#include "scan_mfma16_key.h"

#include <hip/hip_fp16.h>

#include <cstdlib>

namespace mvf {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDP = BK + 4;                    // padded LDS row pitch in floats
constexpr int TILE_F = BM * LDP;               // floats per operand tile per stage
constexpr size_t kLdsBytes = (size_t)4 * TILE_F * 4 + 3 * BM * 4;  // 2 stages x (A + B) + qnorm + tau + prefilter

template <int METRIC>
__global__ void __launch_bounds__(256, 2) scan_mfma_f32_kernel(BatchParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    float* qn_s = lds + 4 * TILE_F;                              // [BM]
    uint32_t* tau_s = reinterpret_cast<uint32_t*>(qn_s + BM);    // [BM]
    float* tql_s = reinterpret_cast<float*>(tau_s + BM);         // [BM] conservative float pre-filter threshold

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // PERSISTENT blocks, XCD-aware tile order.  Blocks b and b+8 share an XCD (round-robin dispatch): block b is
    // lane `ls` of XCD `xcd` and walks the slots ls, ls+nls, ls+2*nls, ...; slot -> (corpus tile, query tile) puts
    // the query tiles of one corpus tile on consecutive lanes of one XCD, so the corpus tile is fetched from HBM
    // once and re-served from that XCD's L2.  The load pipeline runs ACROSS tile boundaries, so only the first
    // tile of a block pays the HBM round trips of a prologue (in-kernel stamps: prologue + dispatch gap + the
    // old epilogue were 30 % of a one-tile-per-block launch).
    const uint32_t xcd = blockIdx.x & 7u, ls = blockIdx.x >> 3, nls = gridDim.x >> 3;
    auto slot_tile = [&](uint32_t n, uint32_t& nt, uint32_t& mt) {  // n-th work item of this block
        const uint32_t slot = ls + n * nls;
        nt = (slot / p.mtiles) * 8u + xcd;
        mt = slot % p.mtiles;
        return nt < p.ntiles;
    };
    uint32_t my_tiles = 0;
    {
        // slots are visited in increasing order and nt is monotone in the slot: count the valid prefix
        const uint32_t max_slot_excl = ((p.ntiles + 7u - xcd) / 8u) * p.mtiles;  // first slot whose nt >= ntiles
        if (ls < max_slot_excl) my_tiles = (max_slot_excl - ls + nls - 1) / nls;
    }
    if (my_tiles == 0) return;
    const uint32_t G = my_tiles * p.KT;  // flat k-tile count of this block

    auto load_query_consts = [&](uint32_t q0) {
        if (tid < BM) {
            const float qn = p.qnorm[q0 + tid];
            const uint32_t tau = p.tau[q0 + tid];
            qn_s[tid] = qn;
            tau_s[tid] = tau;
            // Pre-filter: "key <= tau" <=> "score >= ts" (ts = the k-th best score; NaN when there is none yet, and
            // every comparison with NaN passes).  The epilogue tests y = dot * (1/|x|) against ts*|q| lowered by a
            // 2e-6 relative margin (>> the rounding difference to the exact dot/(|q||x|)), so it never rejects a
            // row the exact test would accept; the exact key is only computed for rows that pass.
            const float ts = score_from_key(tau, METRIC);
            if (METRIC == MVF_METRIC_L2) {
                // batched L2 selects on the GEMM-form squared distance s2 = qq + xx - 2 dot (exact distances are
                // re-scored afterwards); tau holds ord(thr).  s2 <= thr  <=>  2 dot - xx >= qq - thr.
                const float qq = qn * qn;
                const float cq = qq - ts;
                tql_s[tid] = cq - fabsf(cq) * 2e-6f - (qq + p.xxmax[0]) * 4e-7f;
            } else {
                const float tq = METRIC == MVF_METRIC_COSINE ? ts * qn : ts;
                tql_s[tid] = tq - fabsf(tq) * 2e-6f;
            }
            // padding queries of the batch's last tile (zero vectors) never pass: without this their "no threshold yet"
            // sent every 32 x 32 tile of their wave through the exact path
            if (q0 + tid >= p.nq) {
                tau_s[tid] = 0u;
                tql_s[tid] = __builtin_inff();
            }
        }
    };

    // ---- staging maps: thread -> (row sr + 32*i, float4 column sc) ------------------------------------------
    // Loads are branch-free: rows past row_end read row 0 instead (their output columns are discarded in the
    // epilogue); k beyond the row's pitch (last k-tile when dim % 32 != 0) reads the row start and is zeroed with
    // a select at LDS-store time, because 0 * garbage could be 0 * inf.
    const int sr = tid >> 3, sc = tid & 7;
    // load cursors: flat position (tile ordinal, k-tile) of the next A / B global load
    uint32_t a_n = 0, a_kt = 0, b_n = 0, b_kt = 0;
    const float* qsrc = nullptr;
    const unsigned char* xsrc[4];
    auto set_a_tile = [&](uint32_t n) {
        uint32_t nt, mt;
        slot_tile(n, nt, mt);
        qsrc = p.qmat + (size_t)(mt * BM + sr) * p.KP + sc * 4;
    };
    auto set_b_tile = [&](uint32_t n) {
        uint32_t nt, mt;
        slot_tile(n, nt, mt);
        const uint32_t r0 = p.row_begin + nt * BN;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t r = r0 + sr + 32 * i;
            xsrc[i] = p.rows + (size_t)(r < p.row_end ? r : 0u) * p.pitch;
        }
    };
    f32x4 ra[4], rb[4];
    auto load_a = [&]() {  // next A k-tile of the flat sequence -> ra
#pragma unroll
        for (int i = 0; i < 4; i++) ra[i] = *reinterpret_cast<const f32x4*>(qsrc + (size_t)i * 32 * p.KP + a_kt * BK);
        if (++a_kt == p.KT) {
            a_kt = 0;
            if (++a_n < my_tiles) set_a_tile(a_n);
        }
    };
    auto load_b = [&]() {  // next B k-tile of the flat sequence -> rb
        const uint32_t v = b_kt * 8 + sc;  // 16-B vector index within the row
        const size_t xoff = v < p.V ? (size_t)v * 16 : 0;
#pragma unroll
        for (int i = 0; i < 4; i++) rb[i] = *reinterpret_cast<const f32x4*>(xsrc[i] + xoff);
        if (++b_kt == p.KT) {
            b_kt = 0;
            if (++b_n < my_tiles) set_b_tile(b_n);
        }
    };
    auto store_a = [&](int stage) {
        float* a = lds + stage * 2 * TILE_F;
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<f32x4*>(a + (sr + 32 * i) * LDP + sc * 4) = ra[i];
    };
    auto store_b = [&](int stage, uint32_t kt) {  // kt = the k-tile (within its tile) held in rb
        float* bb = lds + stage * 2 * TILE_F + TILE_F;
        const bool vok = kt * 8 + sc < p.V;
#pragma unroll
        for (int i = 0; i < 4; i++)
            *reinterpret_cast<f32x4*>(bb + (sr + 32 * i) * LDP + sc * 4) = vok ? rb[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    };

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    };
    zero_acc();

    // compute cursor
    uint32_t c_n = 0, c_kt = 0, c_nt, c_mt;
    slot_tile(0, c_nt, c_mt);
    load_query_consts(c_mt * BM);

    // Pipeline.  LDS stage g&1 holds flat k-tile g; the staging registers hold k-tile g+1.  A k-tile's 64 MFMAs
    // run as four k-groups of 16; the operand fragments of group x+1 are fetched while group x computes; the LDS
    // stores of k-tile g+1 ride in front of groups 1 (A) and 2 (B), the global loads of k-tile g+2 in front of
    // group 3.  (Spreading them one by one between MFMA pairs was measured 15 % SLOWER: per-op waits and
    // branches; loading B two k-tiles ahead changed nothing.)  One barrier per k-tile.
    set_a_tile(0);
    set_b_tile(0);
    load_a();
    load_b();
    store_a(0);
    store_b(0, 0);
    if (G > 1) {
        load_a();
        load_b();
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    struct Frag {
        f32x4 a[2], b[2];
    };
    auto fetch = [&](Frag& f, const float* a, const float* bb, int ks) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            f.a[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDP + ks * 8);
            f.b[i] = *reinterpret_cast<const f32x4*>(bb + i * 32 * LDP + ks * 8);
        }
    };
    auto mfma16 = [&](const Frag& f) {
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][s], f.b[j][s], acc[i][j], 0, 0, 0);
    };

    // ---- epilogue of one finished tile ---------------------------------------------------------------------
    // C/D map of the 32x32 MFMA: col = lane&31 (corpus row), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (query), so
    // the 4 queries of one e>>2 group are contiguous in LDS (one b128 read).  Fast path per score: one multiply,
    // one compare.  Survivors are ~1 per tile in the late phases, so the exact path (IEEE division as in K1,
    // order key, atomic append) runs only for 32x32 tiles where the wave-wide ballot found a candidate.
    auto epilogue = [&](uint32_t nt, uint32_t mt) {
        const uint32_t q0 = mt * BM, r0 = p.row_begin + nt * BN;
        // keep the epilogue's address arithmetic inside the epilogue: hoisted out of the k-tile loop it costs
        // ~60 VGPRs there and spills
        int lane_q = wm * 64 + 4 * fh, lane_r = wn * 64 + fr;
        asm volatile("" : "+v"(lane_q), "+v"(lane_r));
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t r = r0 + lane_r + j * 32;
            bool rok = r < p.row_end;
            // deleted rows (tombstone bitmap): never appended; the direct phase stores padding in their slots
            const bool dead = p.tomb && rok && ((p.tomb[r >> 5] >> (r & 31)) & 1u);
            if (!p.direct) rok = rok && !dead;
            float xn = 0.f, rx = 1.f, xx = 0.f;
            if (METRIC == MVF_METRIC_COSINE) {
                if (rok) xn = p.xnorm[r];
                rx = xn > 0.0f ? __builtin_amdgcn_rcpf(xn) : 0.0f;
            }
            if (METRIC == MVF_METRIC_L2 && rok) xx = p.xx2[r];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                uint32_t m = 0;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const f32x4 tql4 = *reinterpret_cast<const f32x4*>(tql_s + lane_q + i * 32 + 8 * g);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const float y = METRIC == MVF_METRIC_COSINE ? acc[i][j][4 * g + t] * rx
                                        : METRIC == MVF_METRIC_L2   ? fmaf(2.0f, acc[i][j][4 * g + t], -xx)
                                                                    : acc[i][j][4 * g + t];
                        m |= (y < tql4[t] ? 0u : 1u) << (4 * g + t);
                    }
                }
                if (p.direct) m = 0xFFFFu;  // phase 0: every (query, row) pair is a candidate, whatever its score
                if (!rok) m = 0;
                if (__builtin_amdgcn_ballot_w64(m != 0) != 0) {  // wave-uniform: rare
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        if (m & (1u << e)) {
                            const int ql = lane_q + i * 32 + (e & 3) + 8 * (e >> 2);
                            float sc_ = acc[i][j][e];
                            if (METRIC == MVF_METRIC_COSINE) {
                                const float den = qn_s[ql] * xn;
                                sc_ = den > 0.0f ? sc_ / den : 0.0f;
                            }
                            if (METRIC == MVF_METRIC_L2) sc_ = qn_s[ql] * qn_s[ql] + xx - 2.0f * sc_;  // GEMM-form s2
                            const uint32_t key = key_from_score(sc_, METRIC);
                            const uint32_t q = q0 + ql;
                            if (q < p.nq && (p.direct || key <= tau_s[ql])) {
                                // phase 0 (rows <= cap, no threshold yet): the slot is the row's offset -- 32 lanes
                                // bumping one counter per query is what made that 4096-row phase cost milliseconds
                                const uint32_t slot_i = p.direct ? r - p.row_begin : atomicAdd(&p.cnt[q], 1u);
                                if (slot_i < p.cap) p.cand[(size_t)q * p.cap + slot_i] = dead ? kPadComposite : ((uint64_t)key << 32) | r;
                            }
                        }
                    }
                }
            }
        }
    };

    Frag f0, f1;
    for (uint32_t g = 0; g < G; g++) {
        const int cur = g & 1;
        const float* a = lds + cur * 2 * TILE_F + (wm * 64 + fr) * LDP + fh * 4;
        const float* bb = lds + cur * 2 * TILE_F + TILE_F + (wn * 64 + fr) * LDP + fh * 4;
        const bool more = g + 1 < G, more2 = g + 2 < G;
        const uint32_t next_kt = c_kt + 1 == p.KT ? 0u : c_kt + 1;  // k-tile (within its tile) held in the staging registers
        fetch(f0, a, bb, 0);
        fetch(f1, a, bb, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(f0);                       // group 0
        __builtin_amdgcn_sched_barrier(0);
        fetch(f0, a, bb, 2);
        if (more) store_a(cur ^ 1);
        mfma16(f1);                       // group 1
        __builtin_amdgcn_sched_barrier(0);
        fetch(f1, a, bb, 3);
        if (more) store_b(cur ^ 1, next_kt);
        mfma16(f0);                       // group 2
        __builtin_amdgcn_sched_barrier(0);
        if (more2) {
            load_a();
            load_b();
        }
        mfma16(f1);                       // group 3
        __syncthreads();
        if (++c_kt == p.KT) {  // tile finished: the next tile's first k-tile is already in LDS, its loads in flight
            epilogue(c_nt, c_mt);
            zero_acc();
            c_kt = 0;
            if (++c_n < my_tiles) {
                uint32_t nmt;
                slot_tile(c_n, c_nt, nmt);
                if (nmt != c_mt) {  // block-uniform; rare (grid lanes per XCD not a multiple of the query tiles)
                    __syncthreads();
                    load_query_consts(nmt * BM);
                    __syncthreads();
                    c_mt = nmt;
                }
            }
        }
    }
}

// ---- query preparation: zero-padded [nq_pad][KP] f32 copy + norms --------------
__global__ void prep_queries_kernel(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t KP,
                                    float* qmat, float* qnorm) {
    const uint32_t row = blockIdx.x;
    float part = 0.f;
    for (uint32_t c = threadIdx.x; c < KP; c += blockDim.x) {
        const float v = (row < nq && c < dim) ? q[(size_t)row * dim + c] : 0.f;
        qmat[(size_t)row * KP + c] = v;
        part = fmaf(v, v, part);
    }
    __shared__ float red[4];
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (uint32_t w = 0; w < (blockDim.x + 63) / 64; w++) s += red[w];
        qnorm[row] = sqrtf(s);
    }
}

// ---- K4: row norms sqrt(sum x^2), one wave per row (f32 rows) ---------------------
__global__ void __launch_bounds__(256) row_norms_f32_kernel(const unsigned char* rows, uint32_t n, uint32_t pitch,
                                                             uint32_t V, float* xnorm, float* xx2, float* xxmax) {
    float mx = 0.f;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    for (uint32_t r = wave; r < n; r += nwaves) {
        const unsigned char* rp = rows + (size_t)r * pitch;
        float s = 0.f;
        for (uint32_t v = lane; v < V; v += 64) {
            const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp + (size_t)v * 16));
            s = fmaf(x[0], x[0], s);
            s = fmaf(x[1], x[1], s);
            s = fmaf(x[2], x[2], s);
            s = fmaf(x[3], x[3], s);
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) {
            xnorm[r] = sqrtf(s);
            xx2[r] = s;
            if (s > mx) mx = s;  // NaN never wins: the margin uses finite rows only
        }
    }
    if (lane == 0 && mx > 0.f) atomicMax(reinterpret_cast<unsigned int*>(xxmax), __float_as_uint(mx));  // non-negative floats order as uints
}

// ---- candidate hand-off: per-block regions -> per-query lists ---------------------------------------------------
// The narrow-type K2 kernels append {key, row, query} records to per-block regions with an LDS counter (scan_mfma.h);
// this pass files them under their queries.  A record-per-thread version with one global atomicAdd(cnt[q]) each was
// bound by same-address atomics (2 M records on 1024 counters: 150 us per call, 0.9 ms of a 12-ms search).  Here a block
// owns a contiguous share of the regions and aggregates per query in LDS first: histogram with LDS atomics (which also
// hand every record its rank), ONE global atomicAdd per (block, query) to reserve the range, then the stores.
// grid (kScatterBlocks); block 1024; dynamic LDS 4 * nq_pad bytes (queries <= kScatterMaxQueries, else the simple form).
// The block's regions are walked as ONE flat record range (their counts are prefix-summed first): the record loads of
// a pass are independent of each other, where a region-by-region walk paid two dependent round trips per region and pass
// (8 regions x 2 passes: 50-100 us per call, 0.4 ms of a 10.8-ms search).
// (round 4: 256 blocks -- one per CU -- instead of 128: every launch of a 1024-query search 25-30 % shorter, 220 -> 153 us per
// search; 512 blocks measured the same on 1024 queries and worse on 64.)
constexpr uint32_t kScatterBlocks = 256, kScatterMaxQueries = 8192, kScatterMaxPer = 64;

// RAW records {integer sum, row, query, 1} (scan_mfma16_dma.hip's i32-accumulator flavours, scan_mfma16_key.h) are turned
// into keyed ones in pass 1, in place: key computed, the threshold / padding-query / deletion tests applied; a record that
// fails gets the void query and is skipped from then on.
constexpr uint32_t kVoidQuery = 0xFFFFFFFFu;

__device__ __forceinline__ uint4 resolve_record(uint4* slot, const RawKeyArgs& rk) {
    uint4 rec = *slot;
    if (rec.w != 0u) {
        uint32_t key = 0;
        const bool ok = raw_record_key(rk, (int32_t)rec.x, rec.y, rec.z, key);
        rec = make_uint4(key, rec.y, ok ? rec.z : kVoidQuery, 0u);
        *slot = rec;
    }
    return rec;
}

__global__ void __launch_bounds__(1024) scatter_cand_kernel(uint4* blk_cand, uint32_t* blk_cnt, uint32_t blk_cap,
                                                             uint32_t nregions, uint64_t* cand, uint32_t* cnt, uint32_t cap,
                                                             uint32_t nq_pad, RawKeyArgs rk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);  // [nq_pad] records of this block per query, then their global base
    __shared__ uint32_t start_s[kScatterMaxPer + 1];     // flat offset of each of the block's regions
    const uint32_t per = (nregions + gridDim.x - 1) / gridDim.x;  // <= kScatterMaxPer (the launcher sizes the grid)
    const uint32_t r_lo = blockIdx.x * per, r_hi = min(nregions, r_lo + per);
    for (uint32_t i = threadIdx.x; i < nq_pad; i += 1024) hist[i] = 0;
    if (threadIdx.x < 64) {  // one wave: inclusive scan of the region counts
        const uint32_t b = r_lo + threadIdx.x;
        uint32_t incl = b < r_hi ? min(blk_cnt[b], blk_cap) : 0u;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if ((int)threadIdx.x >= off) incl += v;
        }
        start_s[threadIdx.x + 1] = incl;
        if (threadIdx.x == 0) start_s[0] = 0;
    }
    __syncthreads();
    const uint32_t nreg = r_hi > r_lo ? r_hi - r_lo : 0u, total = start_s[nreg];
    auto record = [&](uint32_t e) __attribute__((always_inline)) -> uint4* {
        uint32_t lo = 0, hi = nreg;  // the region with start_s[lo] <= e < start_s[lo + 1]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (start_s[mid] <= e) lo = mid;
            else hi = mid;
        }
        return blk_cand + (size_t)(r_lo + lo) * blk_cap + (e - start_s[lo]);
    };
    // pass 1: keys of the raw records, histogram
    for (uint32_t e = threadIdx.x; e < total; e += 1024) {
        const uint32_t q = resolve_record(record(e), rk).z;
        if (q != kVoidQuery) atomicAdd(&hist[q], 1u);
    }
    __syncthreads();
    // pass 2: reserve the ranges
    for (uint32_t i = threadIdx.x; i < nq_pad; i += 1024) {
        const uint32_t h = hist[i];
        hist[i] = h ? atomicAdd(&cnt[i], h) : 0u;
    }
    __syncthreads();
    // pass 3: store (a query's records take consecutive slots from its base, in arrival order of the LDS atomics;
    // the lists are unordered)
    for (uint32_t e = threadIdx.x; e < total; e += 1024) {
        const uint4 rec = *record(e);  // this thread resolved it in pass 1
        if (rec.z == kVoidQuery) continue;
        const uint32_t slot = atomicAdd(&hist[rec.z], 1u);
        if (slot < cap) cand[(size_t)rec.z * cap + slot] = ((uint64_t)rec.x << 32) | rec.y;
    }
    // re-arm the counters of this block's regions for the next phase (a scan block without work does not write its own)
    if (threadIdx.x < nreg) blk_cnt[r_lo + threadIdx.x] = 0u;
}

__global__ void __launch_bounds__(256) scatter_cand_simple_kernel(uint4* blk_cand, const uint32_t* blk_cnt, uint32_t blk_cap,
                                                                   uint64_t* cand, uint32_t* cnt, uint32_t cap, RawKeyArgs rk) {
    const uint32_t b = blockIdx.y;
    const uint32_t n = min(blk_cnt[b], blk_cap);
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < n; e += gridDim.x * 256u) {
        const uint4 rec = resolve_record(blk_cand + (size_t)b * blk_cap + e, rk);
        if (rec.z == kVoidQuery) continue;
        const uint32_t slot = atomicAdd(&cnt[rec.z], 1u);
        if (slot < cap) cand[(size_t)rec.z * cap + slot] = ((uint64_t)rec.x << 32) | rec.y;
    }
}

// ---- candidate compaction: keep each query's k best, publish the new threshold ----
// grid (nq); block 1024; LDS cap*8.  FINAL additionally formats the results.
__device__ __forceinline__ void write_result_b(uint64_t comp, uint32_t o, const CompactParams& p) {
    if (comp == kPadComposite) {
        p.out_scores[o] = pad_score(p.metric);
        p.out_indices[o] = ~0ull;
        if (p.out_raw) p.out_raw[o] = 0;
        return;
    }
    const uint32_t key = (uint32_t)(comp >> 32);
    float s;
    int32_t raw = 0;
    if (key_is_raw(p.dtype, p.metric)) {
        raw = raw_from_key(key, p.metric);
        s = p.metric == MVF_METRIC_L2 ? sqrtf((float)raw) : (float)raw;
    } else {
        s = score_from_key(key, p.metric);
    }
    p.out_scores[o] = s;
    p.out_indices[o] = p.ids ? p.ids[(uint32_t)comp] : p.index_base + (uint32_t)comp;
    if (p.out_raw) p.out_raw[o] = raw;
}

// After the sort the padding composites (the direct phase's slots of deleted rows) sit at the end: the live entries
// are the prefix in front of the first of them.
__device__ __forceinline__ uint32_t live_prefix(const uint64_t* buf, uint32_t m, int tid, uint32_t* live_s) {
    if (tid == 0) *live_s = 0;
    __syncthreads();
    for (uint32_t i = tid; i < m; i += 1024)
        if (buf[i] != kPadComposite && (i + 1 >= m || buf[i + 1] == kPadComposite)) *live_s = i + 1;
    __syncthreads();
    return *live_s;
}

template <bool FINAL>
__global__ void __launch_bounds__(1024) compact_kernel(CompactParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* buf = reinterpret_cast<uint64_t*>(smem);
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    __shared__ uint32_t live_s;
    const uint32_t raw_cnt = p.direct_cnt ? p.direct_cnt : p.cnt[q];  // direct phase: every row of it, no counter
    uint32_t m = raw_cnt < p.cap ? raw_cnt : p.cap;
    uint64_t* c = p.cand + (size_t)q * p.cap;
    {
        const uint32_t P2 = next_pow2(m < 2 ? 2 : m);
        for (uint32_t i = tid; i < P2; i += 1024) buf[i] = i < m ? c[i] : kPadComposite;
        __syncthreads();
        // (the usual list between two phases holds a few hundred entries -- k carried + ~k (g - 1) new: one per thread, in registers)
        if (P2 <= 1024) bitonic_sort_u64_reg<1024, 1>(buf, P2, tid);
        else bitonic_sort_u64<1024>(buf, P2, tid);
        m = live_prefix(buf, m, tid, &live_s);
    }
    const uint32_t keep = m < p.k ? m : p.k;
    if (FINAL) {
        for (uint32_t i = tid; i < p.k; i += 1024) write_result_b(i < keep ? buf[i] : kPadComposite, q * p.k + i, p);
    } else {
        for (uint32_t i = tid; i < keep; i += 1024) c[i] = buf[i];
    }
    if (tid == 0) {
        if (raw_cnt > p.cap) p.overflow[q] = 1u;  // survivors were dropped: the host repairs this query exactly
        p.cnt[q] = FINAL ? 0u : keep;
        p.tau[q] = (!FINAL && keep == p.k) ? (uint32_t)(buf[p.k - 1] >> 32) : kNanKey;
    }
}

// ---- approximate selection: margin-aware compaction ---------------------------------------------------------
// Keys are ord(s~) of an APPROXIMATE score with a proven bound |s~ - s| <= delta:
//   float L2 (f32 and f16 rows): s~ = qq + xx - 2 dot (GEMM form),   delta = eps * (qq + xxmax)
//   f16 rows, cosine:            s~ = dot~ / (|q| |x|),              delta = eps
//   f16 rows, inner product:     s~ = dot~,                          delta = eps * |q| * sqrt(xxmax)
// (f16 rows use ONE f16 query plane: |dot~ - dot| <= 2^-11 |q||x| by Cauchy-Schwarz, plus the f32 accumulation.)
// With v_k the k-th best s~ seen, the true k-th best exact value is no worse than v_k -/+ delta, so every true
// top-k row has s~ within 2 delta of v_k: keep all of those (not just k) and publish tau = ord(v_k -/+ 2 delta).
// The kept rows are re-scored exactly at the end (rescore_kernel).
__global__ void __launch_bounds__(1024) compact_margin_kernel(CompactParams p) {
    // The kept rows are re-scored and sorted by rescore_kernel, so nothing here needs the candidates in order: what is
    // needed is the k-th best approximate KEY (a radix select over the 32-bit keys: four 8-bit histogram passes in LDS)
    // and a filter.  (Round 1 sorted all cap entries: 55 us at cap 4096, 118 us at 8192, per phase.)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* buf = reinterpret_cast<uint64_t*>(smem);
    __shared__ uint32_t hist[256];
    __shared__ uint32_t live_s, keep_s, sel_prefix, sel_remaining;
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t raw_cnt = p.direct_cnt ? p.direct_cnt : p.cnt[q];  // direct phase: every row of it, no counter
    const uint32_t m = raw_cnt < p.cap ? raw_cnt : p.cap;
    uint64_t* c = p.cand + (size_t)q * p.cap;
    if (tid == 0) live_s = 0, keep_s = 0, sel_prefix = 0, sel_remaining = p.k;
    __syncthreads();
    uint32_t mylive = 0;
    for (uint32_t i = tid; i < m; i += 1024) {
        const uint64_t e = c[i];
        buf[i] = e;
        mylive += e != kPadComposite;  // padding = the direct phase's slots of deleted rows
    }
    for (int off = 32; off > 0; off >>= 1) mylive += __shfl_xor(mylive, off, 64);
    if ((tid & 63) == 0 && mylive) atomicAdd(&live_s, mylive);
    __syncthreads();
    const uint32_t live = live_s;
    uint32_t tkey = kNanKey;
    if (live >= p.k) {
        uint32_t mask = 0;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = sel_prefix;
            for (uint32_t i = tid; i < m; i += 1024) {
                const uint64_t e = buf[i];
                const uint32_t key = (uint32_t)(e >> 32);
                if (e != kPadComposite && ((key ^ prefix) & mask) == 0) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < 64) {  // one wave: bin of the k-th among the entries that share the prefix
                const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
                uint32_t incl = h0 + h1 + h2 + h3;
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t v = __shfl_up(incl, off, 64);
                    if (tid >= off) incl += v;
                }
                const uint32_t excl = incl - (h0 + h1 + h2 + h3), rem = sel_remaining;
                if (excl < rem && rem <= incl) {  // exactly one lane
                    uint32_t r = rem - excl, bin = 4 * tid;
                    if (r > h0) { r -= h0; bin++; if (r > h1) { r -= h1; bin++; if (r > h2) { r -= h2; bin++; } } }
                    sel_prefix = prefix | (bin << shift);
                    sel_remaining = r;
                }
            }
            mask |= 255u << shift;
            __syncthreads();
        }
        const float vk = score_from_key(sel_prefix, p.metric);  // the k-th best approximate score
        const float qn = p.qnorm[q];
        float thr;
        if (p.delta) thr = p.metric == MVF_METRIC_L2 ? vk + 2.0f * p.delta[q] : vk - 2.0f * p.delta[q];  // int8-shadow selection
        else if (p.metric == MVF_METRIC_L2 && p.l2_is_distance)
            thr = vk + 2.0f * (p.eps * sqrtf(p.xxmax[0]) + p.eps_acc * (qn + sqrtf(p.xxmax[0])));
        else if (p.metric == MVF_METRIC_L2) thr = vk + 2.0f * p.eps * (qn * qn + p.xxmax[0]);
        else if (p.metric == MVF_METRIC_COSINE) thr = vk - 2.0f * p.eps;
        else thr = vk - 2.0f * p.eps * qn * sqrtf(p.xxmax[0]);
        tkey = key_from_score(thr, p.metric);
        // a threshold refined after an earlier phase (launch_refine_tau) may be tighter than this list's own: both hold
        const uint32_t prev = p.tau[q];
        if (prev < tkey) tkey = prev;
    }
    // filter: every live entry whose key is within the margin (all of them while there is no threshold)
    const uint32_t keep_cap = p.cap / 2;
    const bool ordered = p.ntop != nullptr && live >= p.k;  // the k best (and their ties) first: refine scores those
    if (ordered) {
        const uint32_t kth = sel_prefix;
        for (uint32_t i = tid; i < m; i += 1024) {
            const uint64_t e = buf[i];
            if (e != kPadComposite && (uint32_t)(e >> 32) <= kth && (uint32_t)(e >> 32) <= tkey) {
                const uint32_t slot = atomicAdd(&keep_s, 1u);
                if (slot < keep_cap) c[slot] = e;
            }
        }
        __syncthreads();
        if (tid == 0) live_s = keep_s;  // (live is in a register already) the number of top entries
        __syncthreads();
        for (uint32_t i = tid; i < m; i += 1024) {
            const uint64_t e = buf[i];
            if (e != kPadComposite && (uint32_t)(e >> 32) > kth && (uint32_t)(e >> 32) <= tkey) {
                const uint32_t slot = atomicAdd(&keep_s, 1u);
                if (slot < keep_cap) c[slot] = e;
            }
        }
    } else {
        for (uint32_t i = tid; i < m; i += 1024) {
            const uint64_t e = buf[i];
            if (e != kPadComposite && (tkey == kNanKey || (uint32_t)(e >> 32) <= tkey)) {
                const uint32_t slot = atomicAdd(&keep_s, 1u);
                if (slot < keep_cap) c[slot] = e;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t keep = keep_s;
        bool over = raw_cnt > p.cap;
        if (keep > keep_cap) {  // too many near-ties to carry: this query is redone exactly with K1
            keep = keep_cap;
            over = true;
        }
        if (p.truncated_at && m >= p.truncated_at && keep_s >= live) over = true;  // the margin reaches past the cut
        if (over) p.overflow[q] = 1u;
        p.cnt[q] = keep;
        p.tau[q] = tkey;
        if (p.ntop) p.ntop[q] = ordered ? min(live_s, keep) : 0u;
    }
}

// ---- exact re-scoring of the kept candidates, final top-k ------------------------------------------------------
// Two kernels.  rescore_score_kernel: one WAVE per candidate row, four rows in flight per wave -- the score is
// recomputed from the caller's f32 query and the stored row with K1's formulas (sqrt(sum (q-x)^2); sum q x;
// sum q x / (sqrt(qq) sqrt(xx))) and the candidate's key is replaced in place.  grid (B, nq): block b of a query
// takes the 16-candidate slices b, b + B, ... (round 2 had one block per query walk all of them, four at a time:
// 0.46 ms for the ~700 candidates of an int8-selected query whatever the batch size -- a fifth of a 64-query search).
// rescore_select_kernel: one block per query sorts the re-scored candidates and formats the k best.
template <int METRIC, bool REFINE = false>
__global__ void __launch_bounds__(256) rescore_score_kernel(RescoreParams p, const uint32_t* ntop = nullptr, uint32_t* lkey = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t dim4 = (p.dim + 7u) & ~7u;  // zero-padded to a multiple of 8 (one f16 vector)
    float* qs = reinterpret_cast<float*>(smem);
    __shared__ float qq_part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t q = blockIdx.y;
    const uint32_t keep_cap = p.cap / 2;
    const uint32_t m = REFINE ? (ntop[q] >= p.k ? min(ntop[q], keep_cap) : 0u) : min(p.cnt[q], keep_cap);  // REFINE: the best k (+ ties) only
    if (blockIdx.x * 16u >= m) return;  // block-uniform
    float qq = 0.f;
    for (uint32_t e = tid; e < dim4; e += 256) {
        const float v = e < p.dim ? p.queries[(size_t)q * p.dim + e] : 0.f;
        qs[e] = v;
        qq = fmaf(v, v, qq);
    }
    if (METRIC == MVF_METRIC_COSINE) {
        for (int off = 32; off > 0; off >>= 1) qq += __shfl_xor(qq, off, 64);
        if (lane == 0) qq_part[wave] = qq;
    }
    __syncthreads();
    if (METRIC == MVF_METRIC_COSINE) qq = (qq_part[0] + qq_part[1]) + (qq_part[2] + qq_part[3]);
    uint64_t* c = p.cand + (size_t)q * p.cap;
    const uint32_t V = p.pitch / 16;
    const uint32_t tau_q = p.tau[q];
    const uint32_t done = (!REFINE && p.head_done && p.head_done[q] >= p.k) ? min(p.head_done[q], keep_cap) : 0u;
    for (uint32_t c0 = blockIdx.x * 16u; c0 < m; c0 += gridDim.x * 16u) {
        // this wave's four candidates: c0 + wave * 4 + u (a row past the end repeats the slice's first; not written)
        uint32_t r[4];
        const unsigned char* rp[4];
        float s[4], xx[4];
        bool skip[4], head[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t ci = c0 + (uint32_t)wave * 4u + u;
            const uint64_t ce = c[ci < m ? ci : c0];
            r[u] = (uint32_t)ce;
            // outside the (possibly refined) threshold: cannot be in the top-k; not fetched, not scored
            skip[u] = !REFINE && tau_q != kNanKey && (uint32_t)(ce >> 32) > tau_q;
            head[u] = !REFINE && ci < done;  // scored by the refinement of the final lists already: not fetched, not written
            rp[u] = p.rows + (size_t)r[u] * p.pitch;
            s[u] = 0.f;
            xx[u] = 0.f;
        }
        for (uint32_t v = lane; v < V; v += 64) {
            u32x4 x[4];
#pragma unroll
            for (int u = 0; u < 4; u++) x[u] = (skip[u] || head[u]) ? u32x4{0, 0, 0, 0} : *reinterpret_cast<const u32x4*>(rp[u] + (size_t)v * 16);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                auto term = [&](float qv, float xv) __attribute__((always_inline)) {
                    if (METRIC == MVF_METRIC_L2) {
                        const float t = qv - xv;
                        s[u] = fmaf(t, t, s[u]);
                    } else {
                        s[u] = fmaf(qv, xv, s[u]);
                        if (METRIC == MVF_METRIC_COSINE) xx[u] = fmaf(xv, xv, xx[u]);
                    }
                };
                if (p.dtype == MVF_DTYPE_FLOAT32) {
                    const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + v * 4);
#pragma unroll
                    for (int w = 0; w < 4; w++) term(qv[w], __uint_as_float(x[u][w]));
                } else {
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        term(qs[v * 8 + 2 * w], __half2float(__ushort_as_half((unsigned short)(x[u][w] & 0xFFFFu))));
                        term(qs[v * 8 + 2 * w + 1], __half2float(__ushort_as_half((unsigned short)(x[u][w] >> 16))));
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            for (int off = 32; off > 0; off >>= 1) {
                s[u] += __shfl_xor(s[u], off, 64);
                if (METRIC == MVF_METRIC_COSINE) xx[u] += __shfl_xor(xx[u], off, 64);
            }
            float sc = s[u];
            if (METRIC == MVF_METRIC_L2 && !REFINE) sc = sqrtf(s[u]);  // REFINE: the squared distance the selection works on
            if (METRIC == MVF_METRIC_COSINE) {
                const float den = sqrtf(qq) * sqrtf(xx[u]);
                sc = den > 0.0f ? s[u] / den : 0.0f;
            }
            const uint32_t ci = c0 + (uint32_t)wave * 4u + u;
            if (REFINE) {
                if (lane == 0 && ci < m) {
                    atomicMax(&lkey[q], key_from_score(sc, METRIC));  // the worst of the exact scores
                    if (p.write_head) c[keep_cap + ci] = ((uint64_t)key_from_score(METRIC == MVF_METRIC_L2 ? sqrtf(s[u]) : sc, METRIC) << 32) | r[u];
                }
            } else if (lane == 0 && ci < m && !head[u]) {  // (out of place: the list's free upper half -- see rescore_wave_kernel)
                c[keep_cap + ci] = skip[u] ? kPadComposite : ((uint64_t)key_from_score(sc, METRIC) << 32) | r[u];
            }
        }
    }
}

// ---- the same scoring, WAVE-AUTONOMOUS (round 4) -----------------------------------------------------------------------
// rescore_score_kernel above stages the query in LDS per block (a global read, a barrier, THEN the rows) and its blocks live
// for one or two 16-row rounds: the refinement passes ran at 2.5 TB/s and the final pass at 3.8 where the same access shape --
// one wave per row, four rows in flight, random 3-KB rows out of 30 GB -- holds 6.0 TB/s in a bare kernel
// (scripts/probe_gather.hip, profiles/r04_gather_random_rows.txt).  Here nothing is shared between waves: a persistent wave
// takes (query, 64-candidate slice) items; every lane holds ONE candidate of the slice and its own share of the query in
// registers (the vectors v = lane + 64 j it will meet in every row: L2-resident reads, no LDS, no barrier); the LIVE candidates
// (inside the -- possibly refined -- threshold) are compacted with a ballot and scored four at a time, all twelve row loads of
// a round in flight together.  Per-lane element assignment and the xor reduction are rescore_score_kernel's.
// VPL = 16-byte row vectors per lane (rows of up to 64 VPL vectors: 768 x f32 = 3, 1024 x f16 = 2); longer rows keep the
// block kernel.
template <int METRIC, bool REFINE, bool F16ROWS, int VPL>
__global__ void __launch_bounds__(256) rescore_wave_kernel(RescoreParams p, uint32_t nq, uint32_t slices, uint32_t split, const uint32_t* ntop,
                                                            uint32_t* lkey) {
    constexpr int EPV = F16ROWS ? 8 : 4;  // query floats per row vector
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    const uint32_t keep_cap = p.cap / 2;
    const uint32_t V = p.pitch / 16;
    // items in SLICE-major order (item = slice * nq + query): every wave meets low slices -- the ones that exist -- first, the
    // empty tail of the longest possible list costs each wave a few reads of cnt[q].  (Query-major order with a power-of-two
    // slice count gave wave w the slice w % slices of every query it met: two thirds of the waves never found work.)
    // small batches: `split` waves share a slice -- every one of them reads its 64 candidates and forms the same ballot, wave
    // `part` scores the rounds part, part + split, ... (64 queries: 192 refinement items became 3072)
    for (uint32_t item = wave; item < nq * slices * split; item += nwaves) {
        const uint32_t q = item % nq, sl0 = (item / nq) % slices, part = item / (nq * slices);
        const uint32_t m = REFINE ? (ntop[q] >= p.k ? min(ntop[q], keep_cap) : 0u) : min(p.cnt[q], keep_cap);
        if (sl0 * 64u >= m) continue;  // wave-uniform
        // this lane's share of the query (zero beyond the dimension: the rows' padding is zero too, but 0 x NaN is not)
        float qf[VPL][EPV];
        float qq = 0.f;
        const float* qp = p.queries + (size_t)q * p.dim;
#pragma unroll
        for (int j = 0; j < VPL; j++)
#pragma unroll
            for (int w = 0; w < EPV; w++) {
                const uint32_t e = ((uint32_t)lane + 64u * j) * EPV + w;
                qf[j][w] = e < p.dim ? qp[e] : 0.f;
                if (METRIC == MVF_METRIC_COSINE) qq = fmaf(qf[j][w], qf[j][w], qq);
            }
        if (METRIC == MVF_METRIC_COSINE)
            for (int off = 32; off > 0; off >>= 1) qq += __shfl_xor(qq, off, 64);
        const uint32_t tau_q = p.tau[q];
        const uint32_t done = (!REFINE && p.head_done && p.head_done[q] >= p.k) ? min(p.head_done[q], keep_cap) : 0u;
        uint64_t* c = p.cand + (size_t)q * p.cap;
        uint32_t worst = 0;  // REFINE: the worst exact key of this wave's rows
        for (uint32_t sl = sl0; sl * 64u < m; sl += slices) {
            const uint32_t ci = sl * 64u + (uint32_t)lane;
            const uint64_t ce = ci < m ? c[ci] : kPadComposite;
            // outside the (possibly refined) threshold: cannot be in the top-k; not fetched, not scored
            const bool head = !REFINE && ci < done;  // scored by the refinement of the final lists already: its exact composite is in place
            const bool live = ci < m && !head && (REFINE || tau_q == kNanKey || (uint32_t)(ce >> 32) <= tau_q);
            if (!REFINE && part == 0 && ci < m && !live && !head) c[keep_cap + ci] = kPadComposite;
            const uint32_t myrow = (uint32_t)ce;
            unsigned long long mask = __builtin_amdgcn_ballot_w64(live);
            for (uint32_t rnd = 0; mask; rnd++) {  // wave-uniform
                if (split > 1 && rnd % split != part) {  // another wave's round: drop its four candidates
#pragma unroll
                    for (int u = 0; u < 4; u++) mask &= mask - 1;
                    continue;
                }
                int l[4];
                uint32_t r[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    ok[u] = mask != 0;
                    l[u] = ok[u] ? __builtin_ctzll(mask) : l[0];
                    if (ok[u]) mask &= mask - 1;
                    r[u] = (uint32_t)__builtin_amdgcn_readlane((int)myrow, l[u]);
                }
                u32x4 x[4][VPL];
#pragma unroll
                for (int j = 0; j < VPL; j++) {
                    const uint32_t v = (uint32_t)lane + 64u * j;
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        x[u][j] = (ok[u] && v < V) ? *reinterpret_cast<const u32x4*>(p.rows + (size_t)r[u] * p.pitch + (size_t)v * 16) : u32x4{0, 0, 0, 0};
                }
                float s[4] = {0.f, 0.f, 0.f, 0.f}, xx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < VPL; j++)
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        auto term = [&](float qv, float xv) __attribute__((always_inline)) {
                            if (METRIC == MVF_METRIC_L2) {
                                const float t = qv - xv;
                                s[u] = fmaf(t, t, s[u]);
                            } else {
                                s[u] = fmaf(qv, xv, s[u]);
                                if (METRIC == MVF_METRIC_COSINE) xx[u] = fmaf(xv, xv, xx[u]);
                            }
                        };
                        if constexpr (!F16ROWS) {
#pragma unroll
                            for (int w = 0; w < 4; w++) term(qf[j][w], __uint_as_float(x[u][j][w]));
                        } else {
#pragma unroll
                            for (int w = 0; w < 4; w++) {
                                term(qf[j][2 * w], __half2float(__ushort_as_half((unsigned short)(x[u][j][w] & 0xFFFFu))));
                                term(qf[j][2 * w + 1], __half2float(__ushort_as_half((unsigned short)(x[u][j][w] >> 16))));
                            }
                        }
                    }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    for (int off = 32; off > 0; off >>= 1) {
                        s[u] += __shfl_xor(s[u], off, 64);
                        if (METRIC == MVF_METRIC_COSINE) xx[u] += __shfl_xor(xx[u], off, 64);
                    }
                    float sc = s[u];
                    if (METRIC == MVF_METRIC_L2 && !REFINE) sc = sqrtf(s[u]);  // REFINE: the squared distance the selection works on
                    if (METRIC == MVF_METRIC_COSINE) {
                        const float den = sqrtf(qq) * sqrtf(xx[u]);
                        sc = den > 0.0f ? s[u] / den : 0.0f;
                    }
                    const uint32_t key = key_from_score(sc, METRIC);
                    if (REFINE) {
                        if (ok[u]) worst = max(worst, key);
                        if (p.write_head && ok[u] && lane == l[u])
                            c[keep_cap + ci] = ((uint64_t)key_from_score(METRIC == MVF_METRIC_L2 ? sqrtf(s[u]) : sc, METRIC) << 32) | myrow;
                    } else if (ok[u] && lane == l[u]) {
                        c[keep_cap + ci] = ((uint64_t)key << 32) | myrow;
                    }
                }
            }
        }
        if (REFINE && lane == 0 && worst) atomicMax(&lkey[q], worst);  // one atomic per wave and query
    }
}

template <int METRIC, bool REFINE>
bool launch_rescore_wave(const RescoreParams& p, uint32_t nq, uint32_t slices, const uint32_t* ntop, uint32_t* lkey, hipStream_t s) {
    const uint32_t V = p.pitch / 16, vpl = (V + 63u) / 64u;
    if (vpl == 0 || vpl > 4u || (p.dtype != MVF_DTYPE_FLOAT32 && p.dtype != MVF_DTYPE_FLOAT16)) return false;  // longer rows: the block kernel
    // waves that share a slice: enough items to fill the chip with a small batch (a slice holds at most 16 rounds).  The final
    // pass too since round 5: it writes the exact keys into the list's free UPPER half (a query keeps at most cap / 2 candidates;
    // rescore_select_kernel reads them there), so a wave that starts late still finds the approximate keys its ballot needs.
    // (In place, the final pass could not split: the k best sit at the head of the list -- the refinement's order -- so its first
    // two slices were 16 dependent rounds of random row fetches each, 40 us whatever the batch: 16 queries on 1M x 768 spent a
    // seventh of the search there.)  Items to aim for: 8192 (final pass, us, 16 / 64 / 128 / 256 queries: 2048 -> 25 / 43 / 53 / 65; 4096 -> 19 / 43 / 51 / 63;
    // 8192 -> 13 / 25 / 51 / 65; 16384 -> 18 / 28 / 54 / 64; 65536 -> 18 / 48 / 70 / 72: profiles/r05_k2_walk_and_phase_costs.txt 13).
    const uint32_t split = std::max(1u, std::min(16u, 8192u / std::max(1u, nq * slices)));
    const uint32_t items = nq * slices * split;
    const dim3 grid(std::max(1u, std::min((items + 3u) / 4u, 2048u)));
    const bool h = p.dtype == MVF_DTYPE_FLOAT16;
#define MVF_RW(VPL_)                                                                                                          \
    do {                                                                                                                      \
        if (h) hipLaunchKernelGGL((rescore_wave_kernel<METRIC, REFINE, true, VPL_>), grid, dim3(256), 0, s, p, nq, slices, split, ntop, lkey);  \
        else hipLaunchKernelGGL((rescore_wave_kernel<METRIC, REFINE, false, VPL_>), grid, dim3(256), 0, s, p, nq, slices, split, ntop, lkey);   \
    } while (0)
    switch (vpl) {
    case 1: MVF_RW(1); break;
    case 2: MVF_RW(2); break;
    case 3: MVF_RW(3); break;
    default: MVF_RW(4); break;
    }
#undef MVF_RW
    return true;
}

// One thread per query: tau[q] = the tighter of itself and ord(L -/+ delta), L = the worst exact score of its k best
// approximate candidates (rescore_score_kernel<., true>); re-arms lkey.  The 2 % on delta covers the f32 rounding of the
// exact scores themselves (the final ranking is by those f32 values).
__global__ void __launch_bounds__(256) refine_tau_kernel(uint32_t* tau, const uint32_t* ntop, uint32_t* lkey, const float* delta,
                                                        int metric, uint32_t k, uint32_t nq) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= nq) return;
    if (ntop[q] >= k) {
        const float L = score_from_key(lkey[q], metric), d = delta[q] * 1.02f;
        const uint32_t key = key_from_score(metric == MVF_METRIC_L2 ? L + d : L - d, metric);
        if (key < tau[q]) tau[q] = key;
    }
    lkey[q] = 0u;
}

__global__ void __launch_bounds__(1024) rescore_select_kernel(RescoreParams p, int metric) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* buf = reinterpret_cast<uint64_t*>(smem);
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t keep_cap = p.cap / 2;
    const uint32_t m0 = min(p.cnt[q], keep_cap);
    const uint64_t* c = p.cand + (size_t)q * p.cap + keep_cap;  // the scoring pass's output: the list's upper half
    // only the candidates the scoring pass kept (it pads the ones outside the refined threshold: four fifths of an
    // int8-selected list) are sorted: ~300 of ~1500 on cfg3, a 512-entry network instead of a 2048-entry one
    __shared__ uint32_t live_s;
    if (tid == 0) live_s = 0;
    __syncthreads();
    for (uint32_t i = tid; i < m0; i += 1024) {
        const uint64_t e = c[i];
        if (e != kPadComposite) buf[atomicAdd(&live_s, 1u)] = e;
    }
    __syncthreads();
    const uint32_t m = live_s;
    const uint32_t P2 = next_pow2(m < 2 ? 2 : m);
    for (uint32_t i = m + tid; i < P2; i += 1024) buf[i] = kPadComposite;
    __syncthreads();
    if (P2 <= 1024) bitonic_sort_u64_reg<1024, 1>(buf, P2, tid);
    else bitonic_sort_u64<1024>(buf, P2, tid);
    for (uint32_t i = tid; i < p.k; i += 1024) {
        const uint32_t o = q * p.k + i;
        const uint64_t comp = i < m ? buf[i] : kPadComposite;
        if (comp == kPadComposite) {
            p.out_scores[o] = pad_score(metric);
            p.out_indices[o] = ~0ull;
        } else {
            p.out_scores[o] = score_from_key((uint32_t)(comp >> 32), metric);
            p.out_indices[o] = p.ids ? p.ids[(uint32_t)comp] : p.index_base + (uint32_t)comp;
        }
        if (p.out_raw) p.out_raw[o] = 0;
    }
    if (tid == 0) {
        p.cnt[q] = 0;
        p.tau[q] = kNanKey;
    }
}

}  // namespace

size_t scan_mfma_lds_bytes() { return kLdsBytes; }

hipError_t launch_scan_mfma_f32(const BatchParams& p, int metric, int num_cus, int force_persistent, hipStream_t s) {
    // Grid: a multiple of 8 (one lane set per XCD).  The kernel is written persistent, but by default it is
    // launched with the FULL grid (every block owns exactly one tile): the dispatcher then starts the query
    // tiles of one corpus tile together, and the corpus tile is served from L2 to 7 of them (FETCH_SIZE 30 GB per
    // 24 GB algorithmic).  Truly persistent blocks (2 per CU) drift apart in time: same speed (+0.5 %), but HBM
    // traffic 102 GB — wasted re-reads.
    const uint32_t total = ((p.ntiles + 7) / 8) * p.mtiles * 8;
    uint32_t nls = std::max(1u, (uint32_t)num_cus * 2u / 8u);
    if (nls > p.mtiles) nls -= nls % p.mtiles;
    const bool persistent = force_persistent > 0;  // MVF_K2_PERSISTENT (read once per handle)
    const dim3 grid(persistent ? std::min(total, nls * 8u) : total);
    // > 64 KiB of dynamic LDS needs the attribute; it is per device, and one process may drive several devices
    const void* fn = metric == MVF_METRIC_COSINE ? reinterpret_cast<const void*>(&scan_mfma_f32_kernel<MVF_METRIC_COSINE>)
                     : metric == MVF_METRIC_L2   ? reinterpret_cast<const void*>(&scan_mfma_f32_kernel<MVF_METRIC_L2>)
                                                 : reinterpret_cast<const void*>(&scan_mfma_f32_kernel<MVF_METRIC_INNER_PRODUCT>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) return e;
    if (metric == MVF_METRIC_COSINE)
        hipLaunchKernelGGL(scan_mfma_f32_kernel<MVF_METRIC_COSINE>, grid, dim3(256), kLdsBytes, s, p);
    else if (metric == MVF_METRIC_L2)
        hipLaunchKernelGGL(scan_mfma_f32_kernel<MVF_METRIC_L2>, grid, dim3(256), kLdsBytes, s, p);
    else
        hipLaunchKernelGGL(scan_mfma_f32_kernel<MVF_METRIC_INNER_PRODUCT>, grid, dim3(256), kLdsBytes, s, p);
    return hipGetLastError();
}

hipError_t launch_prep_queries(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t KP, float* qmat,
                               float* qnorm, hipStream_t s) {
    hipLaunchKernelGGL(prep_queries_kernel, dim3(nq_pad), dim3(256), 0, s, q, nq, nq_pad, dim, KP, qmat, qnorm);
    return hipGetLastError();
}

hipError_t launch_row_norms_f32(const unsigned char* rows, uint32_t n, uint32_t pitch, float* xnorm, float* xx2,
                                float* xxmax, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 3) / 4, 256u * 8u);
    hipLaunchKernelGGL(row_norms_f32_kernel, dim3(blocks), dim3(256), 0, s, rows, n, pitch, pitch / 16, xnorm, xx2, xxmax);
    return hipGetLastError();
}

uint32_t scatter_rearm_max_queries() { return kScatterMaxQueries; }

hipError_t launch_scatter_cand(const Batch16Params& p, uint32_t nblocks, int metric, int dtype, hipStream_t s) {
    if (!p.blk_cand || nblocks == 0) return hipSuccess;
    const RawKeyArgs rk = raw_key_args(p, metric, dtype);
    if (p.nq_pad <= kScatterMaxQueries) {
        const size_t lds = (size_t)p.nq_pad * 4;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scatter_cand_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const uint32_t grid = std::max(std::min(kScatterBlocks, nblocks), (nblocks + kScatterMaxPer - 1) / kScatterMaxPer);
        hipLaunchKernelGGL(scatter_cand_kernel, dim3(grid), dim3(1024), lds, s, p.blk_cand, p.blk_cnt, p.blk_cap, nblocks, p.cand,
                           p.cnt, p.cap, p.nq_pad, rk);
    } else {
        hipLaunchKernelGGL(scatter_cand_simple_kernel, dim3(4, nblocks), dim3(256), 0, s, p.blk_cand, p.blk_cnt, p.blk_cap, p.cand,
                           p.cnt, p.cap, rk);
    }
    return hipGetLastError();
}

hipError_t launch_compact_margin(const CompactParams& p, uint32_t nq, hipStream_t s) {
    if ((size_t)p.cap * 8 > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&compact_margin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)((size_t)p.cap * 8));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(compact_margin_kernel, dim3(nq), dim3(1024), (size_t)p.cap * 8, s, p);
    return hipGetLastError();
}

hipError_t launch_rescore(const RescoreParams& p, int metric, uint32_t nq, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    {
        const uint32_t sl = (p.cap / 2 + 63u) / 64u;  // a query keeps at most cap / 2 candidates; empty slices cost a read of cnt[q]
        const bool done = metric == MVF_METRIC_L2       ? launch_rescore_wave<MVF_METRIC_L2, false>(p, nq, sl, nullptr, nullptr, s)
                          : metric == MVF_METRIC_COSINE ? launch_rescore_wave<MVF_METRIC_COSINE, false>(p, nq, sl, nullptr, nullptr, s)
                                                        : launch_rescore_wave<MVF_METRIC_INNER_PRODUCT, false>(p, nq, sl, nullptr, nullptr, s);
        if (done) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(rescore_select_kernel, dim3(nq), dim3(1024), (size_t)(p.cap / 2) * 8, s, p, metric);
            return hipGetLastError();
        }
    }
    const size_t lds = (size_t)((p.dim + 7u) & ~7u) * 4;
    // blocks per query: enough to fill the chip with a small batch, few enough that a block's staged query serves
    // several 16-candidate slices with a large one
    const uint32_t slices = (p.cap / 2 + 15u) / 16u;
    const uint32_t B = std::min(slices, std::max(8u, std::min(64u, 2048u / nq)));
    const dim3 grid(B, nq);
    if (metric == MVF_METRIC_L2) hipLaunchKernelGGL(rescore_score_kernel<MVF_METRIC_L2>, grid, dim3(256), lds, s, p);
    else if (metric == MVF_METRIC_COSINE) hipLaunchKernelGGL(rescore_score_kernel<MVF_METRIC_COSINE>, grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL(rescore_score_kernel<MVF_METRIC_INNER_PRODUCT>, grid, dim3(256), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rescore_select_kernel, dim3(nq), dim3(1024), (size_t)(p.cap / 2) * 8, s, p, metric);
    return hipGetLastError();
}

hipError_t launch_refine_tau(const RescoreParams& p, int metric, uint32_t nq, const uint32_t* ntop, uint32_t* lkey,
                             const float* delta, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    {
        const uint32_t sl = (p.k + 63u) / 64u + 1u;  // k best + ties (a longer head is walked in further rounds of the same waves)
        const bool done = metric == MVF_METRIC_L2       ? launch_rescore_wave<MVF_METRIC_L2, true>(p, nq, sl, ntop, lkey, s)
                          : metric == MVF_METRIC_COSINE ? launch_rescore_wave<MVF_METRIC_COSINE, true>(p, nq, sl, ntop, lkey, s)
                                                        : launch_rescore_wave<MVF_METRIC_INNER_PRODUCT, true>(p, nq, sl, ntop, lkey, s);
        if (done) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(refine_tau_kernel, dim3((nq + 255u) / 256u), dim3(256), 0, s, p.tau, ntop, lkey, delta, metric, p.k, nq);
            return hipGetLastError();
        }
    }
    const size_t lds = (size_t)((p.dim + 7u) & ~7u) * 4;
    const uint32_t slices = (p.k + 15u) / 16u + 1u;  // k best + a few ties
    const dim3 grid(std::min(slices, 64u), nq);
    if (metric == MVF_METRIC_L2) hipLaunchKernelGGL((rescore_score_kernel<MVF_METRIC_L2, true>), grid, dim3(256), lds, s, p, ntop, lkey);
    else if (metric == MVF_METRIC_COSINE)
        hipLaunchKernelGGL((rescore_score_kernel<MVF_METRIC_COSINE, true>), grid, dim3(256), lds, s, p, ntop, lkey);
    else hipLaunchKernelGGL((rescore_score_kernel<MVF_METRIC_INNER_PRODUCT, true>), grid, dim3(256), lds, s, p, ntop, lkey);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(refine_tau_kernel, dim3((nq + 255u) / 256u), dim3(256), 0, s, p.tau, ntop, lkey, delta, metric, p.k, nq);
    return hipGetLastError();
}

hipError_t launch_compact(const CompactParams& p, uint32_t nq, bool final_stage, hipStream_t s) {
    const size_t lds = (size_t)p.cap * 8;
    if (final_stage) hipLaunchKernelGGL(compact_kernel<true>, dim3(nq), dim3(1024), lds, s, p);
    else hipLaunchKernelGGL(compact_kernel<false>, dim3(nq), dim3(1024), lds, s, p);
    return hipGetLastError();
}

}  // namespace mvf
