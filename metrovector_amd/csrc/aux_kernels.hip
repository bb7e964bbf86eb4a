// aux_kernels.hip — K3 (candidate-list merge + result formatting), the
// cross-shard merge C1's device half, and the synthetic generators.
//
// K3 replaces the tail of reference examples/similarity_search.rs:172-173
// (heap.into_iter().collect(); sort_by(score)) — here a bitonic sort of u64
// composites (order key << 32 | local row) in LDS.

#include "aux_kernels.h"
#include "bitonic.h"
#include "mvf_common.h"

#include <hip/hip_fp16.h>

#include <algorithm>

namespace mvf {
namespace {

__device__ __forceinline__ void write_result(uint64_t comp, size_t o, const SelectParams& p) {
    const uint32_t key = (uint32_t)(comp >> 32);
    if (comp == kPadComposite) {
        p.out_scores[o] = pad_score(p.metric);
        p.out_indices[o] = ~0ull;
        if (p.out_raw) p.out_raw[o] = 0;
        return;
    }
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

// K3 final select.  grid (nq); block 1024; dynamic LDS P*8 + 16.
// Input: nlists sorted lists per query (one per scan block).
//   1. threshold: the k-th smallest KEY among the lists' first `heads` entries is the k-th smallest of a subset of
//      the corpus, hence >= the true k-th best.  With the top-k spread over many lists it is almost exact.  Found by a
//      radix select over the 32-bit keys (four 8-bit histogram passes in LDS; round 1 sorted all heads: with the final
//      sort below that was 30 of the kernel's 41 us, 1 % of a 10M x 768 scan and most of a small corpus' search);
//   2. every list's prefix with a key <= that is appended to LDS (one thread per list, usually 0-2 entries each);
//   3. the survivors are sorted -- up to 1024 of them by RANK COUNTING (one per thread: its rank is the number of
//      smaller survivors, read as LDS broadcasts; composites are distinct), more by the bitonic network -- and the
//      first k formatted.
// If the survivors overflow LDS (adversarial clustering, mass ties) the lists are folded group by group into a running
// top-k instead.
__global__ void __launch_bounds__(1024) select_final_kernel(SelectParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* buf = reinterpret_cast<uint64_t*>(smem);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(buf + p.P);
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel_prefix, sel_remaining;
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    uint32_t qout = q;  // the query whose result row this block writes
    if (p.redo_list) {
        if (p.redo_base + q >= *p.redo_cnt) return;  // block-uniform: nothing (more) to repair
        qout = p.redo_list[p.redo_base + q];
    }
    const uint64_t* lists = p.lists + (size_t)q * p.nlists * p.kcap;

    // 1. threshold from the list heads (H >= k by the choice of `heads`, padding entries included: they carry the
    // largest key, and a threshold that lands on one means "everything")
    const uint32_t H = p.nlists * p.heads;
    for (uint32_t i = tid; i < H; i += 1024) buf[i] = lists[(size_t)(i / p.heads) * p.kcap + (i % p.heads)];
    if (tid == 0) *cnt = 0, sel_prefix = 0, sel_remaining = p.k;
    __syncthreads();
    uint64_t tau = kPadComposite;
    if (H >= p.k) {
        // the k-th smallest 64-bit composite of the heads: radix select over the key (high word), then -- only when
        // several heads share that key -- over the row (low word) among those; a unique k-th key is simply looked up
        __shared__ uint32_t sel_count;
        __shared__ uint64_t sel_tau;
        uint32_t key_k = 0;
        for (int word = 1; word >= 0; word--) {
            uint32_t mask = 0;
            if (tid == 0) sel_prefix = 0;
            __syncthreads();
            for (int shift = 24; shift >= 0; shift -= 8) {
                if (tid < 256) hist[tid] = 0;
                __syncthreads();
                const uint32_t prefix = sel_prefix;
                for (uint32_t i = tid; i < H; i += 1024) {
                    const uint64_t e = buf[i];
                    const uint32_t w = word ? (uint32_t)(e >> 32) : (uint32_t)e;
                    if ((word || (uint32_t)(e >> 32) == key_k) && ((w ^ prefix) & mask) == 0) atomicAdd(&hist[(w >> shift) & 255u], 1u);
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
                        uint32_t r = rem - excl, bin = 4 * tid, hb = h0;
                        if (r > h0) { r -= h0; bin++; hb = h1; if (r > h1) { r -= h1; bin++; hb = h2; if (r > h2) { r -= h2; bin++; hb = h3; } } }
                        sel_prefix = prefix | (bin << shift);
                        sel_remaining = r;
                        sel_count = hb;
                    }
                }
                mask |= 255u << shift;
                __syncthreads();
            }
            const uint32_t found = sel_prefix, found_count = sel_count;
            __syncthreads();  // everyone has read the selection before the next word's passes reset it
            if (word) {
                key_k = found;
                if (found_count == 1) {  // block-uniform: the k-th key is unique among the heads
                    for (uint32_t i = tid; i < H; i += 1024)
                        if ((uint32_t)(buf[i] >> 32) == key_k) sel_tau = buf[i];
                    __syncthreads();
                    break;
                }
            } else {
                if (tid == 0) sel_tau = ((uint64_t)key_k << 32) | found;
                __syncthreads();
            }
        }
        tau = sel_tau;
    }
    __syncthreads();

    // 2. gather every list's prefix <= tau
    for (uint32_t l = tid; l < p.nlists; l += 1024) {
        const uint64_t* li = lists + (size_t)l * p.kcap;
        for (uint32_t i = 0; i < p.kcap; i++) {
            const uint64_t c = li[i];
            if (c > tau || c == kPadComposite) break;
            const uint32_t slot = atomicAdd(cnt, 1u);
            if (slot < p.P) buf[slot] = c;
        }
    }
    __syncthreads();
    uint32_t m = *cnt;
    __syncthreads();

    if (m <= 1024) {  // rank counting: one survivor per thread
        const uint64_t mine = (uint32_t)tid < m ? buf[tid] : kPadComposite;
        uint32_t rank = 0;
        for (uint32_t i = 0; i < m; i++) rank += buf[i] < mine ? 1u : 0u;
        __syncthreads();
        if ((uint32_t)tid < m) buf[rank] = mine;
        __syncthreads();
    } else if (m <= p.P) {
        const uint32_t P2 = next_pow2(m);
        for (uint32_t i = m + tid; i < P2; i += 1024) buf[i] = kPadComposite;
        __syncthreads();
        bitonic_sort_u64<1024>(buf, P2, tid);
    } else {
        // overflow: fold groups of lists into a running top-k (always fits: k + F*kcap <= P)
        const uint32_t F = (p.P - p.k) / p.kcap;
        uint32_t kcur = 0;
        for (uint32_t l0 = 0; l0 < p.nlists; l0 += F) {
            const uint32_t nl = min(F, p.nlists - l0);
            __syncthreads();
            for (uint32_t i = tid; i < nl * p.kcap; i += 1024) buf[kcur + i] = lists[(size_t)l0 * p.kcap + i];
            const uint32_t tot = kcur + nl * p.kcap;
            const uint32_t P2 = next_pow2(tot < 2 ? 2 : tot);
            for (uint32_t i = tot + tid; i < P2; i += 1024) buf[i] = kPadComposite;
            __syncthreads();
            bitonic_sort_u64<1024>(buf, P2, tid);
            kcur = tot < p.k ? tot : p.k;
        }
        m = kcur;
    }
    if (p.delta) {  // margin mode: everything within 2 delta of the k-th best, from every list
        __shared__ uint32_t cut_s;
        uint32_t tkey = kNanKey;  // fewer than k rows in all: everything is a candidate
        if (m >= p.margin_rank) {
            const float vk = score_from_key((uint32_t)(buf[p.margin_rank - 1] >> 32), p.metric), d = 2.0f * p.delta[q];
            tkey = key_from_score(p.metric == MVF_METRIC_L2 ? vk + d : vk - d, p.metric);
        }
        __syncthreads();
        if (tid == 0) *cnt = 0, cut_s = 0;
        __syncthreads();
        for (uint32_t l = tid; l < p.nlists; l += 1024) {
            const uint64_t* li = lists + (size_t)l * p.kcap;
            for (uint32_t i = 0; i < p.k; i++) {
                const uint64_t c = li[i];
                if (c == kPadComposite || (tkey != kNanKey && (uint32_t)(c >> 32) > tkey)) break;
                const uint32_t slot = atomicAdd(cnt, 1u);
                if (slot < p.keep_cap) p.out_cand[(size_t)q * p.cand_cap + slot] = c;
                if (i + 1 == p.k) cut_s = 1;  // the block kept k rows and the last of them is still inside the bound
            }
        }
        __syncthreads();
        if (tid == 0) {
            const uint32_t inside = *cnt;
            if (inside > p.keep_cap || cut_s) p.out_overflow[q] = 1u;
            p.out_cnt[q] = inside < p.keep_cap ? inside : p.keep_cap;
            p.out_tau[q] = tkey;
        }
        return;
    }
    if (p.out_cand) {
        const uint32_t keep = m < p.k ? m : p.k;
        for (uint32_t i = tid; i < keep; i += 1024) p.out_cand[(size_t)q * p.cand_cap + i] = buf[i];
        if (tid == 0) p.out_cnt[q] = keep;
        return;
    }
    const uint32_t stride = p.out_stride ? p.out_stride : p.k;
    for (uint32_t i = tid; i < p.k; i += 1024) write_result(i < m ? buf[i] : kPadComposite, (size_t)qout * stride + p.out_offset + i, p);
    if (p.out_floor1 && tid == 0) p.out_floor1[qout] = m >= p.k ? buf[p.k - 1] + 1ull : ~0ull;
    if (p.gather_out) {  // block-uniform: the k rows behind the k results (buf[0 .. m) is the sorted list; one wave per row)
        const uint32_t lane = (uint32_t)tid & 63u, wave = (uint32_t)tid >> 6;
        const uint32_t rb = p.gather_row_bytes;
        const bool wide = ((rb | (uint32_t)reinterpret_cast<uintptr_t>(p.gather_out)) & 15u) == 0;  // stored rows start on 16-byte pitches
        for (uint32_t i = wave; i < p.k; i += 16) {
            const bool ok = i < m;
            const unsigned char* src = p.gather_rows + (size_t)(ok ? (uint32_t)buf[i] : 0u) * p.gather_pitch;
            unsigned char* dst = p.gather_out + ((size_t)qout * stride + p.out_offset + i) * rb;
            if (wide) {
                for (uint32_t b = lane; b < rb / 16; b += 64)
                    reinterpret_cast<uint4*>(dst)[b] = ok ? reinterpret_cast<const uint4*>(src)[b] : uint4{0, 0, 0, 0};
            } else {
                for (uint32_t b = lane; b < rb; b += 64) dst[b] = ok ? src[b] : (unsigned char)0;
            }
        }
    }
    if (p.done_flag) {  // block-uniform: tell the waiting host call (results in pinned host memory) that everything is there
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            bool last = true;
            if (gridDim.x > 1) {
                const uint32_t t = __hip_atomic_fetch_add(p.done_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                last = t + 1 == gridDim.x;
                if (last) __hip_atomic_store(p.done_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (last) __hip_atomic_store(p.done_flag, p.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The tail of a search that ranked the WHOLE shard (api.hip: search_sorted_k): `sorted` holds the n rank entries of one query in
// result order (deleted rows dead, hence last); the first k become the query's result row at
// out[out_base ...], entries beyond the rows that exist the padding result.  Only metric / dtype / index_base / ids / out_* / k
// of the parameter block are read.
__global__ void __launch_bounds__(256) write_sorted_kernel(SelectParams p, const uint64_t* sorted, uint32_t n, size_t out_base) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < p.k; i += (size_t)gridDim.x * 256)
        write_result(i < n ? composite_of_rank_entry(sorted[i]) : kPadComposite, out_base + i, p);
}

// The K2 compactions flag queries whose candidate budget overflowed (overflow[q] != 0).  One block turns the flags into
// a dense list for the repair launches and clears them.  Order within the list is irrelevant.
__global__ void __launch_bounds__(1024) flag_compact_kernel(uint32_t* overflow, uint32_t nq, uint32_t* redo_list, uint32_t* redo_cnt,
                                                             uint32_t* host_mirror) {
    __shared__ uint32_t n_s;
    if (threadIdx.x == 0) n_s = 0;
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < nq; q += 1024) {
        if (overflow[q]) {
            redo_list[atomicAdd(&n_s, 1u)] = q;
            overflow[q] = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        *redo_cnt = n_s;
        if (host_mirror) *host_mirror = n_s;  // pinned host memory: the repair feedback reads it two searches later, behind an event
    }
}

// Cross-shard merge of formatted results.  An entry is the u64 composite (order key << 32 | slot), slot = list * k +
// rank: ties are broken by list order, then by rank inside the list.  The lists arrive in ascending row-range order
// (rank order of the all-gather) and each is sorted by (key, row position), so this IS "ascending global row
// position" -- without needing the position, which a shard that reports vector ids no longer carries.  Padding
// (index UINT64_MAX) becomes the all-ones composite and sorts behind every real entry, NaN scores included.
// grid (nq); block 256; LDS P * 8 bytes (P <= 8192: 64 KiB).
__global__ void __launch_bounds__(256) merge_shards_kernel(ShardMergeParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* buf = reinterpret_cast<uint64_t*>(smem);  // [P]
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t total = p.nlists * p.k;
    const bool use_raw = key_is_raw(p.dtype, p.metric) && p.raw != nullptr;
    for (uint32_t i = tid; i < p.P; i += 256) {
        uint64_t comp = kPadComposite;
        if (i < total) {
            const uint32_t l = i / p.k, j = i % p.k;
            const size_t rem = (size_t)q * p.k + j;
            if (p.indices[l * p.ls_indices + rem] != ~0ull) {
                const uint32_t ky = use_raw ? key_from_raw(p.raw[l * p.ls_raw + rem], p.metric)
                                            : key_from_score(p.scores[l * p.ls_scores + rem], p.metric);
                comp = ((uint64_t)ky << 32) | i;
            }
        }
        buf[i] = comp;
    }
    __syncthreads();
    bitonic_sort_u64<256>(buf, p.P, tid);
    for (uint32_t i = tid; i < p.k; i += 256) {
        const size_t o = (size_t)q * p.k + i;
        const uint64_t comp = i < p.P ? buf[i] : kPadComposite;
        if (comp == kPadComposite) {
            p.out_scores[o] = pad_score(p.metric);
            p.out_indices[o] = ~0ull;
            if (p.out_raw) p.out_raw[o] = 0;
        } else {
            const uint32_t slot = (uint32_t)comp, l = slot / p.k, j = slot % p.k;
            const size_t rem = (size_t)q * p.k + j;
            p.out_scores[o] = p.scores[l * p.ls_scores + rem];
            p.out_indices[o] = p.indices[l * p.ls_indices + rem];
            if (p.out_raw) p.out_raw[o] = p.raw ? p.raw[l * p.ls_raw + rem] : 0;
        }
    }
}

// The same merge for lists too long for one block's LDS (nlists * k > kMergeMaxEntries; any k since round 4): the composites
// of ONE query go to HBM (merge_build_kernel), a device-wide sort orders them (sort_topk.hip) and merge_write_kernel gathers
// the first k.  Same composite, same tie rule, same padding as merge_shards_kernel.
__global__ void __launch_bounds__(256) merge_build_kernel(ShardMergeParams p, uint32_t q, uint64_t* comps) {
    const size_t total = (size_t)p.nlists * p.k;
    const bool use_raw = key_is_raw(p.dtype, p.metric) && p.raw != nullptr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t l = i / p.k, j = i % p.k, rem = (size_t)q * p.k + j;
        uint32_t ky = 0;
        const bool live = p.indices[l * p.ls_indices + rem] != ~0ull;
        if (live)
            ky = use_raw ? key_from_raw(p.raw[l * p.ls_raw + rem], p.metric) : key_from_score(p.scores[l * p.ls_scores + rem], p.metric);
        comps[i] = rank_entry(ky, (uint32_t)i, !live);
    }
}

__global__ void __launch_bounds__(256) merge_write_kernel(ShardMergeParams p, uint32_t q, const uint64_t* sorted) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < p.k; i += (size_t)gridDim.x * 256) {
        const size_t o = (size_t)q * p.k + i;
        const uint64_t comp = composite_of_rank_entry(sorted[i]);  // nlists * k >= k entries
        if (comp == kPadComposite) {
            p.out_scores[o] = pad_score(p.metric);
            p.out_indices[o] = ~0ull;
            if (p.out_raw) p.out_raw[o] = 0;
        } else {
            const size_t slot = (uint32_t)comp, l = slot / p.k, j = slot % p.k, rem = (size_t)q * p.k + j;
            p.out_scores[o] = p.scores[l * p.ls_scores + rem];
            p.out_indices[o] = p.indices[l * p.ls_indices + rem];
            if (p.out_raw) p.out_raw[o] = p.raw ? p.raw[l * p.ls_raw + rem] : 0;
        }
    }
}

// One thread per 16-B device vector: (row, v) -> up to 16/ES elements.
__global__ void synth_rows_kernel(unsigned char* rows, uint64_t nvec, uint32_t V, uint32_t dim, uint32_t pitch,
                                  uint8_t dtype, uint64_t base, uint64_t row0) {
    const uint32_t es = elem_size(dtype);
    const uint32_t epv = 16 / es;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvec; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = t / V;
        const uint32_t v = (uint32_t)(t % V);
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t e = 0; e < epv; e++) {
            const uint32_t c = v * epv + e;
            if (c >= dim) break;
            const uint64_t u = mix64(base + (r + row0) * dim + c);
            if (dtype == MVF_DTYPE_FLOAT32) {
                w[e] = __float_as_uint(synth_f32(u));
            } else if (dtype == MVF_DTYPE_FLOAT16) {
                const uint32_t h = __half_as_ushort(__float2half_rn(synth_f32(u)));
                w[e >> 1] |= h << (16 * (e & 1));
            } else {
                w[e >> 2] |= (uint32_t)(uint8_t)(u >> 56) << (8 * (e & 3));
            }
        }
        *reinterpret_cast<uint4*>(rows + r * pitch + (size_t)v * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Re-pitch host rows (row_bytes each, src_stride apart, uploaded as they lie with one 1-D copy) into the device layout
// (pitch = row_bytes rounded up to 16, zero padded; only row_bytes of each source row are read).  One thread per 16-B
// output vector; GRAN = 4 / 2 / 1 is the widest access both layouts are aligned for.
template <int GRAN>
__global__ void repack_rows_kernel(const unsigned char* src, unsigned char* dst, uint64_t nvec, uint32_t V,
                                   uint32_t row_bytes, uint64_t src_stride, uint32_t pitch) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvec; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = t / V;
        const uint32_t v = (uint32_t)(t % V);
        const unsigned char* sp = src + r * src_stride + (size_t)v * 16;
        const uint32_t have = row_bytes - v * 16 < 16u ? row_bytes - v * 16 : 16u;
        uint32_t w[4] = {0, 0, 0, 0};
        if (GRAN == 4) {
            for (uint32_t i = 0; i < have / 4; i++) w[i] = reinterpret_cast<const uint32_t*>(sp)[i];
        } else if (GRAN == 2) {
            for (uint32_t i = 0; i < have / 2; i++) w[i >> 1] |= (uint32_t)reinterpret_cast<const uint16_t*>(sp)[i] << (16 * (i & 1));
        } else {
            for (uint32_t i = 0; i < have; i++) w[i >> 2] |= (uint32_t)sp[i] << (8 * (i & 3));
        }
        *reinterpret_cast<uint4*>(dst + r * pitch + (size_t)v * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Payload gather: rows[idx[i] - index_base] -> out[i] (tightly packed, row_bytes each); out-of-range / padding
// indices give zero rows.  One wave per row, byte granular (result sets are tiny).
template <typename T>  // the copy unit: 16 bytes when the row size allows it (stored rows start on 16-byte pitches), else 4 or 1
__global__ void __launch_bounds__(256) gather_rows_kernel(const unsigned char* rows, uint64_t n, uint32_t pitch,
                                                           uint32_t row_bytes, uint64_t index_base, const uint64_t* idx,
                                                           uint32_t count, unsigned char* out, uint32_t idx_stride, uint32_t per_list) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    const uint32_t units = row_bytes / (uint32_t)sizeof(T);
    for (uint32_t i = wave; i < count; i += nwaves) {
        const uint64_t g = idx[(size_t)(i / per_list) * idx_stride + i % per_list];  // out row i = entry i % per_list of list i / per_list
        const bool ok = g >= index_base && g - index_base < n;
        const T* src = reinterpret_cast<const T*>(rows + (ok ? (g - index_base) : 0) * pitch);
        T* dst = reinterpret_cast<T*>(out + (size_t)i * row_bytes);
        for (uint32_t b = lane; b < units; b += 64) dst[b] = ok ? src[b] : T{};
    }
}

// Tightly packed elements (queries): one thread per element.
__global__ void synth_packed_kernel(void* out, uint64_t nelem, uint8_t dtype, uint64_t base) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nelem; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t u = mix64(base + t);
        if (dtype == MVF_DTYPE_FLOAT32) reinterpret_cast<float*>(out)[t] = synth_f32(u);
        else if (dtype == MVF_DTYPE_FLOAT16) reinterpret_cast<__half*>(out)[t] = __float2half_rn(synth_f32(u));
        else reinterpret_cast<uint8_t*>(out)[t] = (uint8_t)(u >> 56);
    }
}

}  // namespace

hipError_t launch_select_final(const SelectParams& p, uint32_t nq, hipStream_t s) {
    hipLaunchKernelGGL(select_final_kernel, dim3(nq), dim3(1024), (size_t)p.P * 8 + 16, s, p);
    return hipGetLastError();
}

hipError_t launch_write_sorted(const SelectParams& p, const uint64_t* sorted, uint32_t n, size_t out_base, hipStream_t s) {
    const uint32_t blocks = (uint32_t)std::min<size_t>(((size_t)p.k + 255) / 256, 4096);
    hipLaunchKernelGGL(write_sorted_kernel, dim3(blocks), dim3(256), 0, s, p, sorted, n, out_base);
    return hipGetLastError();
}

hipError_t launch_merge_build(const ShardMergeParams& p, uint32_t q, uint64_t* comps, hipStream_t s) {
    const size_t total = (size_t)p.nlists * p.k;
    hipLaunchKernelGGL(merge_build_kernel, dim3((uint32_t)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, p, q, comps);
    return hipGetLastError();
}

hipError_t launch_merge_write(const ShardMergeParams& p, uint32_t q, const uint64_t* sorted, hipStream_t s) {
    hipLaunchKernelGGL(merge_write_kernel, dim3((uint32_t)std::min<size_t>(((size_t)p.k + 255) / 256, 4096)), dim3(256), 0, s, p, q, sorted);
    return hipGetLastError();
}

hipError_t launch_flag_compact(uint32_t* overflow, uint32_t nq, uint32_t* redo_list, uint32_t* redo_cnt, uint32_t* host_mirror, hipStream_t s) {
    hipLaunchKernelGGL(flag_compact_kernel, dim3(1), dim3(1024), 0, s, overflow, nq, redo_list, redo_cnt, host_mirror);
    return hipGetLastError();
}

hipError_t launch_merge_shards(const ShardMergeParams& p, hipStream_t s) {
    const size_t lds = (size_t)p.P * 8;  // <= 64 KiB (kMergeMaxEntries); above 48 KiB the attribute is set per launch (per device)
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&merge_shards_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(merge_shards_kernel, dim3(p.nq), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_synth_rows(unsigned char* rows, uint64_t n, uint32_t dim, uint32_t pitch, uint8_t dtype,
                             uint64_t seed, uint64_t row0, hipStream_t s) {
    const uint32_t V = pitch / 16;
    const uint64_t nvec = n * V;
    if (nvec == 0) return hipSuccess;
    const uint64_t blocks = (nvec + 255) / 256;
    hipLaunchKernelGGL(synth_rows_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, rows, nvec,
                       V, dim, pitch, dtype, mix64(seed), row0);
    return hipGetLastError();
}

hipError_t launch_repack_rows(const unsigned char* src, unsigned char* dst, uint64_t n, uint32_t row_bytes,
                              uint64_t src_stride, uint32_t pitch, hipStream_t s) {
    const uint32_t V = pitch / 16;
    const uint64_t nvec = n * V;
    if (nvec == 0) return hipSuccess;
    const uint64_t blocks64 = (nvec + 255) / 256;
    const dim3 grid((unsigned)(blocks64 < 65536 ? blocks64 : 65536));
    const uint64_t al = row_bytes | src_stride;  // both the row size and the row starts must be aligned for GRAN
    if (al % 4 == 0) hipLaunchKernelGGL(repack_rows_kernel<4>, grid, dim3(256), 0, s, src, dst, nvec, V, row_bytes, src_stride, pitch);
    else if (al % 2 == 0) hipLaunchKernelGGL(repack_rows_kernel<2>, grid, dim3(256), 0, s, src, dst, nvec, V, row_bytes, src_stride, pitch);
    else hipLaunchKernelGGL(repack_rows_kernel<1>, grid, dim3(256), 0, s, src, dst, nvec, V, row_bytes, src_stride, pitch);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const unsigned char* rows, uint64_t n, uint32_t pitch, uint32_t row_bytes, uint64_t index_base,
                              const uint64_t* d_idx, uint32_t count, unsigned char* d_out, hipStream_t s, uint32_t idx_stride, uint32_t per_list) {
    if (count == 0) return hipSuccess;
    if (per_list == 0) idx_stride = per_list = count;  // one list
    const uint32_t blocks = std::min<uint32_t>((count + 3) / 4, 2048u);
    const uint64_t al = row_bytes | reinterpret_cast<uintptr_t>(d_out);
    if (al % 16 == 0) hipLaunchKernelGGL(gather_rows_kernel<uint4>, dim3(blocks), dim3(256), 0, s, rows, n, pitch, row_bytes, index_base, d_idx, count, d_out, idx_stride, per_list);
    else if (al % 4 == 0) hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(blocks), dim3(256), 0, s, rows, n, pitch, row_bytes, index_base, d_idx, count, d_out, idx_stride, per_list);
    else hipLaunchKernelGGL(gather_rows_kernel<unsigned char>, dim3(blocks), dim3(256), 0, s, rows, n, pitch, row_bytes, index_base, d_idx, count, d_out, idx_stride, per_list);
    return hipGetLastError();
}

hipError_t launch_synth_packed(void* out, uint64_t nelem, uint8_t dtype, uint64_t seed, hipStream_t s) {
    if (nelem == 0) return hipSuccess;
    const uint64_t blocks = (nelem + 255) / 256;
    hipLaunchKernelGGL(synth_packed_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, out,
                       nelem, dtype, mix64(seed));
    return hipGetLastError();
}

}  // namespace mvf
