// sort_topk.hip -- the k best of a whole shard's rank entries: radix SELECT + sort of the survivors (api.hip: search_sorted_k,
// the long cross-shard merges).
//
// The reference keeps a heap of k + 1 entries whatever k is (examples/similarity_search.rs:143, :159-168) and sorts what is
// left (:172-173).  For k in the thousands and beyond the streaming kernel writes every row's RANK ENTRY instead of selecting
// (mvf_common.h: position << 32 | key', 8 bytes per row beside the dim * es it reads; entries arrive in ascending position,
// a dead row's key' is 0xFFFFFFFF) and this file orders the first k of them by (key', position) -- the composites' order.
// Rounds 3-4 sorted ALL n entries with the library's radix sort (rocPRIM onesweep: 72 bytes of traffic per row; its small-input
// path needed a workaround).  Round 5 -- hand-written, no library on the path:
//
//   SELECT (k < n / 2): three histogram passes over the 32-bit keys (11 + 11 + 10 bits; each pass counts the digit of the
//     entries that match the prefix found so far -- every block derives that prefix itself from the earlier histograms) give
//     the k-th smallest key T, the count L of keys below it and the E ties on it.  A count pass and a write pass then copy
//     the survivors -- key' < T, and the FIRST k - L ties by position -- IN POSITION ORDER: every wave owns a contiguous run of
//     entries, its output offset is the prefix sum of the waves' counts, inside the run a ballot ranks the lanes.  Exactly
//     min(k, n) survivors, 40 bytes of reads per row.
//   SORT of the m survivors (all n entries when k >= n / 2): up to 2048 in ONE block's LDS (the composites' bitonic network),
//     beyond that a stable LSD radix sort on the key half -- up to 131072 entries three passes of 11 / 11 / 10 bits, two
//     launches each (histogram; scatter, every block summing the few tiles in front of it itself), longer lists four 8-bit
//     passes of histogram / scan / scatter.  Stable, and the input is in position order, so equal keys keep it: the position
//     half needs no pass.
//
// Measured (profiles/r05_any_k.txt): 10M x 768 f32, one query, k = 16384: 5.21 ms (library sort of all 10M entries) -> 4.9-5.0.
#include "aux_kernels.h"
#include "bitonic.h"

namespace mvf {
namespace {

constexpr int kSelBins = 2048;                  // bins of a select pass (passes 0 / 1: 11 bits, pass 2: 10)
constexpr uint32_t kSmallSort = 2048;           // entries one block sorts in LDS (bitonic network; 16384 took 0.12 ms, the radix passes 0.06)
constexpr int kRsItems = 8, kRsTile = 256 * kRsItems;  // radix sort: entries per thread / per block tile
constexpr uint32_t kRsFewBlocks = 64;           // up to this many tiles (131072 entries: L2-resident) the digits are 11 bits wide, three passes
constexpr uint32_t kRsSelfSum = 16;             // up to this many tiles a scatter block sums the tiles in front of it itself (no scan launch)
constexpr uint32_t kSelWavesMax = 4096;         // waves of the count / write passes (one contiguous run of entries each)

__device__ __forceinline__ uint32_t sel_digit(uint32_t key, int pass) {
    return pass == 0 ? key >> 21 : pass == 1 ? (key >> 10) & 0x7FFu : key & 0x3FFu;
}

// Block-wide (256 threads): the bin of `hist` (nbins <= 2048, a multiple of 256) that holds rank `krem` (1-based: the smallest
// bin whose inclusive prefix count reaches it) and the count in front of that bin.  red: 256 + 2 words of LDS.
__device__ __forceinline__ void find_bin(const uint32_t* hist, int nbins, uint64_t krem, uint64_t* red, uint32_t* bin, uint64_t* before) {
    const int tid = threadIdx.x, per = nbins / 256;
    uint64_t mine = 0;
    for (int i = 0; i < per; i++) mine += hist[tid * per + i];
    red[tid] = mine;
    __syncthreads();
    if (tid == 0) {  // 256 partial sums: a serial walk is a microsecond, once per block
        uint64_t run = 0;
        int t = 0;
        for (; t < 255 && run + red[t] < krem; t++) run += red[t];
        red[256] = (uint64_t)t;
        red[257] = run;
    }
    __syncthreads();
    const int t = (int)red[256];
    uint64_t run = red[257];
    __syncthreads();
    if (tid == 0) {
        int i = 0;
        for (; i < per - 1 && run + hist[t * per + i] < krem; i++) run += hist[t * per + i];
        red[256] = (uint64_t)(t * per + i);
        red[257] = run;
    }
    __syncthreads();
    *bin = (uint32_t)red[256];
    *before = red[257];
    __syncthreads();
}

// What the passes before `pass` found: the key prefix / mask the pass filters on, the rank left inside that prefix and the
// count of keys below it.  pass == 3: the threshold key itself.
struct SelState {
    uint32_t prefix, mask;
    uint64_t krem, below;
};
__device__ __forceinline__ SelState sel_state(const uint32_t* hists, int pass, uint64_t k, uint64_t* red) {
    SelState st{0u, 0u, k, 0ull};
    for (int ps = 0; ps < pass; ps++) {
        uint32_t bin;
        uint64_t before;
        find_bin(hists + ps * kSelBins, ps == 2 ? 1024 : 2048, st.krem, red, &bin, &before);
        st.krem -= before;
        st.below += before;
        st.prefix |= ps == 0 ? bin << 21 : ps == 1 ? bin << 10 : bin;
        st.mask |= ps == 0 ? 0xFFE00000u : ps == 1 ? 0x001FFC00u : 0x000003FFu;
    }
    return st;
}

template <int PASS>
__global__ void __launch_bounds__(256) select_hist_kernel(const uint64_t* e, size_t n, uint64_t k, uint32_t* hists, size_t stride) {
    __shared__ uint32_t lh[kSelBins];
    __shared__ uint64_t red[258];
    e += (size_t)blockIdx.y * stride, hists += (size_t)blockIdx.y * 3 * kSelBins;  // blockIdx.y: which of the batch's lists
    const SelState st = sel_state(hists, PASS, k, red);
    for (int i = threadIdx.x; i < kSelBins; i += 256) lh[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t key = (uint32_t)e[i];
        if ((key & st.mask) == st.prefix) atomicAdd(&lh[sel_digit(key, PASS)], 1u);
    }
    __syncthreads();
    uint32_t* out = hists + PASS * kSelBins;
    for (int i = threadIdx.x; i < kSelBins; i += 256)
        if (lh[i]) atomicAdd(&out[i], lh[i]);
}

// wave w owns entries [w run, (w + 1) run); cnt[w] = keys below T, cnt[W + w] = keys equal to T in its run
__global__ void __launch_bounds__(256) select_count_kernel(const uint64_t* e, size_t n, uint64_t k, const uint32_t* hists, size_t run,
                                                           uint32_t* cnt, uint32_t W, size_t stride) {
    __shared__ uint64_t red[258];
    e += (size_t)blockIdx.y * stride, hists += (size_t)blockIdx.y * 3 * kSelBins, cnt += (size_t)blockIdx.y * 2 * kSelWavesMax;
    const SelState st = sel_state(hists, 3, k, red);  // prefix = T
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= W) return;
    const size_t lo = (size_t)w * run, hi = lo + run < n ? lo + run : n;
    uint32_t less = 0, ties = 0;
    for (size_t i = lo + lane; i < hi; i += 64) {
        const uint32_t key = (uint32_t)e[i];
        less += key < st.prefix;
        ties += key == st.prefix;
    }
    for (int off = 32; off > 0; off >>= 1) {
        less += __shfl_xor(less, off, 64);
        ties += __shfl_xor(ties, off, 64);
    }
    if (lane == 0) cnt[w] = less, cnt[W + w] = ties;
}

__global__ void __launch_bounds__(256) select_write_kernel(const uint64_t* e, size_t n, uint64_t k, const uint32_t* hists, size_t run,
                                                           const uint32_t* cnt, uint32_t W, uint64_t* out, size_t stride) {
    __shared__ uint64_t red[258];
    e += (size_t)blockIdx.y * stride, out += (size_t)blockIdx.y * stride, hists += (size_t)blockIdx.y * 3 * kSelBins;
    cnt += (size_t)blockIdx.y * 2 * kSelWavesMax;
    const SelState st = sel_state(hists, 3, k, red);
    const uint64_t need = k - st.below;  // ties wanted: the first `need` by position (1 <= need <= E)
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= W) return;
    const size_t lo = (size_t)w * run, hi = lo + run < n ? lo + run : n;
    uint64_t less_run = 0, ties_run = 0;  // keys below T / ties in front of this run: the waves' counts summed (W <= 4096: 64 per lane)
    for (uint32_t i = lane; i < w; i += 64) less_run += cnt[i], ties_run += cnt[W + i];
    for (int off = 32; off > 0; off >>= 1) {
        less_run += __shfl_xor(less_run, off, 64);
        ties_run += __shfl_xor(ties_run, off, 64);
    }
    uint64_t pos_run = less_run + (ties_run < need ? ties_run : need);  // survivors in front of it
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (size_t i0 = lo; i0 < hi; i0 += 64) {
        const size_t i = i0 + lane;
        const uint64_t ent = i < hi ? e[i] : ~0ull;
        const uint32_t key = (uint32_t)ent;
        const bool tie = i < hi && key == st.prefix;
        const unsigned long long tm = __builtin_amdgcn_ballot_w64(tie);
        const bool emit = i < hi && (key < st.prefix || (tie && ties_run + (uint64_t)__builtin_popcountll(tm & lt) < need));
        const unsigned long long em = __builtin_amdgcn_ballot_w64(emit);
        if (emit) out[pos_run + (uint64_t)__builtin_popcountll(em & lt)] = ent;
        ties_run += (uint64_t)__builtin_popcountll(tm);
        pos_run += (uint64_t)__builtin_popcountll(em);
    }
}

// ---- sort of up to 16384 entries in one block's LDS ---------------------------------------------------------------------
__global__ void __launch_bounds__(1024) sort_small_kernel(const uint64_t* in, uint64_t* out, uint32_t m, uint32_t P, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* buf = reinterpret_cast<uint64_t*>(smem);
    in += (size_t)blockIdx.y * stride, out += (size_t)blockIdx.y * stride;
    for (uint32_t i = threadIdx.x; i < P; i += 1024) {
        const uint64_t ent = i < m ? in[i] : ~0ull;
        buf[i] = i < m ? (ent << 32) | (ent >> 32) : ~0ull;  // rank entry -> (key' << 32 | position): distinct, their order is the result's
    }
    __syncthreads();
    if (P <= 1024) bitonic_sort_u64_reg<1024, 1>(buf, P, threadIdx.x);
    else if (P <= 2048) bitonic_sort_u64_reg<1024, 2>(buf, P, threadIdx.x);
    else if (P <= 4096) bitonic_sort_u64_reg<1024, 4>(buf, P, threadIdx.x);
    else bitonic_sort_u64<1024>(buf, P, threadIdx.x);
    for (uint32_t i = threadIdx.x; i < m; i += 1024) out[i] = (buf[i] << 32) | (buf[i] >> 32);
}

// ---- stable LSD radix sort on the key half: 8-bit digits (four passes) for long lists, 11-bit digits (three passes: 11 + 11 +
// 10) for lists that stay in L2, where 2048 output streams per tile cost nothing ---------------------------------------------
template <int BITS>
__global__ void __launch_bounds__(256) rs_hist_kernel(const uint64_t* in, size_t m, int shift, uint32_t* bh, uint32_t NB, size_t stride,
                                                       size_t bh_stride) {
    constexpr int NBIN = 1 << BITS;
    __shared__ uint32_t lh[NBIN];
    in += (size_t)blockIdx.y * stride, bh += (size_t)blockIdx.y * bh_stride;
    for (int i = threadIdx.x; i < NBIN; i += 256) lh[i] = 0;
    __syncthreads();
    const size_t t0 = (size_t)blockIdx.x * kRsTile;
#pragma unroll
    for (int r = 0; r < kRsItems; r++) {
        const size_t i = t0 + (size_t)r * 256 + threadIdx.x;
        if (i < m) atomicAdd(&lh[((uint32_t)in[i] >> shift) & (NBIN - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NBIN; i += 256) bh[(size_t)i * NB + blockIdx.x] = lh[i];
}

// block d: exclusive scan of digit d's per-block counts in place, the digit's total to tot[d]
__global__ void __launch_bounds__(256) rs_scan_kernel(uint32_t* bh, uint32_t NB, uint32_t* tot, size_t bh_stride) {
    __shared__ uint32_t part[256];
    bh += (size_t)blockIdx.y * bh_stride, tot += (size_t)blockIdx.y * 2048;
    uint32_t* row = bh + (size_t)blockIdx.x * NB;
    const uint32_t per = (NB + 255) / 256, lo = threadIdx.x * per;
    uint32_t s = 0;
    for (uint32_t i = lo; i < lo + per && i < NB; i++) s += row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 256; t++) {
            const uint32_t v = part[t];
            part[t] = run;
            run += v;
        }
        tot[blockIdx.x] = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < lo + per && i < NB; i++) {
        const uint32_t v = row[i];
        row[i] = run;
        run += v;
    }
}

// Block b scatters its tile in order.  Wave w owns the tile's entries [w TW, (w + 1) TW) and walks them in rounds of 64: a
// lane's rank among the entries of its digit = the wave's running count of the digit (LDS) + its rank among the round's lanes
// with that digit (one ballot per digit bit).  Waves in order, rounds in order, lanes in order: stable.
// SCANNED: bh holds the tiles' counts already scanned per digit and tot the digits' totals (rs_scan_kernel); otherwise (few
// tiles) the block sums the raw counts of the tiles in front of it, and of all tiles, itself.
template <int BITS, bool SCANNED>
__global__ void __launch_bounds__(256) rs_scatter_kernel(const uint64_t* in, uint64_t* out, size_t m, int shift, const uint32_t* bh,
                                                          const uint32_t* tot, uint32_t NB, size_t stride, size_t bh_stride) {
    constexpr int NBIN = 1 << BITS, PER = NBIN / 256;
    in += (size_t)blockIdx.y * stride, out += (size_t)blockIdx.y * stride, bh += (size_t)blockIdx.y * bh_stride, tot += (size_t)blockIdx.y * 2048;
    __shared__ uint32_t dbase[NBIN];     // digit d's first output slot + what the tiles in front of this one hold of it
    __shared__ uint32_t cw[4][NBIN];     // per wave: running count per digit, then the wave's offset inside the block's share
    __shared__ uint32_t part[256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    {
        // digit totals and this tile's share: thread t owns digits [t PER, (t + 1) PER)
        uint32_t mine = 0;
        for (int x = 0; x < PER; x++) {
            const int d = tid * PER + x;
            uint32_t total, before;
            if (SCANNED) {
                total = tot[d];
                before = bh[(size_t)d * NB + blockIdx.x];
            } else {
                total = before = 0;
                for (uint32_t b = 0; b < NB; b++) {
                    const uint32_t v = bh[(size_t)d * NB + b];
                    total += v;
                    before += b < blockIdx.x ? v : 0u;
                }
            }
            cw[0][d] = total;
            dbase[d] = before;
            mine += total;
        }
        part[tid] = mine;
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (int t = 0; t < 256; t++) {
                const uint32_t v = part[t];
                part[t] = run;
                run += v;
            }
        }
        __syncthreads();
        uint32_t run = part[tid];
        for (int x = 0; x < PER; x++) {
            const int d = tid * PER + x;
            const uint32_t total = cw[0][d];
            dbase[d] += run;
            run += total;
        }
        __syncthreads();
        for (int w = 0; w < 4; w++)
            for (int x = 0; x < PER; x++) cw[w][tid * PER + x] = 0;
        __syncthreads();
    }
    constexpr int TW = kRsTile / 4, ROUNDS = TW / 64;
    const size_t w0 = (size_t)blockIdx.x * kRsTile + (size_t)wave * TW;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint64_t el[ROUNDS];
    uint32_t lr[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const size_t i = w0 + (size_t)r * 64 + lane;
        const bool valid = i < m;
        el[r] = valid ? in[i] : 0ull;
        const uint32_t d = ((uint32_t)el[r] >> shift) & (NBIN - 1);
        unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const unsigned long long bm = __builtin_amdgcn_ballot_w64((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bm : ~bm;
        }
        const uint32_t rank = (uint32_t)__builtin_popcountll(peers & lt);
        const uint32_t base = valid ? cw[wave][d] : 0u;  // a wave's LDS operations execute in order: every peer reads before the leader writes
        if (valid && rank == 0) cw[wave][d] = base + (uint32_t)__builtin_popcountll(peers);
        lr[r] = base + rank;
    }
    __syncthreads();
    for (int x = 0; x < PER; x++) {  // the waves' offsets inside the block's share of the thread's digits
        const int d = tid * PER + x;
        uint32_t run = 0;
        for (int w = 0; w < 4; w++) {
            const uint32_t v = cw[w][d];
            cw[w][d] = run;
            run += v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const size_t i = w0 + (size_t)r * 64 + lane;
        if (i < m) {
            const uint32_t d = ((uint32_t)el[r] >> shift) & (NBIN - 1);
            out[(size_t)dbase[d] + cw[wave][d] + lr[r]] = el[r];
        }
    }
}

template <int BITS>
void radix_sort_entries(uint64_t*& cur, uint64_t*& oth, size_t m, uint32_t* bh, uint32_t* tot, size_t stride, uint32_t nb, size_t bh_stride,
                        hipStream_t s) {
    const uint32_t NB = (uint32_t)((m + kRsTile - 1) / kRsTile);
    for (int shift = 0; shift < 32; shift += BITS) {
        hipLaunchKernelGGL(rs_hist_kernel<BITS>, dim3(NB, nb), dim3(256), 0, s, cur, m, shift, bh, NB, stride, bh_stride);
        if (NB <= kRsSelfSum) {
            hipLaunchKernelGGL((rs_scatter_kernel<BITS, false>), dim3(NB, nb), dim3(256), 0, s, cur, oth, m, shift, bh, tot, NB, stride, bh_stride);
        } else {
            hipLaunchKernelGGL(rs_scan_kernel, dim3(1 << BITS, nb), dim3(256), 0, s, bh, NB, tot, bh_stride);
            hipLaunchKernelGGL((rs_scatter_kernel<BITS, true>), dim3(NB, nb), dim3(256), 0, s, cur, oth, m, shift, bh, tot, NB, stride, bh_stride);
        }
        std::swap(cur, oth);
    }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

// The first min(k, n) of the n rank entries in a (ascending position) in result order -> *sorted (a or b; the other buffer and
// tmp are scratch) -- for each of `nb` lists of the same n and k, list l at a + l stride / b + l stride (and *sorted + l stride).
// tmp == nullptr: *tmp_bytes receives the scratch size for nb lists of n entries.  Asynchronous on s.
hipError_t sort_composites(void* tmp, size_t* tmp_bytes, uint64_t* a, uint64_t* b, size_t n, size_t k, uint64_t** sorted, hipStream_t s,
                           uint32_t nb, size_t stride) {
    const uint32_t NBmax = (uint32_t)((n + kRsTile - 1) / kRsTile);
    // per-tile digit counts: 256 bins for long lists, 2048 for lists of up to kRsFewBlocks tiles (the digit width follows the list length)
    const size_t bh_words = std::max<size_t>((size_t)256 * std::max(NBmax, 1u), (size_t)2048 * std::min(std::max(NBmax, 1u), kRsFewBlocks));
    const size_t o_cnt = align256((size_t)nb * 3 * kSelBins * 4), o_bh = o_cnt + align256((size_t)nb * 2 * kSelWavesMax * 4),
                 o_tot = o_bh + align256((size_t)nb * bh_words * 4), need = o_tot + align256((size_t)nb * 2048 * 4);
    if (!tmp) {
        *tmp_bytes = need;
        return hipSuccess;
    }
    if (*tmp_bytes < need) return hipErrorInvalidValue;
    if (n == 0 || nb == 0) {
        if (sorted) *sorted = a;
        return hipSuccess;
    }
    unsigned char* t = static_cast<unsigned char*>(tmp);
    uint32_t* hists = reinterpret_cast<uint32_t*>(t);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(t + o_cnt);
    uint32_t* bh = reinterpret_cast<uint32_t*>(t + o_bh);
    uint32_t* tot = reinterpret_cast<uint32_t*>(t + o_tot);

    uint64_t *cur = a, *oth = b;
    size_t m = n;
    if (k < n / 2) {  // SELECT: the survivors, in position order, into b
        hipError_t e = hipMemsetAsync(hists, 0, (size_t)nb * 3 * kSelBins * 4, s);
        if (e != hipSuccess) return e;
        const uint32_t hb = (uint32_t)std::min<size_t>((n + 256 * 16 - 1) / (256 * 16), 1024);
        hipLaunchKernelGGL(select_hist_kernel<0>, dim3(hb, nb), dim3(256), 0, s, a, n, (uint64_t)k, hists, stride);
        hipLaunchKernelGGL(select_hist_kernel<1>, dim3(hb, nb), dim3(256), 0, s, a, n, (uint64_t)k, hists, stride);
        hipLaunchKernelGGL(select_hist_kernel<2>, dim3(hb, nb), dim3(256), 0, s, a, n, (uint64_t)k, hists, stride);
        const uint32_t W = (uint32_t)std::min<size_t>((n + 1023) / 1024, kSelWavesMax);  // >= 1024 entries per wave
        const size_t run = ((n + W - 1) / W + 63) & ~(size_t)63;
        hipLaunchKernelGGL(select_count_kernel, dim3((W + 3) / 4, nb), dim3(256), 0, s, a, n, (uint64_t)k, hists, run, cnt, W, stride);
        hipLaunchKernelGGL(select_write_kernel, dim3((W + 3) / 4, nb), dim3(256), 0, s, a, n, (uint64_t)k, hists, run, cnt, W, b, stride);
        cur = b, oth = a, m = k;
    }
    if (m <= kSmallSort) {
        uint32_t P = 2;
        while (P < m) P <<= 1;
        hipLaunchKernelGGL(sort_small_kernel, dim3(1, nb), dim3(1024), (size_t)P * 8, s, cur, oth, (uint32_t)m, P, stride);
        cur = oth;
    } else if ((m + kRsTile - 1) / kRsTile <= kRsFewBlocks) {
        radix_sort_entries<11>(cur, oth, m, bh, tot, stride, nb, bh_words, s);  // 11 + 11 + 10 bits: three passes
    } else {
        radix_sort_entries<8>(cur, oth, m, bh, tot, stride, nb, bh_words, s);   // four passes
    }
    if (sorted) *sorted = cur;
    return hipGetLastError();
}

}  // namespace mvf
