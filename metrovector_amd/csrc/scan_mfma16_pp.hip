// scan_mfma16_pp.hip — K2 for the narrow types, 256-query tile: PING-PONG schedule.
//
// Same tile (256 queries x 256 corpus rows, 8 waves as 2 x 4, 128 x 64 outputs per wave), LDS-DMA staging, XOR-swizzled
// 64-byte k-tiles, MFMA shape (v_mfma_f32_16x16x32_f16 / v_mfma_i32_16x16x64_i8) and epilogue as scan_mfma16_dma.hip;
// what differs is WHEN the waves do what.  That kernel runs all 8 waves in lockstep: one barrier per k-tile, and
// behind it every wave first waits for its LDS fragment reads -- both waves of a SIMD stall together and the matrix pipe
// idles (SQ_VALU_MFMA_BUSY 54 %, 36 % of wave cycles parked; profiles/r01_cfg5_f16_mfma16_dma_pmc.json).  Here:
//
//   * a k-tile is TWO PHASES of 16 MFMAs (queries 0-63 of the wave's 128, then 64-127; the four B fragments are read
//     once and kept).  A phase is   L: fragment reads + 2 DMA pieces | s_barrier | M: 16 MFMAs | s_barrier.
//   * the two wave groups (waves 0-3 = query half 0, waves 4-7 = half 1; wave w and w+4 share a SIMD) run ONE BARRIER
//     APART: while one group is in M the other is in L, so each SIMD's matrix pipe always has one wave in an MFMA
//     cluster and the LDS / DMA issue of its partner rides underneath (the 8-phase GEMM schedule of
//     cdna_hip_programming.md §5, which holds 1.3-1.5 PFLOP/s on random bf16 data).
//   * the groups also split the staging: group 0 issues every A (query) piece, group 1 every B (corpus) piece, each
//     into its own ring -- a wave's vmcnt is in issue order, so only separate issuers can give the two operands
//     different depths: A (always L2-resident) 4 stages, B (HBM) 5 stages with three to four k-tiles = 48-64 KB of
//     corpus bytes per CU in flight (the lockstep kernel: 32 KB, which is what held cfg4 at 3.3 TB/s).
//
// Barrier numbering (b1, b2, ...; every wave executes every barrier, counts match at exit):
//     group 0:      L(0) b1 M(0) b2 L(1) b3 M(1) b4 L(2) ...          phase ph = 2 t + h of k-tile t
//     group 1:   b1 L(0) b2 M(0) b3 L(1) b4 M(1) b5 ...               (one extra barrier first, one fewer last)
// RAW  k-tile t is first read by group 0 in L(2t), after b(4t).  Every issuer waits for its own pieces of k-tile t
//      (counted vmcnt) in its L(2t-1): group 0 before b(4t-1), group 1 before b(4t) -- a barrier later the data is
//      everyone's.  WAR  A: in iteration t group 0 refills the slot of k-tile t-2 (group 1 reads k-tile t-1 until
//      b(4t)); B: in iteration t group 1 refills the slot of k-tile t-1, whose last fragment reads (phase 0 only)
//      retired before b(4t-1).  Exit drains vmcnt(0): stray DMAs must not land in a successor block's LDS.
// A block keeps ONE query tile for its whole life (the launcher sizes the grid so), so the per-query constants are
// loaded once and no block-wide barrier ever sits inside the k-loop.

#include "scan_mfma.h"

#include "mvf_common.h"

#include <hip/hip_fp16.h>

#include <cstdlib>
#include <type_traits>

namespace mvf {
namespace {

#include "scan_mfma16_common.inc"

constexpr int DKB = 64;                    // k-tile bytes per row
constexpr int BMQ = 256, BR = 256;         // block tile: queries x corpus rows
constexpr int WQ = 128, WR = 64;           // wave tile
constexpr int NSA = 4, NSB = 5;            // ring depths (stages of 16 KB)
constexpr int STG = 256 * DKB;             // one stage of either operand: 256 rows x 64 B
constexpr int PPW = 4;                     // 1-KB DMA pieces per wave per k-tile (16 pieces of its operand / 4 waves)
constexpr int NRC = 4;                     // row-constant buffers (tile ordinal mod 4), two 1-KB arrays each
constexpr size_t PP_LDS = (size_t)(NSA + NSB) * STG + 4 * BMQ * 4 + NRC * 2 * BR * 4 + 16;  // + the candidate counter

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SH = 16;

__device__ __forceinline__ uint32_t slot_swz(uint32_t x) { return (0x78u >> (2u * x)) & 3u; }  // scan_mfma16_dma.hip

#define PP_BARRIER()                            \
    do {                                        \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_s_barrier();           \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

template <int DT, int METRIC, bool DIRECT, bool XS>
__global__ void __launch_bounds__(512, 2) scan_mfma16_pp_kernel(Batch16Params p) {
    using AccT = typename std::conditional<DT == MVF_DTYPE_FLOAT16, f32x4, i32x4>::type;
    constexpr int NI = WQ / SH, NJ = WR / SH, NE = SH * SH / 64, HI = NI / 2;
    constexpr bool U8 = DT == MVF_DTYPE_UINT8;
    constexpr bool QS = DT == MVF_DTYPE_INT8 && XS;             // int8 shadow of a float corpus: float scores (common.inc)
    constexpr bool F16 = DT == MVF_DTYPE_FLOAT16 || QS;         // ... so the per-row constants are the float ones
    // per-row constants the epilogue needs (scan_mfma16_common.inc): array 0 = norms, array 1 = shadow scale / UInt8 bias
    constexpr bool NEED0 = METRIC != MVF_METRIC_INNER_PRODUCT;
    constexpr bool NEED1 = F16 ? XS : (U8 && METRIC != MVF_METRIC_L2);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ringA = smem;
    unsigned char* ringB = smem + NSA * STG;
    float* qa_s = reinterpret_cast<float*>(smem + (NSA + NSB) * STG);
    uint32_t* tau_s = reinterpret_cast<uint32_t*>(qa_s + BMQ);
    float* qb_s = reinterpret_cast<float*>(tau_s + BMQ);
    float* thr_s = qb_s + BMQ;
    uint32_t* rc_s = reinterpret_cast<uint32_t*>(thr_s + BMQ);  // [NRC][2][BR]
    uint32_t* bc_s = rc_s + NRC * 2 * BR;                       // records in the block's candidate region
    const uint32_t* arr0 = F16 ? reinterpret_cast<const uint32_t*>(METRIC == MVF_METRIC_COSINE ? p.xnorm_f : p.xx2)
                               : reinterpret_cast<const uint32_t*>(p.xnorm_i);
    const uint32_t* arr1 = F16 ? reinterpret_cast<const uint32_t*>(p.xscale) : reinterpret_cast<const uint32_t*>(p.xbias_i);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;  // group = query half (wm), wq = row quarter (wn) and DMA piece quarter
    const int wm = grp, wn = wq;

    // persistent blocks, XCD-aware tile order; slot % mtiles is the same for every slot of a block (launcher)
    const uint32_t xcd = blockIdx.x & 7u, ls = blockIdx.x >> 3, nls = gridDim.x >> 3;
    const uint32_t mt = ls % p.mtiles;
    auto slot_nt = [&](uint32_t n) -> uint32_t { return ((ls + n * nls) / p.mtiles) * 8u + xcd; };
    uint32_t my_tiles = 0;
    {
        const uint32_t max_slot_excl = ((p.ntiles + 7u - xcd) / 8u) * p.mtiles;
        if (ls < max_slot_excl) my_tiles = (max_slot_excl - ls + nls - 1) / nls;
    }
    if (my_tiles == 0) return;
    const uint32_t G = my_tiles * p.KT;

    // ---- DMA: group 0 -> A pieces, group 1 -> B pieces; wave wq owns pieces [4 wq, 4 wq + 4) of its operand ----------
    const uint32_t rl = (uint32_t)lane >> 2;
    const uint32_t cl = ((uint32_t)lane & 3u) ^ slot_swz(((uint32_t)lane >> 4) & 3u);
    const unsigned char* src[PPW];  // this lane's row of its operand: query row (A) / corpus row, clamped (B)
    unsigned char* const my_ring = grp == 0 ? ringA : ringB;
    const uint32_t my_ns = grp == 0 ? (uint32_t)NSA : (uint32_t)NSB;
    const uint32_t vlim = grp == 0 ? 0xFFFFFFFFu : p.V;  // B: k beyond the row's pitch reads zeros (0 x Inf would poison f16)
    uint32_t d_n = 0, d_kt = 0, d_st = 0;  // DMA cursor: tile ordinal, k-tile, ring slot
    auto set_dma_tile = [&](uint32_t n) {
        if (grp == 0) {
#pragma unroll
            for (int j = 0; j < PPW; j++) src[j] = p.qprep + ((size_t)mt * BMQ + ((uint32_t)wq * PPW + j) * 16u + rl) * p.KPB;
        } else {
            const uint32_t r0 = p.row_begin + slot_nt(n) * BR;
#pragma unroll
            for (int j = 0; j < PPW; j++) {
                const uint32_t r = r0 + ((uint32_t)wq * PPW + j) * 16u + rl;
                src[j] = p.rows + (size_t)(r < p.row_end ? r : r0) * p.pitch;
            }
            // the tile's per-row constants ride along: one 1-KB piece per array (rows r0 .. r0 + 255; the arrays are
            // padded by 256 entries, api.hip).  Extra pieces only make this wave's counted waits stricter.
            uint32_t* dst = rc_s + (n & (NRC - 1)) * 2 * BR;
            if (NEED0 && wq == 0)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(arr0 + r0 + 4u * (uint32_t)lane), (lds_ptr_t)dst, 16, 0, 0);
            if (NEED1 && wq == 1)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(arr1 + r0 + 4u * (uint32_t)lane), (lds_ptr_t)(dst + BR), 16, 0, 0);
        }
    };
    auto dma_piece = [&](int j) __attribute__((always_inline)) {
        const uint32_t v = d_kt * 4u + cl;  // 16-B vector of the row (the swizzle is applied on the source chunk)
        const unsigned char* s = v < vlim ? src[j] + (size_t)v * 16u : p.zeros;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)s, (lds_ptr_t)(my_ring + d_st * STG + (wq * PPW + j) * (16 * DKB)), 16, 0, 0);
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {  // cursors clamp on the block's last tile
        d_st = d_st + 1 == my_ns ? 0 : d_st + 1;
        if (++d_kt == p.KT) {
            d_kt = 0;
            if (++d_n < my_tiles && grp != 0) set_dma_tile(d_n);
        }
    };

    AccT acc[NI][NJ];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int e = 0; e < NE; e++) acc[i][j][e] = 0;
    };
    zero_acc();

    load_query_consts16<DT, METRIC, BMQ, QS>(p, mt * BMQ, tid, qa_s, qb_s, tau_s, thr_s);
    if (tid == 0) *bc_s = 0;
    set_dma_tile(0);
    // prologue: A k-tiles 0, 1 (group 0) / B k-tiles 0 .. 3 (group 1)
    {
        const int pre = grp == 0 ? NSA - 2 : NSB - 1;
        for (int t = 0; t < pre; t++) {
#pragma unroll
            for (int j = 0; j < PPW; j++) dma_piece(j);
            dma_advance();
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): once per block
    __syncthreads();                     // k-tile 0 is everyone's; also publishes the query constants

    const uint32_t frow = (uint32_t)lane & (SH - 1);
    const uint32_t fchunk = (uint32_t)lane >> 4;
    const uint32_t fslot = (fchunk ^ slot_swz((frow >> 2) & 3u)) & 3u;
    const uint32_t a_off = ((uint32_t)wm * WQ + frow) * DKB + fslot * 16u;
    const uint32_t b_off = ((uint32_t)wn * WR + frow) * DKB + fslot * 16u;
    auto read_a = [&](const unsigned char* st, int i) __attribute__((always_inline)) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(st + a_off + i * SH * DKB);
    };
    auto read_b = [&](const unsigned char* st, int j) __attribute__((always_inline)) -> u32x4 {
        u32x4 x = *reinterpret_cast<const u32x4*>(st + b_off + j * SH * DKB);
        if (U8) x ^= u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
        return x;
    };
    auto mfma1 = [&](AccT& c, const u32x4& fa, const u32x4& fb) __attribute__((always_inline)) {
        if constexpr (DT == MVF_DTYPE_FLOAT16) {
#ifdef MVF_DIAG_BF16  // diagnostic build only (power / clock of the bf16 MFMA on the same operand bits; results are garbage)
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), c, 0, 0, 0);
#else
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fa), __builtin_bit_cast(half8, fb), c, 0, 0, 0);
#endif
        } else
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, fa), __builtin_bit_cast(i32x4, fb), c, 0, 0, 0);
    };

    if (grp == 1) PP_BARRIER();  // the stagger: group 1 runs one barrier behind group 0

    uint32_t sa = 0, sb = 0;     // compute slots of the two rings
    uint32_t c_n = 0, c_kt = 0, c_nt = slot_nt(0);
    for (uint32_t g = 0; g < G; g++) {
        const unsigned char* stA = ringA + sa * STG;
        const unsigned char* stB = ringB + sb * STG;
        u32x4 fb[NJ], fa[HI];
        // ---- phase 0: queries 0-63 of the wave's 128 ---------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < NJ; j++) fb[j] = read_b(stB, j);
#pragma unroll
        for (int i = 0; i < HI; i++) fa[i] = read_a(stA, i);
        dma_piece(0);
        dma_piece(1);
        PP_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < HI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) mfma1(acc[i][j], fa[i], fb[j]);
        __builtin_amdgcn_s_setprio(0);
        PP_BARRIER();
        // ---- phase 1: queries 64-127 (same B fragments) ------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < HI; i++) fa[i] = read_a(stA, HI + i);
        dma_piece(2);
        dma_piece(3);
        dma_advance();
        // this wave's pieces of k-tile g + 1 have landed; the younger k-tiles stay in flight across the barriers
        if (grp == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | (PPW * (NSA - 3)));        // vmcnt(4)
        else __builtin_amdgcn_s_waitcnt(0x0F70 | (PPW * (NSB - 2)));                 // vmcnt(12)
        PP_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < HI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) mfma1(acc[HI + i][j], fa[i], fb[j]);
        __builtin_amdgcn_s_setprio(0);
        PP_BARRIER();
        sa = sa + 1 == NSA ? 0 : sa + 1;
        sb = sb + 1 == NSB ? 0 : sb + 1;
        if (++c_kt == p.KT) {  // tile finished (this group's half of it); the partner group is inside an MFMA cluster
            const uint32_t* rc = rc_s + (c_n & (NRC - 1)) * 2 * BR;
#ifdef MVF_DIAG_NOEPI  // diagnostic build only: the k-loop alone (the sums are kept alive, nothing is selected)
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(acc[i][j]));
            if (false)
#endif
            epilogue16<DT, METRIC, DIRECT, XS, BMQ, SH, WQ, WR, BR, true>(p, acc, c_nt, mt, wm, wn, lane, qa_s, qb_s, tau_s, thr_s,
                                                                          rc, rc + BR, p.blk_cand ? bc_s : nullptr);
            zero_acc();
            c_kt = 0;
            if (++c_n < my_tiles) c_nt = slot_nt(c_n);
        }
    }
    if (grp == 0) PP_BARRIER();                  // group 1 ran one barrier ahead at the start: even the counts
    __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00);  // vmcnt(0): the DMAs issued past the end target this block's own LDS
    __syncthreads();                             // every wave's last epilogue has counted its candidates
    if (tid == 0 && p.blk_cnt) p.blk_cnt[blockIdx.x] = min(*bc_s, p.blk_cap);
}

static_assert(PPW * (NSB - 2) < 16 && PPW * (NSA - 3) < 16, "vmcnt low field");

template <int DT, int METRIC>
hipError_t launch_dtm(const Batch16Params& p, dim3 grid, hipStream_t s) {
    void (*fn)(Batch16Params) = p.direct ? &scan_mfma16_pp_kernel<DT, METRIC, true, false> : &scan_mfma16_pp_kernel<DT, METRIC, false, false>;
    if constexpr (DT == MVF_DTYPE_FLOAT16 || DT == MVF_DTYPE_INT8)  // rows are a scaled shadow (f16, or the int8 shadow: QS)
        if (p.xscale) fn = p.direct ? &scan_mfma16_pp_kernel<DT, METRIC, true, true> : &scan_mfma16_pp_kernel<DT, METRIC, false, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PP_LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, grid, dim3(512), PP_LDS, s, p);
    return hipGetLastError();
}

template <int DT>
hipError_t launch_dt(const Batch16Params& p, int metric, dim3 grid, hipStream_t s) {
    switch (metric) {
    case MVF_METRIC_L2: return launch_dtm<DT, MVF_METRIC_L2>(p, grid, s);
    case MVF_METRIC_INNER_PRODUCT: return launch_dtm<DT, MVF_METRIC_INNER_PRODUCT>(p, grid, s);
    default: return launch_dtm<DT, MVF_METRIC_COSINE>(p, grid, s);
    }
}

}  // namespace

// A block must keep one query tile for life: lanes per XCD (num_cus / 8) must be a multiple of mtiles, i.e.
// mtiles <= num_cus / 8 (8192 queries on 256 CUs); larger batches stay on the lockstep kernel.
// KT >= 2: the loader runs NSB - 1 k-tiles ahead of the epilogues, and the row constants have NRC buffers.
bool scan_mfma16_pp_usable(uint32_t mtiles, int num_cus, uint32_t KT) {
    return KT >= 2 && mtiles >= 1 && mtiles <= std::max(1u, (uint32_t)num_cus / 8u);
}

// p as for launch_scan_mfma16_dma with the 256-query tile.
hipError_t launch_scan_mfma16_pp(const Batch16Params& p, int dtype, int metric, int num_cus, hipStream_t s) {
    const uint32_t total = ((p.ntiles + 7) / 8) * p.mtiles * 8;
    uint32_t nls = std::max(1u, (uint32_t)num_cus / 8u);
    nls -= nls % p.mtiles;  // >= mtiles by scan_mfma16_pp_usable
    const dim3 grid(std::min(total, nls * 8u));
    if (dtype == MVF_DTYPE_FLOAT16) return launch_dt<MVF_DTYPE_FLOAT16>(p, metric, grid, s);
    if (dtype == MVF_DTYPE_UINT8) return launch_dt<MVF_DTYPE_UINT8>(p, metric, grid, s);
    return launch_dt<MVF_DTYPE_INT8>(p, metric, grid, s);
}

}  // namespace mvf
