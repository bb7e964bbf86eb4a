// scan_mfma16_dma.hip — K2 for the narrow types with LDS-DMA staging.
//
// Same role, persistent XCD-aware schedule and epilogue as scan_mfma16.hip (which stages through registers and is kept
// as the A/B reference); what differs is how the operands reach LDS, the MFMA shape and a second, HBM-bound block shape
// for small batches.  The register-staged kernel spends ~830 of ~3600 cycles per 128-B
// k-tile on the VGPR->LDS store path alone (ds_write_b128 moves ~79 B/clk/CU) plus the waits in front of it.  Here
// every operand byte goes global -> LDS directly (global_load_lds_dwordx4, no VGPRs, no ds_write):
//
//   * k-tile = 64 bytes per row (64 int8 / 32 f16) = the k of one 16x16 MFMA; stage = BMQ query rows + 256 corpus rows x
//     64 B; a RING of S stages (256-query shape: 4 x 32 KB; 64-query shape: 6 x 20 KB, see CfT).  During k-tile g the
//     block computes stage g mod S and issues the DMA of k-tile g + S - 1 into the stage freed by the barrier that
//     ended k-tile g - 1; before the next barrier each wave waits for its own pieces of k-tile g + 1 with a COUNTED
//     s_waitcnt vmcnt(pieces x (S - 2)): S - 2 k-tiles (64 KB of corpus bytes per CU) stay in flight across the
//     barrier, which is what a streamed (HBM-latency) operand needs.
//   * one DMA wave-instruction writes 1 KB contiguously (16 rows x 64 B, lane L -> row L>>2, slot L&3), so the LDS
//     image cannot be padded; it is XOR-swizzled instead: slot = chunk ^ swz((row >> 2) & 3), applied on the per-lane
//     SOURCE address here and on the fragment reads (conflict-free for ds_read_b128's lane groups).
//   * k beyond a row's pitch must read zeros (0 x Inf would poison a Float16 dot): such chunks are fetched from a
//     16-byte zero block instead (p.zeros); rows past row_end re-read the tile's first row (columns discarded).
//   * UInt8's x ^ 0x80 happens on the B fragments after the LDS read.
//
// MFMA shape: v_mfma_f32_16x16x32_f16 / v_mfma_i32_16x16x64_i8 -- one MFMA consumes the whole 64-B k of a fragment
// pair.  Same LDS bytes and matrix-pipe cycles per k-tile as the 32x32 shapes, but this loop is POWER-limited (the
// matrix pipe is ~50 % busy at ~1.8 GHz whatever the staging: the clock falls as the MFMAs pack closer) and the chip
// holds a higher clock on the 16x16 shape: +7 % wall.  Measured alternatives that did not pay are in DESIGN.md.
//
// Hazards: RAW — a wave's vmcnt wait covers only its own DMA pieces, the barrier after it publishes everyone's;
// the data is first read in the NEXT k-tile.  WAR — a stage is re-filled only after the barrier that ends the k-tile
// which read it.  No ordinary global load is in flight inside the k-loop (the epilogue's are consumed inside it).

#include "scan_mfma.h"

#include "mvf_common.h"
#include "scan_mfma16_key.h"

#include <hip/hip_fp16.h>

#include <cstdlib>
#include <type_traits>

namespace mvf {
namespace {

#include "scan_mfma16_common.inc"
#include "scan_mfma16_bias.inc"

constexpr int DKB = 64;                       // k-tile bytes per row

// Three block shapes, all 8 waves:
//   BMQ = 256 queries x 256 corpus rows: waves 2 (queries) x 4 (rows), each 128 x 64 outputs; ring of 4 stages x 32 KB.
//     MFMA-bound (power-limited).
//   BMQ = 64 queries x 512 corpus rows (batches <= 64): waves 1 x 8, each 64 x 64 outputs; a quarter of the MFMA work
//     per corpus byte, so the kernel is HBM-bound: ring of 4 stages x 36 KB, two of them (64 KB of corpus bytes) in
//     flight, and twice the rows per barrier of the other shape (with 256 rows and 8 MFMAs per wave between barriers
//     the k-tile iteration was latency-bound at 4.9 TB/s, whatever the ring depth).
//   BMQ = 128 queries x 512 corpus rows (batches of 65..128): waves 1 x 8, each 128 x 64 outputs -- the 256-query shape's
//     wave tile (32 MFMAs per k-tile and barrier) on half the queries; ring of 3 stages x 40 KB.  With two 64-query tiles
//     such a batch moved every corpus byte into LDS twice and ran at 2/3 of the HBM rate; padded to the 256-query tile it
//     did twice the matrix work it needed (50M x 768 int8, 128 queries: 10.0 ms either way).  A first cut with 64 x 64
//     wave tiles (waves 2 x 4 on 256 rows, 16 MFMAs per barrier) held 1.36 POP/s and 7.6 ms.
template <int BMQ_> struct CfT {
    static constexpr int NW = 8;
    static constexpr int BMQ = BMQ_;                            // queries (A rows) per block
    static constexpr int BR = BMQ == 256 ? 256 : 512;           // corpus rows per block tile
    static constexpr int WQ = BMQ == 64 ? 64 : 128;             // a wave's share of the tile: queries ...
    static constexpr int WR = 64;                               // ... x corpus rows
    static constexpr int WN = BR / WR;                          // waves along the rows (4 or 8); NW / WN along the queries
    static constexpr int NSTAGE = BMQ == 128 ? 3 : 4;           // 128 queries: 40-KB stages, one k-tile (32 KB of corpus bytes) in flight across the barrier
    static constexpr int A_B = BMQ * DKB;                       // bytes of A per stage
    static constexpr int STAGE_B = A_B + BR * DKB;              // A then B
    static constexpr int APIECES = BMQ / 16;                    // 1-KB DMA pieces of A per k-tile (16 or 4)
    static constexpr int APW = APIECES >= NW ? APIECES / NW : 1;  // per wave (with 4 pieces, waves 4..7 re-issue 0..3:
                                                                //  every wave then carries the same vmcnt count)
    static constexpr int BPW = BR / 16 / NW;                    //                                      B
    static constexpr int PIECES = APW + BPW;
    static constexpr int INFLIGHT = PIECES * (NSTAGE - 2);      // pieces left in flight across the barrier
    // The tile's per-row constants (norms, shadow scale, UInt8 bias) ride along as 1-KB DMA pieces into LDS, NRC buffers
    // of two arrays x BR entries (tile ordinal mod NRC: the DMA cursor runs up to NSTAGE k-tiles = tiles ahead).  An
    // ordinary global load in the epilogue waits a full HBM round trip with the matrix pipe idle AND drains the ring
    // (s_waitcnt vmcnt(0)): that was 20 % of the int8-selection scan.  The 64-query shape has no LDS left
    // for them (4 x 36 KB of ring) and keeps the loads.
    static constexpr bool RC_LDS = BMQ != 64;
    static constexpr int NRC = BMQ == 128 ? 4 : 8;              // >= NSTAGE + 1 (one k-tile per tile: the DMA cursor runs NSTAGE - 1 tiles ahead and the next
                                                                //  iteration's requests go out while this tile's buffer is still read); a power of two
    // The tile-invariant bounds of the i32-accumulator flavours (scan_mfma16_bias.inc, round 5): R per row and query half
    // (HALVES = 2 at the 256-query tile, 1 at the 128-query one), NRB buffers by tile ordinal, + the halves' threshold extremes,
    // the block's reference point and -B per query
    static constexpr bool INV_LDS = RC_LDS;
    static constexpr int NRB = 4, HALVES = BMQ / WQ;
    static constexpr size_t LDS = (size_t)NSTAGE * STAGE_B + 4 * BMQ * 4 + (RC_LDS ? NRC * 2 * BR * 4 + 2 * BMQ * 4 : 0) + 16 +  // ring + qaux0 + tau + qaux1 + prefilter [+ row constants + the L2 bounds' per-query pair] + candidate counter
                                  (INV_LDS ? NRB * HALVES * BR * 4 + 64 + BMQ * 4 : 0);  // ... and -B per query
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SH = 16;  // MFMA sub-tile: a wave's WQ x WR outputs are (WQ/16) x (WR/16) of them

// XOR swizzle of the 16-B slot by x = (row >> 2) & 3.  A fragment read takes lane -> (row lane & 15, chunk lane >> 4);
// ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... and the permutation {0, 2, 3, 1}
// makes each group touch 16 distinct (row & 3, slot) pairs (SQ_LDS_BANK_CONFLICT = 0).
__device__ __forceinline__ uint32_t slot_swz(uint32_t x) { return (0x78u >> (2u * x)) & 3u; }

// REG: the launch has per-block candidate regions (p.blk_cand; persistent grids) -- what the folded pre-filter of the
// i32-accumulator flavours hands its raw records to; without them those flavours keep round 2's epilogue.
template <int DT, int METRIC, bool DIRECT, bool XS, int BMQ_, bool REG>
__global__ void __launch_bounds__(512, 2) scan_mfma16_dma_kernel(Batch16Params p) {
    using AccT = typename std::conditional<DT == MVF_DTYPE_FLOAT16, f32x4, i32x4>::type;
    using Cf = CfT<BMQ_>;
    constexpr int NW = Cf::NW, WQ = Cf::WQ, WR = Cf::WR, NI = WQ / SH, NJ = WR / SH, NE = SH * SH / 64;
    constexpr int BMQ = Cf::BMQ, NSTAGE = Cf::NSTAGE, STAGE_B = Cf::STAGE_B, OPER_B = Cf::A_B;  // OPER_B: B's offset in a stage
    constexpr bool U8 = DT == MVF_DTYPE_UINT8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* qa_s = reinterpret_cast<float*>(smem + NSTAGE * STAGE_B);  // [BMQ] f16: 2^-e / i8: qq (as int)
    uint32_t* tau_s = reinterpret_cast<uint32_t*>(qa_s + BMQ);        // [BMQ]
    float* qb_s = reinterpret_cast<float*>(tau_s + BMQ);              // [BMQ] f16: |q|
    float* thr_s = qb_s + BMQ;                                        // [BMQ] pre-filter threshold
    uint32_t* rc_s = reinterpret_cast<uint32_t*>(thr_s + BMQ);        // [NRC][2][BR] per-row constants (RC_LDS)
    uint32_t* bc_s = rc_s + (Cf::RC_LDS ? Cf::NRC * 2 * Cf::BR : 0);  // records in the block's candidate region
    float* thu_s = reinterpret_cast<float*>(bc_s + 4);                // [BMQ] int8 shadow, L2: (th - |th| 1e-6) / (2 s_q) ...
    float* u_s = thu_s + BMQ;                                         // [BMQ] ... and 1 / (2 s_q) (scan_mfma16_bias.inc)
    int32_t* rb_s = reinterpret_cast<int32_t*>(u_s + BMQ);            // [NRB][HALVES][BR] tile-invariant bounds: R(row) per query half
    float* ext_s = reinterpret_cast<float*>(rb_s + (Cf::INV_LDS ? Cf::NRB * Cf::HALVES * Cf::BR : 0));  // [2][4] {Pmin, Pmax, Wmin, Wmax} per half
    uint32_t* ref_s = reinterpret_cast<uint32_t*>(ext_s + 8);         // u_ref, v_ref (bit patterns; +inf = not set)
    int32_t* nbq_s = reinterpret_cast<int32_t*>(ref_s + 8);           // [BMQ] -B per query (the block's reference point: the same for every lane)
    // per-row constants the epilogue needs (scan_mfma16_common.inc): array 0 = norms, array 1 = shadow scale / UInt8 bias
    constexpr bool QSF = DT == MVF_DTYPE_FLOAT16 || (DT == MVF_DTYPE_INT8 && XS);  // float scores
    constexpr bool NEED0 = METRIC != MVF_METRIC_INNER_PRODUCT;
    constexpr bool NEED1 = QSF ? XS : (U8 && METRIC != MVF_METRIC_L2);
    const uint32_t* arr0 = QSF ? reinterpret_cast<const uint32_t*>(METRIC == MVF_METRIC_COSINE ? p.xnorm_f : p.xx2)
                               : reinterpret_cast<const uint32_t*>(p.xnorm_i);
    const uint32_t* arr1 = QSF ? reinterpret_cast<const uint32_t*>(p.xscale) : reinterpret_cast<const uint32_t*>(p.xbias_i);

    // i32 accumulators behind a threshold: the pre-filter is folded into the accumulators (scan_mfma16_bias.inc)
    constexpr bool BIAS = !DIRECT && DT != MVF_DTYPE_FLOAT16 && Cf::RC_LDS && REG;
    // The k-tile's barrier in the MIDDLE of its MFMAs (256-query tile; Int8 / UInt8 rows and the int8 shadow): the next k-tile's B
    // fragments and first A fragment are read under the second half of this one's MFMAs, so no wave starts a k-tile with an LDS
    // round trip in the open and the eight waves' fragment bursts no longer collide behind the barrier.  Round 5, one process,
    // 12 rotated rounds (profiles/r05_k2_shadow_ladder.txt): cfg3's last phase 5.65 -> 5.46 ms, the cfg5 shard's 9.16 -> 8.75,
    // cfg4's 7.60 -> 7.53; the bare k-loop (scripts/probe_k2_w1.hip, W8N -> W8NP) 4.97 -> 4.75 / 8.25 -> 7.88.  14 registers more
    // (the fragments in hand): UInt8 rows under cosine spilled with them until their bounds left the registers (INVB below);
    // Float16 rows, whose long phases take the ping-pong kernel, keep the barrier at the k-tile's end.
#ifdef MVF_K2_ENDBARRIER  // A/B builds only
    constexpr bool MIDB = false;
#else
    constexpr bool MIDB = BMQ_ == 256 && DT != MVF_DTYPE_FLOAT16;
#endif
    // ... and the query half of the bound does not change from tile to tile (scan_mfma16_bias.inc, round 5): -B per query in LDS,
    // R per row from the block's transform pass -- every flavour with the folded pre-filter (INVB is BIAS: kept as the name of
    // the scheme where the code speaks of it)
    constexpr bool INVB = BIAS;
    static_assert(!BIAS || Cf::INV_LDS, "the bounds' LDS");
    constexpr bool INV_PROD = INVB && (QSF || METRIC == MVF_METRIC_COSINE);  // the bound has a product term: a reference point, the halves' extremes

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / Cf::WN, wn = wave % Cf::WN;

    // persistent blocks, XCD-aware tile order (scan_mfma16.hip)
    const uint32_t xcd = blockIdx.x & 7u, ls = blockIdx.x >> 3, nls = gridDim.x >> 3;
    auto slot_tile = [&](uint32_t n, uint32_t& nt, uint32_t& mt) {
        const uint32_t slot = ls + n * nls;
        nt = (slot / p.mtiles) * 8u + xcd;
        mt = slot % p.mtiles;
    };
    uint32_t my_tiles = 0;
    {
        const uint32_t max_slot_excl = ((p.ntiles + 7u - xcd) / 8u) * p.mtiles;
        if (ls < max_slot_excl) my_tiles = (max_slot_excl - ls + nls - 1) / nls;
    }
    if (my_tiles == 0) return;
    const uint32_t G = my_tiles * p.KT;
    if (tid == 0) {           // published by the barrier after the prologue
        *bc_s = 0;
        if constexpr (INVB) ref_s[0] = ref_s[1] = 0x7F800000u;
    }

    // ---- DMA: 1-KB pieces (16 rows x 64 B); wave w fills A pieces [w APW, (w+1) APW) and B pieces [w BPW, (w+1) BPW) ----
    const uint32_t rl = (uint32_t)lane >> 2;                              // row inside a piece
    const uint32_t cl = ((uint32_t)lane & 3u) ^ slot_swz(((uint32_t)lane >> 4) & 3u);  // source chunk: slot ^ swz((row >> 2) & 3)
    uint32_t d_n = 0, d_kt = 0;  // DMA cursor: (tile ordinal, k-tile) of the next stage to fill
    const unsigned char* a_src[Cf::APW];  // per lane: its row of the cursor tile's queries, + chunk offset
    const unsigned char* b_src[Cf::BPW];  // per lane: its corpus row (clamped), without the chunk offset
    auto set_dma_tile = [&](uint32_t n) {
        uint32_t nt, mt;
        slot_tile(n, nt, mt);
        const uint32_t r0 = p.row_begin + nt * Cf::BR;
#pragma unroll
        for (int j = 0; j < Cf::APW; j++)
            a_src[j] = p.qprep + ((size_t)mt * BMQ + (((uint32_t)wave * Cf::APW + j) % Cf::APIECES) * 16u + rl) * p.KPB + cl * 16u;
#pragma unroll
        for (int j = 0; j < Cf::BPW; j++) {
            const uint32_t r = r0 + ((uint32_t)wave * Cf::BPW + j) * 16u + rl;
            b_src[j] = p.rows + (size_t)(r < p.row_end ? r : r0) * p.pitch;
        }
        if constexpr (Cf::RC_LDS) {  // BR / 256 1-KB pieces per array (entries r0 .. r0 + BR - 1; beyond row_end: unused)
            auto rc_tile = [&](uint32_t m) __attribute__((always_inline)) {
                constexpr int RCP = Cf::BR / 256;  // waves [0, RCP) bring array 0, waves [RCP, 2 RCP) array 1
                uint32_t mnt, mmt;
                slot_tile(m, mnt, mmt);
                uint32_t* dst = rc_s + (m & (Cf::NRC - 1)) * 2 * Cf::BR;
                const uint32_t part = (uint32_t)(wave % RCP) * 256u;
                const uint32_t e = min(p.row_begin + mnt * Cf::BR + part + 4u * (uint32_t)lane, (p.row_end - 1u) & ~3u);
                if (NEED0 && wave < RCP) __builtin_amdgcn_global_load_lds((glb_ptr_t)(arr0 + e), (lds_ptr_t)(dst + part), 16, 0, 0);
                if (NEED1 && wave >= RCP && wave < 2 * RCP)
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(arr1 + e), (lds_ptr_t)(dst + Cf::BR + part), 16, 0, 0);
            };
            if constexpr (BIAS) {  // one tile AHEAD: the next tile's bounds are computed while this one is multiplied
                if (n == 0) rc_tile(0);
                if (n + 1 < my_tiles) rc_tile(n + 1);
            } else {
                rc_tile(n);
            }
        }
    };
    // In the k-loop the pieces are issued after the groups of four MFMAs, so the address path works underneath the
    // matrix pipe instead of in a lump after the barrier.
    auto dma_piece = [&](uint32_t stage, int piece) __attribute__((always_inline)) {
        unsigned char* st = smem + stage * STAGE_B;
        if (piece < Cf::APW) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(a_src[piece] + (size_t)d_kt * DKB),
                                             (lds_ptr_t)(st + ((wave * Cf::APW + piece) % Cf::APIECES) * (16 * DKB)), 16, 0, 0);
        } else {
            const int j = piece - Cf::APW;
            const uint32_t v = d_kt * 4u + cl;  // 16-B vector of the row
            const unsigned char* src = v < p.V ? b_src[j] + (size_t)v * 16u : p.zeros;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(st + OPER_B + (wave * Cf::BPW + j) * (16 * DKB)), 16, 0, 0);
        }
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {  // cursors clamp on the block's last tile
        if (++d_kt == p.KT) {
            d_kt = 0;
            if (++d_n < my_tiles) set_dma_tile(d_n);
        }
    };
    auto dma_ktile = [&](uint32_t stage) {
#pragma unroll
        for (int piece = 0; piece < Cf::PIECES; piece++) dma_piece(stage, piece);
        dma_advance();
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
    if constexpr (!BIAS) zero_acc();

    uint32_t c_n = 0, c_kt = 0, c_nt, c_mt;  // compute cursor
    slot_tile(0, c_nt, c_mt);
    load_query_consts16<DT, METRIC, BMQ, (DT == MVF_DTYPE_INT8 && XS)>(p, c_mt * BMQ, tid, qa_s, qb_s, tau_s, thr_s);

    set_dma_tile(0);
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; st++) dma_ktile(st);
    __builtin_amdgcn_s_waitcnt(0x0F70 | Cf::INFLIGHT);  // vmcnt(INFLIGHT): k-tile 0 has landed (this wave's pieces)
    __syncthreads();                                    // also publishes the query constants

    static_assert(Cf::INFLIGHT < 16, "vmcnt low field");
    // fragment addressing: lane reads row base + (lane & 15), chunk lane >> 4, at slot = chunk ^ swz((row >> 2) & 3)
    const uint32_t frow = (uint32_t)lane & (SH - 1);
    const uint32_t fchunk = (uint32_t)lane >> 4;
    const uint32_t fslot = (fchunk ^ slot_swz((frow >> 2) & 3u)) & 3u;
    const uint32_t a_off = ((uint32_t)wm * WQ + frow) * DKB + fslot * 16u;
    const uint32_t b_off = OPER_B + ((uint32_t)wn * WR + frow) * DKB + fslot * 16u;
    auto read_a = [&](const unsigned char* st, int i) __attribute__((always_inline)) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(st + a_off + i * SH * DKB);
    };
    auto read_b = [&](const unsigned char* st, int j) __attribute__((always_inline)) -> u32x4 {
        u32x4 x = *reinterpret_cast<const u32x4*>(st + b_off + j * SH * DKB);
        if (U8) x ^= u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};  // x_u -> x_s
        return x;
    };
    auto mfma1 = [&](AccT& c, const u32x4& fa, const u32x4& fb) __attribute__((always_inline)) {
        if constexpr (DT == MVF_DTYPE_FLOAT16)
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fa), __builtin_bit_cast(half8, fb), c, 0, 0, 0);
        else
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, fa), __builtin_bit_cast(i32x4, fb), c, 0, 0, 0);
    };

    // ---- pre-filter folded into the accumulators (scan_mfma16_bias.inc): per-lane state ------------------------------
    [[maybe_unused]] int32_t br_cur[NJ];   // R(row) of the tile being multiplied (-B sits in LDS: nbq_s)
    // the wave's own slice of the block's candidate region and its running record count (wave-uniform)
    [[maybe_unused]] const uint32_t wcap = p.blk_cap / (uint32_t)NW;
    [[maybe_unused]] uint4* const wbase = p.blk_cand + ((size_t)blockIdx.x * NW + (uint32_t)wave) * wcap;
    [[maybe_unused]] uint32_t wcnt = 0;
    [[maybe_unused]] uint32_t flagged = 0;  // wave-uniform: bit i = query group i of the tile being finished holds a maximum >= 0
    constexpr bool L2Q = QSF && METRIC == MVF_METRIC_L2;
    auto rc_of = [&](uint32_t n) __attribute__((always_inline)) { return rc_s + (n & (Cf::NRC - 1)) * 2 * Cf::BR; };
    // ---- tile-invariant bounds (INVB; scan_mfma16_bias.inc): -B per lane and query once per query tile, R per row and
    // query half once per tile by the block, one tile ahead ----------------------------------------------------------------
    static_assert(!INVB || (NW * 64 == Cf::HALVES * Cf::BR && WQ == 128), "one thread per (row, query half)");
    auto tile_mt = [&](uint32_t n) __attribute__((always_inline)) -> uint32_t {
        uint32_t nt, mt;
        slot_tile(n, nt, mt);
        return mt;
    };
    auto ref_val = [&](int i) __attribute__((always_inline)) -> float {
        const uint32_t b = ref_s[i];
        return b == 0x7F800000u ? 0.0f : __uint_as_float(b);  // no row of the first tile had a finite factor: any point serves
    };
    // R of every row of block tile m, both query halves.  The tile's constants have landed (see the call sites) and no wave
    // reads buffer m mod NRB before the next barrier.
    auto inv_transform = [&](uint32_t m) __attribute__((always_inline)) {
        if constexpr (INVB) {
            uint32_t nt, mt;
            slot_tile(m, nt, mt);
            const uint32_t* rc = rc_of(m);
            const int row = tid & (Cf::BR - 1), half = tid / Cf::BR;
            float u, v;
            int32_t ri;
            inv_row_uv<DT, METRIC, XS>(NEED0 ? rc[row] : 0u, NEED1 ? rc[Cf::BR + row] : 0u, p.dim, u, v, ri);
            rb_s[((m & (Cf::NRB - 1)) * Cf::HALVES + half) * Cf::BR + row] =
                inv_row_bound<DT, METRIC, XS>(u, v, ri, ref_val(0), ref_val(1), ext_s + 4 * half, p.row_begin + nt * Cf::BR + (uint32_t)row < p.row_end);
        }
    };
    auto inv_rows = [&](uint32_t n) __attribute__((always_inline)) {
        if constexpr (INVB) {
            const int32_t* rb = rb_s + ((n & (Cf::NRB - 1)) * Cf::HALVES + wm) * Cf::BR + wn * WR + (lane & (SH - 1));
#pragma unroll
            for (int j = 0; j < NJ; j++) br_cur[j] = rb[j * SH];
        }
    };
    // once per query tile (the caller has published thr_s / qa_s): the per-query pair (P, W), the extremes of each half,
    // -B of the lane's queries; `first`: the block's reference point from the rows of its first tile
    auto inv_query_prep = [&](bool first) {
        if constexpr (INVB) {
            const float inf = __builtin_inff();
            if constexpr (L2Q) {
                if (tid < BMQ) {
                    const float th = thr_s[tid], w = 0.5f * __builtin_amdgcn_rcpf(qa_s[tid]);
                    thu_s[tid] = fabsf(th) < 3.0e38f ? (th - fabsf(th) * 1e-6f) * w : th;
                    u_s[tid] = w;
                }
            }
            if (INV_PROD && first && tid < Cf::BR) {
                uint32_t nt, mt;
                slot_tile(0, nt, mt);
                const uint32_t* rc = rc_of(0);
                float u, v;
                int32_t ri;
                inv_row_uv<DT, METRIC, XS>(NEED0 ? rc[tid] : 0u, NEED1 ? rc[Cf::BR + tid] : 0u, p.dim, u, v, ri);
                if (p.row_begin + nt * Cf::BR + (uint32_t)tid < p.row_end) {
                    if (u >= 0.0f && u < inf) atomicMin(&ref_s[0], __float_as_uint(u));
                    if (L2Q && v >= 0.0f && v < inf) atomicMin(&ref_s[1], __float_as_uint(v));
                }
            }
            __syncthreads();
            const float* parr = L2Q ? thu_s : thr_s;
            if constexpr (INV_PROD) {
                float pmin = inf, pmax = -inf, wmin = inf, wmax = 0.0f;
#pragma unroll
                for (int h = 0; h < WQ / 64; h++) {
                    const int ql = wm * WQ + h * 64 + lane;
                    const float P = parr[ql], W = L2Q ? u_s[ql] : 0.0f;
                    if (fabsf(P) < 3.0e38f && (!L2Q || (W >= 0.0f && W < 3.0e38f))) {
                        pmin = fminf(pmin, P), pmax = fmaxf(pmax, P);
                        wmin = fminf(wmin, W), wmax = fmaxf(wmax, W);
                    }
                }
                for (int off = 32; off > 0; off >>= 1) {
                    pmin = fminf(pmin, __shfl_xor(pmin, off, 64)), pmax = fmaxf(pmax, __shfl_xor(pmax, off, 64));
                    wmin = fminf(wmin, __shfl_xor(wmin, off, 64)), wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
                }
                if (!(pmin <= pmax)) pmin = pmax = wmin = wmax = 0.0f;  // no query of the half has finite constants: B decides alone
                if (wn == 0 && lane == 0) ext_s[4 * wm + 0] = pmin, ext_s[4 * wm + 1] = pmax, ext_s[4 * wm + 2] = wmin, ext_s[4 * wm + 3] = wmax;
            }
            if (tid < BMQ) {
                const uint32_t scb = L2Q ? __float_as_uint(u_s[tid]) : (U8 && METRIC == MVF_METRIC_COSINE) ? __float_as_uint(qb_s[tid]) : 0u;  // W / c_q
                nbq_s[tid] = inv_bias_query<DT, METRIC, XS>(__float_as_uint(parr[tid]), scb, ref_val(0), ref_val(1));
            }
            __syncthreads();  // the extremes and -B are published
        }
    };
    if constexpr (BIAS) {
#ifdef MVF_DIAG_NOBIAS  // diagnostic build only (with MVF_DIAG_NOEPI): the tile's bounds are not computed, the sums start from zero
        if (tid < BMQ) nbq_s[tid] = 0;
        __syncthreads();
#else
        // tile 0 (and 1, see below): their constants were requested in front of k-tile 0, which has landed
        inv_query_prep(true);
        inv_transform(0);
        if (p.KT == 1 && 1 < my_tiles && tile_mt(1) == c_mt) inv_transform(1);
        __syncthreads();
        inv_rows(0);
#endif
    }

    // MIDB: the B fragments and the first A fragment of the k-tile about to be multiplied (read during the k-tile before)
    [[maybe_unused]] u32x4 fbc[NJ], fa_first;
    if constexpr (MIDB) {
        static_assert(!MIDB || (NI == 8 && Cf::PIECES == NI / 2 && NJ == NI / 2), "one DMA piece and one B fragment per group of the second half");
#pragma unroll
        for (int j = 0; j < NJ; j++) fbc[j] = read_b(smem, j);
        fa_first = read_a(smem, 0);
    }
    uint32_t cs = 0, ds = NSTAGE - 1;  // compute stage (k-tile g), DMA target (k-tile g + NSTAGE - 1: the stage g - 1 read)
    // the flagged query groups of a finished wave tile: element walk, raw records into the wave's region
    auto walk_tile = [&](uint32_t flags, uint32_t nt, uint32_t mt) {
        if constexpr (BIAS) {
#pragma unroll
            for (int i = 0; i < NI; i++) {
#ifdef MVF_K2_OLD_WALK
                if (!(flags & (1u << i))) continue;
#else
                if (__builtin_expect(!(flags & (1u << i)), 1)) continue;  // (five groups in six: the walk sits out of line, this falls through)
#endif
                auto nb4 = [&]() __attribute__((always_inline)) { return *reinterpret_cast<const i32x4*>(&nbq_s[wm * WQ + 4 * (lane / SH) + i * SH]); };
                epilogue_group16<SH, NJ>(p, acc[i], nb4, br_cur, mt * BMQ + (uint32_t)(wm * WQ + 4 * (lane / SH) + i * SH),
                                         p.row_begin + nt * Cf::BR + (uint32_t)(wn * WR + (lane & (SH - 1))), wbase, wcap, wcnt);
            }
            wcnt = __builtin_amdgcn_readfirstlane(wcnt);
        }
    };
    for (uint32_t g = 0; g < G; g++) {
        const unsigned char* st = smem + cs * STAGE_B;
        // NI groups of NJ MFMAs (one A fragment x NJ B fragments each, the whole 64-B k in one MFMA); the next group's
        // A fragment is read while this group's MFMAs run; the DMA pieces follow the groups; with BIAS the first k-tile
        // of a tile starts every accumulator from -B
        const bool first_kt = BIAS && c_kt == 0;  // wave-uniform
        // Query group i of the tile that finishes in this iteration: is its running maximum >= 0 somewhere in the wave?  One bit
        // per group in a scalar register.
        auto drain_group = [&](int i) __attribute__((always_inline)) {
            if constexpr (BIAS) {
#ifndef MVF_DIAG_NOEPI
                const int32_t h = drain_group16<BiasTraits<DT, METRIC, XS>::HAS_BR, NJ>(acc[i], br_cur);
                if (__builtin_amdgcn_ballot_w64(h >= 0) != 0) flagged |= 1u << i;
#else
#pragma unroll
                for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(acc[i][j]));
#endif
            }
        };
        auto ktile = [&](auto first_c) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_c)::value;
            u32x4 fb[NJ], fa[2];
            [[maybe_unused]] AccT nbv[2];  // INVB, first k-tile: -B of the group's four queries, from LDS just ahead of its MFMAs
            auto read_nb = [&](int i) __attribute__((always_inline)) -> AccT {
                return *reinterpret_cast<const AccT*>(nbq_s + wm * WQ + 4 * (lane / SH) + i * SH);
            };
#pragma unroll
            for (int j = 0; j < NJ; j++) fb[j] = read_b(st, j);
            fa[0] = read_a(st, 0);
            if constexpr (BIAS && FIRST) nbv[0] = read_nb(0);
#pragma unroll
            for (int i = 0; i < NI; i++) {
#ifdef MVF_K2_OLD_FRAG_ORDER  // A/B builds only: rounds 2-4 requested the next fragment in front of the group
                if (i + 1 < NI) fa[(i + 1) & 1] = read_a(st, i + 1);
#endif
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    if constexpr (BIAS && FIRST) acc[i][j] = nbv[i & 1];  // the MFMA's C operand: no copy is emitted
                    mfma1(acc[i][j], fa[i & 1], fb[j]);
#ifndef MVF_K2_OLD_FRAG_ORDER
                    // The next group's A fragment is requested BEHIND this group's first MFMA, that is behind the wait for this
                    // group's own fragment.  With an LDS-DMA pending hipcc waits for LDS reads with lgkmcnt(0) only (it counts the
                    // DMA as a flat access that may return out of order); requested in front of the group, the new read was inside
                    // that wait and every second group began with a full LDS round trip in the open.
                    if (j == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (i + 1 < NI) {
                            fa[(i + 1) & 1] = read_a(st, i + 1);
                            if constexpr (BIAS && FIRST) nbv[(i + 1) & 1] = read_nb(i + 1);
                        }
                    }
#endif
                }
                if (NI == 8 ? (i & 1) == 0 : true) dma_piece(ds, NI == 8 ? i / 2 : i);          // 4 pieces over 8 groups (a 5th behind group 1), or
                if (NI == 8 && i == 1 && Cf::PIECES > 4) dma_piece(ds, 4);
                if (NI == 4 && i == NI - 1 && Cf::PIECES > NI) dma_piece(ds, NI);                    // 5 pieces over 4 groups
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // MIDB: [groups 0..3 on the fragments in hand] wait(k-tile g + 1 landed) barrier [groups 4..7, reading k-tile g + 1's B
        // fragments and first A fragment underneath; the DMA of k-tile g + 3 goes into the stage k-tile g - 1 used: every wave is
        // past its reads of that stage at this barrier].  One k-tile stays in flight across the barrier.
        auto ktile_mid = [&](auto first_c) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_c)::value;
            const unsigned char* stn = smem + (cs + 1 == NSTAGE ? 0 : cs + 1) * STAGE_B;
            u32x4 fbn[NJ], fa[2];
            [[maybe_unused]] AccT nbv[2];
            auto read_nb = [&](int i) __attribute__((always_inline)) -> AccT {
                return *reinterpret_cast<const AccT*>(nbq_s + wm * WQ + 4 * (lane / SH) + i * SH);
            };
            fa[0] = fa_first;
            if constexpr (BIAS && FIRST) nbv[0] = read_nb(0);
#pragma unroll
            for (int i = 0; i < NI; i++) {
                if (i == NI / 2) {
                    __builtin_amdgcn_s_waitcnt(0x0F70 | Cf::PIECES);  // vmcnt(PIECES): k-tile g + 1 has landed (this wave's pieces)
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    if constexpr (BIAS && FIRST) acc[i][j] = nbv[i & 1];
                    mfma1(acc[i][j], fa[i & 1], fbc[j]);
                    if (j == 0) {  // requests behind the group's first MFMA (see ktile)
                        __builtin_amdgcn_sched_barrier(0);
                        fa[(i + 1) & 1] = i + 1 < NI ? read_a(st, i + 1) : read_a(stn, 0);
                        if (i >= NI / 2) fbn[i - NI / 2] = read_b(stn, i - NI / 2);
                        if constexpr (BIAS && FIRST) {
                            if (i + 1 < NI) nbv[(i + 1) & 1] = read_nb(i + 1);
                        }
                    }
                }
                if (i >= NI / 2) dma_piece(ds, i - NI / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < NJ; j++) fbc[j] = fbn[j];
            fa_first = fa[NI & 1];
        };
        if (first_kt) {
            // The NEXT tile's row constants: requested (set_dma_tile) in front of this tile's first k-tile, which the last wait
            // has seen land -- the constants are older -- and read at this tile's end, at least one barrier from here when a
            // tile has two k-tiles or more.  With ONE k-tile per tile the tile after the next is taken instead: the DMA cursor
            // runs NSTAGE - 1 tiles ahead then, its constants were requested behind the pieces of tile c_n + 1 at the latest,
            // and those are older than everything the last wait left in flight.
#ifndef MVF_DIAG_NOBIAS
            // R of the next tile (the one after it when a tile is ONE k-tile), if it is multiplied with the queries in place
            // now; a change of the query tile computes its own (below)
            const uint32_t m = c_n + (p.KT == 1 ? 2u : 1u);
            if (m < my_tiles && tile_mt(m) == c_mt && (p.KT != 1 || tile_mt(c_n + 1) == c_mt)) inv_transform(m);
#endif
            if constexpr (MIDB) ktile_mid(std::true_type{});
            else ktile(std::true_type{});
        } else {
            if constexpr (MIDB) ktile_mid(std::false_type{});
            else ktile(std::false_type{});
        }
        dma_advance();
        cs = cs + 1 == NSTAGE ? 0 : cs + 1;
        ds = ds + 1 == NSTAGE ? 0 : ds + 1;
        if (++c_kt == p.KT) {  // tile finished: its successor's first k-tiles are already in the ring
            if constexpr (BIAS) {
#ifndef MVF_DIAG_NOEPI
                if (lane == 0) MVF_DIAG_ADD(0, 1);
                // Every group's running maximum is taken HERE, behind the loop, and the rare walk in front of the barrier.  Round 5
                // built both alternatives -- the maxima inside the tile's last k-tile, one group behind the MFMAs; the walk behind
                // the barrier, beside the partner wave's next k-tile -- and each lost: VALU inside the MFMA stream +3.7...4.0 %, the
                // deferred walk another +1.1...1.6 % (profiles/r05_k2_drain_walk_placement_ab.txt).
#pragma unroll
                for (int i = 0; i < NI; i++) drain_group(i);
                if (flagged) walk_tile(flagged, c_nt, c_mt);
                flagged = 0;
                wcnt = __builtin_amdgcn_readfirstlane(wcnt);
#else
#pragma unroll
                for (int i = 0; i < NI; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(acc[i][j]));
#endif
                c_kt = 0;
                if (++c_n < my_tiles) {
                    uint32_t nmt;
                    slot_tile(c_n, c_nt, nmt);
                    if (nmt != c_mt) {  // block-uniform; rare
                        __syncthreads();
                        load_query_consts16<DT, METRIC, BMQ, (DT == MVF_DTYPE_INT8 && XS)>(p, nmt * BMQ, tid, qa_s, qb_s, tau_s, thr_s);
                        c_mt = nmt;
                        __syncthreads();
#ifndef MVF_DIAG_NOBIAS
                        inv_query_prep(false);
                        inv_transform(c_n);
                        if (p.KT == 1 && c_n + 1 < my_tiles && tile_mt(c_n + 1) == c_mt) inv_transform(c_n + 1);
                        __syncthreads();
#endif
                    }
#ifndef MVF_DIAG_NOBIAS
                    inv_rows(c_n);
#endif
                }
            } else {
#ifdef MVF_DIAG_NOEPI  // diagnostic build only: the k-loop alone (the sums are kept alive, nothing is selected)
#pragma unroll
                for (int i = 0; i < NI; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(acc[i][j]));
                if (false)
#endif
                {
                    const uint32_t* rc = rc_s + (c_n & (Cf::NRC - 1)) * 2 * Cf::BR;
                    epilogue16<DT, METRIC, DIRECT, XS, BMQ, SH, WQ, WR, Cf::BR, Cf::RC_LDS>(p, acc, c_nt, c_mt, wm, wn, lane, qa_s, qb_s, tau_s,
                                                                                           thr_s, rc, rc + Cf::BR, p.blk_cand ? bc_s : nullptr);
                }
                zero_acc();
                c_kt = 0;
                if (++c_n < my_tiles) {
                    uint32_t nmt;
                    slot_tile(c_n, c_nt, nmt);
                    if (nmt != c_mt) {  // block-uniform; rare
                        __syncthreads();
                        load_query_consts16<DT, METRIC, BMQ, (DT == MVF_DTYPE_INT8 && XS)>(p, nmt * BMQ, tid, qa_s, qb_s, tau_s, thr_s);
                        c_mt = nmt;
                    }
                }
            }
        }
        if constexpr (!MIDB) {
            // k-tile g + 1 must have landed; the younger k-tile(s) stay in flight across the barrier
            __builtin_amdgcn_s_waitcnt(0x0070 | Cf::INFLIGHT);  // vmcnt(INFLIGHT) lgkmcnt(0)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    // the DMAs issued for k-tiles past the end target this block's own LDS: let them land before the wave exits
    __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00);  // vmcnt(0)
    if constexpr (MIDB && !BIAS) __syncthreads();  // the last epilogues lie behind the loop's last barrier
    if constexpr (BIAS) {
        if (lane == 0) p.blk_cnt[blockIdx.x * NW + (uint32_t)wave] = min(wcnt, wcap);  // one region per wave (scan_mfma16_dma_wave_regions)
    } else {
        if (tid == 0 && p.blk_cnt) p.blk_cnt[blockIdx.x] = min(*bc_s, p.blk_cap);  // behind the loop's last barrier: every epilogue is done
    }
}

template <int DT, int METRIC, int BMQ>
hipError_t launch_dtm(const Batch16Params& p, dim3 grid, hipStream_t s) {
    constexpr bool XSOK = DT == MVF_DTYPE_FLOAT16 || DT == MVF_DTYPE_INT8;  // rows may be a scaled shadow (f16, or the int8 shadow)
    constexpr bool REGOK = DT != MVF_DTYPE_FLOAT16 && CfT<BMQ>::RC_LDS;     // flavours with the folded pre-filter
    const bool xs = XSOK && p.xscale, reg = REGOK && !p.direct && p.blk_cand && p.wave_regions;
    void (*fn)(Batch16Params);
    if (p.direct) fn = xs ? &scan_mfma16_dma_kernel<DT, METRIC, true, XSOK, BMQ, false> : &scan_mfma16_dma_kernel<DT, METRIC, true, false, BMQ, false>;
    else if (reg) fn = xs ? &scan_mfma16_dma_kernel<DT, METRIC, false, XSOK, BMQ, REGOK> : &scan_mfma16_dma_kernel<DT, METRIC, false, false, BMQ, REGOK>;
    else fn = xs ? &scan_mfma16_dma_kernel<DT, METRIC, false, XSOK, BMQ, false> : &scan_mfma16_dma_kernel<DT, METRIC, false, false, BMQ, false>;
    // > 64 KiB of dynamic LDS needs the attribute; it is per device, and one process may drive several devices
    const size_t lds = CfT<BMQ>::LDS;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, grid, dim3(512), lds, s, p);
    return hipGetLastError();
}

template <int DT, int BMQ>
hipError_t launch_dt(const Batch16Params& p, int metric, dim3 grid, hipStream_t s) {
    switch (metric) {
    case MVF_METRIC_L2: return launch_dtm<DT, MVF_METRIC_L2, BMQ>(p, grid, s);
    case MVF_METRIC_INNER_PRODUCT: return launch_dtm<DT, MVF_METRIC_INNER_PRODUCT, BMQ>(p, grid, s);
    default: return launch_dtm<DT, MVF_METRIC_COSINE, BMQ>(p, grid, s);
    }
}

template <int BMQ>
hipError_t launch_bmq(const Batch16Params& p, int dtype, int metric, dim3 grid, hipStream_t s) {
    if (dtype == MVF_DTYPE_FLOAT16) return launch_dt<MVF_DTYPE_FLOAT16, BMQ>(p, metric, grid, s);
    if (dtype == MVF_DTYPE_UINT8) return launch_dt<MVF_DTYPE_UINT8, BMQ>(p, metric, grid, s);
    return launch_dt<MVF_DTYPE_INT8, BMQ>(p, metric, grid, s);
}

}  // namespace

#ifdef MVF_DIAG_COUNT
extern "C" int mvfgpu_diag_bias_counts(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bias_diag), 64) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bias_diag), z, 64) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// queries per block tile for a batch of nq: the 64-query tile (HBM-bound) up to 64 queries, the 128-query tile up to 128,
// else the 256-query tile
uint32_t scan_mfma16_dma_queries_per_block(uint32_t nq, int forced_tile) {
    if (forced_tile) return forced_tile == 64 ? 64u : forced_tile == 128 ? 128u : 256u;  // MVF_K2_TILE
    return nq <= 64 ? 64u : nq <= 128 ? 128u : 256u;
}

uint32_t scan_mfma16_dma_tile_rows(uint32_t bmq) { return bmq == 256 ? CfT<256>::BR : CfT<64>::BR; }  // 128: as 64
static_assert(CfT<128>::BR == CfT<64>::BR && CfT<128>::PIECES == 5 && CfT<256>::PIECES == 4 && CfT<64>::PIECES == 5, "tile shapes");
static_assert(CfT<128>::LDS <= 160 * 1024 && CfT<64>::LDS <= 160 * 1024 && CfT<256>::LDS <= 160 * 1024, "LDS");

// Does a launch with these properties hand RAW records to per-WAVE regions (8 per block, blk_cap / 8 records each, counts
// in blk_cnt[block * 8 + wave]) instead of keyed records to the block's region?  (the folded pre-filter of the
// i32-accumulator flavours, scan_mfma16_bias.inc; the caller passes the scatter pass 8 x the regions at 1/8 the capacity)
bool scan_mfma16_dma_wave_regions(int dtype, uint32_t bmq, bool direct, bool has_regions, uint32_t dim) {
    return dtype != MVF_DTYPE_FLOAT16 && bmq >= 128 && !direct && has_regions && dim <= 8192u;  // wider rows: sums beyond 2^27
}

// p.KPB / p.KT are in 64-byte k-tiles here; p.zeros points at >= 16 zero bytes; p.mtiles = nq_pad / bmq;
// p.ntiles = ceil(rows / scan_mfma16_dma_tile_rows(bmq)).
hipError_t launch_scan_mfma16_dma(const Batch16Params& p, int dtype, int metric, int num_cus, uint32_t bmq, bool persistent,
                                  hipStream_t s) {
    const uint32_t total = ((p.ntiles + 7) / 8) * p.mtiles * 8;
    uint32_t nls = std::max(1u, (uint32_t)num_cus / 8u);
    if (nls > p.mtiles) nls -= nls % p.mtiles;
    const dim3 grid(persistent ? std::min(total, nls * 8u) : total);
    Batch16Params q = p;
    if (grid.x > kBlkMaxBlocks || (q.wave_regions && grid.x > (uint32_t)num_cus))  // one candidate region per block: small grids only
        q.blk_cand = nullptr, q.blk_cnt = nullptr, q.wave_regions = 0;
    return bmq == 64 ? launch_bmq<64>(q, dtype, metric, grid, s) : bmq == 128 ? launch_bmq<128>(q, dtype, metric, grid, s)
                                                                                : launch_bmq<256>(q, dtype, metric, grid, s);
}

}  // namespace mvf
