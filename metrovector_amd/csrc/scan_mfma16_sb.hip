// scan_mfma16_sb.hip — K2 for SMALL BATCHES (up to 64 queries) of the narrow types: a streaming kernel with MFMA dots.
//
// Up to 64 queries the scan is HBM-bound (a quarter of the 256-query tile's MFMA work per corpus byte), so what counts
// is how the corpus bytes are fetched.  The 64-query shape of scan_mfma16_dma.hip moves them in the MFMA fragment's
// own shape -- 16 rows x 64 B per 1-KB DMA piece -- and reaches 5.4 TB/s; that shape tops out at 6.0 TB/s on this part,
// the streaming kernel's (4 rows x 256 contiguous bytes per wave-instruction) at 6.9
// (profiles/r02_access_shape_read_bandwidth.txt).  This kernel fetches in the second shape and lets LDS do the
// transposition into fragments:
//
//   * the query tile stays RESIDENT in LDS for the whole launch, already in fragment order (KT x NI fragments of 1 KB:
//     48 KB for 64 int8 queries of dimension 768), read with lane-contiguous ds_read_b128;
//   * every WAVE streams its own 16-row groups through a private LDS ring: a stage = 16 rows x 256 B = four
//     global_load_lds_dwordx4 (lane L -> row 4 t + L / 16, 16-byte chunk L % 16), i.e. four MFMA k-steps; the B
//     fragment of k-step s is read back as lane -> (row lane % 16, chunk 4 s + lane / 16).  A row is 256 B wide in that
//     image, so the 16 rows of one chunk would share a bank group: the chunk is stored at slot chunk ^ row, applied on
//     the DMA's per-lane SOURCE address (a row's 16 lanes still cover its 256 contiguous bytes);
//   * the group's per-row constants (norms, shadow scale, UInt8 bias) come in as two dword-wide DMA instructions with
//     its first stage;
//   * no barrier anywhere in the loop and nothing shared between waves but the queries: a wave waits for its own DMA
//     (counted s_waitcnt vmcnt), eight waves per CU keep NST - 1 stages each in flight (64 KB of corpus bytes per CU
//     with three stages), and a wave that sits in its epilogue stalls nobody;
//   * same epilogue (scan_mfma16_common.inc), thresholds, candidate regions, phases and compactions as the tile kernels.
//
// k beyond a row's pitch reads zeros (p.zeros); rows past row_end re-read the group's first row (discarded by the
// epilogue); UInt8's x ^ 0x80 is applied to the B fragments.
//
// Hazards (per wave; nothing crosses waves): RAW -- the vmcnt wait covers the stage about to be read, and LDS-DMA data is
// visible to the issuing wave once its vmcnt says so.  WAR -- a stage is re-filled one iteration after it was read; its
// fragment reads have returned by then (their values fed MFMAs), and an explicit lgkmcnt(0) sits in front of the refill.

#include "scan_mfma.h"

#include "mvf_common.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <type_traits>

namespace mvf {
namespace {

#include "scan_mfma16_common.inc"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SB_NW = 8;            // waves per block
// Two tile widths.  NQT = 64 queries (NI = 4 MFMA sub-tiles): ring stages of 16 rows x 256 B (four k-steps, four DMA
// instructions of 4 rows x 256 B).  NQT = 128 queries (NI = 8): the query tile takes up to 96 KB of LDS, so the stages
// are 16 rows x 128 B (two k-steps, two instructions of 8 rows x 128 B -- still whole 128-byte lines).
template <int NQT> struct SbT {
    static constexpr int Q = NQT;
    static constexpr int NI = NQT / 16;
    static constexpr int SRB = NQT == 64 ? 256 : 128;  // stage bytes per row
    static constexpr int CPR = SRB / 16;               // 16-byte chunks per row and stage
    static constexpr int RPI = 64 / CPR;               // rows per DMA instruction
    static constexpr int IPS = 16 / RPI;               // DMA instructions per stage
    static constexpr int KPS = SRB / 64;               // k-steps per stage
    static constexpr int STAGE = 16 * SRB;             // bytes of one ring stage
    static constexpr int MAXST = NQT == 64 ? 6 : 8;    // ring stages per wave at most
};
constexpr int SB_RCSLOTS = 8;       // per wave: row-constant buffers of the groups in flight (group ordinal mod 8 >= the stages a ring holds)
constexpr int SB_RC = SB_NW * SB_RCSLOTS * 2 * 16 * 4;  // two arrays x 16 dwords each
constexpr int sb_aux(int nqt) { return 4 * nqt * 4 + SB_RC + 16; }
constexpr size_t SB_LDS_MAX = 156 * 1024;  // of the CU's 160 KB
// cache policy of the corpus-row DMA: nontemporal (aux bit 1) -- the rows are read once; with the default policy they
// wash through L2 and the same kernel is 7-12 % slower (1.51 vs 1.35 ms per 8-query search of 10M x 768).  (The tile
// kernels must NOT do this: their 64-byte pieces fetch a 128-byte line in two halves, one k-tile apart, and the second
// half then comes from HBM again: cfg4 11.5 -> 13.2 ms.)
constexpr int SB_AUX_POLICY = 2;

// vmcnt(n) only (lgkmcnt / expcnt untouched), n even, up to 20 (larger: 20 -- stricter, still correct)
__device__ __forceinline__ void wait_vmcnt(uint32_t n) {
    switch (n) {
    case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    case 2: __builtin_amdgcn_s_waitcnt(0x0F72); break;
    case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
    case 6: __builtin_amdgcn_s_waitcnt(0x0F76); break;
    case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break;
    case 10: __builtin_amdgcn_s_waitcnt(0x0F7A); break;
    case 12: __builtin_amdgcn_s_waitcnt(0x0F7C); break;
    case 14: __builtin_amdgcn_s_waitcnt(0x0F7E); break;
    case 16: __builtin_amdgcn_s_waitcnt(0x4F70); break;
    case 18: __builtin_amdgcn_s_waitcnt(0x4F72); break;
    default: __builtin_amdgcn_s_waitcnt(0x4F74); break;  // 20
    }
}

template <int DT, int METRIC, bool DIRECT, bool XS, int NQT>
__global__ void __launch_bounds__(512, 1) scan_mfma16_sb_kernel(Batch16Params p, uint32_t nst, uint32_t ni) {
    using AccT = typename std::conditional<DT == MVF_DTYPE_FLOAT16, f32x4, i32x4>::type;
    using Sb = SbT<NQT>;
    constexpr int NI = Sb::NI, SH = 16, SB_Q = NQT, SB_STAGE = Sb::STAGE;
    constexpr bool U8 = DT == MVF_DTYPE_UINT8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t KT = p.KT;                        // 64-byte k-steps of a (padded) query row
    const uint32_t S4 = (KT + Sb::KPS - 1u) / Sb::KPS;  // ring stages per 16-row group
    unsigned char* a_s = smem;                       // [KT][ni][1 KB] query fragments
    unsigned char* ring0 = a_s + (size_t)KT * ni * 1024u;
    float* qa_s = reinterpret_cast<float*>(ring0 + (size_t)SB_NW * nst * SB_STAGE);
    uint32_t* tau_s = reinterpret_cast<uint32_t*>(qa_s + SB_Q);
    float* qb_s = reinterpret_cast<float*>(tau_s + SB_Q);
    float* thr_s = qb_s + SB_Q;
    uint32_t* rc_s = reinterpret_cast<uint32_t*>(thr_s + SB_Q);  // [NW][SB_RCSLOTS][2][16]
    uint32_t* bc_s = rc_s + SB_RC / 4;
    // per-row constants the epilogue needs (scan_mfma16_common.inc): array 0 = norms, array 1 = shadow scale / UInt8 bias.
    // They ride along with the group's first stage as two dword-wide LDS-DMA instructions: a global load in the epilogue
    // would have to wait for every DMA issued before it (vmcnt counts in order) -- the whole ring, once per group.
    constexpr bool QSF = DT == MVF_DTYPE_FLOAT16 || (DT == MVF_DTYPE_INT8 && XS);  // float scores
    constexpr bool NEED0 = METRIC != MVF_METRIC_INNER_PRODUCT;
    constexpr bool NEED1 = QSF ? XS : (U8 && METRIC != MVF_METRIC_L2);
    const uint32_t* arr0 = QSF ? reinterpret_cast<const uint32_t*>(METRIC == MVF_METRIC_COSINE ? p.xnorm_f : p.xx2)
                               : reinterpret_cast<const uint32_t*>(p.xnorm_i);
    const uint32_t* arr1 = QSF ? reinterpret_cast<const uint32_t*>(p.xscale) : reinterpret_cast<const uint32_t*>(p.xbias_i);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* ring = ring0 + (size_t)wave * nst * SB_STAGE;

    // ---- prologue: per-query constants, the query tile in fragment order --------------------------------------------
    load_query_consts16<DT, METRIC, SB_Q, (DT == MVF_DTYPE_INT8 && XS)>(p, 0u, tid, qa_s, qb_s, tau_s, thr_s);
    if (tid == 0) *bc_s = 0;
    for (uint32_t idx = tid; idx < KT * ni * 64u; idx += 512u) {  // one 16-byte chunk per thread and trip
        const uint32_t f = idx >> 6, l = idx & 63u, ks = f / ni, i = f % ni;
        const uint32_t q = i * 16u + (l & 15u), off = ks * 64u + (l >> 4) * 16u;
        *reinterpret_cast<u32x4*>(a_s + (size_t)f * 1024u + l * 16u) = *reinterpret_cast<const u32x4*>(p.qprep + (size_t)q * p.KPB + off);
    }
    __syncthreads();

    // ---- this wave's share of the 16-row groups -------------------------------------------------------------------------
    // (a CONTIGUOUS range per wave: consecutive groups of a wave lie next to each other in memory -- with a round-robin
    // assignment every group opened a new DRAM page and a new TLB entry)
    const uint32_t ngroups = (p.row_end - p.row_begin + 15u) / 16u;
    const uint32_t W = gridDim.x * SB_NW, per = (ngroups + W - 1u) / W;
    const uint32_t gw = (blockIdx.x * SB_NW + (uint32_t)wave) * per;  // first group of this wave
    const uint32_t my_groups = gw < ngroups ? min(per, ngroups - gw) : 0u;
    const uint32_t items = my_groups * S4;  // (group, stage) pairs in issue order

    // Cursors over the wave's (group, stage) items -- one for the DMA, one for the MFMAs -- advanced incrementally (the
    // divisions and remainders of an item number cost more scalar instructions than the item's arithmetic).
    struct Cursor {
        uint32_t g, s4, slot, rcs;  // group, stage within the group, ring slot, row-constant slot
    };
    Cursor dc{gw, 0u, 0u, 0u}, cc{gw, 0u, 0u, 0u};
    auto advance = [&](Cursor& c) __attribute__((always_inline)) {
        c.slot = c.slot + 1u == nst ? 0u : c.slot + 1u;
        if (++c.s4 == S4) {
            c.s4 = 0u;
            c.g += 1u;
            c.rcs = (c.rcs + 1u) & (SB_RCSLOTS - 1u);
        }
    };
    // DMA of the cursor's item: IPS instructions, lane L -> row RPI t + L / CPR, source chunk (L % CPR) ^ swz(row).
    // swz: the fragment reads below take 16 rows of ONE chunk at a time; in the 256-byte image a row is one pass over the
    // banks, so the chunk moves by the row (slot = chunk ^ row); in the 128-byte image two rows share a pass (slot =
    // chunk ^ (row >> 1), the row's parity picks the half).
    const uint32_t lrow = (uint32_t)lane / Sb::CPR, lslot = (uint32_t)lane % Sb::CPR;
    auto swz = [](uint32_t row) __attribute__((always_inline)) -> uint32_t { return Sb::CPR == 16 ? (row & 15u) : ((row >> 1) & 7u); };
    auto issue = [&]() __attribute__((always_inline)) {
        const uint32_t r0 = p.row_begin + dc.g * 16u;
        unsigned char* st = ring + (size_t)dc.slot * SB_STAGE;
#pragma unroll
        for (int t = 0; t < Sb::IPS; t++) {
            const uint32_t rl = (uint32_t)Sb::RPI * t + lrow, r = r0 + rl;
            const uint32_t v = dc.s4 * Sb::CPR + (lslot ^ swz(rl));  // 16-byte vector of the row
            const unsigned char* src = v < p.V ? p.rows + (size_t)(r < p.row_end ? r : r0) * p.pitch + (size_t)v * 16u : p.zeros;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(st + t * 1024), 16, 0, SB_AUX_POLICY);
        }
        if (dc.s4 == 0u) {  // the group's row constants (entry lane % 16; rows past row_end repeat the last valid one)
            uint32_t* dst = rc_s + (((uint32_t)wave * SB_RCSLOTS + dc.rcs) * 2u) * 16u;
            const uint32_t e = min(r0 + ((uint32_t)lane & 15u), p.row_end - 1u);
            if (lane < 16) {  // sixteen lanes, one dword each
                if (NEED0) __builtin_amdgcn_global_load_lds((glb_ptr_t)(arr0 + e), (lds_ptr_t)dst, 4, 0, 0);
                if (NEED1) __builtin_amdgcn_global_load_lds((glb_ptr_t)(arr1 + e), (lds_ptr_t)(dst + 16), 4, 0, 0);
            }
        }
        advance(dc);
    };

    AccT acc[NI][1];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][0][e] = 0;
    };
    zero_acc();
    auto mfma1 = [&](AccT& c, const u32x4& fa, const u32x4& fb) __attribute__((always_inline)) {
        if constexpr (DT == MVF_DTYPE_FLOAT16)
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fa), __builtin_bit_cast(half8, fb), c, 0, 0, 0);
        else
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, fa), __builtin_bit_cast(i32x4, fb), c, 0, 0, 0);
    };

    const uint32_t ahead = nst - 1u;  // stages in flight behind the one being read
    for (uint32_t it = 0; it < ahead && it < items; it++) issue();
    const uint32_t frow = (uint32_t)lane & 15u, fq = (uint32_t)lane >> 4;  // fragment read: row, chunk within the k-step
    for (uint32_t it = 0; it < items; it++) {
        // refill the stage read in the previous iteration (its fragment reads are done), then wait for this one
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        asm volatile("" ::: "memory");
        if (it + ahead < items) issue();
        const uint32_t younger = min(ahead, items - 1u - it);  // stages issued after this one (their row constants, if
        wait_vmcnt((uint32_t)Sb::IPS * younger);                // any, only make the wait stricter)
        asm volatile("" ::: "memory");

        const unsigned char* st = ring + (size_t)cc.slot * SB_STAGE;
#pragma unroll
        for (int k4 = 0; k4 < Sb::KPS; k4++) {
            const uint32_t ks = cc.s4 * Sb::KPS + k4;
            if (ks < KT) {  // wave-uniform
                u32x4 fb = *reinterpret_cast<const u32x4*>(st + frow * (uint32_t)Sb::SRB + (((uint32_t)k4 * 4u + fq) ^ swz(frow)) * 16u);
                if (U8) fb ^= u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};  // x_u -> x_s
                const unsigned char* af = a_s + (size_t)ks * ni * 1024u + (uint32_t)lane * 16u;
#pragma unroll
                for (int i = 0; i < NI; i++)
                    if ((uint32_t)i < ni) mfma1(acc[i][0], *reinterpret_cast<const u32x4*>(af + i * 1024), fb);
            }
        }
        if (cc.s4 + 1u == S4) {  // the group is complete
            const uint32_t* rc = rc_s + (((uint32_t)wave * SB_RCSLOTS + cc.rcs) * 2u) * 16u;
            epilogue16<DT, METRIC, DIRECT, XS, SB_Q, SH, SB_Q, 16, 16, true>(p, acc, cc.g, 0u, 0, 0, lane, qa_s, qb_s, tau_s, thr_s, rc, rc + 16,
                                                                             p.blk_cand ? bc_s : nullptr);
            zero_acc();
        }
        advance(cc);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();                      // every wave's last epilogue has counted its candidates
    if (tid == 0 && p.blk_cnt) p.blk_cnt[blockIdx.x] = min(*bc_s, p.blk_cap);
}

template <int DT, int METRIC, int NQT>
hipError_t launch_dtm(const Batch16Params& p, dim3 grid, size_t lds, uint32_t nst, uint32_t ni, hipStream_t s) {
    void (*fn)(Batch16Params, uint32_t, uint32_t) =
        p.direct ? &scan_mfma16_sb_kernel<DT, METRIC, true, false, NQT> : &scan_mfma16_sb_kernel<DT, METRIC, false, false, NQT>;
    if constexpr (DT == MVF_DTYPE_FLOAT16 || DT == MVF_DTYPE_INT8)  // rows are a scaled shadow (f16, or the int8 shadow)
        if (p.xscale)
            fn = p.direct ? &scan_mfma16_sb_kernel<DT, METRIC, true, true, NQT> : &scan_mfma16_sb_kernel<DT, METRIC, false, true, NQT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, grid, dim3(512), lds, s, p, nst, ni);
    return hipGetLastError();
}

template <int DT, int NQT>
hipError_t launch_dt(const Batch16Params& p, int metric, dim3 grid, size_t lds, uint32_t nst, uint32_t ni, hipStream_t s) {
    switch (metric) {
    case MVF_METRIC_L2: return launch_dtm<DT, MVF_METRIC_L2, NQT>(p, grid, lds, nst, ni, s);
    case MVF_METRIC_INNER_PRODUCT: return launch_dtm<DT, MVF_METRIC_INNER_PRODUCT, NQT>(p, grid, lds, nst, ni, s);
    default: return launch_dtm<DT, MVF_METRIC_COSINE, NQT>(p, grid, lds, nst, ni, s);
    }
}

template <int NQT>
hipError_t launch_nqt(const Batch16Params& p, int dtype, int metric, dim3 grid, size_t lds, uint32_t nst, uint32_t ni, hipStream_t s) {
    if (dtype == MVF_DTYPE_FLOAT16) return launch_dt<MVF_DTYPE_FLOAT16, NQT>(p, metric, grid, lds, nst, ni, s);
    if (dtype == MVF_DTYPE_UINT8) return launch_dt<MVF_DTYPE_UINT8, NQT>(p, metric, grid, lds, nst, ni, s);
    return launch_dt<MVF_DTYPE_INT8, NQT>(p, metric, grid, lds, nst, ni, s);
}

// the tile width for nq queries, the 16-query sub-tiles that hold real queries, and the ring stages per wave that fit
// beside them (0: the query fragments do not leave room for two)
struct SbShape {
    uint32_t nqt, ni, nst;
    size_t lds;
};
SbShape sb_shape(uint32_t KT, uint32_t nq) {
    SbShape sh{};
    sh.nqt = nq <= 64u ? 64u : 128u;
    sh.ni = nq <= 16u ? 1u : nq <= 32u ? 2u : (nq + 15u) / 16u;
    if (sh.nqt == 64u && sh.ni == 3u) sh.ni = 4u;
    const size_t stage = sh.nqt == 64u ? SbT<64>::STAGE : SbT<128>::STAGE;
    const uint32_t maxst = sh.nqt == 64u ? SbT<64>::MAXST : SbT<128>::MAXST;
    const size_t fixed = (size_t)KT * sh.ni * 1024u + sb_aux((int)sh.nqt);
    if (nq > 128u || KT == 0 || fixed + (size_t)SB_NW * 2u * stage > SB_LDS_MAX) return sh;  // nst = 0
    sh.nst = (uint32_t)std::min<size_t>(maxst, (SB_LDS_MAX - fixed) / ((size_t)SB_NW * stage));
    sh.lds = fixed + (size_t)SB_NW * sh.nst * stage;
    return sh;
}

}  // namespace

// One query tile of at most 128 queries (p.nq_pad <= 128: the prepared queries are one [nq_pad][KPB] array) whose
// fragments fit in LDS beside two ring stages per wave; p.KT / p.KPB in 64-byte k-steps as for the LDS-DMA kernel.
bool scan_mfma16_sb_usable(uint32_t nq_pad, uint32_t KT, uint32_t nq) {
    if (nq_pad > 128u || nq > nq_pad) return false;
    return sb_shape(KT, nq).nst >= 2u;
}

hipError_t launch_scan_mfma16_sb(const Batch16Params& p, int dtype, int metric, int num_cus, hipStream_t s) {
    const SbShape sh = sb_shape(p.KT, p.nq);
    if (sh.nst < 2u) return hipErrorInvalidValue;
    const uint32_t ngroups = (p.row_end - p.row_begin + 15u) / 16u;
    const uint32_t blocks = std::max(1u, std::min<uint32_t>((uint32_t)num_cus, (ngroups + SB_NW - 1u) / SB_NW));
    Batch16Params q = p;
    if (blocks > kBlkMaxBlocks) q.blk_cand = nullptr, q.blk_cnt = nullptr;
    const dim3 grid(blocks);
    return sh.nqt == 64u ? launch_nqt<64>(q, dtype, metric, grid, sh.lds, sh.nst, sh.ni, s)
                         : launch_nqt<128>(q, dtype, metric, grid, sh.lds, sh.nst, sh.ni, s);
}

}  // namespace mvf
