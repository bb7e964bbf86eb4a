// scan_mfma16.hip — K2 for the narrow types, REGISTER-STAGED variant (MVF_K2_DMA=0; the default is the LDS-DMA
// ring of scan_mfma16_dma.hip, same results, ~7 % faster — this one is kept as its A/B reference): Float16 rows on
// v_mfma_f32_32x32x16_f16 and Int8 rows on v_mfma_i32_32x32x32_i8.  Both
// instructions take 16 bytes per lane per operand, so staging, LDS image and
// fragment fetches are byte-identical; only the MFMA, the accumulator type and
// the epilogue differ.
//
// Same role as scan_mfma.hip (the batched form of the reference loop,
// examples/similarity_search.rs:147-169), different balance: these MFMAs are
// 16-32x faster than the f32 one, so the tile is 256 corpus rows x 256 A rows
// (8 waves as 2 x 4, each 128 x 64 outputs) — smaller tiles would be bound by
// L2->LDS traffic, not by the matrix cores or HBM.
//
//   Int8    : A rows = 256 queries (int8), exact i32 accumulation -> bit-exact
//             dot / L2 (qq + xx - 2 dot) / cosine, identical to K1 and the CPU.
//   Float16 : the reference semantics are "f32 query x exactly-widened f16 row"
//             (Vector::as_f32, src/vectors/vector.rs:81-89).  An f16 MFMA needs
//             an f16 query: each query is scaled by a power of two (max |q| into
//             [2^14, 2^15)) and ROUNDED to one f16 plane, q~ = f16(q 2^e).  This
//             kernel only SELECTS with it: |q~.x 2^-e - q.x| <= 2^-11 |q||x|
//             (Cauchy-Schwarz over the per-element rounding), so the scores are
//             approximate with a proven bound; compact_margin_kernel keeps
//             every row within twice that bound of the k-th and rescore_kernel
//             recomputes the kept rows from the f32 query (scan_mfma.hip).  An
//             exact hi+lo two-plane split needs twice the MFMA, LDS and L2
//             traffic for precision that only ~k rows per query ever use.
//             A rows = 256 queries, like Int8.
//
// k-tile = 128 bytes per row (128 int8 / 64 f16), 4 MFMA k-steps of 32 bytes;
// LDS rows padded to 144 B (conflict-free ds_read_b128); two LDS stages; the
// exact k order inside a step does not matter (A and B share the lane->k map).

#include "scan_mfma.h"

#include "mvf_common.h"

#include <hip/hip_fp16.h>

#include <cstdlib>

namespace mvf {
namespace {

#include "scan_mfma16_common.inc"

constexpr int BKB = 128;                    // k-tile bytes per row
constexpr int LDPB = BKB + 16;              // padded LDS row pitch in bytes
constexpr int TILE_B = AROWS * LDPB;        // bytes per operand tile per stage
constexpr size_t kLds16 = (size_t)4 * TILE_B + 4 * 256 * 4;  // stages + qaux0 + tau + qaux1 + prefilter

// DIRECT = the phase-0 instantiation (rows <= cap, no threshold yet): every (query, row) pair is a candidate and its
// slot is the row's offset -- no pre-filter, no counter.  A separate instantiation so the steady-state variants
// (at the 256-VGPR limit) carry none of it.
// XS (Float16 only) = the rows are the scaled-f16 SHADOW of a Float32 corpus: row r was multiplied by 2^s_r before
// rounding and p.xscale[r] = 2^-s_r undoes it in the epilogue; norms come from the f32 rows.
template <int DT, int METRIC, bool DIRECT, bool XS>
__global__ void __launch_bounds__(512, 2) scan_mfma16_kernel(Batch16Params p) {
    using Tr = T16<DT>;
    constexpr int PLANES = Tr::PLANES, IT = Tr::IT;
    constexpr int BMQ = AROWS / PLANES;  // queries per block
    constexpr bool U8 = DT == MVF_DTYPE_UINT8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* qa_s = reinterpret_cast<float*>(smem + 4 * TILE_B);    // [256] f16: 2^-e / i8: qq (as int)
    uint32_t* tau_s = reinterpret_cast<uint32_t*>(qa_s + 256);    // [256]
    float* qb_s = reinterpret_cast<float*>(tau_s + 256);          // [256] f16: |q|
    float* thr_s = qb_s + 256;                                    // [256] pre-filter threshold (float, or int bits)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    // PERSISTENT blocks, XCD-aware tile order, load pipeline running across tile boundaries — see
    // scan_mfma.hip.  It matters more here: the LDS image allows ONE block per CU, so with one tile per block
    // every prologue (two HBM round trips), epilogue and dispatch gap was fully exposed (~10 of 18 us per tile).
    const uint32_t xcd = blockIdx.x & 7u, ls = blockIdx.x >> 3, nls = gridDim.x >> 3;
    auto slot_tile = [&](uint32_t n, uint32_t& nt, uint32_t& mt) {
        const uint32_t slot = ls + n * nls;
        nt = (slot / p.mtiles) * 8u + xcd;
        mt = slot % p.mtiles;
        return nt < p.ntiles;
    };
    uint32_t my_tiles = 0;
    {
        const uint32_t max_slot_excl = ((p.ntiles + 7u - xcd) / 8u) * p.mtiles;
        if (ls < max_slot_excl) my_tiles = (max_slot_excl - ls + nls - 1) / nls;
    }
    if (my_tiles == 0) return;
    const uint32_t G = my_tiles * p.KT;

    auto load_query_consts = [&](uint32_t q0) { load_query_consts16<DT, METRIC>(p, q0, tid, qa_s, qb_s, tau_s, thr_s); };

    // ---- staging: thread -> 16-B chunk (row sr + 64*i, column sc) of each tile ----------
    // Branch-free loads: rows past row_end re-read a valid row (their output columns are discarded in the epilogue);
    // k beyond the row's pitch reads the row start and is zeroed with a select at LDS-store time.
    const int sr = tid >> 3, sc = tid & 7;
    uint32_t a_n = 0, a_kt = 0, b_n = 0, b_kt = 0;  // load cursors: (tile ordinal, k-tile) of the next A / B load
    // Addresses are kept as UNIFORM 64-bit bases (SGPR pairs) plus 32-bit per-lane offsets (global_load's
    // saddr + voffset form): one VGPR serves all four A loads and four serve B, instead of sixteen for eight 64-bit
    // pointers -- the int8 variants sit at the 256-VGPR limit.
    const unsigned char* abase[4];  // uniform: row (plane_i, tile query (64 i) % BMQ) of the tile's prepared queries
    const unsigned char* xbase[4];  // uniform: corpus row r0 + 64 i (r0 if that is past the end)
    const uint32_t a_loff = (uint32_t)sr * p.KPB + (uint32_t)sc * 16u;
    uint32_t x_loff[4];
    auto set_a_tile = [&](uint32_t n) {
        uint32_t nt, mt;
        slot_tile(n, nt, mt);
#pragma unroll
        for (int i = 0; i < 4; i++)  // LDS A row sr + 64 i: plane = (64 i) / BMQ, query = sr + (64 i) % BMQ (sr < 64 | BMQ)
            abase[i] = p.qprep + ((size_t)((64 * i) / BMQ) * p.nq_pad + mt * BMQ + (64 * i) % BMQ) * p.KPB;
    };
    auto set_b_tile = [&](uint32_t n) {
        uint32_t nt, mt;
        slot_tile(n, nt, mt);
        const uint32_t r0 = p.row_begin + nt * BROWS;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t rb_ = r0 + 64 * i;
            xbase[i] = p.rows + (size_t)(rb_ < p.row_end ? rb_ : r0) * p.pitch;
            x_loff[i] = rb_ + sr < p.row_end ? (uint32_t)sr * p.pitch : 0u;  // rows past the end re-read a valid row
        }
    };
    // ra: A k-tile one ahead of the LDS stage being computed (queries are L2-hot).  B k-tiles are loaded TWO
    // ahead into alternating sets rb0/rb1: these MFMAs retire a k-tile in ~1 us, less than an HBM round trip.
    u32x4 ra[4], rb0[4], rb1[4];
    // The loads of one k-tile are issued in PARTS between the k-steps rather than in one lump, so the CU's
    // vector-memory front end (~64 B/clk; 64 KB per k-tile) works underneath the MFMAs.
    auto load_a_part = [&](int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; i++) ra[i] = *reinterpret_cast<const u32x4*>(abase[i] + (size_t)a_kt * BKB + a_loff);
    };
    auto advance_a = [&]() {
        if (++a_kt == p.KT) {
            a_kt = 0;
            if (++a_n < my_tiles) set_a_tile(a_n);
        }
    };
    auto load_b_part = [&](u32x4 (&rb)[4], int i0, int i1) {
        const uint32_t v = b_kt * 8 + sc;
        const uint32_t xoff = v < p.V ? v * 16u : 0u;
#pragma unroll
        for (int i = i0; i < i1; i++)
            rb[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xbase[i] + (x_loff[i] + xoff)));
    };
    auto advance_b = [&]() {
        if (++b_kt == p.KT) {
            b_kt = 0;
            if (++b_n < my_tiles) set_b_tile(b_n);
        }
    };
    auto load_a = [&]() {
        load_a_part(0, 4);
        advance_a();
    };
    auto load_b = [&](u32x4 (&rb)[4]) {
        load_b_part(rb, 0, 4);
        advance_b();
    };
    auto store_a = [&](int stage) {
        unsigned char* a = smem + stage * 2 * TILE_B;
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<u32x4*>(a + (sr + 64 * i) * LDPB + sc * 16) = ra[i];
    };
    auto store_b = [&](int stage, uint32_t kt, const u32x4 (&rb)[4]) {  // kt = the k-tile (within its tile) held in rb
        unsigned char* bb = smem + stage * 2 * TILE_B + TILE_B;
        const bool vok = kt * 8 + sc < p.V;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            u32x4 x = vok ? rb[i] : u32x4{0, 0, 0, 0};
            if (U8) x ^= u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};  // x_u -> x_s (k padding: q_s is 0 there)
            *reinterpret_cast<u32x4*>(bb + (sr + 64 * i) * LDPB + sc * 16) = x;
        }
    };

    typename Tr::Acc acc[IT][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < IT; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0;
    };
    zero_acc();

    uint32_t c_n = 0, c_kt = 0, c_nt, c_mt;  // compute cursor
    slot_tile(0, c_nt, c_mt);
    load_query_consts(c_mt * BMQ);

    set_a_tile(0);
    set_b_tile(0);
    load_a();
    load_b(rb0);
    store_a(0);
    store_b(0, 0, rb0);
    // Loads past the last k-tile are NOT branched around: the cursors stay on the block's last tile (valid memory)
    // and the data is never stored to a stage that is read.  Uniform control flow is what lets the compiler count
    // outstanding loads exactly; with `if (more)` around them it fell back to s_waitcnt vmcnt(0) before every load
    // group, i.e. each group waited out the full latency of the previous one (in-kernel stamps: 25 % of a k-tile).
    load_a();
    load_b(rb1);
    load_b(rb0);
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    // one k-step (32 bytes of k): IT*2*PLANES MFMAs
    auto kstep = [&](const unsigned char* a, const unsigned char* bb, int ks) __attribute__((always_inline)) {
        u32x4 fb[2];
#pragma unroll
        for (int j = 0; j < 2; j++) fb[j] = *reinterpret_cast<const u32x4*>(bb + j * 32 * LDPB + ks * 32);
#pragma unroll
        for (int pl = 0; pl < PLANES; pl++) {
#pragma unroll
            for (int i = 0; i < IT; i++) {
                const u32x4 fa = *reinterpret_cast<const u32x4*>(a + (pl * BMQ + i * 32) * LDPB + ks * 32);
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if constexpr (DT == MVF_DTYPE_FLOAT16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa),
                                                                            __builtin_bit_cast(half8, fb[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, fa),
                                                                           __builtin_bit_cast(i32x4, fb[j]), acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    auto epilogue = [&](uint32_t nt, uint32_t mt) __attribute__((always_inline)) {
        epilogue16<DT, METRIC, DIRECT, XS, BMQ, 32, 128, 64>(p, acc, nt, mt, wm, wn, lane, qa_s, qb_s, tau_s, thr_s);
    };

    // one flat k-tile g; rb holds B k-tile g+1 on entry and receives B k-tile g+3.  The LDS stores of k-tile g+1 and
    // the global loads ride between the four k-steps so the matrix pipe only drains at the one barrier per k-tile.
    auto ktile = [&](uint32_t g, u32x4 (&rb)[4]) __attribute__((always_inline)) {
        const int cur = g & 1;
        const unsigned char* a = smem + cur * 2 * TILE_B + (wm * (BMQ / 2) + fr) * LDPB + fh * 16;
        const unsigned char* bb = smem + cur * 2 * TILE_B + TILE_B + (wn * 64 + fr) * LDPB + fh * 16;
        const uint32_t next_kt = c_kt + 1 == p.KT ? 0u : c_kt + 1;
        kstep(a, bb, 0);
        __builtin_amdgcn_sched_barrier(0);
        store_a(cur ^ 1);
        store_b(cur ^ 1, next_kt, rb);
        load_a_part(0, 3);
        kstep(a, bb, 1);
        __builtin_amdgcn_sched_barrier(0);
        load_a_part(3, 4);
        load_b_part(rb, 0, 2);
        kstep(a, bb, 2);
        __builtin_amdgcn_sched_barrier(0);
        load_b_part(rb, 2, 4);
        advance_a();
        advance_b();
        kstep(a, bb, 3);
        __syncthreads();
        if (++c_kt == p.KT) {  // tile finished: the next tile's first k-tile is already in LDS, its loads in flight
            epilogue(c_nt, c_mt);
            zero_acc();
            c_kt = 0;
            if (++c_n < my_tiles) {
                uint32_t nmt;
                slot_tile(c_n, c_nt, nmt);
                if (nmt != c_mt) {  // block-uniform; rare
                    __syncthreads();
                    load_query_consts(nmt * BMQ);
                    __syncthreads();
                    c_mt = nmt;
                }
            }
        }
    };
    uint32_t g = 0;
    for (; g + 1 < G; g += 2) {
        ktile(g, rb1);
        ktile(g + 1, rb0);
    }
    if (g < G) ktile(g, rb1);
}

// ---- query preparation ---------------------------------------------------------------
// f16: per query, scale = 2^e with max|q|*2^e in [2^14, 2^15); one plane q~ = f16(q*2^e) (round to nearest).
//      qaux0 = 2^-e, qaux1 = |q| (f32 norm of the ORIGINAL query).
__global__ void prep_queries_f16_kernel(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t KPB,
                                        unsigned char* qprep, float* qaux0, float* qaux1) {
    const uint32_t row = blockIdx.x;
    const uint32_t KP = KPB / 2;
    __shared__ float red[8];
    float mx = 0.f, ss = 0.f;
    if (row < nq)
        for (uint32_t c = threadIdx.x; c < dim; c += blockDim.x) {
            const float v = q[(size_t)row * dim + c];
            mx = fmaxf(mx, fabsf(v));
            ss = fmaf(v, v, ss);
        }
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        ss += __shfl_xor(ss, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = mx;
        red[4 + (threadIdx.x >> 6)] = ss;
    }
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    ss = red[4] + red[5] + red[6] + red[7];
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) {
        int ex;
        (void)frexpf(mx, &ex);  // mx = m * 2^ex, m in [0.5, 1)
        e = 15 - ex;            // mx * 2^e in [2^14, 2^15)
    }
    const float up = ldexpf(1.0f, e), down = ldexpf(1.0f, -e);
    __half* hi = reinterpret_cast<__half*>(qprep + (size_t)row * KPB);
    for (uint32_t c = threadIdx.x; c < KP; c += blockDim.x) {
        float v = (row < nq && c < dim) ? q[(size_t)row * dim + c] * up : 0.f;
        hi[c] = __float2half_rn(v);
    }
    if (threadIdx.x == 0) {
        qaux0[row] = down;
        qaux1[row] = sqrtf(ss);
    }
}

// i8: zero-padded copy; qaux0 = bit pattern of the i32 sum q^2.
__global__ void prep_queries_i8_kernel(const int8_t* q, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t KPB,
                                       unsigned char* qprep, float* qaux0, float* qaux1) {
    const uint32_t row = blockIdx.x;
    __shared__ int red[4];
    int ss = 0;
    for (uint32_t c = threadIdx.x; c < KPB; c += blockDim.x) {
        const int8_t v = (row < nq && c < dim) ? q[(size_t)row * dim + c] : (int8_t)0;
        reinterpret_cast<int8_t*>(qprep)[(size_t)row * KPB + c] = v;
        ss += (int)v * (int)v;
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        qaux0[row] = __int_as_float(red[0] + red[1] + red[2] + red[3]);
        qaux1[row] = 0.f;
    }
}

// u8: shifted int8 copy (q ^ 0x80), zero padded in the SIGNED domain; qaux0 = bits of sum q_s^2,
// qaux1 = bits of 128 * sum q_s + 16384 * dim.
__global__ void prep_queries_u8_kernel(const uint8_t* q, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t KPB,
                                       unsigned char* qprep, float* qaux0, float* qaux1) {
    const uint32_t row = blockIdx.x;
    __shared__ int red[8];
    int ss = 0, su = 0;
    for (uint32_t c = threadIdx.x; c < KPB; c += blockDim.x) {
        const int v = (row < nq && c < dim) ? (int)q[(size_t)row * dim + c] - 128 : 0;
        reinterpret_cast<int8_t*>(qprep)[(size_t)row * KPB + c] = (int8_t)v;
        ss += v * v;
        su += v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off, 64);
        su += __shfl_xor(su, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = ss;
        red[4 + (threadIdx.x >> 6)] = su;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        qaux0[row] = __int_as_float(red[0] + red[1] + red[2] + red[3]);
        qaux1[row] = __int_as_float(128 * (red[4] + red[5] + red[6] + red[7]) + 16384 * (int)dim);
    }
}

// ---- K4 for the narrow types: one wave per row ----------------------------------------------
__global__ void __launch_bounds__(256) row_norms_f16_kernel(const unsigned char* rows, uint32_t n, uint32_t pitch,
                                                             uint32_t V, float* xnorm, float* xx2, float* xxmax) {
    float mx = 0.f;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    for (uint32_t r = wave; r < n; r += nwaves) {
        const unsigned char* rp = rows + (size_t)r * pitch;
        float s = 0.f;
        for (uint32_t v = lane; v < V; v += 64) {
            const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rp + (size_t)v * 16));
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const float a = __half2float(__ushort_as_half((unsigned short)(x[w] & 0xFFFFu)));
                const float b2 = __half2float(__ushort_as_half((unsigned short)(x[w] >> 16)));
                s = fmaf(a, a, s);
                s = fmaf(b2, b2, s);
            }
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) {
            xnorm[r] = sqrtf(s);
            xx2[r] = s;
            if (s > mx) mx = s;
        }
    }
    if (lane == 0 && mx > 0.f) atomicMax(reinterpret_cast<unsigned int*>(xxmax), __float_as_uint(mx));
}

// Scaled-f16 SHADOW of a Float32 corpus, used for selection only (api.hip): row r is multiplied by 2^s_r with
// max|x| 2^s_r in [2^14, 2^15) -- nothing overflows f16 and every element keeps 11 significant bits relative to
// itself (elements more than 2^29 below the row's largest fall into the f16 subnormals: absolute error 2^-25, i.e.
// < 2^-39 of the largest) -- and rounded to nearest; xscale[r] = 2^-s_r.  Rows holding Inf keep s_r = 0.
__global__ void __launch_bounds__(256) shadow_f16_kernel(const unsigned char* rows32, uint32_t n, uint32_t pitch32,
                                                          uint32_t dim, unsigned char* rows16, uint32_t pitch16,
                                                          float* xscale) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    const uint32_t V32 = pitch32 / 16, V16 = pitch16 / 16;
    for (uint32_t r = wave; r < n; r += nwaves) {
        const unsigned char* rp = rows32 + (size_t)r * pitch32;
        float mx = 0.f;
        for (uint32_t v = lane; v < V32; v += 64) {
            const u32x4 x = *reinterpret_cast<const u32x4*>(rp + (size_t)v * 16);
#pragma unroll
            for (int w = 0; w < 4; w++) mx = fmaxf(mx, fabsf(__uint_as_float(x[w])));
        }
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        int sh = 0;
        if (mx > 0.f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);  // mx = m * 2^ex, m in [0.5, 1)
            sh = 15 - ex;
        }
        unsigned char* op = rows16 + (size_t)r * pitch16;
        for (uint32_t v = lane; v < V16; v += 64) {  // one 16-B f16 vector = two f32 vectors
            u32x4 o;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t v32 = 2 * v + h;
                u32x4 x = u32x4{0, 0, 0, 0};
                if (v32 < V32) x = *reinterpret_cast<const u32x4*>(rp + (size_t)v32 * 16);  // f32 padding is zero
#pragma unroll
                for (int w = 0; w < 2; w++) {
                    const unsigned short lo = __half_as_ushort(__float2half_rn(ldexpf(__uint_as_float(x[2 * w]), sh)));
                    const unsigned short hi = __half_as_ushort(__float2half_rn(ldexpf(__uint_as_float(x[2 * w + 1]), sh)));
                    o[2 * h + w] = (uint32_t)lo | ((uint32_t)hi << 16);
                }
            }
            *reinterpret_cast<u32x4*>(op + (size_t)v * 16) = o;
        }
        if (lane == 0) xscale[r] = ldexpf(1.0f, -sh);
    }
}

__global__ void __launch_bounds__(256) row_norms_i8_kernel(const unsigned char* rows, uint32_t n, uint32_t pitch,
                                                            uint32_t V, int32_t* xx) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    for (uint32_t r = wave; r < n; r += nwaves) {
        const unsigned char* rp = rows + (size_t)r * pitch;
        int s = 0;
        for (uint32_t v = lane; v < V; v += 64) {
            const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rp + (size_t)v * 16));
#pragma unroll
            for (int w = 0; w < 4; w++) s = __builtin_amdgcn_sdot4((int)x[w], (int)x[w], s, false);
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) xx[r] = s;
    }
}

// UInt8 rows: xx[r] = sum (x-128)^2, xbias[r] = 128 * sum (x-128), over the row's REAL elements
__global__ void __launch_bounds__(256) row_norms_u8_kernel(const unsigned char* rows, uint32_t n, uint32_t pitch,
                                                            uint32_t V, uint32_t dim, int32_t* xx, int32_t* xbias) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    for (uint32_t r = wave; r < n; r += nwaves) {
        const unsigned char* rp = rows + (size_t)r * pitch;
        int s2 = 0, s1 = 0;
        for (uint32_t v = lane; v < V; v += 64) {
            const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rp + (size_t)v * 16));
#pragma unroll
            for (int w = 0; w < 4; w++) {
                // bytes past `dim` inside the last vector are zero padding: keep them 0 in the signed domain too
                const uint32_t e0 = v * 16 + w * 4;
                uint32_t mask = e0 + 4 <= dim ? 0xFFFFFFFFu : e0 >= dim ? 0u : (0xFFFFFFFFu >> (8 * (4 - (dim - e0))));
                const uint32_t xs = (x[w] ^ 0x80808080u) & mask;
                s2 = __builtin_amdgcn_sdot4((int)xs, (int)xs, s2, false);
                s1 = __builtin_amdgcn_sdot4((int)xs, 0x01010101, s1, false);
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            s2 += __shfl_xor(s2, off, 64);
            s1 += __shfl_xor(s1, off, 64);
        }
        if (lane == 0) {
            xx[r] = s2;
            xbias[r] = 128 * s1;
        }
    }
}

template <int DT, int METRIC>
hipError_t launch_dtm(const Batch16Params& p, dim3 grid, hipStream_t s) {
    // > 64 KiB of dynamic LDS needs the attribute; it is per device, and one process may drive several devices
    void (*fn)(Batch16Params) = p.direct ? &scan_mfma16_kernel<DT, METRIC, true, false> : &scan_mfma16_kernel<DT, METRIC, false, false>;
    if constexpr (DT == MVF_DTYPE_FLOAT16)
        if (p.xscale) fn = p.direct ? &scan_mfma16_kernel<DT, METRIC, true, true> : &scan_mfma16_kernel<DT, METRIC, false, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds16);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, grid, dim3(512), kLds16, s, p);
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

uint32_t scan_mfma16_queries_per_block(int) { return 256u; }

hipError_t launch_scan_mfma16(const Batch16Params& p, int dtype, int metric, int num_cus, int force_persistent, hipStream_t s) {
    // Grid: a multiple of 8 (one lane set per XCD).  Persistent (one block per CU, LDS-limited) or one tile per
    // block — the same kernel: with the full grid every block owns exactly one tile.
    const uint32_t total = ((p.ntiles + 7) / 8) * p.mtiles * 8;
    uint32_t nls = std::max(1u, (uint32_t)num_cus / 8u);
    if (nls > p.mtiles) nls -= nls % p.mtiles;
    bool persistent = dtype != MVF_DTYPE_FLOAT16;  // measured: int8 equal either way, f16 13 % faster with one tile per block
    if (force_persistent >= 0) persistent = force_persistent != 0;  // MVF_K2_PERSISTENT16 (read once per handle)
    const dim3 grid(persistent ? std::min(total, nls * 8u) : total);
    if (dtype == MVF_DTYPE_FLOAT16) return launch_dt<MVF_DTYPE_FLOAT16>(p, metric, grid, s);
    if (dtype == MVF_DTYPE_UINT8) return launch_dt<MVF_DTYPE_UINT8>(p, metric, grid, s);
    return launch_dt<MVF_DTYPE_INT8>(p, metric, grid, s);
}

hipError_t launch_prep_queries16(const void* q, int dtype, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t KPB,
                                 unsigned char* qprep, float* qaux0, float* qaux1, hipStream_t s) {
    if (dtype == MVF_DTYPE_FLOAT16)
        hipLaunchKernelGGL(prep_queries_f16_kernel, dim3(nq_pad), dim3(256), 0, s, static_cast<const float*>(q), nq, nq_pad,
                           dim, KPB, qprep, qaux0, qaux1);
    else if (dtype == MVF_DTYPE_UINT8)
        hipLaunchKernelGGL(prep_queries_u8_kernel, dim3(nq_pad), dim3(256), 0, s, static_cast<const uint8_t*>(q), nq, nq_pad,
                           dim, KPB, qprep, qaux0, qaux1);
    else
        hipLaunchKernelGGL(prep_queries_i8_kernel, dim3(nq_pad), dim3(256), 0, s, static_cast<const int8_t*>(q), nq, nq_pad,
                           dim, KPB, qprep, qaux0, qaux1);
    return hipGetLastError();
}

hipError_t launch_shadow_f16(const unsigned char* rows32, uint32_t n, uint32_t pitch32, uint32_t dim, unsigned char* rows16,
                             uint32_t pitch16, float* xscale, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 3) / 4, 256u * 16u);
    hipLaunchKernelGGL(shadow_f16_kernel, dim3(blocks), dim3(256), 0, s, rows32, n, pitch32, dim, rows16, pitch16, xscale);
    return hipGetLastError();
}

hipError_t launch_row_norms16(const unsigned char* rows, int dtype, uint32_t n, uint32_t pitch, uint32_t dim, void* out,
                              float* xx2, float* xxmax, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 3) / 4, 256u * 8u);
    if (dtype == MVF_DTYPE_FLOAT16)
        hipLaunchKernelGGL(row_norms_f16_kernel, dim3(blocks), dim3(256), 0, s, rows, n, pitch, pitch / 16, static_cast<float*>(out),
                           xx2, xxmax);
    else if (dtype == MVF_DTYPE_UINT8)
        hipLaunchKernelGGL(row_norms_u8_kernel, dim3(blocks), dim3(256), 0, s, rows, n, pitch, pitch / 16, dim,
                           static_cast<int32_t*>(out), reinterpret_cast<int32_t*>(xx2));
    else
        hipLaunchKernelGGL(row_norms_i8_kernel, dim3(blocks), dim3(256), 0, s, rows, n, pitch, pitch / 16, static_cast<int32_t*>(out));
    return hipGetLastError();
}

}  // namespace mvf
