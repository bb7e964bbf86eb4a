/*
 * mvf_oracle.c — CPU ORACLE (test infrastructure; see mvf_oracle.h header).
 * PARITY STATUS: "parity unpinned" — the Rust reference cannot be built here.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile).
 * -ffp-contract=off matters: rustc never fuses a*b+c, gcc would on FMA targets.
 */
#include "mvf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- half ---- */

/* crate half 2.6.0 f16::to_f32: exact widening incl. subnormals/inf/NaN
 * (call sites: src/vectors/vector.rs:85-86). */
float mvfo_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                man <<= 1;
                e++;
            } while ((man & 0x400u) == 0);
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* crate half 2.6.0 f16::from_f32: IEEE RNE, overflow -> inf, NaN stays NaN
 * (call site: src/builder.rs:187). */
uint16_t mvfo_f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t exp = (x >> 23) & 0xFFu;
    uint32_t man = x & 0x7FFFFFu;
    if (exp == 255) { /* inf / NaN */
        if (man == 0) return (uint16_t)(sign | 0x7C00u);
        uint32_t m = man >> 13;
        return (uint16_t)(sign | 0x7C00u | 0x0200u | m); /* quiet */
    }
    int32_t e = (int32_t)exp - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u); /* overflow -> inf */
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign; /* underflow -> signed zero */
        man |= 0x800000u;                   /* implicit 1 */
        uint32_t shift = (uint32_t)(14 - e); /* 14..24 */
        uint32_t half_m = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half_m & 1u))) half_m++;
        return (uint16_t)(sign | half_m);
    }
    uint32_t half_m = man >> 13;
    uint32_t rem = man & 0x1FFFu;
    uint16_t h = (uint16_t)(sign | ((uint32_t)e << 10) | half_m);
    if (rem > 0x1000u || (rem == 0x1000u && (half_m & 1u))) h++; /* may carry to inf: correct */
    return h;
}

uint32_t mvfo_elem_size(uint8_t dtype) {
    /* src/vectors/vector_space.rs:122-127 */
    switch (dtype) {
    case MVFO_F32: return 4;
    case MVFO_F16: return 2;
    case MVFO_I8:
    case MVFO_U8: return 1;
    default: return 0;
    }
}

/* ---------------------------------------------------------------- keys ---- */

static inline uint32_t ord_f32(float x) {
    if (x != x) return 0xFFFFFFFFu; /* NaN last */
    x = x + 0.0f;                   /* -0.0 -> +0.0 */
    uint32_t b;
    memcpy(&b, &x, 4);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

uint32_t mvfo_key_from_score(float score, uint8_t metric) {
    return metric == MVFO_L2 ? ord_f32(score) : ord_f32(-score);
}

uint32_t mvfo_key_from_raw(int32_t raw, uint8_t metric) {
    uint32_t u = (uint32_t)raw ^ 0x80000000u; /* ascending i32 */
    return metric == MVFO_L2 ? u : ~u;        /* IP: descending */
}

static inline float canon_score(float s) {
    if (s != s) {
        uint32_t q = 0x7FC00000u;
        memcpy(&s, &q, 4);
        return s;
    }
    return s + 0.0f;
}

/* -------------------------------------------------------------- scoring ---- */

/* examples/similarity_search.rs:152-157: zip/map/sum/sqrt; sum::<f32>() is a
 * strict left fold from 0.0, every op rounded to f32, no FMA. */
static float l2_f32(const float* q, const float* x, uint32_t d) {
    float s = 0.0f;
    for (uint32_t j = 0; j < d; j++) {
        float t = q[j] - x[j];
        float sq = t * t; /* powi(2) == t*t */
        s = s + sq;
    }
    return sqrtf(s);
}

static float dot_f32(const float* q, const float* x, uint32_t d) {
    float s = 0.0f;
    for (uint32_t j = 0; j < d; j++) {
        float p = q[j] * x[j];
        s = s + p;
    }
    return s;
}

static float cos_from_parts(float dot, float qq, float xx) {
    float den = sqrtf(qq) * sqrtf(xx);
    if (!(den > 0.0f)) return 0.0f; /* zero-norm (or NaN norm) => 0 */
    return dot / den;
}

/* src/vectors/vector.rs:71-92 (as_f32) without the allocation */
static void decode_row_f32(const uint8_t* row, uint32_t d, uint8_t dtype, float* out) {
    if (dtype == MVFO_F32) {
        memcpy(out, row, (size_t)d * 4); /* from_le_bytes on an LE host */
    } else {
        for (uint32_t j = 0; j < d; j++) {
            uint16_t h = (uint16_t)(row[2 * j] | (row[2 * j + 1] << 8));
            out[j] = mvfo_f16_to_f32(h);
        }
    }
}

int mvfo_scores(const void* rows, uint64_t n, uint32_t dim, uint8_t dtype,
                uint64_t stride, uint8_t metric, const void* query,
                float* out_scores, uint32_t* out_keys, int32_t* out_raw) {
    uint32_t es = mvfo_elem_size(dtype);
    if (es == 0) return MVFO_ERR_BUILD;
    if (metric > MVFO_COS) return MVFO_ERR_ARG;
    if (dim == 0 || stride < (uint64_t)dim * es) return MVFO_ERR_ARG;
    const uint8_t* base = (const uint8_t*)rows;

    if (dtype == MVFO_F32 || dtype == MVFO_F16) {
        const float* q = (const float*)query;
        float qq = (metric == MVFO_COS) ? dot_f32(q, q, dim) : 0.0f;
#pragma omp parallel
        {
            float* x = (float*)malloc((size_t)dim * 4);
#pragma omp for schedule(static)
            for (int64_t i = 0; i < (int64_t)n; i++) {
                decode_row_f32(base + (uint64_t)i * stride, dim, dtype, x);
                float s;
                if (metric == MVFO_L2) s = l2_f32(q, x, dim);
                else if (metric == MVFO_IP) s = dot_f32(q, x, dim);
                else s = cos_from_parts(dot_f32(q, x, dim), qq, dot_f32(x, x, dim));
                out_scores[i] = canon_score(s);
                out_keys[i] = mvfo_key_from_score(s, metric);
                if (out_raw) out_raw[i] = 0;
            }
            free(x);
        }
        return MVFO_OK;
    }

    /* Int8 / UInt8: exact integer accumulation (defined here; reference has
     * no integer path — SURVEY.md F2/F3).  i64 internally, must fit i32. */
    int64_t qq = 0;
    for (uint32_t j = 0; j < dim; j++) {
        int32_t a = dtype == MVFO_I8 ? ((const int8_t*)query)[j] : ((const uint8_t*)query)[j];
        qq += (int64_t)a * a;
    }
    int overflow = 0;
#pragma omp parallel for schedule(static) reduction(| : overflow)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const uint8_t* row = base + (uint64_t)i * stride;
        int64_t dot = 0, xx = 0, l2 = 0;
        for (uint32_t j = 0; j < dim; j++) {
            int32_t a, b;
            if (dtype == MVFO_I8) {
                a = ((const int8_t*)query)[j];
                b = ((const int8_t*)row)[j];
            } else {
                a = ((const uint8_t*)query)[j];
                b = row[j];
            }
            dot += (int64_t)a * b;
            xx += (int64_t)b * b;
            l2 += (int64_t)(a - b) * (a - b);
        }
        if (l2 > INT32_MAX || xx > INT32_MAX || qq > INT32_MAX) overflow |= 1;
        float s;
        int32_t raw = 0;
        uint32_t key;
        if (metric == MVFO_L2) {
            raw = (int32_t)l2;
            s = sqrtf((float)raw);
            key = mvfo_key_from_raw(raw, metric);
        } else if (metric == MVFO_IP) {
            raw = (int32_t)dot;
            s = (float)raw;
            key = mvfo_key_from_raw(raw, metric);
        } else {
            float den = sqrtf((float)(int32_t)qq) * sqrtf((float)(int32_t)xx);
            s = (den > 0.0f) ? (float)(int32_t)dot / den : 0.0f;
            key = mvfo_key_from_score(s, metric);
        }
        out_scores[i] = canon_score(s);
        out_keys[i] = key;
        if (out_raw) out_raw[i] = raw;
    }
    return overflow ? MVFO_ERR_ARG : MVFO_OK;
}

/* ---------------------------------------------------------------- top-k ---- */

typedef struct {
    uint32_t key;
    uint64_t idx;
} cand_t;

static inline int cand_less(const cand_t* a, const cand_t* b) {
    return a->key < b->key || (a->key == b->key && a->idx < b->idx);
}

static int cand_cmp(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a;
    const cand_t* y = (const cand_t*)b;
    if (cand_less(x, y)) return -1;
    if (cand_less(y, x)) return 1;
    return 0;
}

/* bounded max-heap (worst candidate at the root) */
static void heap_sift_down(cand_t* h, uint32_t n, uint32_t i) {
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && cand_less(&h[m], &h[l])) m = l;
        if (r < n && cand_less(&h[m], &h[r])) m = r;
        if (m == i) return;
        cand_t t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

static void heap_sift_up(cand_t* h, uint32_t i) {
    while (i > 0) {
        uint32_t p = (i - 1) / 2;
        if (!cand_less(&h[p], &h[i])) return;
        cand_t t = h[i];
        h[i] = h[p];
        h[p] = t;
        i = p;
    }
}

static uint32_t select_k(const uint32_t* keys, const uint64_t* idx_or_null, uint64_t n,
                         uint32_t k, cand_t* heap /* k */) {
    uint32_t cnt = 0;
    for (uint64_t i = 0; i < n; i++) {
        cand_t c = {keys[i], idx_or_null ? idx_or_null[i] : i};
        if (idx_or_null && c.idx == UINT64_MAX) continue; /* padding */
        if (cnt < k) {
            heap[cnt] = c;
            heap_sift_up(heap, cnt);
            cnt++;
        } else if (k > 0 && cand_less(&c, &heap[0])) {
            heap[0] = c;
            heap_sift_down(heap, cnt, 0);
        }
    }
    qsort(heap, cnt, sizeof(cand_t), cand_cmp);
    return cnt;
}

int mvfo_topk_from_keys(const uint32_t* keys, uint64_t n, uint32_t k, uint64_t* out_idx) {
    cand_t* heap = (cand_t*)malloc(sizeof(cand_t) * (k ? k : 1));
    if (!heap) return MVFO_ERR_ARG;
    uint32_t cnt = select_k(keys, NULL, n, k, heap);
    for (uint32_t i = 0; i < k; i++) out_idx[i] = i < cnt ? heap[i].idx : UINT64_MAX;
    free(heap);
    return MVFO_OK;
}

static float pad_score(uint8_t metric) { return metric == MVFO_L2 ? INFINITY : -INFINITY; }

int mvfo_search(const void* rows, uint64_t n, uint32_t dim, uint8_t dtype,
                uint64_t stride, uint8_t metric, const void* queries,
                uint32_t nq, uint32_t k, uint64_t index_base,
                float* out_scores, uint64_t* out_idx, int32_t* out_raw) {
    uint32_t es = mvfo_elem_size(dtype);
    if (es == 0) return MVFO_ERR_BUILD;
    uint32_t qes = (dtype == MVFO_F32 || dtype == MVFO_F16) ? 4 : 1;
    float* sc = (float*)malloc(sizeof(float) * (n ? n : 1));
    uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    int32_t* raw = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    uint64_t* sel = (uint64_t*)malloc(sizeof(uint64_t) * (k ? k : 1));
    int rc = MVFO_OK;
    for (uint32_t qi = 0; qi < nq && rc == MVFO_OK; qi++) {
        const uint8_t* q = (const uint8_t*)queries + (size_t)qi * dim * qes;
        rc = mvfo_scores(rows, n, dim, dtype, stride, metric, q, sc, keys, raw);
        if (rc != MVFO_OK) break;
        rc = mvfo_topk_from_keys(keys, n, k, sel);
        for (uint32_t j = 0; j < k; j++) {
            size_t o = (size_t)qi * k + j;
            if (sel[j] == UINT64_MAX) {
                out_idx[o] = UINT64_MAX;
                out_scores[o] = pad_score(metric);
                if (out_raw) out_raw[o] = 0;
            } else {
                out_idx[o] = sel[j] + index_base;
                out_scores[o] = sc[sel[j]];
                if (out_raw) out_raw[o] = raw[sel[j]];
            }
        }
    }
    free(sc);
    free(keys);
    free(raw);
    free(sel);
    return rc;
}

int mvfo_merge_topk(const float* scores, const uint64_t* idx, const int32_t* raw,
                    uint32_t nlists, uint32_t nq, uint32_t k, uint8_t metric,
                    uint8_t dtype, float* out_scores, uint64_t* out_idx,
                    int32_t* out_raw) {
    int use_raw = (dtype == MVFO_I8 || dtype == MVFO_U8) && metric != MVFO_COS && raw;
    size_t m = (size_t)nlists * k;
    uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * (m ? m : 1));
    uint64_t* ids = (uint64_t*)malloc(sizeof(uint64_t) * (m ? m : 1));
    size_t* src = (size_t*)malloc(sizeof(size_t) * (m ? m : 1));
    cand_t* heap = (cand_t*)malloc(sizeof(cand_t) * (k ? k : 1));
    for (uint32_t qi = 0; qi < nq; qi++) {
        for (uint32_t l = 0; l < nlists; l++)
            for (uint32_t j = 0; j < k; j++) {
                size_t s = ((size_t)l * nq + qi) * k + j;
                size_t t = (size_t)l * k + j;
                /* ties: list order, then rank in the list (the lists come in ascending row-range order and each is
                 * sorted by (key, position), so this is ascending global position -- also when a shard reports vector
                 * ids instead of positions); padding sorts behind everything */
                int pad = idx[s] == UINT64_MAX;
                ids[t] = pad ? UINT64_MAX : (uint64_t)t;
                keys[t] = pad ? 0xFFFFFFFFu
                        : use_raw ? mvfo_key_from_raw(raw[s], metric)
                                  : mvfo_key_from_score(scores[s], metric);
                src[t] = s;
            }
        uint32_t cnt = select_k(keys, ids, m, k, heap);
        for (uint32_t j = 0; j < k; j++) {
            size_t o = (size_t)qi * k + j;
            if (j >= cnt) {
                out_idx[o] = UINT64_MAX;
                out_scores[o] = pad_score(metric);
                if (out_raw) out_raw[o] = 0;
                continue;
            }
            if (heap[j].idx == UINT64_MAX) { /* fewer than k real entries */
                out_idx[o] = UINT64_MAX;
                out_scores[o] = pad_score(metric);
                if (out_raw) out_raw[o] = 0;
                continue;
            }
            size_t s = src[heap[j].idx];
            out_idx[o] = idx[s];
            out_scores[o] = scores[s];
            if (out_raw) out_raw[o] = raw ? raw[s] : 0;
        }
    }
    free(keys);
    free(ids);
    free(src);
    free(heap);
    return MVFO_OK;
}

/* ------------------------------------------- faithful find_top_k_similar ---- */

/* ScoredVector, examples/similarity_search.rs:14-19 */
typedef struct {
    uint64_t index;
    float score;
    float* vector; /* Vec<f32> payload (owned) */
} scored_t;

/* Ord::cmp as the heap sees it.  farthest=1: similarity_search.rs:23-31
 * (other.score.partial_cmp(&self.score).unwrap_or(Equal)); farthest=0: the
 * natural order (intended semantics). Returns -1/0/1 for a<b, a==b, a>b. */
static int scored_cmp(const scored_t* a, const scored_t* b, int farthest) {
    float x = farthest ? b->score : a->score;
    float y = farthest ? a->score : b->score;
    if (x < y) return -1;
    if (x > y) return 1;
    return 0; /* equal or unordered (NaN) -> Equal */
}

/* std::collections::BinaryHeap (max-heap) push = sift_up */
static void bh_push(scored_t* h, uint32_t* len, scored_t item, int farthest) {
    uint32_t pos = (*len)++;
    h[pos] = item;
    while (pos > 0) {
        uint32_t parent = (pos - 1) / 2;
        if (scored_cmp(&h[pos], &h[parent], farthest) <= 0) break;
        scored_t t = h[pos];
        h[pos] = h[parent];
        h[parent] = t;
        pos = parent;
    }
}

/* BinaryHeap::pop = take last, swap into root, sift_down_to_bottom, sift_up */
static scored_t bh_pop(scored_t* h, uint32_t* len, int farthest) {
    scored_t item = h[--(*len)];
    uint32_t n = *len;
    if (n > 0) {
        scored_t top = h[0];
        h[0] = item;
        item = top;
        uint32_t pos = 0, child = 1;
        scored_t hole = h[0];
        while (child + 1 < n) { /* child <= end.saturating_sub(2) */
            if (scored_cmp(&h[child], &h[child + 1], farthest) <= 0) child++;
            h[pos] = h[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == n - 1) {
            h[pos] = h[child];
            pos = child;
        }
        h[pos] = hole;
        while (pos > 0) {
            uint32_t parent = (pos - 1) / 2;
            if (scored_cmp(&h[pos], &h[parent], farthest) <= 0) break;
            scored_t t = h[pos];
            h[pos] = h[parent];
            h[parent] = t;
            pos = parent;
        }
    }
    return item;
}

int mvfo_find_top_k_similar_faithful(const uint8_t* block, uint64_t block_len,
                                     uint64_t total_vectors, uint32_t dim,
                                     uint8_t dtype, const float* query,
                                     uint32_t query_len, uint32_t k,
                                     int farthest, uint64_t* out_idx,
                                     float* out_scores, uint32_t* out_count) {
    scored_t* heap = (scored_t*)malloc(sizeof(scored_t) * ((size_t)k + 2));
    uint32_t len = 0;
    int rc = MVFO_OK;
    for (uint64_t i = 0; i < total_vectors; i++) { /* similarity_search.rs:147 */
        /* get_vector: vector_space.rs:101-142 */
        uint32_t element_size = mvfo_elem_size(dtype);
        if (element_size == 0) { rc = MVFO_ERR_BUILD; break; } /* :126 */
        uint64_t vector_size = (uint64_t)dim * element_size;
        uint64_t vector_offset = i * vector_size;
        if (vector_offset + vector_size > block_len) { rc = MVFO_ERR_INDEX; break; } /* :132 */
        const uint8_t* vector_data = block + vector_offset;

        /* as_f32: vector.rs:71-92 — fresh Vec<f32>, per-element decode */
        if (dtype != MVFO_F32 && dtype != MVFO_F16) { rc = MVFO_ERR_BUILD; break; } /* :90 */
        uint32_t n_elems = (uint32_t)(vector_size / element_size); /* chunks_exact */
        float* data = (float*)malloc(sizeof(float) * (n_elems ? n_elems : 1));
        if (dtype == MVFO_F32) {
            for (uint32_t j = 0; j < n_elems; j++) {
                uint32_t b = (uint32_t)vector_data[4 * j] | ((uint32_t)vector_data[4 * j + 1] << 8) |
                             ((uint32_t)vector_data[4 * j + 2] << 16) | ((uint32_t)vector_data[4 * j + 3] << 24);
                memcpy(&data[j], &b, 4);
            }
        } else {
            for (uint32_t j = 0; j < n_elems; j++)
                data[j] = mvfo_f16_to_f32((uint16_t)(vector_data[2 * j] | (vector_data[2 * j + 1] << 8)));
        }

        /* similarity_search.rs:152-157 — zip truncates to the shorter side */
        uint32_t m = query_len < n_elems ? query_len : n_elems;
        float distance = l2_f32(query, data, m);

        scored_t item = {i, distance, data};
        bh_push(heap, &len, item, farthest); /* :159-163 */
        if (len > k) {                       /* :166-168 */
            scored_t ev = bh_pop(heap, &len, farthest);
            free(ev.vector);
        }
    }
    if (rc == MVFO_OK) {
        /* :172-173 — heap.into_iter() is the backing-vec order; sort_by is a
         * stable sort ascending by score; partial_cmp().unwrap() panics on NaN */
        for (uint32_t a = 0; a < len && rc == MVFO_OK; a++)
            if (heap[a].score != heap[a].score) rc = MVFO_ERR_ARG; /* would panic */
        if (rc == MVFO_OK) {
            for (uint32_t a = 1; a < len; a++) { /* stable insertion sort */
                scored_t t = heap[a];
                uint32_t b = a;
                while (b > 0 && heap[b - 1].score > t.score) {
                    heap[b] = heap[b - 1];
                    b--;
                }
                heap[b] = t;
            }
            for (uint32_t a = 0; a < len; a++) {
                out_idx[a] = heap[a].index;
                out_scores[a] = heap[a].score;
            }
            *out_count = len;
        }
    }
    for (uint32_t a = 0; a < len; a++) free(heap[a].vector);
    free(heap);
    return rc;
}

/* ------------------------------------------------------------ generator ---- */

static inline uint64_t mix64(uint64_t z) { /* splitmix64 output function */
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void mvfo_synth_rows(uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim,
                     uint8_t dtype, void* out) {
    const uint64_t base = mix64(seed);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)nrows; r++) {
        for (uint32_t c = 0; c < dim; c++) {
            uint64_t e = ((uint64_t)r + row0) * dim + c;
            uint64_t u = mix64(base + e);
            size_t o = (size_t)r * dim + c;
            if (dtype == MVFO_F32 || dtype == MVFO_F16) {
                float f = (float)(u >> 40) * 0x1p-23f - 1.0f;
                if (dtype == MVFO_F32) ((float*)out)[o] = f;
                else ((uint16_t*)out)[o] = mvfo_f32_to_f16(f);
            } else {
                ((uint8_t*)out)[o] = (uint8_t)(u >> 56);
            }
        }
    }
}
