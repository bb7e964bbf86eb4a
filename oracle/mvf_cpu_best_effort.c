/*
 * mvf_cpu_best_effort.c — what the HOST can do on this path when it tries.
 *
 * TEST / BENCH INFRASTRUCTURE, and NOT the checker: bench.py's
 * `cpu_baseline_best_effort` leg times it beside the GPU so that the GPU / CPU
 * ratio is not inflated by the reference's single thread, per-row allocation
 * and strict serial f32 fold (examples/similarity_search.rs:140-176).  Nothing
 * compares results against it; parity is judged against mvf_oracle.c only.
 *
 *   - every CPU the cgroup grants (the caller passes the thread count);
 *   - no per-row allocation, no decode copy: rows are read where they lie;
 *   - 16 partial sums per row so the compiler vectorises the fold (AVX-512 /
 *     AVX2 clones picked at load time) — therefore NOT bit-exact with the
 *     reference's left-to-right sum ("SIMD-friendly order");
 *   - one bounded heap per thread, merged at the end.
 *
 * Float32 rows, one query, L2 / InnerProduct / Cosine.
 */
#include "mvf_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float key; /* ascending = best first (L2: distance^2; others: -score) */
    uint64_t idx;
} be_cand;

static inline int be_less(const be_cand* a, const be_cand* b) { return a->key < b->key || (a->key == b->key && a->idx < b->idx); }

static void be_sift_down(be_cand* h, uint32_t n, uint32_t i) {
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && be_less(&h[m], &h[l])) m = l;
        if (r < n && be_less(&h[m], &h[r])) m = r;
        if (m == i) return;
        be_cand t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

static void be_offer(be_cand* h, uint32_t* cnt, uint32_t k, be_cand c) {
    if (*cnt < k) {
        uint32_t i = (*cnt)++;
        h[i] = c;
        while (i > 0) {
            uint32_t p = (i - 1) / 2;
            if (!be_less(&h[p], &h[i])) break;
            be_cand t = h[i];
            h[i] = h[p];
            h[p] = t;
            i = p;
        }
    } else if (be_less(&c, &h[0])) {
        h[0] = c;
        be_sift_down(h, *cnt, 0);
    }
}

static int be_cmp(const void* a, const void* b) {
    const be_cand* x = (const be_cand*)a;
    const be_cand* y = (const be_cand*)b;
    return be_less(x, y) ? -1 : be_less(y, x) ? 1 : 0;
}

/* sum (q-x)^2, sum q*x and sum x*x of one row in 16 independent partial sums (gcc vector extension: one zmm, two ymm
 * or four xmm registers per sum depending on the clone) */
typedef float be_v16 __attribute__((vector_size(64), aligned(4)));

__attribute__((target_clones("avx512f", "avx2", "default"), optimize("O3")))
static void be_row_sums(const float* q, const float* x, uint32_t d, int want_l2, int want_xx, float* l2, float* dot, float* xx) {
    be_v16 a = {0}, b = {0}, c = {0};
    uint32_t j = 0;
    if (want_l2) {
        for (; j + 16 <= d; j += 16) {
            const be_v16 t = *(const be_v16*)(q + j) - *(const be_v16*)(x + j);
            a += t * t;
        }
    } else if (want_xx) {
        for (; j + 16 <= d; j += 16) {
            const be_v16 xv = *(const be_v16*)(x + j);
            b += *(const be_v16*)(q + j) * xv;
            c += xv * xv;
        }
    } else {
        for (; j + 16 <= d; j += 16) b += *(const be_v16*)(q + j) * *(const be_v16*)(x + j);
    }
    float sa = 0, sb = 0, sc = 0;
    for (int u = 0; u < 16; u++) {
        sa += a[u];
        sb += b[u];
        sc += c[u];
    }
    for (; j < d; j++) {
        const float t = q[j] - x[j];
        sa += t * t;
        sb += q[j] * x[j];
        sc += x[j] * x[j];
    }
    *l2 = sa;
    *dot = sb;
    *xx = sc;
}

int mvfo_search_best_effort_f32(const float* rows, uint64_t n, uint32_t dim, uint8_t metric, const float* query,
                                uint32_t k, uint64_t index_base, int threads, float* out_scores, uint64_t* out_idx) {
    if (!rows || !query || !out_scores || !out_idx || dim == 0 || k == 0 || metric > MVFO_COS) return MVFO_ERR_ARG;
    if (threads < 1) threads = 1;
    float qq = 0;
    for (uint32_t j = 0; j < dim; j++) qq += query[j] * query[j];
    const float qn = sqrtf(qq);
    be_cand* heaps = (be_cand*)malloc(sizeof(be_cand) * (size_t)threads * k);
    uint32_t* cnts = (uint32_t*)calloc((size_t)threads, sizeof(uint32_t));
    if (!heaps || !cnts) {
        free(heaps);
        free(cnts);
        return MVFO_ERR_ARG;
    }
#pragma omp parallel num_threads(threads)
    {
        const int t = omp_get_thread_num();
        be_cand* h = heaps + (size_t)t * k;
        uint32_t cnt = 0;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            float l2, dot, xx;
            be_row_sums(query, rows + (size_t)i * dim, dim, metric == MVFO_L2, metric == MVFO_COS, &l2, &dot, &xx);
            float key;
            if (metric == MVFO_L2) key = l2;
            else if (metric == MVFO_IP) key = -dot;
            else {
                float den = qn * sqrtf(xx);
                key = den > 0.0f ? -(dot / den) : 0.0f;
            }
            be_cand c = {key, (uint64_t)i};
            be_offer(h, &cnt, k, c);
        }
        cnts[t] = cnt;
    }
    /* merge the per-thread heaps */
    size_t total = 0;
    for (int t = 0; t < threads; t++) {
        memmove(heaps + total, heaps + (size_t)t * k, sizeof(be_cand) * cnts[t]);
        total += cnts[t];
    }
    qsort(heaps, total, sizeof(be_cand), be_cmp);
    for (uint32_t j = 0; j < k; j++) {
        if (j < total) {
            out_idx[j] = heaps[j].idx + index_base;
            out_scores[j] = metric == MVFO_L2 ? sqrtf(heaps[j].key) : -heaps[j].key;
        } else {
            out_idx[j] = UINT64_MAX;
            out_scores[j] = metric == MVFO_L2 ? INFINITY : -INFINITY;
        }
    }
    free(heaps);
    free(cnts);
    return MVFO_OK;
}
