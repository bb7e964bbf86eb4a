/*
 * similarity_search.c — a plain-C consumer of the two C ABIs, following the flow of the reference's
 * examples/similarity_search.rs (build a small .mvf, reopen it, scan a vector space for the nearest rows)
 * with the scan done by libmvf_gpu.so instead of the inline Rust loop (similarity_search.rs:140-176).
 *
 *   cc -std=c99 -I include examples/c/similarity_search.c -L metrovector_amd -lmvf_gpu -lmvf_host -o ss
 *   ./ss /tmp/example.mvf
 *
 * It is what a maintainer's FFI would do, minus the language: map_vector_range() hands over (ptr, stride,
 * count, dtype) of the mmap'd block, mvfgpu_corpus_create() uploads once, mvfgpu_search() replaces
 * find_top_k_similar().  tests/test_c_consumer.py compiles it (CPU tier) and checks its output against the
 * known answers of the reference example (GPU tier).
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mvf_file.h"
#include "mvf_gpu.h"
#include "mvf_status.h"

#define ROWS 60
#define DIM 4
#define TOP_K 5

static void die_host(const char* what, int rc) {
    fprintf(stderr, "%s: %s (%s)\n", what, mvf_strerror(rc), mvf_last_error_message());
    exit(1);
}
static void die_gpu(const char* what, int rc) {
    fprintf(stderr, "%s: %s (%s)\n", what, mvfgpu_strerror(rc), mvfgpu_last_error_message());
    exit(1);
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "similarity_search_example.mvf";
    /* three clusters of 20 rows drifting away from their centres (the example's dataset) */
    static const float centre[3][DIM] = {{1, 1, 1, 1}, {5, 5, 5, 5}, {-2, 3, 0, 4}};
    static const float drift[3][DIM] = {{1, -1, 0.5f, -0.5f}, {1, -1, 0.5f, -0.5f}, {1, -1, 1, -0.5f}};
    float rows[ROWS][DIM];
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 20; i++) {
            const float noise = (float)i * 0.1f;
            for (int j = 0; j < DIM; j++) rows[c * 20 + i][j] = centre[c][j] + noise * drift[c][j];
        }

    /* ---- write the file (MvfBuilder) ---- */
    mvf_builder* b = NULL;
    int rc = mvf_builder_new(&b);
    if (rc) die_host("mvf_builder_new", rc);
    rc = mvf_builder_add_vector_space(b, "embeddings", DIM, MVF_VECTOR_DENSE, MVF_METRIC_L2, MVF_DTYPE_FLOAT32, NULL);
    if (rc) die_host("add_vector_space", rc);
    rc = mvf_builder_add_vectors_f32(b, "embeddings", &rows[0][0], ROWS, DIM);
    if (rc) die_host("add_vectors", rc);
    rc = mvf_builder_save(b, path, 0);
    if (rc) die_host("save", rc);
    mvf_builder_free(b);

    /* ---- reopen it (MvfReader) and take the zero-copy view of the rows ---- */
    mvf_reader* r = NULL;
    rc = mvf_reader_open(path, &r);
    if (rc) die_host("mvf_reader_open", rc);
    rc = mvf_reader_validate_with_checksum(r);
    if (rc) die_host("validate_with_checksum", rc);
    mvf_vector_space space;
    rc = mvf_reader_vector_space(r, "embeddings", &space);
    if (rc) die_host("vector_space", rc);
    mvf_vector_slice slice;
    rc = mvf_space_map_vector_range(&space, 0, space.total_vectors, &slice);
    if (rc) die_host("map_vector_range", rc);
    printf("space %.*s: %" PRIu64 " vectors x %u, dtype %u, metric %u\n", (int)space.name_len, space.name,
           space.total_vectors, space.dimension, space.data_type, space.distance_metric);

    /* ---- upload once, search many ---- */
    mvfgpu_corpus* corpus = NULL;
    rc = mvfgpu_corpus_create(slice.data, slice.count, space.dimension, slice.data_type, slice.stride, 0, 0, &corpus);
    if (rc) die_gpu("mvfgpu_corpus_create", rc);
    mvf_reader_close(r); /* the rows were only borrowed for the duration of the call */

    for (int c = 0; c < 3; c++) {
        float scores[TOP_K];
        uint64_t indices[TOP_K];
        rc = mvfgpu_search(corpus, space.distance_metric, centre[c], MVF_DTYPE_FLOAT32, DIM, 1, TOP_K, scores, indices, NULL);
        if (rc) die_gpu("mvfgpu_search", rc);
        printf("query %d:", c);
        for (int i = 0; i < TOP_K; i++) {
            uint32_t bits;
            memcpy(&bits, &scores[i], 4);
            printf(" %" PRIu64 ":%08" PRIx32, indices[i], bits);
        }
        printf("\n");
        float payload[DIM]; /* ScoredVector.vector of the best hit, served from HBM */
        rc = mvfgpu_corpus_gather_rows(corpus, &indices[0], 1, payload);
        if (rc) die_gpu("mvfgpu_corpus_gather_rows", rc);
        printf("best %d: [%g, %g, %g, %g]\n", c, payload[0], payload[1], payload[2], payload[3]);
    }

    /* a wrong-length query is refused (the reference's zip would silently truncate) */
    {
        float s[1];
        uint64_t ix[1];
        rc = mvfgpu_search(corpus, MVF_METRIC_L2, centre[0], MVF_DTYPE_FLOAT32, DIM - 1, 1, 1, s, ix, NULL);
        printf("short query -> %s\n", mvfgpu_strerror(rc));
    }
    mvfgpu_corpus_destroy(corpus);
    return 0;
}
