// similarity_search.cpp — the reference's examples/similarity_search.rs, line for line in its flow, written against
// include/mvf.hpp (the C++ mirror of the reference's host API): build three clusters of 20 vectors, save, reopen,
// take the first vector space, run the example's four queries through find_top_k_similar(&space, &query, k) -- whose
// scan (similarity_search.rs:140-176) runs on the GPU behind the C ABI -- and print rank, index, distance and payload.
//
//   g++ -std=c++17 -I include examples/cpp/similarity_search.cpp -L metrovector_amd -lmvf_gpu -lmvf_host -o ss && ./ss out.mvf
//   ./ss out.mvf --host-only      the host half only (no GPU needed): builder, reader, get_vector, the error variants
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mvf.hpp"

using namespace mvf;

static std::vector<std::vector<float>> create_clustered_vectors() {  // similarity_search.rs:42-76 (deterministic drift for noise)
    const float centre[3][4] = {{1, 1, 1, 1}, {5, 5, 5, 5}, {-2, 3, 0, 4}};
    const float drift[3][4] = {{1, -1, 0.5f, -0.5f}, {1, -1, 0.5f, -0.5f}, {1, -1, 1, -0.5f}};
    std::vector<std::vector<float>> v;
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 20; i++) {
            std::vector<float> row(4);
            for (int j = 0; j < 4; j++) row[j] = centre[c][j] + (float)i * 0.1f * drift[c][j];
            v.push_back(row);
        }
    return v;
}

static const char* variant(const MvfError& e) {
    static const char* names[] = {"Ok", "Io", "InvalidFormat", "UnsupportedVersion", "VectorSpaceNotFound", "IndexOutOfBounds",
                                  "DimensionMismatch", "InvalidVectorType", "CorruptedData", "Extension", "Build", "Device", "InvalidArgument"};
    return e.code() >= 0 && e.code() <= 12 ? names[e.code()] : "?";
}

int main(int argc, char** argv) {
    const std::string path = argc > 1 ? argv[1] : "similarity_search_example.mvf";
    const bool host_only = argc > 2 && std::strcmp(argv[2], "--host-only") == 0;
    try {
        // ---- similarity_search.rs:80-98: build and save ----
        MvfBuilder builder;
        builder.add_vector_space("embeddings", 4, VectorType::Dense, DistanceMetric::L2, DataType::Float32);
        builder.add_vectors("embeddings", create_clustered_vectors());
        builder.build().save(path);

        // ---- :100-101: reopen, first vector space ----
        MvfReader mvf_file = MvfReader::open(path);
        mvf_file.validate();
        mvf_file.validate_with_checksum();
        const VectorSpace space = mvf_file.vector_space(mvf_file.vector_space_names().front());
        std::printf("space %s: %" PRIu64 " vectors x %u, version %u, %zu space(s), %" PRIu64 " bytes\n", space.name().c_str(),
                    space.total_vectors(), space.dimension(), (unsigned)mvf_file.version(), mvf_file.num_vector_spaces(),
                    mvf_file.file_size());
        const std::vector<float> v25 = space.get_vector(25).as_f32();
        std::printf("vector 25: [%g, %g, %g, %g]\n", v25[0], v25[1], v25[2], v25[3]);

        // ---- the error variants a caller of the reference sees ----
        try { space.get_vector(60); } catch (const MvfError& e) { std::printf("get_vector(60) -> %s\n", variant(e)); }
        try { mvf_file.vector_space("nope"); } catch (const MvfError& e) { std::printf("vector_space(nope) -> %s\n", variant(e)); }
        try { MvfReader::open(path + ".missing"); } catch (const MvfError& e) { std::printf("open(missing) -> %s\n", variant(e)); }
        try {
            MvfBuilder b2;
            b2.add_vector_space("s", 4);
            b2.add_vectors("s", {{1, 2, 3, 4}, {1, 2, 3}});
        } catch (const MvfError& e) { std::printf("add_vectors(ragged) -> %s: %s\n", variant(e), e.what()); }
        if (host_only) return 0;

        // ---- :104-135: the four queries ----
        const std::vector<std::pair<std::vector<float>, const char*>> queries = {
            {{1, 1, 1, 1}, "Near cluster 1"}, {{5, 5, 5, 5}, "Near cluster 2"}, {{-2, 3, 0, 4}, "Near cluster 3"}, {{0, 0, 0, 0}, "At origin"}};
        const GpuVectorSpace resident(space);  // upload once for the four queries
        int qn = 0;
        for (const auto& [query, description] : queries) {
            const size_t k = 5;
            const std::vector<ScoredVector> top_k = qn == 0 ? find_top_k_similar(space, query, k)  // the reference's signature
                                                            : resident.find_top_k_similar(query, k);
            std::printf("=== Query: %s ===\nquery %d:", description, qn);
            for (const ScoredVector& s : top_k) {
                uint32_t bits;
                std::memcpy(&bits, &s.score, 4);
                std::printf(" %" PRIu64 ":%08" PRIx32, s.index, bits);
            }
            std::printf("\n");
            for (size_t rank = 0; rank < top_k.size(); rank++)
                std::printf("  %zu. Vector %" PRIu64 " (distance: %.3f): [%g, %g, %g, %g]\n", rank + 1, top_k[rank].index, top_k[rank].score,
                            top_k[rank].vector[0], top_k[rank].vector[1], top_k[rank].vector[2], top_k[rank].vector[3]);
            qn++;
        }
        try { resident.find_top_k_similar({1, 1, 1}, 1); } catch (const MvfError& e) { std::printf("short query -> %s\n", variant(e)); }
        return 0;
    } catch (const MvfError& e) {
        std::fprintf(stderr, "MvfError::%s: %s\n", variant(e), e.what());
        return 1;
    }
}
