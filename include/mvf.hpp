// mvf.hpp — header-only C++17 mirror of the reference's host API for the search path, over the two C ABIs
// (include/mvf_file.h -> libmvf_host.so, include/mvf_gpu.h -> libmvf_gpu.so).
//
// The reference is a Rust crate; no Rust toolchain exists in this image, so the host side above the C ABI is written
// in C++ with the reference's names, argument meaning and error behaviour:
//
//   reference (Rust)                                          here
//   MvfError (src/errors.rs:8-40)                             mvf::MvfError (one code per variant, mvf_status.h)
//   MvfReader::open / version / num_vector_spaces /           mvf::MvfReader (src/reader.rs:45-172)
//     vector_space_names / vector_space / file_size /
//     has_metadata / metadata_column_names / validate /
//     validate_with_checksum
//   VectorSpace::name / dimension / total_vectors /           mvf::VectorSpace (src/vectors/vector_space.rs:62-188)
//     vector_type / distance_metric / data_type /
//     get_vector / map_vector_range
//   Vector::dimension / data_type / as_bytes / as_f32         mvf::Vector (src/vectors/vector.rs:51-92)
//   VectorSlice (as_ptr, stride, count)                       mvf::VectorSlice (src/vectors/mem.rs:24-77)
//   MvfBuilder::new / add_vector_space / add_vectors /        mvf::MvfBuilder, mvf::BuiltMvf (src/builder.rs:93-558)
//     build -> BuiltMvf::save / to_bytes
//   ScoredVector, find_top_k_similar(&space, &query, k)       mvf::ScoredVector, mvf::find_top_k_similar
//     (examples/similarity_search.rs:14-37, :140-176)           -- the scan runs on the GPU (mvfgpu_search)
//
// find_top_k_similar(space, query, k) uploads the space's rows on every call, as the reference's function walks the file
// on every call; a caller with more than one query keeps a mvf::GpuVectorSpace (upload once, search many).
// Results follow the INTENDED semantics of the example (the k nearest for L2, best first; UPSTREAM.md F5) with the metric
// the space declares, not the example's hard-wired Euclidean distance.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mvf_file.h"
#include "mvf_gpu.h"
#include "mvf_status.h"

namespace mvf {

enum class DataType : uint8_t { Float32 = 0, Float16 = 1, Int8 = 2, UInt8 = 3 };       // schema/types.fbs:3-11 (vector types)
enum class VectorType : uint8_t { Dense = 0, Sparse = 1 };                              // schema/types.fbs:14-17
enum class DistanceMetric : uint8_t { L2 = 0, InnerProduct = 1, Cosine = 2, Custom = 255 };  // schema/types.fbs:20-25

// MvfError (src/errors.rs:8-40): code() is the variant, what() the reference's message text.
class MvfError : public std::runtime_error {
public:
    MvfError(int code, const std::string& msg) : std::runtime_error(msg), code_(code) {}
    int code() const noexcept { return code_; }

private:
    int code_;
};

namespace detail {
inline void check_host(int rc) {
    if (rc != MVF_OK) throw MvfError(rc, std::string(mvf_strerror(rc)) + ": " + mvf_last_error_message());
}
inline void check_gpu(int rc) {
    if (rc != MVF_OK) throw MvfError(rc, std::string(mvfgpu_strerror(rc)) + ": " + mvfgpu_last_error_message());
}
}  // namespace detail

// VectorSlice<'a> (src/vectors/mem.rs:24-30): a borrowed, strided view of rows inside the mapping.
struct VectorSlice {
    const void* data = nullptr;
    uint64_t stride = 0;  // bytes between rows
    uint64_t count = 0;
    DataType data_type = DataType::Float32;
    template <class T> const T* as_ptr() const { return static_cast<const T*>(data); }  // mem.rs:75-77
};

// Vector<'a> (src/vectors/vector.rs:28-33): one row, borrowed from the mapping.
class Vector {
public:
    Vector(const void* data, uint64_t len, uint32_t dimension, DataType dt) : data_(data), len_(len), dim_(dimension), dt_(dt) {}
    uint32_t dimension() const { return dim_; }
    DataType data_type() const { return dt_; }
    std::pair<const uint8_t*, uint64_t> as_bytes() const { return {static_cast<const uint8_t*>(data_), len_}; }
    std::vector<float> as_f32() const {  // vector.rs:71-92 (Float32 / Float16; the integer types widen)
        std::vector<float> out(dim_);
        uint64_t n = 0;
        detail::check_host(mvf_vector_as_f32(data_, len_, (uint8_t)dt_, out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }

private:
    const void* data_;
    uint64_t len_;
    uint32_t dim_;
    DataType dt_;
};

class MvfReader;

// VectorSpace<'a> (src/vectors/vector_space.rs:34-39): valid while its reader lives.
class VectorSpace {
public:
    std::string name() const { return std::string(s_.name, s_.name_len); }
    uint32_t dimension() const { return s_.dimension; }
    uint64_t total_vectors() const { return s_.total_vectors; }
    VectorType vector_type() const { return (VectorType)s_.vector_type; }
    DistanceMetric distance_metric() const { return (DistanceMetric)s_.distance_metric; }
    DataType data_type() const { return (DataType)s_.data_type; }
    Vector get_vector(uint64_t index) const {  // :101-142: IndexOutOfBounds beyond total_vectors
        const void* d = nullptr;
        uint64_t len = 0;
        detail::check_host(mvf_space_get_vector(&s_, index, &d, &len));
        return Vector(d, len, s_.dimension, data_type());
    }
    VectorSlice map_vector_range(uint64_t start, uint64_t count) const {  // :155-188
        mvf_vector_slice v;
        detail::check_host(mvf_space_map_vector_range(&s_, start, count, &v));
        VectorSlice out;
        out.data = v.data, out.stride = v.stride, out.count = v.count, out.data_type = (DataType)v.data_type;
        return out;
    }
    const mvf_vector_space& raw() const { return s_; }

private:
    friend class MvfReader;
    explicit VectorSpace(const mvf_vector_space& s) : s_(s) {}
    mvf_vector_space s_;
};

// MvfReader (src/reader.rs:27-33): the mapping and the verified footer.
class MvfReader {
public:
    static MvfReader open(const std::string& path) {  // :45-79
        mvf_reader* r = nullptr;
        detail::check_host(mvf_reader_open(path.c_str(), &r));
        return MvfReader(r);
    }
    MvfReader(MvfReader&& o) noexcept : r_(o.r_) { o.r_ = nullptr; }
    MvfReader& operator=(MvfReader&& o) noexcept {
        if (this != &o) {
            if (r_) mvf_reader_close(r_);
            r_ = o.r_, o.r_ = nullptr;
        }
        return *this;
    }
    MvfReader(const MvfReader&) = delete;
    MvfReader& operator=(const MvfReader&) = delete;
    ~MvfReader() {
        if (r_) mvf_reader_close(r_);
    }
    uint16_t version() const {
        uint16_t v = 0;
        detail::check_host(mvf_reader_version(r_, &v));
        return v;
    }
    size_t num_vector_spaces() const {
        uint64_t n = 0;
        detail::check_host(mvf_reader_num_vector_spaces(r_, &n));
        return (size_t)n;
    }
    std::vector<std::string> vector_space_names() const {  // :92-98
        std::vector<std::string> out;
        for (uint64_t i = 0, n = num_vector_spaces(); i < n; i++) {
            const char* s = nullptr;
            uint32_t len = 0;
            detail::check_host(mvf_reader_vector_space_name(r_, i, &s, &len));
            out.emplace_back(s, len);
        }
        return out;
    }
    VectorSpace vector_space(const std::string& name) const {  // :104-119: VectorSpaceNotFound
        mvf_vector_space s;
        detail::check_host(mvf_reader_vector_space(r_, name.c_str(), &s));
        return VectorSpace(s);
    }
    uint64_t file_size() const {
        uint64_t n = 0;
        detail::check_host(mvf_reader_file_size(r_, &n));
        return n;
    }
    bool has_metadata() const {
        int b = 0;
        detail::check_host(mvf_reader_has_metadata(r_, &b));
        return b != 0;
    }
    std::vector<std::string> metadata_column_names() const {  // :132-143
        std::vector<std::string> out;
        uint64_t n = 0;
        detail::check_host(mvf_reader_num_metadata_columns(r_, &n));
        for (uint64_t i = 0; i < n; i++) {
            const char* s = nullptr;
            uint32_t len = 0;
            detail::check_host(mvf_reader_metadata_column_name(r_, i, &s, &len));
            out.emplace_back(s, len);
        }
        return out;
    }
    void validate() const { detail::check_host(mvf_reader_validate(r_)); }                              // :149-162
    void validate_with_checksum() const { detail::check_host(mvf_reader_validate_with_checksum(r_)); }  // :172-220 (todo!() upstream)

private:
    explicit MvfReader(mvf_reader* r) : r_(r) {}
    mvf_reader* r_;
};

// BuiltMvf (src/builder.rs:395-558)
class BuiltMvf {
public:
    void save(const std::string& path) const { detail::check_host(mvf_builder_save(b_, path.c_str(), 0)); }  // :408-411
    std::vector<uint8_t> to_bytes() const {                                                                  // :417-558
        uint8_t* p = nullptr;
        uint64_t len = 0;
        detail::check_host(mvf_builder_to_bytes(b_, 0, &p, &len));
        std::vector<uint8_t> out(p, p + len);
        mvf_free(p);
        return out;
    }
    BuiltMvf(BuiltMvf&& o) noexcept : b_(o.b_) { o.b_ = nullptr; }
    BuiltMvf(const BuiltMvf&) = delete;
    BuiltMvf& operator=(const BuiltMvf&) = delete;
    ~BuiltMvf() {
        if (b_) mvf_builder_free(b_);
    }

private:
    friend class MvfBuilder;
    explicit BuiltMvf(mvf_builder* b) : b_(b) {}
    mvf_builder* b_;
};

// MvfBuilder (src/builder.rs:44-308)
class MvfBuilder {
public:
    MvfBuilder() { detail::check_host(mvf_builder_new(&b_)); }  // :93-95
    MvfBuilder(const MvfBuilder&) = delete;
    MvfBuilder& operator=(const MvfBuilder&) = delete;
    ~MvfBuilder() {
        if (b_) mvf_builder_free(b_);
    }
    // :113-135 + VectorSpaceBuilderRef's setters (:339-357) in one call
    MvfBuilder& add_vector_space(const std::string& name, uint32_t dimension, VectorType vt = VectorType::Dense,
                                 DistanceMetric dm = DistanceMetric::L2, DataType dt = DataType::Float32) {
        detail::check_host(mvf_builder_add_vector_space(b_, name.c_str(), dimension, (uint8_t)vt, (uint8_t)dm, (uint8_t)dt, nullptr));
        return *this;
    }
    // :151-196: rows of `dimension` floats, stored as the space's type (Float32 / Float16; Build error otherwise);
    // DimensionMismatch names the offending row's length
    MvfBuilder& add_vectors(const std::string& space_name, const std::vector<std::vector<float>>& vectors) {
        if (vectors.empty()) return *this;
        const size_t dim = vectors.front().size();
        std::vector<float> flat;
        flat.reserve(vectors.size() * dim);
        for (const auto& v : vectors) {
            if (v.size() != dim)
                throw MvfError(MVF_ERR_DIMENSION_MISMATCH, "Dimension mismatch: expected " + std::to_string(dim) + ", got " + std::to_string(v.size()));
            flat.insert(flat.end(), v.begin(), v.end());
        }
        detail::check_host(mvf_builder_add_vectors_f32(b_, space_name.c_str(), flat.data(), vectors.size(), (uint32_t)dim));
        return *this;
    }
    BuiltMvf build() {  // :241-308 (consumes the builder)
        mvf_builder* b = b_;
        b_ = nullptr;
        return BuiltMvf(b);
    }

private:
    mvf_builder* b_ = nullptr;
};

// ScoredVector (examples/similarity_search.rs:14-19)
struct ScoredVector {
    uint64_t index;
    float score;
    std::vector<float> vector;
};

// A vector space resident in HBM: upload once (VectorSpace::map_vector_range -> mvfgpu_corpus_create), search many.
class GpuVectorSpace {
public:
    explicit GpuVectorSpace(const VectorSpace& space, int device = 0)
        : dim_(space.dimension()), metric_(space.distance_metric()), dt_(space.data_type()) {
        if (mvfgpu_abi_version() != MVFGPU_ABI_VERSION)  // a library built from another header: its structs may differ
            throw MvfError(MVF_ERR_INVALID_ARGUMENT, "libmvf_gpu speaks ABI version " + std::to_string(mvfgpu_abi_version()) +
                                                         ", this program was built against " + std::to_string(MVFGPU_ABI_VERSION));
        const VectorSlice s = space.map_vector_range(0, space.total_vectors());
        detail::check_gpu(mvfgpu_corpus_create(s.data, s.count, dim_, (uint8_t)s.data_type, s.stride, 0, device, &c_));
    }
    GpuVectorSpace(const GpuVectorSpace&) = delete;
    GpuVectorSpace& operator=(const GpuVectorSpace&) = delete;
    ~GpuVectorSpace() {
        if (c_) mvfgpu_corpus_destroy(c_);
    }
    // examples/similarity_search.rs:140-176 with the metric the space declares; DimensionMismatch for a query of another
    // length (the reference's zip would truncate silently); the payload of every hit is fetched from HBM
    std::vector<ScoredVector> find_top_k_similar(const std::vector<float>& query, size_t k, bool with_vectors = true) const {
        if (dt_ != DataType::Float32 && dt_ != DataType::Float16)
            throw MvfError(MVF_ERR_BUILD, "find_top_k_similar takes f32 queries: Float32 / Float16 spaces");
        std::vector<float> scores(k);
        std::vector<uint64_t> idx(k);
        const size_t es = dt_ == DataType::Float32 ? 4 : 2;
        // one query: the library writes the first min(k, rows) payload rows only (include/mvf_gpu.h) -- any k, as the reference takes
        mvfgpu_corpus_info info{};
        info.struct_size = sizeof info;
        detail::check_gpu(mvfgpu_corpus_get_info(c_, &info));
        std::vector<uint8_t> rows(with_vectors ? std::min<uint64_t>(k, info.rows) * dim_ * es : 0);
        if (with_vectors)  // the k best and their rows in one call
            detail::check_gpu(mvfgpu_search_fetch(c_, (uint8_t)metric_, query.data(), MVF_DTYPE_FLOAT32, (uint32_t)query.size(), 1, (uint32_t)k,
                                                  scores.data(), idx.data(), nullptr, rows.data()));
        else
            detail::check_gpu(mvfgpu_search(c_, (uint8_t)metric_, query.data(), MVF_DTYPE_FLOAT32, (uint32_t)query.size(), 1, (uint32_t)k,
                                            scores.data(), idx.data(), nullptr));
        std::vector<ScoredVector> out;
        for (size_t i = 0; i < k && idx[i] != ~0ull; i++) {  // fewer than k rows: the tail is padding
            out.push_back({idx[i], scores[i], {}});
            if (with_vectors) out.back().vector = Vector(rows.data() + i * dim_ * es, dim_ * es, dim_, dt_).as_f32();
        }
        return out;
    }
    mvfgpu_corpus* raw() const { return c_; }

private:
    mvfgpu_corpus* c_ = nullptr;
    uint32_t dim_;
    DistanceMetric metric_;
    DataType dt_;
};

// fn find_top_k_similar(space: &VectorSpace, query: &[f32], k: usize) -> Result<Vec<ScoredVector>, _>
inline std::vector<ScoredVector> find_top_k_similar(const VectorSpace& space, const std::vector<float>& query, size_t k) {
    return GpuVectorSpace(space).find_top_k_similar(query, k);
}

}  // namespace mvf
