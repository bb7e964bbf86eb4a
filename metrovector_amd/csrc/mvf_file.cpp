// mvf_file.cpp — libmvf_host.so: C++ MVF reader/writer (include/mvf_file.h).
//
// Mirrors the reference's Rust host side, function for function:
//   MvfReader::open / validate_*   src/reader.rs:45-79, :225-278
//   VectorSpace::get_vector        src/vectors/vector_space.rs:101-142
//   VectorSpace::map_vector_range  src/vectors/vector_space.rs:155-188
//   Vector::as_f32                 src/vectors/vector.rs:71-92
//   MvfBuilder / BuiltMvf          src/builder.rs:113-308, :417-558
// The footer codec below implements the public FlatBuffers binary format for
// the tables of schema/{mvf,core,index}.fbs by hand (there is no flatc here).

#include "../../include/mvf_file.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int st, const std::string& msg) {
    g_err = msg;
    return st;
}

const uint8_t kMagic[4] = {'M', 'V', 'F', '1'};  // src/lib.rs:25
constexpr size_t kFooterSizeField = 4;           // src/lib.rs:26

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

uint32_t elem_size(uint8_t dt) {  // vector_space.rs:122-127
    switch (dt) {
    case MVF_DTYPE_FLOAT32: return 4;
    case MVF_DTYPE_FLOAT16: return 2;
    case MVF_DTYPE_INT8:
    case MVF_DTYPE_UINT8: return 1;
    default: return 0;
    }
}

// ------------------------------------------------------------------ half ----
float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

uint16_t f32_to_f16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u, exp = (x >> 23) & 0xFFu, man = x & 0x7FFFFFu;
    if (exp == 255) return (uint16_t)(man ? (sign | 0x7E00u | (man >> 13)) : (sign | 0x7C00u));
    int32_t e = (int32_t)exp - 112;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        uint32_t shift = (uint32_t)(14 - e), hm = man >> shift, rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) hm++;
        return (uint16_t)(sign | hm);
    }
    uint32_t hm = man >> 13, rem = man & 0x1FFFu;
    uint16_t h = (uint16_t)(sign | ((uint32_t)e << 10) | hm);
    if (rem > 0x1000u || (rem == 0x1000u && (hm & 1u))) h++;
    return h;
}

// ----------------------------------------------------------------- crc32 ----
// crc32fast::hash (src/builder.rs:251) == CRC-32/ISO-HDLC (poly 0xEDB88320, reflected, init / xorout 0xFFFFFFFF).
// Slicing-by-8 on one core; blocks of 8 MiB and more are cut into one segment per host thread and the segment CRCs
// are joined with the GF(2) "append len zero bytes" operator (crc(A|B) = shift(crc(A), |B|) ^ crc(B)): a multi-GB
// vector block is checked at memory speed beside its upload instead of at 0.4 GB/s.
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int k = 1; k < 8; k++) t[k][i] = t[0][t[k - 1][i] & 0xFF] ^ (t[k - 1][i] >> 8);
    }
};
const CrcTables g_crc;  // built at load time: no lazy-init race between the upload and the checksum thread

// raw register update (no init / final xor), so that segments can be chained
uint32_t crc32_update(uint32_t c, const uint8_t* p, uint64_t n) {
    const auto& t = g_crc.t;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
        c = t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
        n--;
    }
    while (n >= 8) {
        uint64_t w;
        std::memcpy(&w, p, 8);
        const uint32_t lo = (uint32_t)w ^ c, hi = (uint32_t)(w >> 32);
        c = t[7][lo & 0xFF] ^ t[6][(lo >> 8) & 0xFF] ^ t[5][(lo >> 16) & 0xFF] ^ t[4][lo >> 24] ^
            t[3][hi & 0xFF] ^ t[2][(hi >> 8) & 0xFF] ^ t[1][(hi >> 16) & 0xFF] ^ t[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c;
}

uint32_t gf2_times(const uint32_t* mat, uint32_t vec) {
    uint32_t sum = 0;
    for (; vec; vec >>= 1, mat++)
        if (vec & 1) sum ^= *mat;
    return sum;
}

void gf2_square(uint32_t* sq, const uint32_t* mat) {
    for (int n = 0; n < 32; n++) sq[n] = gf2_times(mat, mat[n]);
}

// CRC of A|B from the finished CRCs of A and B and |B| (the construction zlib's crc32_combine uses: the operator that
// advances the register over one zero BIT, squared up to |B| zero bytes)
uint32_t crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    if (len_b == 0) return crc_a;
    uint32_t even[32], odd[32];
    odd[0] = 0xEDB88320u;
    for (int n = 1; n < 32; n++) odd[n] = 1u << (n - 1);
    gf2_square(even, odd);  // two zero bits
    gf2_square(odd, even);  // four zero bits
    do {
        gf2_square(even, odd);  // first pass: one zero byte
        if (len_b & 1) crc_a = gf2_times(even, crc_a);
        len_b >>= 1;
        if (!len_b) break;
        gf2_square(odd, even);
        if (len_b & 1) crc_a = gf2_times(odd, crc_a);
        len_b >>= 1;
    } while (len_b);
    return crc_a ^ crc_b;
}

uint32_t crc32_ieee(const uint8_t* p, uint64_t n) {
    unsigned threads = 1;
    if (n >= ((uint64_t)8 << 20)) {
        threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char* e = std::getenv("MVF_CRC_THREADS")) threads = (unsigned)std::max(1, std::atoi(e));
    }
    if (threads <= 1) return crc32_update(0xFFFFFFFFu, p, n) ^ 0xFFFFFFFFu;
    const uint64_t seg = ((n + threads - 1) / threads + 63) & ~(uint64_t)63;
    std::vector<uint32_t> part(threads, 0);
    std::vector<uint64_t> len(threads, 0);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; t++) {
        const uint64_t o = (uint64_t)t * seg;
        if (o >= n) break;
        len[t] = std::min(seg, n - o);
        pool.emplace_back([&part, &len, p, o, t] { part[t] = crc32_update(0xFFFFFFFFu, p + o, len[t]) ^ 0xFFFFFFFFu; });
    }
    for (auto& th : pool) th.join();
    uint32_t crc = part[0];
    for (unsigned t = 1; t < threads && len[t]; t++) crc = crc32_join(crc, part[t], len[t]);
    return crc;
}

// ------------------------------------------------- FlatBuffers: reading ----
struct FbView {
    const uint8_t* b = nullptr;
    size_t n = 0;
    bool in(size_t pos, size_t len) const { return pos <= n && len <= n - pos; }
};

struct FbTable {
    const FbView* v = nullptr;
    size_t pos = 0, vt = 0;
    uint16_t vt_size = 0, tb_size = 0;
    bool ok = false;

    // absolute position of field `slot`, 0 if absent
    size_t field(unsigned slot) const {
        size_t e = 4 + 2 * (size_t)slot;
        if (e + 2 > vt_size) return 0;
        uint16_t off = rd16(v->b + vt + e);
        return off ? pos + off : 0;
    }
};

bool fb_table_at(const FbView& v, size_t pos, FbTable* t, std::string* why) {
    if (pos % 4 || !v.in(pos, 4)) { *why = "table offset out of range"; return false; }
    int32_t so = (int32_t)rd32(v.b + pos);
    int64_t vt = (int64_t)pos - so;
    if (vt < 0 || (vt % 2) || !v.in((size_t)vt, 4)) { *why = "vtable out of range"; return false; }
    uint16_t vs = rd16(v.b + vt), ts = rd16(v.b + vt + 2);
    if (vs < 4 || (vs % 2) || !v.in((size_t)vt, vs)) { *why = "bad vtable size"; return false; }
    if (ts < 4 || !v.in(pos, ts)) { *why = "bad table size"; return false; }
    for (size_t e = 4; e + 2 <= vs; e += 2)
        if (rd16(v.b + vt + e) >= ts && rd16(v.b + vt + e) != 0) { *why = "field offset beyond table"; return false; }
    t->v = &v;
    t->pos = pos;
    t->vt = (size_t)vt;
    t->vt_size = vs;
    t->tb_size = ts;
    t->ok = true;
    return true;
}

// follow a uoffset stored at `fpos`
bool fb_indirect(const FbView& v, size_t fpos, size_t* target) {
    if (!v.in(fpos, 4)) return false;
    uint64_t t = (uint64_t)fpos + rd32(v.b + fpos);
    if (t >= v.n) return false;
    *target = (size_t)t;
    return true;
}

bool fb_vector(const FbView& v, size_t fpos, size_t elem, size_t* first, uint32_t* count) {
    size_t vp;
    if (!fb_indirect(v, fpos, &vp) || !v.in(vp, 4)) return false;
    uint32_t c = rd32(v.b + vp);
    if (!v.in(vp + 4, (size_t)c * elem)) return false;
    *first = vp + 4;
    *count = c;
    return true;
}

// strict UTF-8 check (the Rust verifier rejects footers whose strings are not valid UTF-8)
bool utf8_valid(const uint8_t* p, size_t n) {
    size_t i = 0;
    while (i < n) {
        const uint8_t c = p[i];
        size_t need;
        uint32_t cp;
        if (c < 0x80) { i++; continue; }
        else if ((c & 0xE0) == 0xC0) { need = 1; cp = c & 0x1F; }
        else if ((c & 0xF0) == 0xE0) { need = 2; cp = c & 0x0F; }
        else if ((c & 0xF8) == 0xF0) { need = 3; cp = c & 0x07; }
        else return false;
        if (n - i <= need) return false;
        for (size_t k = 1; k <= need; k++) {
            if ((p[i + k] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (p[i + k] & 0x3F);
        }
        if ((need == 1 && cp < 0x80) || (need == 2 && cp < 0x800) || (need == 3 && cp < 0x10000)) return false;  // overlong
        if (cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += need + 1;
    }
    return true;
}

bool fb_string(const FbView& v, size_t fpos, const char** s, uint32_t* len) {
    size_t first;
    uint32_t c;
    if (!fb_vector(v, fpos, 1, &first, &c)) return false;
    if (!v.in(first, (size_t)c + 1) || v.b[first + c] != 0) return false;  // NUL terminator (verifier rule)
    if (!utf8_valid(v.b + first, c)) return false;
    *s = reinterpret_cast<const char*>(v.b + first);
    *len = c;
    return true;
}

// ------------------------------------------------- FlatBuffers: writing ----
// Back-to-front builder following the public FlatBuffers encoding rules.
class FbBuilder {
  public:
    uint32_t used() const { return (uint32_t)(buf_.size() - head_); }

    void prep(size_t size, size_t additional) {
        if (size > minalign_) minalign_ = size;
        size_t pad = (~(used() + additional) + 1) & (size - 1);
        make_room(pad + size + additional);
        for (size_t i = 0; i < pad; i++) buf_[--head_] = 0;
    }
    template <typename T> void place(T v) {
        head_ -= sizeof(T);
        std::memcpy(&buf_[head_], &v, sizeof(T));  // little-endian host
    }
    template <typename T> void push(T v) {
        prep(sizeof(T), 0);
        place(v);
    }
    void push_uoffset_to(uint32_t target) {
        prep(4, 0);
        place<uint32_t>(used() - target + 4);
    }
    uint32_t create_string(const std::string& s) {
        prep(4, s.size() + 1);
        make_room(s.size() + 1);
        buf_[--head_] = 0;
        head_ -= s.size();
        std::memcpy(&buf_[head_], s.data(), s.size());
        push<uint32_t>((uint32_t)s.size());
        return used();
    }
    uint32_t create_byte_vector(const uint8_t* p, size_t n) {
        prep(4, n);
        make_room(n);
        head_ -= n;
        if (n) std::memcpy(&buf_[head_], p, n);
        push<uint32_t>((uint32_t)n);
        return used();
    }
    uint32_t create_offset_vector(const std::vector<uint32_t>& offs) {
        prep(4, offs.size() * 4);
        for (size_t i = offs.size(); i-- > 0;) push_uoffset_to(offs[i]);
        push<uint32_t>((uint32_t)offs.size());
        return used();
    }
    uint32_t create_struct_vector(const uint8_t* p, size_t elem, size_t count, size_t align) {
        prep(4, elem * count);
        prep(align, elem * count);
        make_room(elem * count);
        head_ -= elem * count;
        if (count) std::memcpy(&buf_[head_], p, elem * count);
        push<uint32_t>((uint32_t)count);
        return used();
    }
    // ---- tables
    void start_table(unsigned nfields) {
        slots_.assign(nfields, 0);
        object_start_ = used();
    }
    template <typename T> void add_scalar(unsigned slot, T v, T def) {
        if (v == def) return;  // the Rust builder omits defaults (force_defaults off)
        push<T>(v);
        slots_[slot] = used();
    }
    void add_offset(unsigned slot, uint32_t target) {
        if (!target) return;
        push_uoffset_to(target);
        slots_[slot] = used();
    }
    uint32_t end_table() {
        prep(4, 0);
        place<int32_t>(0);
        const uint32_t object_off = used();
        size_t nf = slots_.size();
        while (nf > 0 && slots_[nf - 1] == 0) nf--;
        std::vector<uint16_t> vt(2 + nf);
        vt[0] = (uint16_t)((2 + nf) * 2);
        vt[1] = (uint16_t)(object_off - object_start_);
        for (size_t i = 0; i < nf; i++) vt[2 + i] = slots_[i] ? (uint16_t)(object_off - slots_[i]) : 0;
        // reuse an identical vtable if one was already written
        uint32_t vt_off = 0;
        for (uint32_t prev : vtables_) {
            const uint8_t* pv = &buf_[buf_.size() - prev];
            if (rd16(pv) == vt[0] && std::memcmp(pv, vt.data(), vt[0]) == 0) {
                vt_off = prev;
                break;
            }
        }
        if (!vt_off) {
            make_room(vt[0]);
            head_ -= vt[0];
            std::memcpy(&buf_[head_], vt.data(), vt[0]);
            vt_off = used();
            vtables_.push_back(vt_off);
        }
        int32_t so = (int32_t)vt_off - (int32_t)object_off;
        std::memcpy(&buf_[buf_.size() - object_off], &so, 4);
        return object_off;
    }
    std::vector<uint8_t> finish_minimal(uint32_t root) {  // builder.rs:545
        prep(minalign_, 4);
        push_uoffset_to(root);
        return std::vector<uint8_t>(buf_.begin() + head_, buf_.end());
    }

  private:
    void make_room(size_t need) {
        if (head_ >= need) return;
        size_t old = buf_.size(), grow = std::max<size_t>(need - head_, old ? old : 1024);
        std::vector<uint8_t> nb(old + grow, 0);
        if (old > head_) std::memcpy(nb.data() + grow + head_, buf_.data() + head_, old - head_);  // (memcpy from an empty vector's null data() is undefined, even for 0 bytes)
        buf_.swap(nb);
        head_ += grow;
    }
    std::vector<uint8_t> buf_;
    size_t head_ = 0;
    size_t minalign_ = 1;
    std::vector<uint32_t> slots_;
    uint32_t object_start_ = 0;
    std::vector<uint32_t> vtables_;
};

// --------------------------------------------------------------- reader ----
struct SpaceRec {
    std::string name;
    const char* name_ptr = nullptr;
    uint32_t name_len = 0;
    uint32_t dimension = 0;
    uint64_t total_vectors = 0;
    uint8_t vector_type = 0, distance_metric = 0, data_type = 0, index_type = 0;
    uint32_t vectors_block_index = 0, vector_ids_block_index = 0;
    bool sparse = false, tombstones = false;
    uint8_t tomb_format = 0;        // TombstoneInfo, schema/core.fbs:35-39
    uint32_t tomb_block_index = 0;
    uint64_t tomb_deleted_count = 0;
};

}  // namespace

struct mvf_reader {
    const uint8_t* data = nullptr;  // whole file
    size_t len = 0;
    bool mapped = false;
    std::vector<uint8_t> owned;
    uint16_t version = 0;
    std::vector<SpaceRec> spaces;
    std::vector<mvf_data_block> blocks;
    bool has_metadata = false;
    std::vector<std::string> metadata_names;
};

namespace {

// validate_file_structure, reader.rs:259-278
int validate_file_structure(const uint8_t* m, size_t len) {
    if (len < sizeof(kMagic) + kFooterSizeField + sizeof(kMagic)) return fail(MVF_ERR_INVALID_FORMAT, "File too small");
    if (std::memcmp(m, kMagic, 4) != 0) return fail(MVF_ERR_INVALID_FORMAT, "Invalid magic bytes at start of file");
    if (std::memcmp(m + len - 4, kMagic, 4) != 0)
        return fail(MVF_ERR_INVALID_FORMAT, "Invalid magic bytes at end of file, it may be corrupted");
    return MVF_OK;
}

int parse_footer(mvf_reader* r, size_t fs, size_t fe) {
    FbView v{r->data + fs, fe - fs};
    std::string why;
    auto bad = [&](const std::string& w) { return fail(MVF_ERR_INVALID_FORMAT, "Failed to parse footer: " + w); };
    size_t root_pos;
    if (v.n < 8 || !fb_indirect(v, 0, &root_pos)) return bad("root offset out of range");
    FbTable ft;
    if (!fb_table_at(v, root_pos, &ft, &why)) return bad(why);

    // FileFooter, schema/mvf.fbs:12-30 (slots in declaration order)
    size_t f = ft.field(0);
    r->version = f ? (v.in(f, 2) ? rd16(v.b + f) : 0) : 3;  // schema default 3
    size_t f_spaces = ft.field(1), f_blocks = ft.field(2);
    if (!f_spaces) return bad("missing required field vector_spaces");
    if (!f_blocks) return bad("missing required field block_manifest");

    size_t first;
    uint32_t cnt;
    if (!fb_vector(v, f_blocks, 40, &first, &cnt)) return bad("block_manifest out of range");
    if (first % 8) return bad("block_manifest misaligned");
    r->blocks.resize(cnt);
    for (uint32_t i = 0; i < cnt; i++) {  // DataBlock struct, schema/core.fbs:7-13: 40 B, align 8
        const uint8_t* p = v.b + first + (size_t)i * 40;
        r->blocks[i].offset = rd64(p);
        r->blocks[i].size = rd64(p + 8);
        r->blocks[i].compression = p[16];
        r->blocks[i].compressed_size = rd64(p + 24);
        r->blocks[i].checksum = rd32(p + 32);
    }

    if (!fb_vector(v, f_spaces, 4, &first, &cnt)) return bad("vector_spaces out of range");
    r->spaces.resize(cnt);
    for (uint32_t i = 0; i < cnt; i++) {
        size_t tp;
        FbTable st;
        if (!fb_indirect(v, first + (size_t)i * 4, &tp) || !fb_table_at(v, tp, &st, &why)) return bad("vector space table: " + why);
        SpaceRec& s = r->spaces[i];
        // VectorSpace, schema/core.fbs:42-57
        size_t fn = st.field(0);
        if (!fn || !fb_string(v, fn, &s.name_ptr, &s.name_len)) return bad("vector space name missing or malformed");
        s.name.assign(s.name_ptr, s.name_len);
        auto u8f = [&](unsigned slot) -> uint8_t { size_t p = st.field(slot); return p && v.in(p, 1) ? v.b[p] : 0; };
        auto u32f = [&](unsigned slot) -> uint32_t { size_t p = st.field(slot); return p && v.in(p, 4) ? rd32(v.b + p) : 0; };
        s.dimension = u32f(1);
        size_t ptv = st.field(2);
        s.total_vectors = ptv && v.in(ptv, 8) ? rd64(v.b + ptv) : 0;
        s.vector_type = u8f(3);
        s.distance_metric = u8f(4);
        s.data_type = u8f(5);
        s.vectors_block_index = u32f(6);
        s.index_type = u8f(7);
        s.vector_ids_block_index = u32f(9);
        s.sparse = st.field(10) != 0;
        s.tombstones = st.field(11) != 0;
        if (s.tombstones) {  // TombstoneInfo { format:ubyte; data_block_index:uint; deleted_count:ulong }
            size_t tp2;
            FbTable tt;
            if (!fb_indirect(v, st.field(11), &tp2) || !fb_table_at(v, tp2, &tt, &why)) return bad("tombstone table: " + why);
            size_t p0 = tt.field(0), p1 = tt.field(1), p2 = tt.field(2);
            s.tomb_format = p0 && v.in(p0, 1) ? v.b[p0] : 0;
            s.tomb_block_index = p1 && v.in(p1, 4) ? rd32(v.b + p1) : 0;
            s.tomb_deleted_count = p2 && v.in(p2, 8) ? rd64(v.b + p2) : 0;
        }
    }

    size_t f_meta = ft.field(3);
    r->has_metadata = f_meta != 0;  // reader.rs:127-129
    if (f_meta) {
        if (!fb_vector(v, f_meta, 4, &first, &cnt)) return bad("metadata_columns out of range");
        for (uint32_t i = 0; i < cnt; i++) {
            size_t tp;
            FbTable mt;
            if (!fb_indirect(v, first + (size_t)i * 4, &tp) || !fb_table_at(v, tp, &mt, &why)) return bad("metadata column: " + why);
            const char* nm;
            uint32_t nl;
            size_t fn = mt.field(0);
            if (!fn || !fb_string(v, fn, &nm, &nl)) return bad("metadata column name missing or malformed");
            r->metadata_names.emplace_back(nm, nl);
        }
    }
    return MVF_OK;
}

// validate_footer_bounds + the parse of open(), reader.rs:225-256 and :63-76
int open_common(mvf_reader* r) {
    int rc = validate_file_structure(r->data, r->len);
    if (rc) return rc;
    const size_t tail = kFooterSizeField + sizeof(kMagic);
    const size_t fl_pos = r->len - tail;
    const size_t footer_len = rd32(r->data + fl_pos);
    if (footer_len + tail > r->len - sizeof(kMagic)) return fail(MVF_ERR_INVALID_FORMAT, "Invalid footer length");
    const size_t fs = r->len - tail - footer_len, fe = r->len - tail;
    rc = parse_footer(r, fs, fe);
    if (rc) return rc;
    if (r->version != 1)
        return fail(MVF_ERR_UNSUPPORTED_VERSION,
                    "Unsupported version: got " + std::to_string(r->version) + ", expected 1");
    return MVF_OK;
}

// ---------------------------------------------------------------- builder ----
struct SpaceB {
    std::string name;
    uint32_t dimension;
    uint8_t vector_type, distance_metric, data_type;
    std::vector<uint8_t> vectors;
    std::vector<uint8_t> ids;         // u64 LE per row (vector_ids block), empty = positions are the ids
    std::vector<uint8_t> tombstones;  // Bitmap or SortedList payload
    uint8_t tomb_format = 0;
    uint64_t tomb_deleted = 0;
};
struct ColumnB {
    std::string name;
    uint8_t data_type;
    std::vector<uint8_t> data;
};

}  // namespace

struct mvf_builder {
    std::vector<SpaceB> spaces;
    std::vector<ColumnB> columns;
};

extern "C" {

const char* mvf_last_error_message(void) { return g_err.c_str(); }

const char* mvf_strerror(int status) {
    switch (status) {
    case MVF_OK: return "ok";
    case MVF_ERR_IO: return "I/O error";
    case MVF_ERR_INVALID_FORMAT: return "Invalid file format";
    case MVF_ERR_UNSUPPORTED_VERSION: return "Unsupported version";
    case MVF_ERR_SPACE_NOT_FOUND: return "Vector space not found";
    case MVF_ERR_INDEX_OUT_OF_BOUNDS: return "Index out of bounds";
    case MVF_ERR_DIMENSION_MISMATCH: return "Dimension mismatch";
    case MVF_ERR_INVALID_VECTOR_TYPE: return "Invalid vector type";
    case MVF_ERR_CORRUPTED_DATA: return "Corrupted data";
    case MVF_ERR_EXTENSION: return "Extension error";
    case MVF_ERR_BUILD: return "Build error";
    case MVF_ERR_DEVICE: return "Device error";
    case MVF_ERR_INVALID_ARGUMENT: return "Invalid argument";
    default: return "unknown status";
    }
}

uint32_t mvf_crc32(const void* data, uint64_t len) { return crc32_ieee(static_cast<const uint8_t*>(data), len); }
uint16_t mvf_f32_to_f16(float f) { return f32_to_f16(f); }
float mvf_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
void mvf_free(void* p) { std::free(p); }

// ---- MvfReader ---------------------------------------------------------------

int mvf_reader_open(const char* path, mvf_reader** out) {
    if (!path || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    int fd = ::open(path, O_RDONLY);  // File::open, reader.rs:46
    if (fd < 0) return fail(MVF_ERR_IO, std::string("I/O error: ") + std::strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) {
        int e = errno;
        ::close(fd);
        return fail(MVF_ERR_IO, std::string("I/O error: ") + std::strerror(e));
    }
    if (st.st_size == 0) {  // memmap2 refuses zero-length maps
        ::close(fd);
        return fail(MVF_ERR_IO, "I/O error: memory map must have a non-zero length");
    }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);  // Mmap::map, reader.rs:47
    int e = errno;
    ::close(fd);
    if (m == MAP_FAILED) return fail(MVF_ERR_IO, std::string("I/O error: ") + std::strerror(e));
    auto* r = new mvf_reader();
    r->data = static_cast<const uint8_t*>(m);
    r->len = (size_t)st.st_size;
    r->mapped = true;
    int rc = open_common(r);
    if (rc) {
        mvf_reader_close(r);
        return rc;
    }
    *out = r;
    return MVF_OK;
}

int mvf_reader_open_bytes(const void* bytes, uint64_t len, mvf_reader** out) {
    if ((!bytes && len) || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    auto* r = new mvf_reader();
    r->owned.assign(static_cast<const uint8_t*>(bytes), static_cast<const uint8_t*>(bytes) + len);
    r->data = r->owned.data();
    r->len = (size_t)len;
    int rc = open_common(r);
    if (rc) {
        mvf_reader_close(r);
        return rc;
    }
    *out = r;
    return MVF_OK;
}

void mvf_reader_close(mvf_reader* r) {
    if (!r) return;
    if (r->mapped && r->data) munmap(const_cast<uint8_t*>(r->data), r->len);
    delete r;
}

int mvf_reader_version(const mvf_reader* r, uint16_t* out) {
    if (!r || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = r->version;
    return MVF_OK;
}

int mvf_reader_num_vector_spaces(const mvf_reader* r, uint64_t* out) {
    if (!r || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = r->spaces.size();
    return MVF_OK;
}

int mvf_reader_vector_space_name(const mvf_reader* r, uint64_t i, const char** name, uint32_t* len) {
    if (!r || !name || !len) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    if (i >= r->spaces.size())
        return fail(MVF_ERR_INDEX_OUT_OF_BOUNDS, "Index out of bounds: " + std::to_string(i) + " >= " + std::to_string(r->spaces.size()));
    *name = r->spaces[i].name.c_str();
    *len = (uint32_t)r->spaces[i].name.size();
    return MVF_OK;
}

static void fill_space(const mvf_reader* r, size_t i, mvf_vector_space* out) {
    const SpaceRec& s = r->spaces[i];
    out->reader = r;
    out->index = (uint32_t)i;
    out->name = s.name.c_str();
    out->name_len = (uint32_t)s.name.size();
    out->dimension = s.dimension;
    out->total_vectors = s.total_vectors;
    out->vector_type = s.vector_type;
    out->distance_metric = s.distance_metric;
    out->data_type = s.data_type;
    out->index_type = s.index_type;
    out->vectors_block_index = s.vectors_block_index;
    out->vector_ids_block_index = s.vector_ids_block_index;
    out->has_sparse_metadata = s.sparse;
    out->has_tombstones = s.tombstones;
    out->tombstone_format = s.tomb_format;
    out->tombstone_block_index = s.tomb_block_index;
    out->tombstone_deleted_count = s.tomb_deleted_count;
}

int mvf_reader_vector_space(const mvf_reader* r, const char* name, mvf_vector_space* out) {
    if (!r || !name || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    for (size_t i = 0; i < r->spaces.size(); i++)  // linear scan by name, reader.rs:107-111
        if (r->spaces[i].name == name) {
            fill_space(r, i, out);
            return MVF_OK;
        }
    return fail(MVF_ERR_SPACE_NOT_FOUND, std::string("Vector space '") + name + "' not found");
}

int mvf_reader_vector_space_at(const mvf_reader* r, uint64_t i, mvf_vector_space* out) {
    if (!r || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    if (i >= r->spaces.size())
        return fail(MVF_ERR_INDEX_OUT_OF_BOUNDS, "Index out of bounds: " + std::to_string(i) + " >= " + std::to_string(r->spaces.size()));
    fill_space(r, (size_t)i, out);
    return MVF_OK;
}

int mvf_reader_file_size(const mvf_reader* r, uint64_t* out) {
    if (!r || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = r->len;
    return MVF_OK;
}

int mvf_reader_has_metadata(const mvf_reader* r, int* out) {
    if (!r || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = r->has_metadata ? 1 : 0;
    return MVF_OK;
}

int mvf_reader_num_metadata_columns(const mvf_reader* r, uint64_t* out) {
    if (!r || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = r->metadata_names.size();
    return MVF_OK;
}

int mvf_reader_metadata_column_name(const mvf_reader* r, uint64_t i, const char** name, uint32_t* len) {
    if (!r || !name || !len) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    if (i >= r->metadata_names.size())
        return fail(MVF_ERR_INDEX_OUT_OF_BOUNDS, "Index out of bounds: " + std::to_string(i) + " >= " + std::to_string(r->metadata_names.size()));
    *name = r->metadata_names[i].c_str();
    *len = (uint32_t)r->metadata_names[i].size();
    return MVF_OK;
}

int mvf_reader_num_blocks(const mvf_reader* r, uint64_t* out) {
    if (!r || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = r->blocks.size();
    return MVF_OK;
}

int mvf_reader_block(const mvf_reader* r, uint64_t i, mvf_data_block* out) {
    if (!r || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    if (i >= r->blocks.size())
        return fail(MVF_ERR_INDEX_OUT_OF_BOUNDS, "Index out of bounds: " + std::to_string(i) + " >= " + std::to_string(r->blocks.size()));
    *out = r->blocks[i];
    return MVF_OK;
}

int mvf_reader_validate(const mvf_reader* r) {  // reader.rs:149-162
    if (!r) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    for (const auto& b : r->blocks) {
        uint64_t end = b.offset + b.size;
        if (end > r->len || end < b.offset)
            return fail(MVF_ERR_CORRUPTED_DATA, "Block extends beyond file: offset=" + std::to_string(b.offset) +
                                                    ", size=" + std::to_string(b.size) + ", file_size=" + std::to_string(r->len));
    }
    return MVF_OK;
}

int mvf_reader_validate_with_checksum(const mvf_reader* r) {
    int rc = mvf_reader_validate(r);
    if (rc) return rc;
    for (const auto& b : r->blocks) {
        if (b.checksum == 0) continue;  // reader.rs:184
        uint64_t start = 4 + b.offset, end = start + b.size;  // the bytes builder.rs:251 hashed
        if (end > r->len || end < start)
            return fail(MVF_ERR_CORRUPTED_DATA, "Invalid block range after adjustment: " + std::to_string(start) + ".." + std::to_string(end));
        uint32_t got = crc32_ieee(r->data + start, b.size);
        if (got != b.checksum)
            return fail(MVF_ERR_CORRUPTED_DATA, "Block checksum mismatch: expected " + std::to_string(b.checksum) + ", got " + std::to_string(got));
    }
    return MVF_OK;
}

// ---- VectorSpace / Vector -------------------------------------------------------

// shared prologue of get_vector / map_vector_range: the block bytes
// One block of the manifest as bytes of the mapping.  Compressed blocks are refused: the reference's builder only ever
// writes CompressionAlgorithm::None (src/builder.rs:249) and nothing in the reference can read LZ4/Zstd blocks
// (schema/types.fbs:28-32 is an enum without code) -- scanning the compressed bytes as rows would be silent garbage.
static int manifest_block(const mvf_reader* r, uint32_t index, const char* what, const uint8_t** block, uint64_t* block_len) {
    if (index >= r->blocks.size())
        return fail(MVF_ERR_CORRUPTED_DATA, std::string("Invalid ") + what + " block index");  // vector_space.rs:110-112
    const mvf_data_block& b = r->blocks[index];
    if (b.compression != 0)
        return fail(MVF_ERR_BUILD, std::string("Unsupported compression algorithm ") + std::to_string(b.compression) + " on the " +
                                       what + " block (only CompressionAlgorithm::None is implemented)");
    uint64_t start = 4 + b.offset;  // METRO_MAGIC.len() + block.offset, vector_space.rs:118-119
    if (start < b.offset || start + b.size > r->len || start + b.size < start)
        return fail(MVF_ERR_CORRUPTED_DATA, std::string(what) + " block extends beyond file");  // the reference would panic on the slice
    *block = r->data + start;
    *block_len = b.size;
    return MVF_OK;
}

static int space_block(const mvf_vector_space* s, const uint8_t** block, uint64_t* block_len) {
    int rc = manifest_block(s->reader, s->vectors_block_index, "vector", block, block_len);
    if (rc == MVF_ERR_CORRUPTED_DATA && s->vectors_block_index < s->reader->blocks.size())
        return fail(MVF_ERR_CORRUPTED_DATA, "Vector block extends beyond file");
    return rc;
}

int mvf_space_get_vector(const mvf_vector_space* s, uint64_t index, const void** data, uint64_t* len) {
    if (!s || !s->reader || !data || !len) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    if (index >= s->total_vectors)  // vector_space.rs:102-107
        return fail(MVF_ERR_INDEX_OUT_OF_BOUNDS, "Index out of bounds: " + std::to_string(index) + " >= " + std::to_string(s->total_vectors));
    const uint8_t* block;
    uint64_t block_len;
    int rc = space_block(s, &block, &block_len);
    if (rc) return rc;
    uint32_t es = elem_size(s->data_type);
    if (!es) return fail(MVF_ERR_BUILD, "Unsupported vector data type");  // :126
    // dimension and total_vectors come from the (untrusted) footer: no product may wrap
    uint64_t vector_size = (uint64_t)s->dimension * es, vector_offset = 0;
    if (__builtin_mul_overflow(index, vector_size, &vector_offset) || vector_offset > block_len ||
        vector_size > block_len - vector_offset)  // :132-137
        return fail(MVF_ERR_INDEX_OUT_OF_BOUNDS, "Index out of bounds: " + std::to_string(index) + " >= " +
                                                      std::to_string(vector_size ? block_len / vector_size : 0));
    *data = block + vector_offset;
    *len = vector_size;
    return MVF_OK;
}

int mvf_space_map_vector_range(const mvf_vector_space* s, uint64_t start, uint64_t count, mvf_vector_slice* out) {
    if (!s || !s->reader || !out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    if (start + count > s->total_vectors || start + count < start)  // vector_space.rs:156-161
        return fail(MVF_ERR_INDEX_OUT_OF_BOUNDS, "Index out of bounds: " + std::to_string(start + count) + " >= " + std::to_string(s->total_vectors));
    const uint8_t* block;
    uint64_t block_len;
    int rc = space_block(s, &block, &block_len);
    if (rc) return rc;
    uint32_t es = elem_size(s->data_type);
    if (!es) return fail(MVF_ERR_BUILD, "Unsupported vector data type");  // :174
    uint64_t vector_size = (uint64_t)s->dimension * es;
    uint64_t range_offset = 0, range_size = 0;  // footer fields are untrusted: no product may wrap
    if (__builtin_mul_overflow(start, vector_size, &range_offset) || __builtin_mul_overflow(count, vector_size, &range_size) ||
        range_offset > block_len || range_size > block_len - range_offset)
        return fail(MVF_ERR_CORRUPTED_DATA, "Vector range out of bounds");  // :181-183
    out->data = block + range_offset;
    out->stride = vector_size;
    out->count = count;
    out->data_type = s->data_type;
    return MVF_OK;
}

// ---- vector ids and tombstones (schema/core.fbs:35-39, :54, :56) ------------------------------------------------
// The reference parses neither (its builder never writes them: src/builder.rs:483-485), so the layout is defined
// here in the schema's words: the id block is one u64 LE per row ("0 = use positions as IDs": block 0 is always the
// first space's vectors, so 0 can mean "none"); a Bitmap tombstone block holds bit r of byte r >> 3 (LSB first) per
// row POSITION; a SortedList block holds ascending u64 LE "deleted IDs" -- vector ids when the space has an id block,
// positions otherwise.
int mvf_space_vector_ids(const mvf_vector_space* s, const void** ids_le, uint64_t* count) {
    if (!s || !s->reader || !ids_le || !count) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    *ids_le = nullptr;
    *count = 0;
    if (s->vector_ids_block_index == 0) return MVF_OK;
    const uint8_t* block;
    uint64_t len;
    int rc = manifest_block(s->reader, s->vector_ids_block_index, "vector id", &block, &len);
    if (rc) return rc;
    if (len / 8 < s->total_vectors) return fail(MVF_ERR_CORRUPTED_DATA, "Vector id block shorter than total_vectors");
    *ids_le = block;
    *count = s->total_vectors;
    return MVF_OK;
}

int mvf_space_tombstones(const mvf_vector_space* s, uint8_t* format, const void** data, uint64_t* size, uint64_t* deleted_count) {
    if (!s || !s->reader || !format || !data || !size) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    *format = 0;
    *data = nullptr;
    *size = 0;
    if (deleted_count) *deleted_count = 0;
    if (!s->has_tombstones || s->tombstone_format == 0 || s->tombstone_block_index == 0) return MVF_OK;  // "0 if no deletions"
    if (s->tombstone_format > 2) return fail(MVF_ERR_BUILD, "Unsupported tombstone format " + std::to_string(s->tombstone_format));
    const uint8_t* block;
    uint64_t len;
    int rc = manifest_block(s->reader, s->tombstone_block_index, "tombstone", &block, &len);
    if (rc) return rc;
    *format = s->tombstone_format;
    *data = block;
    *size = len;
    if (deleted_count) *deleted_count = s->tombstone_deleted_count;
    return MVF_OK;
}

int mvf_space_tombstone_bitmap(const mvf_vector_space* s, uint8_t* bitmap, uint64_t nbytes, uint64_t* deleted) {
    if (!s || !s->reader || (!bitmap && nbytes)) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    const uint64_t n = s->total_vectors, need = (n + 7) / 8;
    if (nbytes < need) return fail(MVF_ERR_INVALID_ARGUMENT, "bitmap buffer too small");
    std::memset(bitmap, 0, nbytes);
    if (deleted) *deleted = 0;
    uint8_t fmt;
    const void* data;
    uint64_t size;
    int rc = mvf_space_tombstones(s, &fmt, &data, &size, nullptr);
    if (rc || fmt == 0) return rc;
    const uint8_t* p = static_cast<const uint8_t*>(data);
    uint64_t cnt = 0;
    if (fmt == 1) {  // Bitmap: bit per vector
        if (size < need) return fail(MVF_ERR_CORRUPTED_DATA, "Tombstone bitmap shorter than total_vectors");
        std::memcpy(bitmap, p, need);
        if (n % 8) bitmap[need - 1] &= (uint8_t)((1u << (n % 8)) - 1);  // bits past the last row do not count
        for (uint64_t i = 0; i < need; i++) cnt += (uint64_t)__builtin_popcount(bitmap[i]);
    } else {  // SortedList of deleted ids
        const void* ids_le;
        uint64_t nids;
        rc = mvf_space_vector_ids(s, &ids_le, &nids);
        if (rc) return rc;
        std::vector<std::pair<uint64_t, uint64_t>> by_id;  // (id, position), only when an id block exists
        if (ids_le) {
            by_id.reserve(n);
            for (uint64_t r = 0; r < n; r++) by_id.emplace_back(rd64(static_cast<const uint8_t*>(ids_le) + 8 * r), r);
            std::sort(by_id.begin(), by_id.end());
        }
        uint64_t prev = 0;
        for (uint64_t i = 0; i + 8 <= size; i += 8) {
            const uint64_t id = rd64(p + i);
            if (i && id < prev) return fail(MVF_ERR_CORRUPTED_DATA, "Tombstone list is not sorted");
            prev = id;
            uint64_t pos = id;
            if (ids_le) {
                auto it = std::lower_bound(by_id.begin(), by_id.end(), std::make_pair(id, (uint64_t)0));
                if (it == by_id.end() || it->first != id) continue;  // deleting an id the space does not hold: no-op
                pos = it->second;
            } else if (id >= n) {
                continue;
            }
            if (!(bitmap[pos >> 3] & (1u << (pos & 7)))) cnt++;
            bitmap[pos >> 3] |= (uint8_t)(1u << (pos & 7));
        }
    }
    if (deleted) *deleted = cnt;
    return MVF_OK;
}

int mvf_vector_as_f32(const void* data, uint64_t len, uint8_t data_type, float* out, uint64_t cap, uint64_t* n_out) {
    if ((!data && len) || !n_out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    const uint8_t* p = static_cast<const uint8_t*>(data);
    uint64_t n;
    if (data_type == MVF_DTYPE_FLOAT32) n = len / 4;  // chunks_exact(4), vector.rs:75
    else if (data_type == MVF_DTYPE_FLOAT16) n = len / 2;  // chunks_exact(2), vector.rs:83
    else return fail(MVF_ERR_BUILD, "Cannot convert to f32");  // vector.rs:90
    *n_out = n;
    if (!out) return MVF_OK;
    if (cap < n) return fail(MVF_ERR_INVALID_ARGUMENT, "output buffer too small");
    for (uint64_t j = 0; j < n; j++) {
        if (data_type == MVF_DTYPE_FLOAT32) {
            uint32_t b = rd32(p + 4 * j);
            std::memcpy(&out[j], &b, 4);
        } else {
            out[j] = f16_to_f32(rd16(p + 2 * j));
        }
    }
    return MVF_OK;
}

// ---- MvfBuilder --------------------------------------------------------------------

int mvf_builder_new(mvf_builder** out) {
    if (!out) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = new mvf_builder();
    return MVF_OK;
}

void mvf_builder_free(mvf_builder* b) { delete b; }

int mvf_builder_add_vector_space(mvf_builder* b, const char* name, uint32_t dimension, uint8_t vector_type,
                                 uint8_t distance_metric, uint8_t data_type, uint64_t* index_out) {
    if (!b || !name) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    b->spaces.push_back(SpaceB{name, dimension, vector_type, distance_metric, data_type, {}, {}, {}, 0, 0});
    if (index_out) *index_out = b->spaces.size() - 1;
    return MVF_OK;
}

static SpaceB* find_space(mvf_builder* b, const char* name) {
    for (auto& s : b->spaces)
        if (s.name == name) return &s;
    return nullptr;
}

int mvf_builder_add_vectors_f32(mvf_builder* b, const char* space_name, const float* values, uint64_t n_vectors,
                                uint32_t dimension) {
    if (!b || !space_name || (!values && n_vectors)) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    SpaceB* s = find_space(b, space_name);
    if (!s) return fail(MVF_ERR_SPACE_NOT_FOUND, std::string("Vector space '") + space_name + "' not found");  // builder.rs:155-159
    if (n_vectors == 0) return MVF_OK;  // :161-163
    if (s->dimension == 0) s->dimension = dimension;  // :166-167
    else if (s->dimension != dimension)               // :168-173
        return fail(MVF_ERR_DIMENSION_MISMATCH, "Dimension mismatch: expected " + std::to_string(s->dimension) + ", got " + std::to_string(dimension));
    const uint64_t n = n_vectors * dimension;
    if (s->data_type == MVF_DTYPE_FLOAT32) {  // :176-183
        size_t o = s->vectors.size();
        s->vectors.resize(o + n * 4);
        std::memcpy(s->vectors.data() + o, values, n * 4);
    } else if (s->data_type == MVF_DTYPE_FLOAT16) {  // :184-191
        size_t o = s->vectors.size();
        s->vectors.resize(o + n * 2);
        for (uint64_t i = 0; i < n; i++) {
            uint16_t h = f32_to_f16(values[i]);
            s->vectors[o + 2 * i] = (uint8_t)(h & 0xFF);
            s->vectors[o + 2 * i + 1] = (uint8_t)(h >> 8);
        }
    } else {
        return fail(MVF_ERR_BUILD, "Unsupported data type for vectors");  // :192
    }
    return MVF_OK;
}

int mvf_builder_add_vectors_raw(mvf_builder* b, const char* space_name, const void* rows, uint64_t n_vectors,
                                uint32_t dimension) {
    if (!b || !space_name || (!rows && n_vectors)) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    SpaceB* s = find_space(b, space_name);
    if (!s) return fail(MVF_ERR_SPACE_NOT_FOUND, std::string("Vector space '") + space_name + "' not found");
    if (n_vectors == 0) return MVF_OK;
    uint32_t es = elem_size(s->data_type);
    if (!es) return fail(MVF_ERR_BUILD, "Unsupported data type for vectors");
    if (s->dimension == 0) s->dimension = dimension;
    else if (s->dimension != dimension)
        return fail(MVF_ERR_DIMENSION_MISMATCH, "Dimension mismatch: expected " + std::to_string(s->dimension) + ", got " + std::to_string(dimension));
    const uint8_t* p = static_cast<const uint8_t*>(rows);
    uint64_t nbytes = 0;
    if (__builtin_mul_overflow(n_vectors, (uint64_t)dimension * es, &nbytes)) return fail(MVF_ERR_INVALID_ARGUMENT, "n_vectors * dimension overflows");
    s->vectors.insert(s->vectors.end(), p, p + nbytes);
    return MVF_OK;
}

// EXTENSION: room for n_vectors rows up front.  Vec<u8>-style doubling (the reference, builder.rs:176-191) re-copies and
// re-faults a multi-GB block several times on its way up; one reservation, with transparent huge pages asked for, does not.
int mvf_builder_reserve_vectors(mvf_builder* b, const char* space_name, uint64_t n_vectors) {
    if (!b || !space_name) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    SpaceB* s = find_space(b, space_name);
    if (!s) return fail(MVF_ERR_SPACE_NOT_FOUND, std::string("Vector space '") + space_name + "' not found");
    uint32_t es = elem_size(s->data_type);
    if (!es) return fail(MVF_ERR_BUILD, "Unsupported data type for vectors");
    uint64_t nbytes = 0;
    if (__builtin_mul_overflow(n_vectors, (uint64_t)s->dimension * es, &nbytes)) return fail(MVF_ERR_INVALID_ARGUMENT, "n_vectors * dimension overflows");
    try {
        s->vectors.reserve((size_t)nbytes);
    } catch (const std::exception&) {
        return fail(MVF_ERR_IO, "I/O error: out of memory");
    }
    if (nbytes >= ((uint64_t)64 << 20)) {
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(s->vectors.data()) + ((1u << 21) - 1)) & ~(uintptr_t)((1u << 21) - 1);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(s->vectors.data()) + s->vectors.capacity()) & ~(uintptr_t)((1u << 21) - 1);
        if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);  // advice only
    }
    return MVF_OK;
}

// EXTENSION (the reference's builder carries `vector_ids` / `tombstones` fields but has no public setter,
// src/builder.rs:61-63): attach an id per row / a deletion block to a space.  Layouts: mvf_space_vector_ids above.
int mvf_builder_set_vector_ids(mvf_builder* b, const char* space_name, const uint64_t* ids, uint64_t n) {
    if (!b || !space_name || (!ids && n)) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    SpaceB* s = find_space(b, space_name);
    if (!s) return fail(MVF_ERR_SPACE_NOT_FOUND, std::string("Vector space '") + space_name + "' not found");
    s->ids.resize(n * 8);
    for (uint64_t i = 0; i < n; i++)
        for (int k = 0; k < 8; k++) s->ids[i * 8 + k] = (uint8_t)(ids[i] >> (8 * k));  // to_le_bytes, builder.rs:258-261
    return MVF_OK;
}

int mvf_builder_set_tombstones(mvf_builder* b, const char* space_name, uint8_t format, const void* data, uint64_t len,
                               uint64_t deleted_count) {
    if (!b || !space_name || (!data && len)) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    if (format > 2) return fail(MVF_ERR_BUILD, "Unsupported tombstone format " + std::to_string(format));
    SpaceB* s = find_space(b, space_name);
    if (!s) return fail(MVF_ERR_SPACE_NOT_FOUND, std::string("Vector space '") + space_name + "' not found");
    const uint8_t* p = static_cast<const uint8_t*>(data);
    s->tombstones.assign(p, p + len);
    s->tomb_format = format;
    s->tomb_deleted = deleted_count;
    return MVF_OK;
}

int mvf_builder_add_metadata_column(mvf_builder* b, const char* name, uint8_t data_type, const void* bytes, uint64_t len) {
    if (!b || !name || (!bytes && len)) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    const uint8_t* p = static_cast<const uint8_t*>(bytes);
    b->columns.push_back(ColumnB{name, data_type, std::vector<uint8_t>(p, p + len)});
    return MVF_OK;
}

struct Blk {
    uint64_t offset, size;
    uint32_t crc;
    const std::vector<uint8_t>* bytes;
};

// build() + the footer half of to_bytes(): the block layout (one DataBlock per space, then per metadata column,
// builder.rs:241-285), every block's CRC and the finished FlatBuffers footer (builder.rs:427-547).  `cur` = bytes of
// the data section.
static void layout_image(const mvf_builder* b, uint32_t quirks, std::vector<Blk>& blks, uint64_t& cur, std::vector<uint8_t>& footer) {
    cur = 0;
    std::vector<uint32_t> vec_blk(b->spaces.size(), 0), ids_blk(b->spaces.size(), 0), tomb_blk(b->spaces.size(), 0);
    for (size_t i = 0; i < b->spaces.size(); i++) {
        const auto& s = b->spaces[i];
        vec_blk[i] = (uint32_t)blks.size();
        blks.push_back({cur, s.vectors.size(), crc32_ieee(s.vectors.data(), s.vectors.size()), &s.vectors});
        cur += s.vectors.size();
        if (!s.ids.empty()) {  // right behind the space's vectors, builder.rs:257-271
            ids_blk[i] = (uint32_t)blks.size();
            blks.push_back({cur, s.ids.size(), crc32_ieee(s.ids.data(), s.ids.size()), &s.ids});
            cur += s.ids.size();
        }
        if (s.tomb_format != 0 && !s.tombstones.empty()) {
            tomb_blk[i] = (uint32_t)blks.size();
            blks.push_back({cur, s.tombstones.size(), crc32_ieee(s.tombstones.data(), s.tombstones.size()), &s.tombstones});
            cur += s.tombstones.size();
        }
    }
    const uint32_t first_column_blk = (uint32_t)blks.size();
    for (const auto& c : b->columns) {
        blks.push_back({cur, c.data.size(), crc32_ieee(c.data.data(), c.data.size()), &c.data});
        cur += c.data.size();
    }

    // to_bytes(): footer (builder.rs:427-547)
    FbBuilder fb;
    std::vector<uint32_t> space_offs;
    for (size_t i = 0; i < b->spaces.size(); i++) {
        const SpaceB& s = b->spaces[i];
        uint32_t name = fb.create_string(s.name);
        fb.start_table(0);  // FlatIndex {} (builder.rs:465-468; union tag 1, schema/index.fbs:6-11)
        uint32_t flat = fb.end_table();
        uint64_t denom = (uint64_t)s.dimension * ((quirks & MVF_QUIRK_TOTAL_VECTORS_DIV4) ? 4u : elem_size(s.data_type));
        uint64_t total = denom ? s.vectors.size() / denom : 0;  // builder.rs:476 divides by dimension*4
        uint32_t tomb = 0;
        if (tomb_blk[i]) {
            fb.start_table(3);  // TombstoneInfo, schema/core.fbs:35-39
            fb.add_scalar<uint64_t>(2, s.tomb_deleted, 0);
            fb.add_scalar<uint32_t>(1, tomb_blk[i], 0);
            fb.add_scalar<uint8_t>(0, s.tomb_format, 0);
            tomb = fb.end_table();
        }
        fb.start_table(12);
        fb.add_scalar<uint64_t>(2, total, 0);
        fb.add_offset(11, tomb);
        fb.add_offset(8, flat);
        fb.add_scalar<uint32_t>(9, ids_blk[i], 0);
        // vectors_block_index: the reference writes the space ORDINAL (builder.rs:480), which is the block's index only
        // while no id / tombstone blocks exist -- all it can produce; with them the real index is the only readable one
        fb.add_scalar<uint32_t>(6, vec_blk[i], 0);
        fb.add_scalar<uint32_t>(1, s.dimension, 0);
        fb.add_offset(0, name);
        fb.add_scalar<uint8_t>(7, 1, 0);  // index_type_type = FlatIndex
        fb.add_scalar<uint8_t>(5, s.data_type, 0);
        fb.add_scalar<uint8_t>(4, s.distance_metric, 0);
        fb.add_scalar<uint8_t>(3, s.vector_type, 0);
        space_offs.push_back(fb.end_table());
    }
    uint32_t spaces_vec = fb.create_offset_vector(space_offs);

    std::vector<uint8_t> raw(blks.size() * 40, 0);  // DataBlock structs, schema/core.fbs:7-13
    for (size_t i = 0; i < blks.size(); i++) {
        uint8_t* p = raw.data() + i * 40;
        std::memcpy(p, &blks[i].offset, 8);
        std::memcpy(p + 8, &blks[i].size, 8);
        p[16] = 0;  // CompressionAlgorithm::None
        uint64_t z = 0;
        std::memcpy(p + 24, &z, 8);
        std::memcpy(p + 32, &blks[i].crc, 4);
    }
    uint32_t blocks_vec = fb.create_struct_vector(raw.data(), 40, blks.size(), 8);

    uint32_t meta_vec = 0;
    if (!b->columns.empty()) {
        std::vector<uint32_t> col_offs;
        for (size_t i = 0; i < b->columns.size(); i++) {
            uint32_t name = fb.create_string(b->columns[i].name);
            fb.start_table(6);  // MetadataColumn, schema/core.fbs:16-25
            fb.add_scalar<uint64_t>(3, 0, 0);
            fb.add_scalar<uint32_t>(2, first_column_blk + (uint32_t)i, 0);  // builder.rs:512 (spaces.len() + i: the same without id blocks)
            fb.add_offset(0, name);
            fb.add_scalar<uint8_t>(1, b->columns[i].data_type, 0);
            col_offs.push_back(fb.end_table());
        }
        meta_vec = fb.create_offset_vector(col_offs);
    }

    fb.start_table(8);  // FileFooter, schema/mvf.fbs:12-30
    fb.add_offset(3, meta_vec);
    fb.add_offset(2, blocks_vec);
    fb.add_offset(1, spaces_vec);
    fb.add_scalar<uint16_t>(6, 1, 3);  // compatibility_version: 1 (schema default 3), builder.rs:540
    fb.add_scalar<uint16_t>(0, 1, 3);  // format_version: 1 (schema default 3), builder.rs:531
    uint32_t root = fb.end_table();
    footer = fb.finish_minimal(root);
}


int mvf_builder_to_bytes(const mvf_builder* b, uint32_t quirks, uint8_t** out, uint64_t* len) {
    if (!b || !out || !len) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    std::vector<Blk> blks;
    std::vector<uint8_t> footer;
    uint64_t cur = 0;
    layout_image(b, quirks, blks, cur, footer);
    // MVF1 | blocks | footer | u32 footer_len | MVF1   (builder.rs:417-557)
    const uint64_t total_len = 4 + cur + footer.size() + 4 + 4;
    uint8_t* img = static_cast<uint8_t*>(std::malloc(total_len));
    if (!img) return fail(MVF_ERR_IO, "I/O error: out of memory");
    uint8_t* w = img;
    std::memcpy(w, kMagic, 4);
    w += 4;
    for (const auto& k : blks) {
        if (k.size) std::memcpy(w, k.bytes->data(), k.size);
        w += k.size;
    }
    std::memcpy(w, footer.data(), footer.size());
    w += footer.size();
    uint32_t fl = (uint32_t)footer.size();
    std::memcpy(w, &fl, 4);
    w += 4;
    std::memcpy(w, kMagic, 4);
    *out = img;
    *len = total_len;
    return MVF_OK;
}

// BuiltMvf::save (builder.rs:408-411 -> MvfWriter, io.rs:29-46).  The reference assembles the whole image in memory
// (to_bytes) and writes it; here the blocks go to the file as they lie in the builder -- the same bytes without a
// second copy of a multi-GB vector block.
int mvf_builder_save(const mvf_builder* b, const char* path, uint32_t quirks) {
    if (!b || !path) return fail(MVF_ERR_INVALID_ARGUMENT, "NULL argument");
    std::vector<Blk> blks;
    std::vector<uint8_t> footer;
    uint64_t cur = 0;
    layout_image(b, quirks, blks, cur, footer);
    FILE* f = std::fopen(path, "wb");  // MvfWriter::create, io.rs:29-35
    if (!f) return fail(MVF_ERR_IO, std::string("I/O error: ") + std::strerror(errno));
    bool ok = std::fwrite(kMagic, 1, 4, f) == 4;
    int e = ok ? 0 : errno;
    for (size_t i = 0; ok && i < blks.size(); i++) {
        const uint8_t* p = blks[i].bytes->data();
        for (uint64_t o = 0; ok && o < blks[i].size;) {  // 1-GiB pieces: a single fwrite of > 2 GiB is not portable
            const size_t piece = (size_t)std::min<uint64_t>(blks[i].size - o, (uint64_t)1 << 30);
            ok = std::fwrite(p + o, 1, piece, f) == piece;
            if (!ok) e = errno;
            o += piece;
        }
    }
    const uint32_t fl = (uint32_t)footer.size();
    if (ok) {
        ok = std::fwrite(footer.data(), 1, footer.size(), f) == footer.size() && std::fwrite(&fl, 1, 4, f) == 4 &&
             std::fwrite(kMagic, 1, 4, f) == 4;
        if (!ok) e = errno;
    }
    if (std::fclose(f) != 0 && ok) {
        ok = false;
        e = errno;
    }
    if (!ok) return fail(MVF_ERR_IO, std::string("I/O error: ") + std::strerror(e));
    return MVF_OK;
}

}  // extern "C"
