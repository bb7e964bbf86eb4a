/*
 * mvf_file.h — C ABI of libmvf_host.so: a C++ reader/writer for MVF files
 * ("MVF1" | raw LE row-major blocks | FlatBuffers footer | u32 footer_len |
 * "MVF1"), the host side above the GPU boundary where the reference uses its
 * Rust reader.  It mirrors, call for call:
 *   MvfReader    reference src/reader.rs:27-289
 *   VectorSpace  reference src/vectors/vector_space.rs:34-318
 *   Vector       reference src/vectors/vector.rs:28-207 (as_f32 :71-92)
 *   MvfBuilder / BuiltMvf  reference src/builder.rs:44-559
 * Wire format: reference schema/{types,core,mvf,index}.fbs; where
 * schema/FORMAT.md disagrees with the code (offset base, DataBlock fields)
 * the code wins.  The footer is parsed / emitted by a
 * hand-written FlatBuffers codec (no flatc in this image).
 *
 * Every function returns enum mvf_status; mvf_last_error_message() gives the
 * detail text of the calling thread's last failure, phrased like the
 * reference's error strings.
 */
#ifndef MVF_FILE_H
#define MVF_FILE_H

#include <stddef.h>
#include <stdint.h>

#include "mvf_status.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvf_reader mvf_reader;
typedef struct mvf_builder mvf_builder;

/* One DataBlock of the manifest (schema/core.fbs:7-13; 40-byte struct). */
typedef struct mvf_data_block {
    uint64_t offset; /* relative to the data section (file offset 4) — src/builder.rs:243-255 */
    uint64_t size;
    uint8_t compression;
    uint64_t compressed_size;
    uint32_t checksum; /* CRC32 (IEEE) of the block bytes — src/builder.rs:251 */
} mvf_data_block;

/* A borrowed view of one vector space; valid while the reader is open.
 * (VectorSpace<'a>, src/vectors/vector_space.rs:34-39.) */
typedef struct mvf_vector_space {
    const mvf_reader* reader;
    uint32_t index;               /* ordinal in the footer */
    const char* name;             /* not NUL-terminated in general: use name_len */
    uint32_t name_len;
    uint32_t dimension;
    uint64_t total_vectors;
    uint8_t vector_type;          /* enum mvf_vector_type */
    uint8_t distance_metric;      /* enum mvf_distance_metric */
    uint8_t data_type;            /* enum mvf_data_type */
    uint8_t index_type;           /* union tag: 0 none, 1 Flat, 2 IVF, 3 HNSW, 4 Custom */
    uint32_t vectors_block_index;
    uint32_t vector_ids_block_index;
    uint8_t has_sparse_metadata;
    uint8_t has_tombstones;
    uint8_t tombstone_format;         /* TombstoneInfo (schema/core.fbs:35-39): 0 None, 1 Bitmap, 2 SortedList */
    uint32_t tombstone_block_index;   /* "0 if no deletions" */
    uint64_t tombstone_deleted_count;
} mvf_vector_space;

/* VectorSlice (src/vectors/mem.rs:24-30): what the GPU boundary consumes. */
typedef struct mvf_vector_slice {
    const void* data; /* as_ptr(), mem.rs:75-77 — points into the mmap */
    uint64_t stride;  /* bytes between rows = dimension * elem_size */
    uint64_t count;
    uint8_t data_type;
} mvf_vector_slice;

const char* mvf_last_error_message(void);
const char* mvf_strerror(int status);

/* ---- MvfReader ----------------------------------------------------------- */
int mvf_reader_open(const char* path, mvf_reader** out);             /* reader.rs:45-79 */
/* Same validation over an in-memory image (copied). */
int mvf_reader_open_bytes(const void* bytes, uint64_t len, mvf_reader** out);
void mvf_reader_close(mvf_reader* r);
int mvf_reader_version(const mvf_reader* r, uint16_t* out);           /* :82-84  */
int mvf_reader_num_vector_spaces(const mvf_reader* r, uint64_t* out); /* :87-89  */
int mvf_reader_vector_space_name(const mvf_reader* r, uint64_t i, const char** name, uint32_t* len); /* :92-98 */
int mvf_reader_vector_space(const mvf_reader* r, const char* name, mvf_vector_space* out);           /* :104-119 */
int mvf_reader_vector_space_at(const mvf_reader* r, uint64_t i, mvf_vector_space* out);
int mvf_reader_file_size(const mvf_reader* r, uint64_t* out);         /* :122-124 */
int mvf_reader_has_metadata(const mvf_reader* r, int* out);           /* :127-129 */
int mvf_reader_num_metadata_columns(const mvf_reader* r, uint64_t* out);
int mvf_reader_metadata_column_name(const mvf_reader* r, uint64_t i, const char** name, uint32_t* len); /* :132-143 */
int mvf_reader_num_blocks(const mvf_reader* r, uint64_t* out);
int mvf_reader_block(const mvf_reader* r, uint64_t i, mvf_data_block* out);
int mvf_reader_validate(const mvf_reader* r);                         /* :149-162 */
/* What the reference leaves as todo!() (reader.rs:172-221): CRC32 of
 * mmap[4+offset .. 4+offset+size] against DataBlock.checksum. */
int mvf_reader_validate_with_checksum(const mvf_reader* r);

/* ---- VectorSpace / Vector ------------------------------------------------- */
/* get_vector, vector_space.rs:101-142: borrowed row bytes + their length. */
int mvf_space_get_vector(const mvf_vector_space* s, uint64_t index, const void** data, uint64_t* len);
/* map_vector_range, vector_space.rs:155-188. */
int mvf_space_map_vector_range(const mvf_vector_space* s, uint64_t start, uint64_t count, mvf_vector_slice* out);
/*
 * Vector ids and deletions (schema/core.fbs:54, :56, :35-39).  The reference neither writes nor reads them
 * (src/builder.rs:483-485: always 0 / None), so the block layouts are defined here in the schema's words:
 *   vector_ids block : one u64 LE per row; vector_ids_block_index 0 = "use positions as IDs";
 *   Bitmap           : bit (r & 7) of byte (r >> 3) set = the row at POSITION r is deleted;
 *   SortedList       : ascending u64 LE deleted ids -- vector ids when the space has an id block, else positions.
 * mvf_space_vector_ids returns a borrowed, possibly UNALIGNED pointer into the mapping (NULL when positions are
 * the ids); mvf_space_tombstone_bitmap expands either tombstone format into a position bitmap of
 * (total_vectors + 7) / 8 bytes (all zero without deletions) -- the form mvfgpu_corpus_set_tombstones consumes.
 * Compressed blocks (schema/types.fbs:28-32) are refused everywhere with MVF_ERR_BUILD: the reference has no codec.
 */
int mvf_space_vector_ids(const mvf_vector_space* s, const void** ids_le, uint64_t* count);
int mvf_space_tombstones(const mvf_vector_space* s, uint8_t* format, const void** data, uint64_t* size,
                         uint64_t* deleted_count);
int mvf_space_tombstone_bitmap(const mvf_vector_space* s, uint8_t* bitmap, uint64_t nbytes, uint64_t* deleted);
/* Vector::as_f32, vector.rs:71-92: decodes len/elem_size values into out
 * (capacity `cap` floats); Int8/UInt8/others -> MVF_ERR_BUILD "Cannot convert to f32". */
int mvf_vector_as_f32(const void* data, uint64_t len, uint8_t data_type, float* out, uint64_t cap, uint64_t* n_out);

/* ---- MvfBuilder / BuiltMvf -------------------------------------------------- */
int mvf_builder_new(mvf_builder** out);                                /* builder.rs:93-95 */
void mvf_builder_free(mvf_builder* b);
int mvf_builder_add_vector_space(mvf_builder* b, const char* name, uint32_t dimension, uint8_t vector_type,
                                 uint8_t distance_metric, uint8_t data_type, uint64_t* index_out); /* :113-135 */
/* add_vectors, builder.rs:151-196: f32 inputs encoded as Float32 (LE bits) or
 * Float16 (IEEE round-to-nearest-even, crate `half`); any other space dtype ->
 * MVF_ERR_BUILD "Unsupported data type for vectors" (:192). */
int mvf_builder_add_vectors_f32(mvf_builder* b, const char* space_name, const float* values, uint64_t n_vectors,
                                uint32_t dimension);
/* EXTENSION (no reference counterpart): append rows already in
 * the space's storage type — the only way to write Int8/UInt8 spaces. */
int mvf_builder_add_vectors_raw(mvf_builder* b, const char* space_name, const void* rows, uint64_t n_vectors,
                                uint32_t dimension);
/* EXTENSION: reserve room for n_vectors rows of the space (its dimension must be known) before appending a multi-GB
 * block in pieces: no re-copying on the way up, transparent huge pages requested. */
int mvf_builder_reserve_vectors(mvf_builder* b, const char* space_name, uint64_t n_vectors);
/* EXTENSION: the reference's builder has the fields (builder.rs:61-63) but no setter. */
int mvf_builder_set_vector_ids(mvf_builder* b, const char* space_name, const uint64_t* ids, uint64_t n);
int mvf_builder_set_tombstones(mvf_builder* b, const char* space_name, uint8_t format, const void* data, uint64_t len,
                               uint64_t deleted_count);
int mvf_builder_add_metadata_column(mvf_builder* b, const char* name, uint8_t data_type, const void* bytes,
                                    uint64_t len);                      /* :211-236 */
/*
 * build() + to_bytes(), builder.rs:241-308 and :417-558.
 * quirks bit 0 (MVF_QUIRK_TOTAL_VECTORS_DIV4): reproduce the reference's
 * total_vectors = bytes / (dimension*4) (builder.rs:476) instead
 * of the correct bytes / (dimension*elem_size).
 * The image is malloc'd; release with mvf_free.
 */
#define MVF_QUIRK_TOTAL_VECTORS_DIV4 1u
int mvf_builder_to_bytes(const mvf_builder* b, uint32_t quirks, uint8_t** out, uint64_t* len);
int mvf_builder_save(const mvf_builder* b, const char* path, uint32_t quirks); /* BuiltMvf::save :408-411, io.rs:29-46 */
void mvf_free(void* p);

/* helpers shared with tests */
uint32_t mvf_crc32(const void* data, uint64_t len); /* crc32fast::hash == CRC-32/ISO-HDLC */
uint16_t mvf_f32_to_f16(float f);                   /* half::f16::from_f32 */
float mvf_f16_to_f32(uint16_t h);                   /* half::f16::to_f32   */

#ifdef __cplusplus
}
#endif
#endif
