/*
 * mvf_status.h — status codes shared by libmvf_gpu.so and libmvf_host.so.
 *
 * One code per MvfError variant of the reference, in declaration order
 * (reference src/errors.rs:8-40), so a Rust wrapper converts 1:1 into
 * Result<_, MvfError>; two codes are new (DEVICE, INVALID_ARGUMENT).
 * No panics, exceptions or errno cross the boundary.
 */
#ifndef MVF_STATUS_H
#define MVF_STATUS_H

#ifdef __cplusplus
extern "C" {
#endif

enum mvf_status {
    MVF_OK = 0,
    MVF_ERR_IO = 1,                  /* MvfError::Io                  src/errors.rs:9-10  */
    MVF_ERR_INVALID_FORMAT = 2,      /* MvfError::InvalidFormat       src/errors.rs:12-13 */
    MVF_ERR_UNSUPPORTED_VERSION = 3, /* MvfError::UnsupportedVersion  src/errors.rs:15-16 */
    MVF_ERR_SPACE_NOT_FOUND = 4,     /* MvfError::VectorSpaceNotFound src/errors.rs:18-19 */
    MVF_ERR_INDEX_OUT_OF_BOUNDS = 5, /* MvfError::IndexOutOfBounds    src/errors.rs:21-22 */
    MVF_ERR_DIMENSION_MISMATCH = 6,  /* MvfError::DimensionMismatch   src/errors.rs:24-25 */
    MVF_ERR_INVALID_VECTOR_TYPE = 7, /* MvfError::InvalidVectorType   src/errors.rs:27-31 */
    MVF_ERR_CORRUPTED_DATA = 8,      /* MvfError::CorruptedData       src/errors.rs:32-33 */
    MVF_ERR_EXTENSION = 9,           /* MvfError::Extension           src/errors.rs:35-36 */
    MVF_ERR_BUILD = 10,              /* MvfError::Build               src/errors.rs:38-39 */
    MVF_ERR_DEVICE = 11,             /* new: HIP / RCCL failure, or no GPU present        */
    MVF_ERR_INVALID_ARGUMENT = 12    /* new: NULL pointer, k == 0, unknown metric code …  */
};

/* schema/types.fbs:3-11 (DataType) */
enum mvf_data_type {
    MVF_DTYPE_FLOAT32 = 0,
    MVF_DTYPE_FLOAT16 = 1,
    MVF_DTYPE_INT8 = 2,
    MVF_DTYPE_UINT8 = 3,
    MVF_DTYPE_UINT32 = 4,
    MVF_DTYPE_UINT64 = 5,
    MVF_DTYPE_STRINGREF = 6
};

/* schema/types.fbs:14-17 (VectorType) */
enum mvf_vector_type { MVF_VECTOR_DENSE = 0, MVF_VECTOR_SPARSE = 1 };

/* schema/types.fbs:20-25 (DistanceMetric) */
enum mvf_distance_metric {
    MVF_METRIC_L2 = 0,
    MVF_METRIC_INNER_PRODUCT = 1,
    MVF_METRIC_COSINE = 2,
    MVF_METRIC_CUSTOM = 255
};

#ifdef __cplusplus
}
#endif
#endif
