//! Rust binding of `libmvf_gpu.so` (include/mvf_gpu.h) for the `metrovector` crate.
//!
//! UNVERIFIED: this file was written without a Rust toolchain (the build image has
//! none); it is the stub a maintainer would add, not tested code.
//!
//! Drop-in for `find_top_k_similar` (examples/similarity_search.rs:140-176).

use std::ffi::{c_char, c_int, c_void, CStr};

use metrovector::{
    errors::{MvfError, Result},
    mvf_fbs::{DataType, DistanceMetric},
    vectors::vector_space::VectorSpace,
};

#[repr(C)]
pub struct MvfGpuCorpus {
    _private: [u8; 0],
}

#[link(name = "mvf_gpu")]
extern "C" {
    fn mvfgpu_corpus_create(rows: *const c_void, n: u64, dimension: u32, data_type: u8, stride_bytes: u64,
                            device: c_int, index_base: u64, out: *mut *mut MvfGpuCorpus) -> c_int;
    fn mvfgpu_corpus_destroy(corpus: *mut MvfGpuCorpus);
    fn mvfgpu_search(corpus: *const MvfGpuCorpus, metric: u8, queries: *const c_void, query_dtype: u8,
                     query_dim: u32, nq: u32, k: u32, out_scores: *mut f32, out_indices: *mut u64,
                     out_raw: *mut i32) -> c_int;
    fn mvfgpu_last_error_message() -> *const c_char;
}

fn status_to_error(status: c_int) -> MvfError {
    let msg = unsafe { CStr::from_ptr(mvfgpu_last_error_message()) }.to_string_lossy().into_owned();
    match status {
        2 => MvfError::InvalidFormat(msg),
        4 => MvfError::VectorSpaceNotFound(msg),
        5 => MvfError::IndexOutOfBounds { index: 0, len: 0 },
        6 => MvfError::DimensionMismatch { expected: 0, actual: 0 },
        8 => MvfError::CorruptedData(msg),
        9 => MvfError::Extension(msg),
        _ => MvfError::Build(msg), // 10 Build, 11 Device, 12 InvalidArgument
    }
}

fn elem_size(dt: DataType) -> Result<usize> {
    match dt {
        DataType::Float32 => Ok(4),
        DataType::Float16 => Ok(2),
        DataType::Int8 | DataType::UInt8 => Ok(1),
        _ => Err(MvfError::build_error("Unsupported vector data type")),
    }
}

/// One vector space resident in HBM on one MI355X.
pub struct GpuCorpus {
    handle: *mut MvfGpuCorpus,
    dimension: u32,
}

unsafe impl Send for GpuCorpus {}
unsafe impl Sync for GpuCorpus {}

impl GpuCorpus {
    /// Uploads the whole space: `map_vector_range(0, total)` is the hand-off
    /// (src/vectors/vector_space.rs:155-188, src/vectors/mem.rs:75-77).
    pub fn from_space(space: &VectorSpace, device: i32) -> Result<Self> {
        let total = space.total_vectors();
        let slice = space.map_vector_range(0, total)?;
        let stride = space.dimension() as u64 * elem_size(space.data_type())? as u64;
        let mut handle = std::ptr::null_mut();
        let rc = unsafe {
            mvfgpu_corpus_create(slice.as_ptr::<u8>() as *const c_void, total, space.dimension(),
                                 space.data_type().0, stride, device, 0, &mut handle)
        };
        if rc != 0 {
            return Err(status_to_error(rc));
        }
        Ok(Self { handle, dimension: space.dimension() })
    }

    /// k best rows for one f32 query, best first: (index, score).
    pub fn search(&self, metric: DistanceMetric, query: &[f32], k: usize) -> Result<Vec<(u64, f32)>> {
        let mut scores = vec![0f32; k];
        let mut indices = vec![0u64; k];
        let rc = unsafe {
            mvfgpu_search(self.handle, metric.0, query.as_ptr() as *const c_void, DataType::Float32.0,
                          query.len() as u32, 1, k as u32, scores.as_mut_ptr(), indices.as_mut_ptr(),
                          std::ptr::null_mut())
        };
        if rc != 0 {
            return Err(status_to_error(rc));
        }
        let _ = self.dimension;
        Ok(indices.into_iter().zip(scores).take_while(|(i, _)| *i != u64::MAX).collect())
    }
}

impl Drop for GpuCorpus {
    fn drop(&mut self) {
        unsafe { mvfgpu_corpus_destroy(self.handle) }
    }
}

/// Same name and argument meaning as examples/similarity_search.rs:140-144.
pub fn find_top_k_similar(space: &VectorSpace, query: &[f32], k: usize) -> Result<Vec<(u64, f32)>> {
    GpuCorpus::from_space(space, 0)?.search(space.distance_metric(), query, k)
}
