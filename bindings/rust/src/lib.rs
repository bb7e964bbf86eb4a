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
    /// Rows by GLOBAL index from HBM: the `ScoredVector.vector` payload (similarity_search.rs:18).
    fn mvfgpu_corpus_gather_rows(corpus: *const MvfGpuCorpus, indices: *const u64, count: u64,
                                 out_rows: *mut c_void) -> c_int;
    /// 0 automatic, 1 streaming kernel, 2 exact MFMA on the stored rows, 3 MFMA with the f16 shadow (the automatic
    /// choice for Float32 spaces), 4 additionally streams the f16 shadow for 1-2 queries (include/mvf_gpu.h).
    fn mvfgpu_set_scan_path(corpus: *mut MvfGpuCorpus, path: c_int) -> c_int;
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

    /// Batched search: `queries` holds `nq` rows of `dimension` f32 values; returns `nq` lists of up to `k` hits.
    /// Two or more queries on a large space take the MFMA path (include/mvf_gpu.h).
    pub fn search_batch(&self, metric: DistanceMetric, queries: &[f32], nq: usize, k: usize) -> Result<Vec<Vec<(u64, f32)>>> {
        let mut scores = vec![0f32; nq * k];
        let mut indices = vec![0u64; nq * k];
        let rc = unsafe {
            mvfgpu_search(self.handle, metric.0, queries.as_ptr() as *const c_void, DataType::Float32.0,
                          self.dimension, nq as u32, k as u32, scores.as_mut_ptr(), indices.as_mut_ptr(),
                          std::ptr::null_mut())
        };
        if rc != 0 {
            return Err(status_to_error(rc));
        }
        Ok((0..nq)
            .map(|q| (0..k).map(|j| (indices[q * k + j], scores[q * k + j])).take_while(|(i, _)| *i != u64::MAX).collect())
            .collect())
    }

    /// The rows behind a result list, fetched from HBM (f32 spaces): `ScoredVector.vector`.
    pub fn gather_rows_f32(&self, indices: &[u64]) -> Result<Vec<f32>> {
        let mut out = vec![0f32; indices.len() * self.dimension as usize];
        let rc = unsafe {
            mvfgpu_corpus_gather_rows(self.handle, indices.as_ptr(), indices.len() as u64, out.as_mut_ptr() as *mut c_void)
        };
        if rc != 0 {
            return Err(status_to_error(rc));
        }
        Ok(out)
    }

    /// See `mvfgpu_set_scan_path` in include/mvf_gpu.h.
    pub fn set_scan_path(&mut self, path: i32) -> Result<()> {
        let rc = unsafe { mvfgpu_set_scan_path(self.handle, path) };
        if rc != 0 {
            return Err(status_to_error(rc));
        }
        Ok(())
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
