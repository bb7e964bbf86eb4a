//! Rust binding of `libmvf_gpu.so` (include/mvf_gpu.h) for the `metrovector` crate.
//!
//! UNVERIFIED: this file was written without a Rust toolchain (the build image has
//! none); it is the stub a maintainer would add, not tested code.
//!
//! Drop-in for `find_top_k_similar` (examples/similarity_search.rs:140-176).

use std::ffi::{c_char, c_int, c_void, CStr};

use metrovector::{
    errors::{MvfError, Result},
    mvf_fbs::{DataType, DistanceMetric, VectorType},
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
    fn mvfgpu_search_fetch(corpus: *const MvfGpuCorpus, metric: u8, queries: *const c_void, query_dtype: u8,
                           query_dim: u32, nq: u32, k: u32, out_scores: *mut f32, out_indices: *mut u64,
                           out_raw: *mut i32, out_vectors: *mut c_void) -> c_int;
    fn mvfgpu_last_error_message() -> *const c_char;
    /// `MVFGPU_ABI_VERSION` of the loaded library (include/mvf_gpu.h): struct layouts and signatures this file mirrors.
    fn mvfgpu_abi_version() -> u32;
    /// Rows by GLOBAL index from HBM: the `ScoredVector.vector` payload (similarity_search.rs:18).
    fn mvfgpu_corpus_gather_rows(corpus: *const MvfGpuCorpus, indices: *const u64, count: u64,
                                 out_rows: *mut c_void) -> c_int;
    /// 0 automatic, 1 streaming kernel, 2 exact MFMA on the stored rows, 3 MFMA with the f16 shadow (the automatic
    /// choice for Float32 spaces), 4 additionally streams the f16 shadow for 1-2 queries (include/mvf_gpu.h).
    fn mvfgpu_set_scan_path(corpus: *mut MvfGpuCorpus, path: c_int) -> c_int;
    /// Deletion bitmap over the shard's rows / one u64 id per row (include/mvf_gpu.h; schema/core.fbs:35-39, :54).
    fn mvfgpu_corpus_set_tombstones(corpus: *mut MvfGpuCorpus, bitmap: *const u8, first_bit: u64, nbits: u64) -> c_int;
    fn mvfgpu_corpus_set_vector_ids(corpus: *mut MvfGpuCorpus, ids_le: *const c_void, n: u64) -> c_int;
    /// Several GPUs in one process: per-shard searches + ONE packed RCCL all-gather + merge (include/mvf_gpu.h).
    fn mvfgpu_shardset_create(shards: *const *mut MvfGpuCorpus, n_shards: c_int, out: *mut *mut MvfGpuShardset) -> c_int;
    fn mvfgpu_shardset_destroy(set: *mut MvfGpuShardset);
    fn mvfgpu_shardset_search(set: *mut MvfGpuShardset, metric: u8, queries: *const c_void, query_dtype: u8,
                              query_dim: u32, nq: u32, k: u32, out_scores: *mut f32, out_indices: *mut u64,
                              out_raw: *mut i32) -> c_int;
}

#[repr(C)]
pub struct MvfGpuShardset {
    _private: [u8; 0],
}

/// The `MVFGPU_ABI_VERSION` this binding was written against (include/mvf_gpu.h; 3 = round 4).
pub const MVFGPU_ABI_VERSION: u32 = 3;

/// Refuse a library that speaks another ABI version (its out-structs or signatures may differ); call once at start-up.
pub fn check_abi() -> Result<()> {
    let got = unsafe { mvfgpu_abi_version() };
    if got != MVFGPU_ABI_VERSION {
        return Err(MvfError::Extension(format!(
            "libmvf_gpu speaks ABI version {got}, this binding was written against {MVFGPU_ABI_VERSION}"
        )));
    }
    Ok(())
}

/// First two unsigned integers found in `msg` ("Index out of bounds: 7 >= 3", "Dimension mismatch: expected 768,
/// got 4", "Unsupported version: got 2, expected 1"): the C ABI carries the numbers of the structured variants in the
/// detail text, phrased exactly like the reference's `#[error(...)]` strings (src/errors.rs:8-40).
fn two_numbers(msg: &str) -> (usize, usize) {
    let mut it = msg
        .split(|c: char| !c.is_ascii_digit())
        .filter(|t| !t.is_empty())
        .filter_map(|t| t.parse::<usize>().ok());
    (it.next().unwrap_or(0), it.next().unwrap_or(0))
}

/// `enum mvf_status` (include/mvf_status.h) -> `MvfError` (src/errors.rs:8-40), every code explicitly.
/// The two codes without a reference variant (11 Device, 12 InvalidArgument) become `Extension`: "something outside
/// the file format failed", with the code's name in front so callers can still tell them apart.
fn status_to_error(status: c_int) -> MvfError {
    let msg = unsafe { CStr::from_ptr(mvfgpu_last_error_message()) }.to_string_lossy().into_owned();
    match status {
        1 => MvfError::Io(std::io::Error::new(std::io::ErrorKind::Other, msg)),
        2 => MvfError::InvalidFormat(msg),
        3 => {
            let (got, expected) = two_numbers(&msg); // "Unsupported version: got {got}, expected {expected}"
            MvfError::UnsupportedVersion { got: got as u16, expected: expected as u16 }
        }
        4 => MvfError::VectorSpaceNotFound(msg),
        5 => {
            let (index, len) = two_numbers(&msg); // "Index out of bounds: {index} >= {len}"
            MvfError::IndexOutOfBounds { index, len }
        }
        6 => {
            let (expected, actual) = two_numbers(&msg); // "Dimension mismatch: expected {expected}, got {actual}"
            MvfError::DimensionMismatch { expected, actual }
        }
        // the only way the GPU path produces it: a Sparse space where a Dense one is required
        7 => MvfError::InvalidVectorType { expected: VectorType::Dense, actual: VectorType::Sparse },
        8 => MvfError::CorruptedData(msg),
        9 => MvfError::Extension(msg),
        10 => MvfError::Build(msg),
        11 => MvfError::Extension(format!("Device error: {msg}")),
        12 => MvfError::Extension(format!("Invalid argument: {msg}")),
        other => MvfError::Extension(format!("unknown libmvf_gpu status {other}: {msg}")),
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
    rows: u64,
    float32: bool,
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
        Ok(Self { handle, dimension: space.dimension(), rows: total, float32: space.data_type().0 == DataType::Float32.0 })
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

    /// The k best rows WITH their payload -- the reference's `ScoredVector { index, score, vector }`
    /// (examples/similarity_search.rs:14-19) -- in one call: the rows are gathered on the GPU behind the search.
    /// Float32 spaces only (the payload comes back in the stored type; a Float16 space would need the widening of
    /// `Vector::as_f32`, src/vectors/vector.rs:81-89, on the returned halves).
    pub fn search_with_vectors(&self, metric: DistanceMetric, query: &[f32], k: usize) -> Result<Vec<(u64, f32, Vec<f32>)>> {
        if !self.float32 {
            return Err(MvfError::Build("search_with_vectors: Float32 spaces only".to_string()));
        }
        let d = self.dimension as usize;
        let mut scores = vec![0f32; k];
        let mut indices = vec![0u64; k];
        // one query: the library writes the first min(k, rows) payload rows only (include/mvf_gpu.h)
        let mut rows = vec![0f32; k.min(self.rows as usize) * d];
        let rc = unsafe {
            mvfgpu_search_fetch(self.handle, metric.0, query.as_ptr() as *const c_void, DataType::Float32.0,
                                query.len() as u32, 1, k as u32, scores.as_mut_ptr(), indices.as_mut_ptr(),
                                std::ptr::null_mut(), rows.as_mut_ptr() as *mut c_void)
        };
        if rc != 0 {
            return Err(status_to_error(rc));
        }
        Ok((0..k).take_while(|&i| indices[i] != u64::MAX)
            .map(|i| (indices[i], scores[i], rows[i * d..(i + 1) * d].to_vec()))
            .collect())
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

/// One vector space sharded by row range over several GPUs of this node: `devices.len()` shards of
/// near-equal size, searched as a whole -- per-shard top-k, one RCCL all-gather over xGMI, merge.
pub struct ShardedCorpus {
    set: *mut MvfGpuShardset,
    shards: Vec<GpuCorpus>, // borrowed by the set: dropped after it
    dimension: u32,
}

unsafe impl Send for ShardedCorpus {}

impl ShardedCorpus {
    pub fn from_space(space: &VectorSpace, devices: &[i32]) -> Result<Self> {
        let total = space.total_vectors();
        let g = devices.len().max(1) as u64;
        let per = (total + g - 1) / g;
        let stride = space.dimension() as u64 * elem_size(space.data_type())? as u64;
        let mut shards = Vec::new();
        for (i, &dev) in devices.iter().enumerate() {
            let first = (i as u64 * per).min(total);
            let count = per.min(total - first);
            let slice = space.map_vector_range(first, count)?;
            let mut handle = std::ptr::null_mut();
            let rc = unsafe {
                mvfgpu_corpus_create(slice.as_ptr::<u8>() as *const c_void, count, space.dimension(),
                                     space.data_type().0, stride, dev, first, &mut handle)
            };
            if rc != 0 {
                return Err(status_to_error(rc));
            }
            shards.push(GpuCorpus { handle, dimension: space.dimension(), float32: space.data_type().0 == DataType::Float32.0 });
        }
        let handles: Vec<*mut MvfGpuCorpus> = shards.iter().map(|s| s.handle).collect();
        let mut set = std::ptr::null_mut();
        let rc = unsafe { mvfgpu_shardset_create(handles.as_ptr(), handles.len() as c_int, &mut set) };
        if rc != 0 {
            return Err(status_to_error(rc));
        }
        Ok(Self { set, shards, dimension: space.dimension() })
    }

    pub fn search_batch(&mut self, metric: DistanceMetric, queries: &[f32], nq: usize, k: usize) -> Result<Vec<Vec<(u64, f32)>>> {
        let mut scores = vec![0f32; nq * k];
        let mut indices = vec![0u64; nq * k];
        let rc = unsafe {
            mvfgpu_shardset_search(self.set, metric.0, queries.as_ptr() as *const c_void, DataType::Float32.0,
                                   self.dimension, nq as u32, k as u32, scores.as_mut_ptr(), indices.as_mut_ptr(),
                                   std::ptr::null_mut())
        };
        if rc != 0 {
            return Err(status_to_error(rc));
        }
        let _ = &self.shards;
        Ok((0..nq)
            .map(|q| (0..k).map(|j| (indices[q * k + j], scores[q * k + j])).take_while(|(i, _)| *i != u64::MAX).collect())
            .collect())
    }
}

impl Drop for ShardedCorpus {
    fn drop(&mut self) {
        unsafe { mvfgpu_shardset_destroy(self.set) } // before the shards it borrows
    }
}
