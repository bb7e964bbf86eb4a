"""ctypes loaders for the two in-tree native libraries.

  libmvf_gpu.so   — HIP kernels + the C ABI of include/mvf_gpu.h (the drop-in
                    boundary a Rust caller would bind)
  libmvf_host.so  — C++ MVF reader/writer (include/mvf_file.h)

There is no Python or CPU fallback for the search path: if libmvf_gpu.so is
missing the import of anything that needs it raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
GPU_LIB_PATH = os.environ.get("MVF_GPU_LIB_PATH") or os.path.join(_HERE, "libmvf_gpu.so")  # override: diagnostic builds
HOST_LIB_PATH = os.environ.get("MVF_HOST_LIB_PATH") or os.path.join(_HERE, "libmvf_host.so")  # override: another build (sanitizers)

_gpu = None
_host = None


class _OutStruct(C.Structure):
    """Out-structs of the C ABI carry a caller-set `struct_size` first (include/mvf_gpu.h, "OUT-STRUCTS GROW")."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(type(self))


class CorpusInfo(_OutStruct):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("rows", C.c_uint64), ("index_base", C.c_uint64),
                ("dimension", C.c_uint32), ("pitch_bytes", C.c_uint32), ("data_type", C.c_uint8),
                ("has_vector_ids", C.c_uint8), ("shadows", C.c_uint8), ("selection_state", C.c_uint8), ("reserved2", C.c_uint32),
                ("device_bytes", C.c_uint64), ("deleted_rows", C.c_uint64)]


class UploadOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("chunk_mib", C.c_uint32), ("reserved", C.c_uint32)]


UPLOAD_EAGER_NORMS, UPLOAD_EAGER_SHADOW, UPLOAD_PINNED_STAGING, UPLOAD_PAGEABLE = 1, 2, 4, 8


MAX_SHARDS = 64


class ShardsetInfo(_OutStruct):
    _fields_ = [("struct_size", C.c_uint32), ("n_shards", C.c_uint32), ("rccl_ranks", C.c_uint32), ("dimension", C.c_uint32),
                ("data_type", C.c_uint8), ("reserved", C.c_uint8 * 7), ("rows", C.c_uint64)]


class ShardsetTiming(_OutStruct):
    _fields_ = [("struct_size", C.c_uint32), ("n_shards", C.c_uint32), ("searches", C.c_uint64), ("total_ms", C.c_float),
                ("enqueue_ms", C.c_float), ("search_ms", C.c_float), ("exchange_merge_ms", C.c_float),
                ("shard_search_ms", C.c_float * MAX_SHARDS)]


class Timing(_OutStruct):
    _fields_ = [("struct_size", C.c_uint32), ("samples", C.c_uint32), ("scan_ms", C.c_float), ("select_ms", C.c_float),
                ("total_ms", C.c_float), ("scan_ms_avg", C.c_float), ("select_ms_avg", C.c_float),
                ("scan_kernel", C.c_uint32), ("scan_launches", C.c_uint32), ("scan_bytes", C.c_uint64),
                ("scan_flops", C.c_uint64), ("search_ms", C.c_float), ("search_ms_avg", C.c_float),
                ("search_flops", C.c_uint64), ("repaired_queries", C.c_uint32), ("reserved", C.c_uint32)]


ABI_VERSION = 3  # include/mvf_gpu.h MVFGPU_ABI_VERSION


def _preload_torch_hip() -> None:
    # torch ships its own libamdhip64.so (same SONAME as /opt/rocm's).  Loading
    # torch FIRST makes libmvf_gpu.so bind to that copy, so both share one HIP
    # runtime (device pointers, streams).  Loading order reversed would put two
    # runtimes in one process.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing, not a requirement of the library
        pass


def gpu() -> C.CDLL:
    """libmvf_gpu.so, or an ImportError naming the build command."""
    global _gpu
    if _gpu is not None:
        return _gpu
    if not os.path.exists(GPU_LIB_PATH):
        raise ImportError(
            f"{GPU_LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C metrovector_amd/csrc`.  metrovector_amd has no CPU fallback for the search path.")
    _preload_torch_hip()
    lib = C.CDLL(GPU_LIB_PATH)
    vp, u64, u32, u8, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint8, C.c_int
    pp = C.POINTER(C.c_void_p)
    lib.mvfgpu_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.mvfgpu_strerror.restype = C.c_char_p
    lib.mvfgpu_strerror.argtypes = [i32]
    lib.mvfgpu_last_error_message.restype = C.c_char_p
    lib.mvfgpu_last_error_message.argtypes = []
    lib.mvfgpu_corpus_create.argtypes = [vp, u64, u32, u8, u64, i32, u64, pp]
    lib.mvfgpu_corpus_create_ex.argtypes = [vp, u64, u32, u8, u64, i32, u64, C.POINTER(UploadOptions), pp]
    lib.mvfgpu_corpus_set_tombstones.argtypes = [vp, vp, u64, u64]
    lib.mvfgpu_corpus_set_vector_ids.argtypes = [vp, vp, u64]
    lib.mvfgpu_corpus_create_synthetic.argtypes = [u64, u32, u8, u64, u64, i32, pp]
    lib.mvfgpu_shardset_create.argtypes = [C.POINTER(vp), i32, pp]
    lib.mvfgpu_shardset_destroy.restype = None
    lib.mvfgpu_shardset_destroy.argtypes = [vp]
    lib.mvfgpu_shardset_get_info.argtypes = [vp, C.POINTER(ShardsetInfo)]
    lib.mvfgpu_shardset_search.argtypes = [vp, u8, vp, u8, u32, u32, u32, vp, vp, vp]
    lib.mvfgpu_shardset_last_timing.argtypes = [vp, C.POINTER(ShardsetTiming)]
    lib.mvfgpu_corpus_destroy.restype = None
    lib.mvfgpu_corpus_destroy.argtypes = [vp]
    lib.mvfgpu_corpus_get_info.argtypes = [vp, C.POINTER(CorpusInfo)]
    lib.mvfgpu_corpus_read_rows.argtypes = [vp, u64, u64, vp]
    lib.mvfgpu_corpus_gather_rows.argtypes = [vp, vp, u64, vp]
    lib.mvfgpu_search.argtypes = [vp, u8, vp, u8, u32, u32, u32, vp, vp, vp]
    lib.mvfgpu_search_fetch.argtypes = [vp, u8, vp, u8, u32, u32, u32, vp, vp, vp, vp]
    lib.mvfgpu_search_device.argtypes = [vp, u8, vp, u8, u32, u32, u32, vp, vp, vp, vp]
    lib.mvfgpu_merge_topk_host.argtypes = [vp, vp, vp, u32, u32, u32, u8, u8, vp, vp, vp]
    lib.mvfgpu_merge_topk_device.argtypes = [vp, vp, vp, u32, u32, u32, u8, u8, vp, vp, vp, i32, vp]
    lib.mvfgpu_merge_topk_packed_device.argtypes = [vp, u32, u32, u32, u8, u8, vp, vp, vp, i32, vp]
    lib.mvfgpu_synth_queries_device.argtypes = [vp, u32, u32, u8, u64, i32, vp]
    lib.mvfgpu_set_profiling.argtypes = [vp, i32]
    lib.mvfgpu_last_timing.argtypes = [vp, C.POINTER(Timing)]
    lib.mvfgpu_set_scan_path.argtypes = [vp, i32]
    lib.mvfgpu_corpus_reload_tuning.argtypes = [vp]
    lib.mvfgpu_selftest_feedback.argtypes = [vp, u32, vp]
    lib.mvfgpu_selftest_route.argtypes = [u64, u32, u8, u8, u32, u32, vp]
    abi = getattr(lib, "mvfgpu_abi_version", None)  # a library from before round 4 has no such symbol: the same advice, not an AttributeError
    if abi is not None:
        abi.restype = u32
        abi.argtypes = []
    if abi is None or abi() != ABI_VERSION:
        raise ImportError(f"{GPU_LIB_PATH} speaks ABI version {abi() if abi else '< 3 (no mvfgpu_abi_version)'}, this binding was written "
                          f"against {ABI_VERSION} (include/mvf_gpu.h MVFGPU_ABI_VERSION): rebuild the library")
    for name in ("mvfgpu_device_count", "mvfgpu_corpus_create", "mvfgpu_corpus_create_ex", "mvfgpu_corpus_create_synthetic",
                 "mvfgpu_corpus_set_tombstones", "mvfgpu_corpus_set_vector_ids", "mvfgpu_shardset_create",
                 "mvfgpu_shardset_get_info", "mvfgpu_shardset_search", "mvfgpu_shardset_last_timing",
                 "mvfgpu_corpus_get_info", "mvfgpu_corpus_read_rows", "mvfgpu_corpus_gather_rows", "mvfgpu_search", "mvfgpu_search_fetch", "mvfgpu_search_device",
                 "mvfgpu_merge_topk_host", "mvfgpu_merge_topk_device", "mvfgpu_merge_topk_packed_device",
                 "mvfgpu_synth_queries_device",
                 "mvfgpu_set_profiling", "mvfgpu_last_timing", "mvfgpu_set_scan_path", "mvfgpu_corpus_reload_tuning",
                 "mvfgpu_selftest_feedback", "mvfgpu_selftest_route"):
        getattr(lib, name).restype = C.c_int
    sched = getattr(lib, "mvfgpu_selftest_schedule", None)  # added within ABI 3 (round 5): a diagnostic entry point, no layout changed
    if sched is not None:
        sched.argtypes = [u64, u32, u32, C.c_int, vp, u32, vp, vp, vp]
        sched.restype = C.c_int
    _gpu = lib
    return lib


class DataBlock(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("size", C.c_uint64), ("compression", C.c_uint8),
                ("compressed_size", C.c_uint64), ("checksum", C.c_uint32)]


class CVectorSpace(C.Structure):
    _fields_ = [("reader", C.c_void_p), ("index", C.c_uint32), ("name", C.c_void_p), ("name_len", C.c_uint32),
                ("dimension", C.c_uint32), ("total_vectors", C.c_uint64), ("vector_type", C.c_uint8),
                ("distance_metric", C.c_uint8), ("data_type", C.c_uint8), ("index_type", C.c_uint8),
                ("vectors_block_index", C.c_uint32), ("vector_ids_block_index", C.c_uint32),
                ("has_sparse_metadata", C.c_uint8), ("has_tombstones", C.c_uint8), ("tombstone_format", C.c_uint8),
                ("tombstone_block_index", C.c_uint32), ("tombstone_deleted_count", C.c_uint64)]


class CVectorSlice(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_uint64), ("count", C.c_uint64), ("data_type", C.c_uint8)]


def host() -> C.CDLL:
    """libmvf_host.so (C++ MVF reader/writer)."""
    global _host
    if _host is not None:
        return _host
    if not os.path.exists(HOST_LIB_PATH):
        raise ImportError(f"{HOST_LIB_PATH} is missing — build it with `make -C metrovector_amd/csrc host`")
    lib = C.CDLL(HOST_LIB_PATH)
    vp, u64, u32, u8, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint8, C.c_int
    pp = C.POINTER(C.c_void_p)
    lib.mvf_last_error_message.restype = C.c_char_p
    lib.mvf_strerror.restype = C.c_char_p
    lib.mvf_strerror.argtypes = [i32]
    lib.mvf_reader_open.argtypes = [C.c_char_p, pp]
    lib.mvf_reader_open_bytes.argtypes = [vp, u64, pp]
    lib.mvf_reader_close.restype = None
    lib.mvf_reader_close.argtypes = [vp]
    lib.mvf_reader_version.argtypes = [vp, C.POINTER(C.c_uint16)]
    lib.mvf_reader_num_vector_spaces.argtypes = [vp, C.POINTER(u64)]
    lib.mvf_reader_vector_space_name.argtypes = [vp, u64, pp, C.POINTER(u32)]
    lib.mvf_reader_vector_space.argtypes = [vp, C.c_char_p, C.POINTER(CVectorSpace)]
    lib.mvf_reader_vector_space_at.argtypes = [vp, u64, C.POINTER(CVectorSpace)]
    lib.mvf_reader_file_size.argtypes = [vp, C.POINTER(u64)]
    lib.mvf_reader_has_metadata.argtypes = [vp, C.POINTER(i32)]
    lib.mvf_reader_num_metadata_columns.argtypes = [vp, C.POINTER(u64)]
    lib.mvf_reader_metadata_column_name.argtypes = [vp, u64, pp, C.POINTER(u32)]
    lib.mvf_reader_num_blocks.argtypes = [vp, C.POINTER(u64)]
    lib.mvf_reader_block.argtypes = [vp, u64, C.POINTER(DataBlock)]
    lib.mvf_reader_validate.argtypes = [vp]
    lib.mvf_reader_validate_with_checksum.argtypes = [vp]
    lib.mvf_space_get_vector.argtypes = [C.POINTER(CVectorSpace), u64, pp, C.POINTER(u64)]
    lib.mvf_space_map_vector_range.argtypes = [C.POINTER(CVectorSpace), u64, u64, C.POINTER(CVectorSlice)]
    lib.mvf_vector_as_f32.argtypes = [vp, u64, u8, vp, u64, C.POINTER(u64)]
    lib.mvf_space_vector_ids.argtypes = [C.POINTER(CVectorSpace), pp, C.POINTER(u64)]
    lib.mvf_space_tombstones.argtypes = [C.POINTER(CVectorSpace), C.POINTER(u8), pp, C.POINTER(u64), C.POINTER(u64)]
    lib.mvf_space_tombstone_bitmap.argtypes = [C.POINTER(CVectorSpace), vp, u64, C.POINTER(u64)]
    lib.mvf_builder_set_vector_ids.argtypes = [vp, C.c_char_p, vp, u64]
    lib.mvf_builder_set_tombstones.argtypes = [vp, C.c_char_p, u8, vp, u64, u64]
    lib.mvf_builder_new.argtypes = [pp]
    lib.mvf_builder_free.restype = None
    lib.mvf_builder_free.argtypes = [vp]
    lib.mvf_builder_add_vector_space.argtypes = [vp, C.c_char_p, u32, u8, u8, u8, C.POINTER(u64)]
    lib.mvf_builder_add_vectors_f32.argtypes = [vp, C.c_char_p, vp, u64, u32]
    lib.mvf_builder_add_vectors_raw.argtypes = [vp, C.c_char_p, vp, u64, u32]
    lib.mvf_builder_reserve_vectors.argtypes = [vp, C.c_char_p, u64]
    lib.mvf_builder_add_metadata_column.argtypes = [vp, C.c_char_p, u8, vp, u64]
    lib.mvf_builder_to_bytes.argtypes = [vp, u32, pp, C.POINTER(u64)]
    lib.mvf_builder_save.argtypes = [vp, C.c_char_p, u32]
    lib.mvf_free.restype = None
    lib.mvf_free.argtypes = [vp]
    lib.mvf_crc32.restype = u32
    lib.mvf_crc32.argtypes = [vp, u64]
    lib.mvf_f32_to_f16.restype = C.c_uint16
    lib.mvf_f32_to_f16.argtypes = [C.c_float]
    lib.mvf_f16_to_f32.restype = C.c_float
    lib.mvf_f16_to_f32.argtypes = [C.c_uint16]
    _host = lib
    return lib


def host_check(status: int) -> None:
    if status != 0:
        from .errors import raise_for_status
        msg = host().mvf_last_error_message().decode("utf-8", "replace")
        raise_for_status(status, msg or host().mvf_strerror(status).decode())


def gpu_check(status: int) -> None:
    if status != 0:
        from .errors import raise_for_status
        msg = gpu().mvfgpu_last_error_message().decode("utf-8", "replace")
        raise_for_status(status, msg or gpu().mvfgpu_strerror(status).decode())
