"""Calibration workloads for the FETCH_SIZE counter (run under `rocprofv3 --pmc FETCH_SIZE`): launches whose HBM read
traffic is KNOWN because every block reads its corpus rows exactly once and nothing is shared between blocks:
  K1 single query (10M x 768 f32: 30.72 GB);  the exact f32 MFMA kernel with ONE 128-query tile (the last phase reads its
  rows once);  the f16 LDS-DMA kernel with ONE 64-query tile on Float16 rows;  the int8 LDS-DMA kernel with ONE 256-query
  tile on Int8 rows.  Prints, per workload, the rows x bytes of the LAST phase (the largest dispatch of its kernel) as
  `expect <kernel substring> <bytes>` lines that scripts/rocprof_summarize.py calibrate pairs with the counter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
os.environ["MVF_I8_SHADOW"] = "0"


def run(n, dim, dt, metric, nq, path, kname):
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    qdt = {0: torch.float32, 1: torch.float32, 2: torch.int8}[dt]
    dq = torch.empty((nq, dim), dtype=qdt, device="cuda:0")
    _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
    ds = torch.empty((nq, 100), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, 100), dtype=torch.int64, device="cuda:0")
    c.set_scan_path(path); c.set_profiling(True)
    for _ in range(3):
        _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), G.query_dtype_code(dt), dim, nq, 100, ds.data_ptr(), di.data_ptr(), None, None))
    torch.cuda.synchronize()
    tm = c.last_timing()
    print(f"expect {kname} {tm.scan_bytes}", flush=True)
    c.close()


run(10_000_000, 768, 0, 2, 1, 1, "scan_stream_kernel<0, 2")
run(10_000_000, 768, 0, 2, 128, 2, "scan_mfma_f32_kernel<2>")
run(12_500_000, 1024, 1, 0, 64, 2, "scan_mfma16_dma_kernel<1, 0, false, false, 64>")
run(30_000_000, 768, 2, 1, 256, 2, "scan_mfma16_dma_kernel<2, 1, false, false, 256>")
