"""k beyond one pass: passes of the streaming kernel (MVF_LARGE_K=1) against the whole-shard sort (MVF_LARGE_K=2) and the
library's own choice, in one process.  Device-pointer searches, wall clock around 5 calls + a synchronise."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
SHAPES = ((10_000_000, 768, 0, 2, "cfg2: 10M x 768 f32 cosine"), (12_500_000, 1024, 1, 0, "cfg5 shard: 12.5M x 1024 f16 L2"),
          (50_000_000, 768, 2, 1, "cfg4: 50M x 768 int8 dot"), (10_000, 128, 0, 0, "cfg1: 10k x 128 f32 L2"),
          (1_000_000, 128, 0, 0, "1M x 128 f32 L2"), (20_000_000, 64, 2, 1, "20M x 64 int8 dot"), (50_000_000, 16, 2, 1, "50M x 16 int8 dot"))
if len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]
for n, dim, dt, metric, name in SHAPES:
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    tdt = torch.float32 if dt in (0, 1) else torch.int8
    print(f"## {name}", flush=True)
    for nq in (1, 4):
        dq = torch.empty((nq, dim), dtype=tdt, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
        for k in (1024, 1025, 2048, 4096, 16384, 100_000, 1_000_000):
            if k > n:
                continue
            ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
            out, ref = [], None
            for mode in ("1", "2", None):
                if mode == "1" and k > 16384:
                    out.append("passes:      --  ")
                    continue
                if mode is None:
                    os.environ.pop("MVF_LARGE_K", None)
                else:
                    os.environ["MVF_LARGE_K"] = mode
                c.reload_tuning()
                c.set_profiling(True)
                for it in range(2):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(5):
                        _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), 2 if dt == 2 else 0, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
                    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5 * 1e3
                kern = c.last_timing().scan_kernel
                c.set_profiling(False)
                if ref is None:
                    ref = di.clone()
                same = bool((ref == di).all().item())
                out.append(f"{'passes' if mode == '1' else 'sort' if mode == '2' else 'auto(' + ('sort' if kern == 8 else 'passes') + ')'}: {t:8.3f} ms{'' if same else ' DIFFERENT'}")
            print(f"nq={nq} k={k:>8}: " + "   ".join(out), flush=True)
    c.close()
