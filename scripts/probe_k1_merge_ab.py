"""A/B of the streaming kernel's piece merge: survivors joined to the running top-k by counting (MVF_K1_RANK_MERGE=128, the
default) against the sort network on every merge (=0), in one process on one corpus per shape.  Single query and the
four-query pass (scan path 1), k = 10 and 100; ms per search (best of 3 rounds of 10) and GB/s of stored rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G

lib = _lib.gpu()
ES = {0: 4, 1: 2, 2: 1, 3: 1}
SHAPES = [(0, 768), (0, 128), (0, 32), (0, 8), (1, 1024), (1, 64), (1, 16), (2, 768), (2, 128), (2, 64), (2, 32), (3, 256)]
if len(sys.argv) > 1 and sys.argv[1] == "headline":
    SHAPES = [(0, 768)]
# optional: MVF_AB_MODES=a,b (two values of MVF_K1_RANK_MERGE instead of 0,128), MVF_AB_ROWS=n (rows per corpus instead of 4 GiB worth)
MA, MB = (os.environ.get("MVF_AB_MODES") or "0,128").split(",")
VAR = os.environ.get("MVF_AB_VAR") or "MVF_K1_RANK_MERGE"  # the switch the two modes are values of
ROWS = int(os.environ.get("MVF_AB_ROWS") or 0)
print(f"dtype dim row_bytes nq k : {VAR}={MA} ms -> ={MB} ms (GB/s)  ratio", flush=True)
for dt, dim in SHAPES:
    rb = dim * ES[dt]
    n = ROWS or (10_000_000 if (dt, dim) == (0, 768) else min(100_000_000, (4 << 30) // rb))
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    c.set_scan_path(1)
    tdt = torch.float32 if dt in (0, 1) else (torch.int8 if dt == 2 else torch.uint8)
    for nq in (1, 4):
        dq = torch.empty((nq, dim), dtype=tdt, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
        for k in (10, 100):
            ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0")
            di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
            out = {}
            ref = None
            for mode in (MA, MB, MA, MB):
                os.environ[VAR] = mode
                c.reload_tuning()
                best = 1e9
                for _ in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(10):
                        _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), 0 if dt in (0, 1) else dt, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
                    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
                out[mode] = min(out.get(mode, 1e9), best)
                cur = (ds.clone(), di.clone())
                if ref is None:
                    ref = cur
                elif not (torch.equal(ref[0].view(torch.int32), cur[0].view(torch.int32)) and torch.equal(ref[1], cur[1])):
                    print("  RESULTS DIFFER", dt, dim, nq, k, flush=True)
            gbs = n * rb / out[MB] / 1e6
            print(f"{dt} {dim:5d} {rb:5d} nq={nq} k={k:3d}: {out[MA]:8.3f} -> {out[MB]:8.3f} ms ({gbs:7.1f} GB/s)  {out[MA] / out[MB]:.3f}", flush=True)
    c.close()
os.environ.pop(VAR, None)
