"""Per-search cost of the N>1 exchange step (RCCL all-gather of the packed list + merge), measured on ONE GPU with a
1-rank nccl group and always_exchange=True (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29777")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
from metrovector_amd import gpu as G, _lib
from metrovector_amd.sharded import ShardedSearcher
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
c = G.GpuCorpus.synthetic(10_000_000, 768, 0, 0x4D564631)
for nq in (1, 1024):
    dq = torch.empty((nq, 768), dtype=torch.float32, device="cuda")
    _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dq.data_ptr(), nq, 768, 0, 0x4D564632, 0, None))
    res = {}
    for name, ex in (("plain", False), ("exchange", True)):
        s = ShardedSearcher(c, always_exchange=ex)
        for _ in range(5): s.search(dq, 100, 2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        steps = 50 if nq == 1 else 10
        for _ in range(steps): s.search(dq, 100, 2)
        torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / steps * 1e3
    print(f"nq={nq}: plain {res['plain']:.3f} ms, with all-gather + merge {res['exchange']:.3f} ms "
          f"(+{(res['exchange']-res['plain'])*1e3:.0f} us = {100*(res['exchange']/res['plain']-1):.1f} %)", flush=True)
c.close(); dist.destroy_process_group()
