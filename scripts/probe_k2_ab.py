"""A/B timing of the K2 kernels for the narrow types in ONE process (development aid).

For each BASELINE batched config the same resident corpus is searched with the variants interleaved, several
rounds each; prints wall ms per search (device-resident queries and results) and the last phase's scan ms.
Variants are environment switches libmvf_gpu reads per handle (re-read here with mvfgpu_corpus_reload_tuning) (MVF_K2_PP, MVF_K2_GROWTH, MVF_I8_SHADOW): see ALL_VARIANTS.
usage: python scripts/probe_k2_ab.py [cfg4,cfg5,cfg3] [rounds] [variant,variant,...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from metrovector_amd import _lib, gpu as G

CFGS = {
    "cfg4": (50_000_000, 768, 2, 1, 256),     # int8 dot, 256 queries
    "cfg5": (12_500_000, 1024, 1, 0, 1024),   # f16 L2 shard, 1024 queries
    "cfg3": (10_000_000, 768, 0, 2, 1024),    # f32 cosine (f16 shadow), 1024 queries
    "cfg5c": (12_500_000, 1024, 1, 2, 1024),  # f16 cosine
    "u8": (20_000_000, 768, 3, 0, 256),       # uint8 L2
    "d128": (15_000_000, 128, 0, 2, 4096),    # short rows: f32 cosine through the int8 shadow, 2 k-tiles per row
    "d64": (30_000_000, 64, 0, 2, 4096),      # 1 k-tile per row
    "d768": (2_500_000, 768, 0, 2, 4096),     # the same number of elements at 12 k-tiles per row
    "cfg5z": (12_500_000, 1024, 1, 0, 1024),  # cfg5 with all-but-one-element-zero rows (clock / power diagnostic)
    "cfg4z": (50_000_000, 768, 2, 1, 256),
    "cfg5b": (12_500_000, 1024, 1, 0, 1024),  # bf16-compatible f16 bit patterns (MVF_DIAG_BF16 builds)
}
ALL_VARIANTS = {"lockstep": {"MVF_K2_PP": "0", "MVF_K2_GROWTH": None, "MVF_I8_SHADOW": "0"},   # f16 / native kernels, no int8 shadow
                "pingpong": {"MVF_K2_PP": "1", "MVF_K2_GROWTH": None, "MVF_I8_SHADOW": "0"},
                "default": {"MVF_K2_PP": None, "MVF_K2_GROWTH": None, "MVF_I8_SHADOW": None, "MVF_QS_REFINE": None, "MVF_K2_BIAS": None},
                "g3": {"MVF_K2_PP": None, "MVF_K2_GROWTH": "3", "MVF_I8_SHADOW": None, "MVF_QS_REFINE": None},
                "g5": {"MVF_K2_PP": None, "MVF_K2_GROWTH": "5", "MVF_I8_SHADOW": None, "MVF_QS_REFINE": None},
                "g6": {"MVF_K2_PP": None, "MVF_K2_GROWTH": "6", "MVF_I8_SHADOW": None, "MVF_QS_REFINE": None},
                "g8": {"MVF_K2_PP": None, "MVF_K2_GROWTH": "8", "MVF_I8_SHADOW": None, "MVF_QS_REFINE": None},
                "g12": {"MVF_K2_PP": None, "MVF_K2_GROWTH": "12", "MVF_I8_SHADOW": None, "MVF_QS_REFINE": None},
                "oldepi": {"MVF_K2_PP": None, "MVF_K2_GROWTH": None, "MVF_I8_SHADOW": None, "MVF_QS_REFINE": None, "MVF_K2_BIAS": "0"},
                "norefine": {"MVF_K2_PP": None, "MVF_K2_GROWTH": None, "MVF_I8_SHADOW": None, "MVF_QS_REFINE": "0"},
                "pp_g8": {"MVF_K2_PP": "1", "MVF_K2_GROWTH": "8", "MVF_I8_SHADOW": "0"},
                "pp_g4": {"MVF_K2_PP": "1", "MVF_K2_GROWTH": "4", "MVF_I8_SHADOW": "0"},
                "pp_g3": {"MVF_K2_PP": "1", "MVF_K2_GROWTH": "3", "MVF_I8_SHADOW": "0"},
                "f16sel": {"MVF_K2_PP": None, "MVF_K2_GROWTH": None, "MVF_I8_SHADOW": "0"},
                "i8s_ls": {"MVF_K2_PP": "0", "MVF_K2_GROWTH": None, "MVF_I8_SHADOW": "1"},
                "i8s_pp": {"MVF_K2_PP": "1", "MVF_K2_GROWTH": None, "MVF_I8_SHADOW": "1"},
                "ls_g4": {"MVF_K2_PP": "0", "MVF_K2_GROWTH": "4", "MVF_I8_SHADOW": "0"},
                "ls_g3": {"MVF_K2_PP": "0", "MVF_K2_GROWTH": "3", "MVF_I8_SHADOW": "0"},
                # round 4: the f16 selection (no int8 shadow) on the lockstep kernel with / without the folded pre-filter, and on
                # the ping-pong kernel (round 2's epilogue)
                "f16_ls_bias": {"MVF_K2_PP": "0", "MVF_K2_GROWTH": None, "MVF_I8_SHADOW": "0", "MVF_K2_BIAS": None},
                "f16_ls_old": {"MVF_K2_PP": "0", "MVF_K2_GROWTH": None, "MVF_I8_SHADOW": "0", "MVF_K2_BIAS": "0"},
                "f16_pp": {"MVF_K2_PP": "1", "MVF_K2_GROWTH": None, "MVF_I8_SHADOW": "0", "MVF_K2_BIAS": None}}
VARIANTS = [(v, ALL_VARIANTS[v]) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["lockstep", "pingpong"])]


def main():
    names = (sys.argv[1] if len(sys.argv) > 1 else "cfg4,cfg5,cfg3").split(",")
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = _lib.gpu()
    for name in names:
        n, dim, dt, metric, nq = CFGS[name]
        if name.endswith("b"):  # f16 rows in +-[1.875, 2): the same bits read as bf16 are random values in +-[1, 2)
            rng = np.random.default_rng(1)
            host = np.empty((n, dim), np.uint16)
            step = 1_000_000
            for r0 in range(0, n, step):
                h = min(step, n - r0)
                host[r0:r0 + h] = (rng.integers(0, 2, (h, dim), dtype=np.uint16) << 15) | 0x3F80 | rng.integers(0, 128, (h, dim), dtype=np.uint16)
            c = G.GpuCorpus.from_array(host.view(np.float16))
            del host
        elif name.endswith("z"):  # power diagnostic: rows are zero except ONE element each (scores stay distinct)
            host = np.zeros((n, dim), {1: np.float16, 2: np.int8}[dt])
            rng = np.random.default_rng(1)
            host[np.arange(n), rng.integers(0, dim, n)] = (rng.uniform(-1, 1, n) if dt == 1 else rng.integers(-127, 127, n)).astype(host.dtype)
            c = G.GpuCorpus.from_array(host)
            del host
        else:
            c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
        qdt = {0: torch.float32, 1: torch.float32, 2: torch.int8, 3: torch.uint8}[dt]
        dq = torch.empty((nq, dim), dtype=qdt, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
        k = 100
        ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0")
        di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
        dr = torch.empty((nq, k), dtype=torch.int32, device="cuda:0")
        qcode = G.query_dtype_code(dt)

        def search():
            _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), qcode, dim, nq, k, ds.data_ptr(),
                                                    di.data_ptr(), dr.data_ptr(), None))

        ref = None
        res = {v: [] for v, _ in VARIANTS}
        for rnd in range(rounds + 1):  # round 0 = warm-up (norms, shadow, scratch)
            for vname, env in VARIANTS:
                for ek, ev in env.items():
                    if ev is None:
                        os.environ.pop(ek, None)
                    else:
                        os.environ[ek] = ev
                c.reload_tuning()  # the switches are read once per handle
                c.set_profiling(True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    search()
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / 3 * 1e3
                tm = c.last_timing()
                c.set_profiling(False)
                idx = di.cpu().numpy().copy()
                sc = ds.cpu().numpy().copy()
                if ref is None:
                    ref = (idx, sc)
                same = bool((idx == ref[0]).all() and (sc.view(np.uint32) == ref[1].view(np.uint32)).all())
                if rnd:
                    res[vname].append((wall, tm.scan_ms_avg))
                print(f"{name} round {rnd} {vname:9s} wall {wall:7.2f} ms  last-phase {tm.scan_ms_avg:6.2f} ms "
                      f"({tm.scan_flops / max(tm.scan_ms_avg, 1e-9) / 1e9:7.1f} Tops/s, "
                      f"{tm.scan_bytes / max(tm.scan_ms_avg, 1e-9) / 1e6:6.0f} GB/s) launches {tm.scan_launches} "
                      f"kernel {tm.scan_kernel} identical_to_first={same}", flush=True)
        for vname, _ in VARIANTS:
            w = sorted(x[0] for x in res[vname])
            l = sorted(x[1] for x in res[vname])
            print(f"== {name} {vname:9s} wall median {w[len(w) // 2]:.2f} min {w[0]:.2f}   last-phase median {l[len(l) // 2]:.2f} "
                  f"min {l[0]:.2f}", flush=True)
        c.close()


if __name__ == "__main__":
    main()
