"""The ladder VERDICT r4 item 1 asks for, in ONE process on ONE box: the int8-shadow K2 kernel's last phase under several builds
of scan_mfma16_dma.hip (scripts/build_k2_variants.sh) next to the bare k-loop probe (scripts/probe_k2_w1.hip as a shared object)
over the same number of rows, rounds interleaved.

Every library build is its own copy of libmvf_gpu (own handles, own corpus): the corpora are Float16 rows (cfg3's row shape
at half the bytes of the Float32 corpus -- the selection kernel reads the int8 shadow either way; cfg5's own shard), so that six
copies fit the card.

usage: python scripts/k2_ladder.py [cfg3,cfg5] [rounds=3] [tags=main,r4,oldfrag,noepi,nobias] [probe variants=8,8n]
Prints one line per (config, rung): median / min of the last phase's scan ms and of the whole search, POP/s of the last phase.
Under rocprofv3 --pmc the kernels of different builds carry the same name: the script prints `SEQ <tag>` lines in dispatch
order (one per search), which scripts/k2_ladder_pmc.py joins with the counter CSV.
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: one HIP runtime for every library)

from metrovector_amd import _lib  # noqa: E402

CFGS = {"cfg3": (10_000_000, 768, 1, 2, 1024),   # f16 rows, cosine: the int8-shadow kernel <2, 2, false, true, 256, true>
        "cfg5": (12_500_000, 1024, 1, 0, 1024),  # f16 rows, L2: <2, 0, false, true, 256, true>
        "cfg3ip": (10_000_000, 768, 1, 1, 1024),
        "cfg4": (50_000_000, 768, 2, 1, 256),    # int8 rows, dot: <2, 1, false, false, 256, true> (no probe: other row count per byte)
        "u8cos": (20_000_000, 768, 3, 2, 256),   # uint8 rows, cosine: <3, 2, false, false, 256, true>
        "u8l2": (20_000_000, 768, 3, 0, 1024),   # uint8 rows, L2, four query tiles
        "i8l2": (20_000_000, 512, 2, 0, 1024),   # int8 rows, L2
        "i8cos": (20_000_000, 512, 2, 2, 512),   # int8 rows, cosine
        "q100": (10_000_000, 768, 1, 2, 100),    # the 128-query tile: f16 rows, cosine through the int8 shadow
        "q128l2": (12_500_000, 1024, 1, 0, 128), # the 128-query tile, L2
        "i8q128": (20_000_000, 512, 2, 2, 128)}  # the 128-query tile, int8 rows, cosine


def load(tag):
    path = os.path.join(ROOT, "metrovector_amd", "libmvf_gpu.so") if tag == "main" else os.path.join(ROOT, "scripts", "bin", f"libmvf_gpu_{tag}.so")
    lib = C.CDLL(path)
    vp, u64, u32, u8, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint8, C.c_int
    lib.mvfgpu_corpus_create_synthetic.argtypes = [u64, u32, u8, u64, u64, i32, C.POINTER(vp)]
    lib.mvfgpu_corpus_destroy.restype = None
    lib.mvfgpu_corpus_destroy.argtypes = [vp]
    lib.mvfgpu_search_device.argtypes = [vp, u8, vp, u8, u32, u32, u32, vp, vp, vp, vp]
    lib.mvfgpu_synth_queries_device.argtypes = [vp, u32, u32, u8, u64, i32, vp]
    lib.mvfgpu_set_profiling.argtypes = [vp, i32]
    lib.mvfgpu_last_timing.argtypes = [vp, C.POINTER(_lib.Timing)]
    return lib


def main():
    names = (sys.argv[1] if len(sys.argv) > 1 else "cfg3,cfg5").split(",")
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    tags = (sys.argv[3] if len(sys.argv) > 3 else "main,r4,oldfrag,noepi,nobias").split(",")
    pvars = [v for v in (sys.argv[4] if len(sys.argv) > 4 else "8,8n").split(",") if v]
    libs = {t: load(t) for t in tags}
    probe = None
    if pvars:
        probe = C.CDLL(os.path.join(ROOT, "scripts", "bin", "libprobe_k2.so"))
        probe.probe_k2_run.restype = C.c_float
        probe.probe_k2_run.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p]
    for name in names:
        n, dim, dt, metric, nq = CFGS[name]
        k = 100
        qdt = {1: torch.float32, 2: torch.int8, 3: torch.uint8}[dt]
        qcode = 0 if dt == 1 else dt
        dq = torch.empty((nq, dim), dtype=qdt, device="cuda:0")
        ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0")
        di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
        dr = torch.empty((nq, k), dtype=torch.int32, device="cuda:0")
        hs = {}
        for t, lib in libs.items():
            h = C.c_void_p()
            rc = lib.mvfgpu_corpus_create_synthetic(n, dim, dt, 0x4D564631, 0, 0, C.byref(h))
            assert rc == 0, (t, rc)
            hs[t] = h
        rc = libs[tags[0]].mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, qcode, 0x4D564632, 0, None)
        assert rc == 0
        last_rows = None
        res = {t: [] for t in tags}
        pres = {v: [] for v in pvars}
        ref = None
        for rnd in range(rounds + 1):  # round 0 = warm-up (norms, shadow, scratch)
            order = tags if rnd == 0 else tags[(rnd - 1) % len(tags):] + tags[:(rnd - 1) % len(tags)]  # rotated: no build always runs behind the same one
            for t in order:
                lib, h = libs[t], hs[t]
                lib.mvfgpu_set_profiling(h, 1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 2 if rnd else 1
                for _ in range(reps):
                    print(f"SEQ {name} {t}", flush=True)
                    rc = lib.mvfgpu_search_device(h, metric, dq.data_ptr(), qcode, dim, nq, k, ds.data_ptr(), di.data_ptr(), dr.data_ptr(), None)
                    assert rc == 0, (t, rc)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / reps * 1e3
                tm = _lib.Timing()
                lib.mvfgpu_last_timing(h, C.byref(tm))
                lib.mvfgpu_set_profiling(h, 0)
                last_rows = tm.scan_bytes // dim  # int8 shadow / int8 rows: dim bytes per row
                if t in ("main", "r4", "oldfrag") or not t.startswith("no"):
                    idx = di.cpu()
                    if ref is None:
                        ref = idx
                    same = bool((idx == ref).all())
                else:
                    same = None  # the diagnostic builds select nothing
                if rnd:
                    res[t].append((tm.scan_ms_avg, wall))
                print(f"{name} round {rnd} {t:8s} last-phase {tm.scan_ms_avg:7.3f} ms  wall {wall:7.3f} ms  launches {tm.scan_launches} kernel {tm.scan_kernel} "
                      f"repaired {tm.repaired_queries} same_as_first={same}", flush=True)
            if probe is not None and last_rows and nq % 256 == 0:
                for v in pvars:
                    print(f"SEQ {name} probe{v}", flush=True)
                    ms = probe.probe_k2_run(int(last_rows), dim, nq, 3 if rnd else 1, v.encode())
                    if rnd:
                        pres[v].append(ms)
                    print(f"{name} round {rnd} probe{v:4s} {ms:7.3f} ms over {last_rows} rows", flush=True)
        ops = 2.0 * nq * last_rows * dim
        for v in pvars:
            x = sorted(pres[v])
            if not x:  # (the probe takes multiples of 256 queries)
                continue
            print(f"== {name} probe{v:6s} k-loop alone       median {x[len(x) // 2]:7.3f} min {x[0]:7.3f} ms  {ops / x[len(x) // 2] / 1e12:6.3f} POP/s", flush=True)
        for t in tags:
            a = sorted(x[0] for x in res[t])
            w = sorted(x[1] for x in res[t])
            print(f"== {name} {t:11s} last phase ({last_rows} rows) median {a[len(a) // 2]:7.3f} min {a[0]:7.3f} ms  {ops / a[len(a) // 2] / 1e12:6.3f} POP/s   "
                  f"whole search median {w[len(w) // 2]:7.3f} ms", flush=True)
        for t, lib in libs.items():
            lib.mvfgpu_corpus_destroy(hs[t])
        if probe is not None:
            probe.probe_k2_free()
        del dq, ds, di, dr
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
