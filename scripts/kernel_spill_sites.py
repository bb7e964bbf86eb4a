#!/usr/bin/env python3
"""Where do the spilled registers of libmvf_gpu.so's kernels live?    Compiles every .hip of
metrovector_amd/csrc to gfx950 assembly (hipcc -S --cuda-device-only, the Makefile's flags), and for every kernel that
touches scratch counts the scratch_load / scratch_store instructions INSIDE its hot loops -- the innermost loops (a
backward branch to a label) that contain the kernel's streaming loads (global_load_dwordx4 / global_load_lds) or MFMAs --
against those outside.  CSV on stdout:  python scripts/kernel_spill_sites.py > profiles/r04_kernel_spill_sites.csv"""
import concurrent.futures as cf, glob, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "metrovector_amd", "csrc")
srcs = sorted(glob.glob(os.path.join(csrc, "*.hip")))
HOT = re.compile(r"\s(global_load_dwordx4|global_load_lds_dwordx4|v_mfma_\w+|buffer_load_dwordx4)\s")
SCR = re.compile(r"\s(scratch_load_\w+|scratch_store_\w+)\s")


def asm(src, d):
    out = os.path.join(d, os.path.basename(src)[:-4] + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, src],
                   cwd=csrc, capture_output=True, check=True)
    return out


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if "s_endpgm" in line:
                yield name, body
                name = None


rows = []
with tempfile.TemporaryDirectory() as d:
    with cf.ThreadPoolExecutor(8) as ex:
        files = list(ex.map(lambda s: asm(s, d), srcs))
    for f in files:
        for name, body in kernels(f):
            scr = [i for i, l in enumerate(body) if SCR.search(l)]
            if not scr:
                continue
            labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\w+):", l)] if m}
            loops = []
            for i, l in enumerate(body):
                m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", l)
                if m:
                    t = labels.get(m.group(1) or m.group(2))
                    if t is not None and t < i:
                        loops.append((t, i))
            hot = [i for i, l in enumerate(body) if HOT.search(l)]
            hot_loops = set()
            for h in hot:  # the innermost loop around each hot instruction
                inner = [lp for lp in loops if lp[0] <= h <= lp[1]]
                if inner:
                    hot_loops.add(min(inner, key=lambda lp: lp[1] - lp[0]))
            inside = sum(1 for s in scr if any(a <= s <= b for a, b in hot_loops))
            # ... and, finer: in a straight-line stretch (label / branch to label / branch) that itself holds a hot instruction --
            # a K2 kernel's k-tile loop also contains its tile-boundary code (bounds, epilogue), which runs once per KT k-tiles
            cuts = sorted(set([0, len(body)] + list(labels.values()) + [i + 1 for i, l in enumerate(body) if re.search(r"\ss_c?branch", l)]))
            def stretch(i):
                lo = max(c for c in cuts if c <= i)
                hi = min(c for c in cuts if c > i)
                return lo, hi
            hotset = set(hot)
            in_stretch = sum(1 for s in scr if any(j in hotset for j in range(*stretch(s))))
            rows.append((os.path.basename(f)[:-2], name, len(scr), inside, in_stretch, len(hot_loops)))
dem = subprocess.run(["c++filt"], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("# kernels of libmvf_gpu.so that touch scratch: scratch_load/store instructions in all, and how many of them sit INSIDE a hot loop "
      "(innermost loop holding the streaming loads / LDS-DMA / MFMAs); scripts/kernel_spill_sites.py")
print("# in_hot_stretch: of those, the ones in a straight-line stretch of code (between two labels / branches) that itself holds such an instruction")
print("unit,kernel,scratch_instructions,inside_hot_loops,in_hot_stretch,hot_loops")
for (unit, _, n, inside, ins, nl), dn in sorted(zip(rows, dem), key=lambda t: (-t[0][4], -t[0][3], t[0][0], t[1])):
    print(f'{unit},"{dn.replace("mvf::(anonymous namespace)::", "").replace("void ", "")}",{n},{inside},{ins},{nl}')
