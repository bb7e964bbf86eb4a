#!/usr/bin/env python3
"""Register / scratch / LDS use of EVERY kernel in libmvf_gpu.so, from the code objects' own metadata:
unbundle the gfx950 code object of each build/*.o (clang-offload-bundler), read the amdhsa.kernels notes (llvm-readelf),
demangle (llvm-cxxfilt).  Writes CSV to stdout.
    make -C metrovector_amd/csrc && python scripts/kernel_resources.py > profiles/r04_kernel_resources.csv"""
import glob, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
root = os.environ.get("MVF_RES_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
objs = sorted(glob.glob(os.path.join(root, "metrovector_amd", "csrc", "build", "*.o")))
rows = []
with tempfile.TemporaryDirectory() as d:
    for o in objs:
        co, fat = os.path.join(d, os.path.basename(o) + ".co"), os.path.join(d, os.path.basename(o) + ".fat")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", o, fat], capture_output=True)  # the host object's embedded bundle
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            continue
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={fat}", f"--output={co}"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        cur = None
        for line in notes.splitlines():
            if re.match(r"  - \.\w+:", line):  # a new entry of amdhsa.kernels (two-space indent)
                if cur and cur.get("name"):
                    rows.append((os.path.basename(o), cur))
                cur = {}
            m = re.match(r"  [ -] \.(\w+):\s*(.*)", line)
            if cur is None or not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'\"")
            if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                     "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
                cur[k] = v
        if cur and cur.get("name"):
            rows.append((os.path.basename(o), cur))
names = [c["name"] for _, c in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
print("# libmvf_gpu.so, gfx950: per-kernel resources from the code objects' metadata (scripts/kernel_resources.py).  vgpr = arch VGPRs + AGPRs "
      "allocated per lane (512 per SIMD: waves/SIMD = floor(512 / ceil8(vgpr)), at most 8); spill = VGPRs spilled to scratch; scratch = private segment bytes per lane; "
      "lds_static = group segment fixed size (the kernels' dynamic LDS comes on top)")
print("object,kernel,vgpr,agpr,sgpr,vgpr_spill,sgpr_spill,scratch_bytes,lds_static,max_threads,waves_per_simd_by_regs")
for (obj, c), dn in sorted(zip(rows, dem), key=lambda t: (t[0][0], t[1])):
    v = int(c.get("vgpr_count", 0))
    alloc = (v + 7) // 8 * 8 if v else 8
    w = min(8, 512 // alloc)
    dn = dn.replace("mvf::(anonymous namespace)::", "").replace("void ", "")
    print(",".join([obj, '"' + dn + '"', str(v), c.get("agpr_count", "0"), c.get("sgpr_count", "0"), c.get("vgpr_spill_count", "0"),
                    c.get("sgpr_spill_count", "0"), c.get("private_segment_fixed_size", "0"), c.get("group_segment_fixed_size", "0"),
                    c.get("max_flat_workgroup_size", ""), str(w)]))
