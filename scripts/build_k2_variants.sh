#!/bin/bash
# Builds diagnostic variants of libmvf_gpu.so that differ in scan_mfma16_dma.hip only (scripts/k2_ladder.py loads them side by
# side in one process), and the k-loop probe as a shared object:
#   scripts/bin/libmvf_gpu_<tag>.so   tag = noepi   -DMVF_DIAG_NOEPI                     (no drain at a tile's end)
#                                           nobias  -DMVF_DIAG_NOEPI -DMVF_DIAG_NOBIAS    (... and no bounds at its start)
#                                           oldfrag -DMVF_K2_OLD_FRAG_ORDER              (rounds 2-4's fragment request order)
#                                           nowalk  the drain without the element walk (NB: nothing is selected then, every group of the later phases is flagged)
#                                           oldwalk -DMVF_K2_OLD_WALK  a compare + branch per element in the walk of a flagged group (until the middle of round 5)
#                                           endb    -DMVF_K2_ENDBARRIER  the barrier at the end of the k-tile (rounds 2-4) instead of in the middle of its MFMAs
#                                           r4      round 4's kernel sources (git show b12f680:...) against this tree's headers
#   scripts/bin/libprobe_k2.so        scripts/probe_k2_w1.hip -DPROBE_K2_SHARED
# usage: bash scripts/build_k2_variants.sh [tags...]   (after `make -C metrovector_amd/csrc`)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/metrovector_amd/csrc
B=$ROOT/scripts/bin
mkdir -p $B/obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -I$C"
OTHERS=$(ls $C/build/*.o | grep -v scan_mfma16_dma.o)
TAGS=${@:-noepi nobias oldfrag r4}
pids=()
for t in $TAGS; do
  (
    src=$C/scan_mfma16_dma.hip; defs=""
    case $t in
      noepi) defs="-DMVF_DIAG_NOEPI" ;;
      nobias) defs="-DMVF_DIAG_NOEPI -DMVF_DIAG_NOBIAS" ;;
      oldfrag) defs="-DMVF_K2_OLD_FRAG_ORDER" ;;
      nowalk) defs="-DMVF_DIAG_NOWALK" ;;
      endb) defs="-DMVF_K2_ENDBARRIER" ;;
      oldwalk) defs="-DMVF_K2_OLD_WALK" ;;
      head) d=$B/obj/src_$t; mkdir -p $d   # the last commit's kernel sources against the working tree: A/B of an uncommitted change
          for f in scan_mfma16_dma.hip scan_mfma16_bias.inc scan_mfma16_common.inc; do git -C $ROOT show HEAD:metrovector_amd/csrc/$f > $d/$f; done
          src=$d/scan_mfma16_dma.hip ;;
      r4*) d=$B/obj/src_$t; mkdir -p $d
          for f in scan_mfma16_dma.hip scan_mfma16_bias.inc scan_mfma16_common.inc; do git -C $ROOT show b12f680:metrovector_amd/csrc/$f > $d/$f; done
          src=$d/scan_mfma16_dma.hip
          [ $t = r4noepi ] && defs="-DMVF_DIAG_NOEPI"; [ $t = r4nobias ] && defs="-DMVF_DIAG_NOEPI -DMVF_DIAG_NOBIAS" ;;
      *) defs="$K2_VARIANT_DEFS" ;;
    esac
    defs="$defs -Dscan_mfma16_dma_kernel=scan_mfma16_dma_kernel_$t"   # its own kernel name: profiles tell the builds apart
    /opt/rocm/bin/hipcc $FLAGS $defs -c $src -o $B/obj/dma_$t.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libmvf_gpu_$t.so $OTHERS $B/obj/dma_$t.o -ldl 2>/dev/null
    echo "built libmvf_gpu_$t.so ($defs)"
  ) &
  pids+=($!)
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPROBE_K2_SHARED -shared -fPIC -o $B/libprobe_k2.so $ROOT/scripts/probe_k2_w1.hip 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $B/probe_k2_w1 $ROOT/scripts/probe_k2_w1.hip 2>/dev/null
for p in "${pids[@]}"; do wait $p; done
ls -la $B/*.so
