#!/bin/bash
# A/B of MVF_QS_REFINE_PHASES (development aid): whole-search wall times of cfg3 / the cfg5 shard, and cfg3's kernels
for r in 1 2 3 1 2; do
  export MVF_QS_REFINE_PHASES=$r
  echo "== MVF_QS_REFINE_PHASES=$r"
  python scripts/probe_wall_jitter.py 0 0 14 2>/dev/null | tail -1
  python scripts/probe_wall_jitter.py 1 0 14 12500000,1024,0,1024 2>/dev/null | tail -1
done
for r in 1 2; do
  export MVF_QS_REFINE_PHASES=$r
  bash scripts/trace_search.sh rp_$r > /dev/null; grep "last search\|scan launches\|scatter_cand l\|compact_margin l\|rescore_ l" gpurun_out/rp_${r}_kernels.txt
done
