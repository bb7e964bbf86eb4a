#!/bin/bash
# A/B of MVF_K2_DIRECT64 (development aid): whole-search wall times and the first scan launch of cfg3 / cfg5 / cfg4
for d in 0 1; do
  export MVF_K2_DIRECT64=$d
  echo "== MVF_K2_DIRECT64=$d"
  python scripts/probe_wall_jitter.py 0 0 14 2>/dev/null | tail -1
  python scripts/probe_wall_jitter.py 1 0 14 12500000,1024,0,1024 2>/dev/null | tail -1
  python scripts/probe_wall_jitter.py 2 0 14 50000000,768,1,256 2>/dev/null | tail -1
  bash scripts/trace_search.sh d64_$d > /dev/null; grep "scan launches" gpurun_out/d64_${d}_kernels.txt
  bash scripts/trace_search.sh d64c4_$d cfg4 > /dev/null; grep "scan launches" gpurun_out/d64c4_${d}_kernels.txt
done
