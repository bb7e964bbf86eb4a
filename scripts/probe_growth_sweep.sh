for g in 4 5 6 8; do
  export MVF_K2_GROWTH=$g
  python scripts/probe_wall_jitter.py 0 0 14 | tail -1
  python scripts/probe_wall_jitter.py 1 0 14 12500000,1024,0,1024 | tail -1
  python scripts/probe_wall_jitter.py 2 0 14 50000000,768,1,256 | tail -1
done
