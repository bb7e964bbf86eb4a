for o in 9000 10000 11000 12000; do MVF_FUZZ_OFFSET=$o python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -1; done
for v in "MVF_K2_DIRECT64=0" "MVF_QS_REFINE_PHASES=1" "MVF_QS_REFINE_PHASES=3"; do echo "== $v"; env $v python -m pytest tests -q -m gpu -k "batched or k2 or round5 or shadow or fuzz" 2>&1 | tail -1; done
echo "== MVF_I8_SHADOW_ROWS=66000"; MVF_I8_SHADOW_ROWS=66000 python -m pytest tests -q -m gpu -k 'not cfg5_100m and not prefix and not two_ranges and not streamed_int8_shadow and not sane_and_wild_norm' 2>&1 | tail -1
