OUT=gpurun_out/${1:-fold}; mkdir -p $OUT
for f in 0 1; do
  T2S_FOLD_QSCALE=$f timeout -k 10 900 python3 -m pytest tests/test_fulllength_reference_gpu.py -m gpu -q -s -k "bf16" > $OUT/pytest_fold$f.log 2>&1
  echo "=== T2S_FOLD_QSCALE=$f"; grep -E "passed|failed|max abs logit|beyond 3|gradient slices|bf16 gradient norms|FAILED|Error" $OUT/pytest_fold$f.log | cut -c1-900
done
