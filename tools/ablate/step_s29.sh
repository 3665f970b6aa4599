mkdir -p gpurun_out/${1:-r6an}
timeout -k 10 1100 python3 -m pytest tests/test_fulllength_reference_gpu.py -m gpu -q -s -k "gradients_bf16 and peaky" > gpurun_out/${1:-r6an}/pytest_s29.log 2>&1
grep -E "passed|failed|gradient slices|bf16 gradient norms|beyond 3|FAILED|Error|AssertionError" gpurun_out/${1:-r6an}/pytest_s29.log | cut -c1-1200
