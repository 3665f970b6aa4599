OUT=gpurun_out/${1:-finalparity}; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_fulllength_reference_gpu.py -m gpu -q -s > $OUT/pytest_fulllen.log 2>&1; tail -2 $OUT/pytest_fulllen.log
timeout -k 10 600 python3 tools/train_curve.py 150 64 > $OUT/train_curve.txt 2> $OUT/train_curve.err; tail -3 $OUT/train_curve.txt
