OUT=gpurun_out/${1:-gradvec}; mkdir -p $OUT
for c in full_peaky_b2_f100_p100 full_peaky_s29_b2_f100_p100 full_b1_f100_p100; do for f in 0 1; do
  T2S_FOLD_QSCALE=$f timeout -k 10 600 python3 tools/grad_vector_probe.py $c 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gradvec.txt | cut -c1-220
done; done
