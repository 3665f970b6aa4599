mkdir -p gpurun_out/r6w
# (the old library is built first, in the authoring container:  d=$(mktemp -d); git archive <commit> vitxt_gqa_amd/csrc include | tar -x -C $d; cd $d/vitxt_gqa_amd/csrc;
#  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -shared -w -o <repo>/tools/ablate/_build/libt2s_<commit>.so *.hip *.cpp)
for rep in 1 2; do
  T2S_HIP_LIB=$PWD/tools/ablate/_build/libt2s_e7e0c8f.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-dropout0 > gpurun_out/r6w/old_$rep.json 2> gpurun_out/r6w/old_$rep.err || exit 1
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-dropout0 > gpurun_out/r6w/new_$rep.json 2> gpurun_out/r6w/new_$rep.err || exit 1
done
python3 - <<'PY'
import json
for n in ("old_1","new_1","old_2","new_2"):
    d=json.loads(open("gpurun_out/r6w/%s.json"%n).read().strip().splitlines()[-1])
    print("%-6s %8.2f ms/step %7.2f samples/s  bwd frac %.4f  fwd frac %.4f  loss %s" % (n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline_fwd"]["frac"], d.get("loss")))
PY
