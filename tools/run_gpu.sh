#!/bin/bash
# ONE parametrised GPU-call script (round 5: replaced the per-call r4_run_*.sh files; round 6: renamed from r5_run.sh, steps added).
#   gpurun --timeout 1200 -- 'bash tools/run_gpu.sh <tag> <step> [<step> ...]'
# steps: fulllen (reference-pinned tests at L = 10 132), suite (whole -m gpu suite), bench (bench line), benchq (bench without cpu baseline),
#        gemm (own GEMM tests + probe), attn (attention tests + probe), smoke
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5}; shift
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
set -e
cd $REPO
for STEP in "$@"; do
  echo "=== step $STEP" | tee -a $OUT/steps.log
  case $STEP in
    peaky)   # round 6: both reference-pinned full-length fixtures (flat + peaky / unequal samples), every test, no -x: the maxima are the result
             timeout -k 10 900 python3 -m pytest tests/test_fulllength_reference_gpu.py -m gpu -q -s > $OUT/pytest_fulllen.log 2>&1 || true
             grep -E "passed|failed|vs the reference|max abs logit|argmax indices|worst comparable|FAILED|Error" $OUT/pytest_fulllen.log | cut -c1-260
             timeout -k 10 900 python3 -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "fp64_heads" > $OUT/pytest_twins.log 2>&1 || true
             grep -E "passed|failed|peaky twin|FAILED|Error|assert" $OUT/pytest_twins.log | cut -c1-300 ;;
    budget)  timeout -k 10 900 python3 tools/error_budget.py ${BUDGET_CASES} > $OUT/error_budget.txt 2>&1 || { tail -30 $OUT/error_budget.txt; exit 1; }; cut -c1-220 $OUT/error_budget.txt ;;
    diag)    # round 6: the multi-GPU diagnosability fields (two gloo ranks on one card) and the neighbour-stream rehearsal of the hand-off
             timeout -k 10 900 python3 -m pytest tests/test_bench_gpu.py tests/test_handoff_guard_gpu.py tests/test_ddp_gpu.py -m gpu -q -s > $OUT/pytest_diag.log 2>&1 || { grep -E "passed|failed|FAILED|Error|assert" $OUT/pytest_diag.log | cut -c1-300; tail -40 $OUT/pytest_diag.log; exit 1; }
             grep -E "passed|failed|neighbour" $OUT/pytest_diag.log | cut -c1-250 ;;
    fbabl)   # round 6: cycle stamps of the product and of its timing-only ablations (tools/ablate/make_fb_ablation.py), libraries prebuilt by
             # tools/ablate/build_fb_libs.sh stamp_<what>=tools/ablate/_build/src/stamp_<what>.hip (they travel with the snapshot)
             rm -f $OUT/fb_ablation_stamps.txt
             for w in ${FB_ABLS:-product noreads nodq nofinal nopin}; do
               echo "=== $w" >> $OUT/fb_ablation_stamps.txt
               FB_LIB=$REPO/tools/ablate/_build/libt2s_fb_stamp_$w.so FB_SRC=product timeout -k 10 300 python3 tools/fused_stamps2.py 8 0.7 0.1 >> $OUT/fb_ablation_stamps.txt 2>&1 || { tail -20 $OUT/fb_ablation_stamps.txt; exit 1; }
             done
             cat $OUT/fb_ablation_stamps.txt ;;
    forcedist) # the RCCL process-group path with ONE rank (communicator, bucket all-reduces, all_gather of the rank report, MAX reduction) on a 1-GPU box
             T2S_BENCH_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dropout0 > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err || { tail -30 $OUT/bench_forcedist.err; exit 1; }
             python3 -c "import json,sys; d=json.loads(open('$OUT/bench_forcedist.json').read().strip().splitlines()[-1]); print('force-dist:', d['backend'], d['ms_per_step'], 'ms/step; multi_gpu:', json.dumps(d['multi_gpu'])[:900])" ;;
    gemmab)  # round 6: the product GEMM family vs a variant source (GEMM_VARIANT=name, tools/ablate/variants/gemm_bf16_<name>.hip; built beforehand
             # by tools/ablate/gemm_variant.sh): every GEMM test through the C ABI on the variant library, then the probe on both, twice
             T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_gemm_${GEMM_VARIANT}.so timeout -k 10 600 python3 -m pytest tests/test_gemm_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or bert_layer" > $OUT/pytest_gemm_variant.log 2>&1 || { tail -40 $OUT/pytest_gemm_variant.log; exit 1; }; tail -2 $OUT/pytest_gemm_variant.log
             rm -f $OUT/gemm_ab.txt
             for rep in 1 2; do
               echo "== product" >> $OUT/gemm_ab.txt
               timeout -k 10 300 python3 tools/gemm_probe5.py 2>&1 | grep -E "^NT|^dgrad|^FFN|own gemm_nt" >> $OUT/gemm_ab.txt
               echo "== variant ${GEMM_VARIANT}" >> $OUT/gemm_ab.txt
               T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_gemm_${GEMM_VARIANT}.so timeout -k 10 300 python3 tools/gemm_probe5.py 2>&1 | grep -E "^NT|^dgrad|^FFN|own gemm_nt" >> $OUT/gemm_ab.txt
             done
             cat $OUT/gemm_ab.txt | cut -c1-150 ;;
    suiteall) # the whole -m gpu suite WITHOUT -x: every failure is listed (exit status ignored; read the summary)
             timeout -k 10 1150 python3 -m pytest tests -m gpu -q > $OUT/pytest_suite_all.log 2>&1 || true
             grep -E "^FAILED|^ERROR| passed| failed" $OUT/pytest_suite_all.log | cut -c1-250 ;;
    fulllen) timeout -k 10 900 python3 -m pytest tests/test_fulllength_reference_gpu.py -m gpu -x -q -s > $OUT/pytest_fulllen.log 2>&1 || { tail -60 $OUT/pytest_fulllen.log; exit 1; }; tail -30 $OUT/pytest_fulllen.log ;;
    suite)   timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_suite.log 2>&1 || { tail -60 $OUT/pytest_suite.log; exit 1; }; tail -3 $OUT/pytest_suite.log ;;
    smoke)   timeout -k 10 300 python3 __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || { tail -30 $OUT/smoke.log; exit 1; }; tail -1 $OUT/smoke.log ;;
    bench)   timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -30 $OUT/bench.err; exit 1; }; tail -c 1500 $OUT/bench.json ;;
    benchq)  timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-dropout0 > $OUT/benchq.json 2> $OUT/benchq.err || { tail -30 $OUT/benchq.err; exit 1; }; tail -c 1200 $OUT/benchq.json ;;
    gemm)    timeout -k 10 600 python3 -m pytest tests/test_gemm_gpu.py -m gpu -x -q -s > $OUT/pytest_gemm.log 2>&1 || { tail -60 $OUT/pytest_gemm.log; exit 1; }; tail -5 $OUT/pytest_gemm.log
             timeout -k 10 600 python3 tools/gemm_probe5.py > $OUT/gemm_probe.txt 2>&1 || { tail -40 $OUT/gemm_probe.txt; exit 1; }; cat $OUT/gemm_probe.txt ;;
    attn)    timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py -m gpu -x -q > $OUT/pytest_attn.log 2>&1 || { tail -40 $OUT/pytest_attn.log; exit 1; }; tail -2 $OUT/pytest_attn.log
             timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 > $OUT/attn_probe.txt 2>&1 || { tail -30 $OUT/attn_probe.txt; exit 1; }; cut -c1-200 $OUT/attn_probe.txt ;;
    guard)   timeout -k 10 600 python3 -m pytest tests/test_handoff_guard_gpu.py -m gpu -x -q > $OUT/pytest_guard.log 2>&1 || { tail -60 $OUT/pytest_guard.log; exit 1; }; tail -3 $OUT/pytest_guard.log ;;
    benchab) # same box, back to back: the step with the library GEMMs everywhere, then with the own GEMM family (default)
             T2S_OWN_GEMM=none timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-dropout0 > $OUT/bench_lib.json 2> $OUT/bench_lib.err || { tail -30 $OUT/bench_lib.err; exit 1; }
             timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-dropout0 > $OUT/bench_own.json 2> $OUT/bench_own.err || { tail -30 $OUT/bench_own.err; exit 1; }
             T2S_OWN_GEMM=none timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-dropout0 > $OUT/bench_lib2.json 2> $OUT/bench_lib2.err || { tail -30 $OUT/bench_lib2.err; exit 1; }
             timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-dropout0 > $OUT/bench_own2.json 2> $OUT/bench_own2.err || { tail -30 $OUT/bench_own2.err; exit 1; }
             python3 - <<PYEOF
import json
for n in ("bench_lib", "bench_own", "bench_lib2", "bench_own2"):
    d = json.loads(open("$OUT/" + n + ".json").read().strip().splitlines()[-1])
    print("%-11s %8.2f ms/step  %7.2f samples/s  bwd-group frac %.3f  loss_step0 %s" % (n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("loss_step0")))
PYEOF
             ;;
    dropdyn) timeout -k 10 1100 python3 tools/dropout_dynamics.py ${DD_SEEDS:-5} ${DD_STEPS:-200} > $OUT/dropout_dynamics.txt 2> $OUT/dropout_dynamics.err || { tail -30 $OUT/dropout_dynamics.err; exit 1; }; cat $OUT/dropout_dynamics.txt ;;
    stamps2) # FB_SRC=ilv384 stamps the 384-key interleaved variant, default the 256-key variant
             timeout -k 10 600 python3 tools/fused_stamps2.py 8 0.7 0.1 > $OUT/fused_stamps_ilv.txt 2>&1 || { tail -30 $OUT/fused_stamps_ilv.txt; exit 1; }
             timeout -k 10 600 python3 tools/fused_stamps2.py 8 0.7 0.0 >> $OUT/fused_stamps_ilv.txt 2>&1 || { tail -30 $OUT/fused_stamps_ilv.txt; exit 1; }; cat $OUT/fused_stamps_ilv.txt ;;
    fwdstamps) timeout -k 10 600 python3 tools/fwd_stamps.py 8 0.7 0.1 > $OUT/fwd_stamps.txt 2>&1 || { tail -30 $OUT/fwd_stamps.txt; exit 1; }
             timeout -k 10 600 python3 tools/fwd_stamps.py 8 0.7 0.0 >> $OUT/fwd_stamps.txt 2>&1 || { tail -30 $OUT/fwd_stamps.txt; exit 1; }; cat $OUT/fwd_stamps.txt ;;
    fwdab)   # same box, interleaved: the product forward vs a variant source (FWD_VARIANT=name, tools/ablate/variants/attn_fwd_<name>.hip)
             # (built beforehand in the authoring container when possible: the .so travels with the snapshot and no GPU-minute goes into hipcc)
             [ tools/ablate/_build/libt2s_fwd_${FWD_VARIANT}.so -nt tools/ablate/variants/attn_fwd_${FWD_VARIANT}.hip ] || bash tools/ablate/fwd_variant.sh ${FWD_VARIANT} tools/ablate/variants/attn_fwd_${FWD_VARIANT}.hip > $OUT/fwd_variant_build.log 2>&1 || { tail -20 $OUT/fwd_variant_build.log; exit 1; }
             T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fwd_${FWD_VARIANT}.so timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_dropout_gpu.py -m gpu -x -q -k "fwd or forward or dropout or attention" > $OUT/pytest_fwd_variant.log 2>&1 || { tail -30 $OUT/pytest_fwd_variant.log; exit 1; }; tail -2 $OUT/pytest_fwd_variant.log
             rm -f $OUT/fwd_ab.txt
             for rep in 1 2 3; do
               echo "== product" >> $OUT/fwd_ab.txt
               T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 2>&1 | grep "fwd " | sed -E 's/ \| bwd.*$//' >> $OUT/fwd_ab.txt
               echo "== variant ${FWD_VARIANT}" >> $OUT/fwd_ab.txt
               T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fwd_${FWD_VARIANT}.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 2>&1 | grep "fwd " | sed -E 's/ \| bwd.*$//' >> $OUT/fwd_ab.txt
             done
             cat $OUT/fwd_ab.txt | cut -c1-150 ;;
    fbab)    # same box, interleaved: the product fused backward vs a variant source (FB_VARIANT=name, tools/ablate/variants/attn_bwd_fused_bf16_<name>.hip)
             [ tools/ablate/_build/libt2s_fbv_${FB_VARIANT}.so -nt tools/ablate/variants/attn_bwd_fused_bf16_${FB_VARIANT}.hip ] || bash tools/ablate/fb_variant.sh ${FB_VARIANT} tools/ablate/variants/attn_bwd_fused_bf16_${FB_VARIANT}.hip > $OUT/fb_variant_build.log 2>&1 || { tail -20 $OUT/fb_variant_build.log; exit 1; }
             # the variant must be CORRECT before its time means anything: every attention / dropout / hand-off test on the variant library
             T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fbv_${FB_VARIANT}.so timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py tests/test_handoff_guard_gpu.py -m gpu -x -q -k "attention or attn or fused or dropout or handoff or bwd or backward" > $OUT/pytest_fb_variant.log 2>&1 || { tail -40 $OUT/pytest_fb_variant.log; exit 1; }; tail -2 $OUT/pytest_fb_variant.log
             rm -f $OUT/fb_ab.txt
             for dp in 0.1 0.0; do for rep in 1 2 3; do
               echo "== product, dropout $dp" >> $OUT/fb_ab.txt
               T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $dp 2>&1 | grep "bwd fused/handoff  " >> $OUT/fb_ab.txt
               echo "== variant ${FB_VARIANT}, dropout $dp" >> $OUT/fb_ab.txt
               T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fbv_${FB_VARIANT}.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $dp 2>&1 | grep "bwd fused/handoff  " >> $OUT/fb_ab.txt
             done; done
             cat $OUT/fb_ab.txt | cut -c1-150 ;;
    fwdabm)  # several forward variants (FWD_VARIANTS="a b c", libraries prebuilt by tools/ablate/fwd_variant.sh) against the product, same box, interleaved
             rm -f $OUT/fwd_abm.txt
             for v in ${FWD_VARIANTS}; do
               T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fwd_$v.so timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_dropout_gpu.py -m gpu -x -q -k "fwd or forward or dropout or attention" > $OUT/pytest_fwdv_$v.log 2>&1 || { tail -30 $OUT/pytest_fwdv_$v.log; exit 1; }
               echo "variant $v: $(tail -1 $OUT/pytest_fwdv_$v.log)" | tee -a $OUT/fwd_abm.txt
             done
             for dp in ${FWD_DPS:-0.1}; do for rep in $(seq 1 ${FWD_REPS:-3}); do
               echo "== product, dropout $dp" >> $OUT/fwd_abm.txt
               T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $dp 2>&1 | grep "fwd " | sed -E 's/ \| bwd.*$//' >> $OUT/fwd_abm.txt
               for v in ${FWD_VARIANTS}; do
                 echo "== variant $v, dropout $dp" >> $OUT/fwd_abm.txt
                 T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fwd_$v.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $dp 2>&1 | grep "fwd " | sed -E 's/ \| bwd.*$//' >> $OUT/fwd_abm.txt
               done
             done; done
             cat $OUT/fwd_abm.txt | cut -c1-150 ;;
    fbabm)   # several fused-backward variants (FB_VARIANTS="a b c", libraries prebuilt by tools/ablate/fb_variant.sh) against the product, same box,
             # interleaved, dropout 0.1 (FB_DPS overrides): the fused-backward tests on each variant first, then FB_REPS (2) rounds of the probe
             rm -f $OUT/fb_abm.txt
             for v in ${FB_VARIANTS}; do
               T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fbv_$v.so timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py -m gpu -x -q -k "fused or handoff" > $OUT/pytest_fbv_$v.log 2>&1 || { tail -30 $OUT/pytest_fbv_$v.log; exit 1; }
               echo "variant $v: $(tail -1 $OUT/pytest_fbv_$v.log)" | tee -a $OUT/fb_abm.txt
             done
             for dp in ${FB_DPS:-0.1}; do for rep in $(seq 1 ${FB_REPS:-2}); do
               echo "== product, dropout $dp" >> $OUT/fb_abm.txt
               T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $dp 2>&1 | grep "bwd fused/handoff  " >> $OUT/fb_abm.txt
               for v in ${FB_VARIANTS}; do
                 echo "== variant $v, dropout $dp" >> $OUT/fb_abm.txt
                 T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fbv_$v.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $dp 2>&1 | grep "bwd fused/handoff  " >> $OUT/fb_abm.txt
               done
             done; done
             cat $OUT/fb_abm.txt | cut -c1-150 ;;
    scope)   # hand-off cache-scope variants (SCOPE_VARIANTS="name:ld:st ..."): correctness subset, interleaved timing, HBM counters
             rm -f $OUT/scope.txt
             for v in ${SCOPE_VARIANTS:-st0:16:0}; do
               IFS=: read name ld st <<< "$v"
               [ -f tools/ablate/variants/attn_bwd_fused_bf16_scope_$name.hip ] || python3 tools/ablate/make_fb_scope.py $name $ld $st > /dev/null
               [ -f tools/ablate/_build/libt2s_fbv_scope_$name.so ] || bash tools/ablate/fb_variant.sh scope_$name tools/ablate/variants/attn_bwd_fused_bf16_scope_$name.hip > $OUT/scope_build_$name.log 2>&1 || { tail -20 $OUT/scope_build_$name.log; exit 1; }
               echo "== variant $name (load aux $ld, store aux $st): correctness" >> $OUT/scope.txt
               T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fbv_scope_$name.so timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py -q -k "fused" 2>&1 | tail -4 >> $OUT/scope.txt
             done
             for rep in 1 2 3; do
               echo "== product" >> $OUT/scope.txt
               T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 2>&1 | grep "bwd fused/handoff  " >> $OUT/scope.txt || true
               for v in ${SCOPE_VARIANTS:-st0:16:0}; do
                 IFS=: read name ld st <<< "$v"
                 echo "== variant $name" >> $OUT/scope.txt
                 T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fbv_scope_$name.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 2>&1 | grep "bwd fused/handoff  " >> $OUT/scope.txt || true
               done
             done
             cd /tmp && export TMPDIR=/tmp
             for v in product ${SCOPE_VARIANTS:-st0:16:0}; do
               IFS=: read name ld st <<< "$v"
               lib=$REPO/vitxt_gqa_amd/libt2s_hip.so; [ $name != product ] && lib=$REPO/tools/ablate/_build/libt2s_fbv_scope_$name.so
               for c in FETCH_SIZE WRITE_SIZE; do
                 T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$lib timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/scope_pmc_${name}_$c -- python3 $REPO/tools/attn_probe.py 8 10120 0.7 12 2 0.1 > $OUT/scope_pmc_${name}_$c.log 2>&1 || true
               done
               python3 $REPO/tools/ablate/scope_pmc_sum.py $OUT scope_pmc_$name >> $OUT/scope.txt
             done
             cd $REPO
             cat $OUT/scope.txt | cut -c1-170 ;;
    scopeab) # the shipped XCD-local running sums vs the write-through form (T2S_FB_HANDOFF_SCOPE=agent), same library, interleaved; then HBM counters
             rm -f $OUT/scopeab.txt
             for rep in 1 2 3; do for sc in xcd agent; do
               echo "== sums: $sc" >> $OUT/scopeab.txt
               T2S_FB_HANDOFF_SCOPE=$sc T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 2>&1 | grep "bwd fused/handoff  " >> $OUT/scopeab.txt || true
             done; done
             cd /tmp && export TMPDIR=/tmp
             for sc in xcd agent; do for c in FETCH_SIZE WRITE_SIZE; do
               T2S_FB_HANDOFF_SCOPE=$sc T2S_PROBE_FORMS=shipped timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/scopeab_pmc_${sc}_$c -- python3 $REPO/tools/attn_probe.py 8 10120 0.7 12 2 0.1 > $OUT/scopeab_pmc_${sc}_$c.log 2>&1 || true
             done; python3 $REPO/tools/ablate/scope_pmc_sum.py $OUT scopeab_pmc_$sc >> $OUT/scopeab.txt; done
             cd $REPO
             cat $OUT/scopeab.txt | cut -c1-170 ;;
    scopestep) # same box, back to back: the whole step with the write-through running sums (round-4 form), then the XCD-local ones (shipped), twice
             for i in 1 2; do for sc in agent xcd; do
               T2S_FB_HANDOFF_SCOPE=$sc timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-dropout0 > $OUT/bench_${sc}$i.json 2> $OUT/bench_${sc}$i.err || { tail -30 $OUT/bench_${sc}$i.err; exit 1; }
             done; done
             python3 - $OUT <<'PYEOF' | tee $OUT/scope_step_ab.txt
import json, sys
out = sys.argv[1]
for name in ("agent1", "xcd1", "agent2", "xcd2"):
    b = json.loads([l for l in open("%s/bench_%s.json" % (out, name)) if l.startswith("{")][-1])
    r = b["roofline"]
    print("%-7s %8.1f ms/step  %6.2f samples/s   attention backward %6.1f ms/step (frac %.3f)  forward attention %6.1f ms/step" % (
        name, b["ms_per_step"], b["value"], r["ms_per_step"], r["frac"], b.get("roofline_fwd", {}).get("ms_per_step", float("nan"))))
PYEOF
             ;;
    fbstep)  # same box, back to back: the whole step with a variant fused backward (FB_VARIANT), then the product, twice
             bash tools/ablate/fb_variant.sh ${FB_VARIANT} tools/ablate/variants/attn_bwd_fused_bf16_${FB_VARIANT}.hip > $OUT/fb_variant_build.log 2>&1 || { tail -20 $OUT/fb_variant_build.log; exit 1; }
             for i in 1 2; do
               T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fbv_${FB_VARIANT}.so timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-dropout0 > $OUT/bench_variant$i.json 2> $OUT/bench_variant$i.err || { tail -30 $OUT/bench_variant$i.err; exit 1; }
               timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-dropout0 > $OUT/bench_product$i.json 2> $OUT/bench_product$i.err || { tail -30 $OUT/bench_product$i.err; exit 1; }
             done
             python3 - $OUT ${FB_VARIANT} <<'PYEOF' | tee $OUT/fb_step_ab.txt
import json, sys
out, var = sys.argv[1], sys.argv[2]
for name in ("variant1", "product1", "variant2", "product2"):
    b = json.loads([l for l in open("%s/bench_%s.json" % (out, name)) if l.startswith("{")][-1])
    r = b["roofline"]
    print("%-22s %8.1f ms/step  %6.2f samples/s   attention backward %6.1f ms/step (frac %.3f)  forward attention %6.1f ms/step" % (
        name.replace("variant", var + " "), b["ms_per_step"], b["value"], r["ms_per_step"], r["frac"], b.get("roofline_fwd", {}).get("ms_per_step", float("nan"))))
PYEOF
             ;;
    ntmap)   # the NT GEMM's work map: N-tiles per group, timing + one FETCH_SIZE pass (tools/gemm_nt_map_probe.py)
             timeout -k 10 600 python3 -m pytest tests/test_gemm_gpu.py -m gpu -x -q > $OUT/pytest_gemm.log 2>&1 || { tail -60 $OUT/pytest_gemm.log; exit 1; }; tail -2 $OUT/pytest_gemm.log
             timeout -k 10 600 python3 tools/gemm_nt_map_probe.py > $OUT/gemm_nt_map.txt 2>&1 || { tail -40 $OUT/gemm_nt_map.txt; exit 1; }
             cd /tmp && export TMPDIR=/tmp
             PROBE_PMC=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/ntmap_pmc -- python3 $REPO/tools/gemm_nt_map_probe.py > $OUT/ntmap_pmc.log 2>&1 || true
             cd $REPO
             python3 - $OUT >> $OUT/gemm_nt_map.txt <<'PYEOF'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/ntmap_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm_nt_bf16" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:60], float(r["Counter_Value"])))
rows.sort()
print("# FETCH_SIZE per dispatch of the NT kernel, in dispatch order (PROBE_PMC=1: per form and width one warm-up call and one timed call); GB = 2 x KB x 1024")
for d, k, v in rows:
    print("  dispatch %4d  %-60s  %7.2f GB" % (d, k, 2 * v * 1024 / 1e9))
PYEOF
             cat $OUT/gemm_nt_map.txt | cut -c1-150 ;;
    *) echo "unknown step $STEP"; exit 2 ;;
  esac
done
