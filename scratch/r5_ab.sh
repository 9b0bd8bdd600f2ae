#!/bin/bash
# round 5: A / B of library variants on one box.  usage: r5_ab.sh "<variant ...>" [streams ...]   (variant "cur" = the built library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_ab; mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_kat.py tests/test_gpu_synth.py tests/test_gpu_seam_fuzz.py tests/test_gpu_bipred.py tests/test_gpu_main_profile.py tests/test_gpu_batch_shapes.py -x -q -k "not bench_s_own" > $out/tests.log 2>&1 || { echo "gpu tests failed"; tail -30 $out/tests.log; exit 1; }
  tail -1 $out/tests.log
fi
for st in ${2:-2048}; do
for rep in 1 2; do
for v in $1; do
  L=""; [ $v != cur ] && L=$GRAFT_REPO_ROOT/scratch/lib_$v.so
  echo -n "$v streams $st: "
  P264AMD_TIMING_BUILD_OK=1 P264AMD_BENCH_NO_GOLDEN=${NOGOLD:-} P264AMD_LIB=$L python bench.py --steps ${STEPS:-12} --warmup 2 --streams $st --no-cpu-baseline --no-extras 2>$out/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), d['ms_per_step'], 'inter', k['inter']['avg_ms'], 'intra', k['intra']['avg_ms'], 'deblock', k['deblock']['avg_ms'], 'golden', d['golden_check'].get('checked'))" || tail -3 $out/err_$v.log
done
done
done 2>&1 | tee $out/ab_$(date +%H%M%S).log
