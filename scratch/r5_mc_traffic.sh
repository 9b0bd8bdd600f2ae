#!/bin/bash
# round 5: the MC stage's time and memory reads by locality band height and workgroups per picture (FETCH_SIZE / WRITE_SIZE of k_mc)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_mc_traffic; mkdir -p $out
for cfg in ${@:-"4:48" "3:48" "2:48" "4:96" "3:96" "5:48"}; do
  bl=${cfg%%:*}; wg=${cfg##*:}
  export P264AMD_MC_BAND_LOG2=$bl P264AMD_MC_WGS_PER_PIC=$wg
  t=$(python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels']['inter']['avg_ms'])")
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 110 rocprofv3 --pmc $ctr --kernel-include-regex "^k_mc" --output-format csv -d $out/p_${bl}_${wg}_$ctr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/log_${bl}_${wg}_$ctr.txt 2>&1 || echo "pass failed $cfg $ctr"
  done
  python3 - $out $bl $wg $t <<'PY'
import csv, glob, sys, collections
out, bl, wg, t = sys.argv[1:5]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/p_%s_%s_*/**/*counter_collection.csv" % (out, bl, wg), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
def avg(k, c): v = agg[k][c]; return sum(v) / max(1, len(v))
print("band_log2 %s wgs %s: inter %s ms | k_mc read %.2f GB write %.2f GB | k_mc_sort read %.2f write %.2f GB" % (bl, wg, t,
      2 * avg("k_mc", "FETCH_SIZE") * 1024 / 1e9, avg("k_mc", "WRITE_SIZE") * 1024 / 1e9, 2 * avg("k_mc_sort", "FETCH_SIZE") * 1024 / 1e9, avg("k_mc_sort", "WRITE_SIZE") * 1024 / 1e9))
PY
  rm -rf $out/p_${bl}_${wg}_*
done 2>&1 | tee $out/result.txt
