#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in st2nt st2n; do
  out=gpurun_out/r3_wr_$lib; rm -rf $out; mkdir -p $out
  for ctr in FETCH_SIZE WRITE_SIZE; do
    P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$lib.so timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out/$ctr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/$ctr.log 2>&1
  done
  python3 - $out $lib <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_mc"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: {c: round(sum(v) / len(v) / 1e6, 3) for c, v in d.items()} for k, d in agg.items()})
PY
done
bash scratch/variants_run.sh "st2nt st2n st2nt st2n"
