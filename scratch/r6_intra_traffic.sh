#!/bin/bash
# round 6: HBM traffic of the k_intra_sparse launch (intra roles + the edge-info role) - FETCH_SIZE / WRITE_SIZE passes, bench's batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_intra_traffic; rm -rf $out; mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-include-regex "k_intra_sparse" --output-format csv -d $out/$ctr -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $out/$ctr.log 2>&1 || { echo "pass $ctr failed"; tail -3 $out/$ctr.log; exit 1; }
done
python3 - <<'PY'
import csv, glob
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/r6_intra_traffic/%s/**/*counter_collection.csv" % c, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "k_intra_sparse" in r["Kernel_Name"]]
    tot[c] = sum(v) / len(v)
print(tot, "HBM bytes per launch: %.3f GB" % ((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / 1e9))
PY
