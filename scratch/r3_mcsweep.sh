#!/bin/bash
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels']['inter']['avg_ms'], end=' ')"; }
for rep in 1 2; do
for cfg in "4 0" "5 0" "3 0" "4 40" "4 56" "4 64" "5 56"; do
  set -- $cfg
  export P264AMD_MC_BAND_LOG2=$1; if [ "$2" = "0" ]; then unset P264AMD_MC_WGS_PER_PIC; else export P264AMD_MC_WGS_PER_PIC=$2; fi
  echo -n "band_log2=$1 wgs=$2: "; run; run; echo
done; done
