#!/bin/bash
# bench stage times for variations of the synthetic stream (what bounds the MC stage?)
run() { echo -n "[$1] "; P264AMD_BENCH_SYNTH_EXTRA="$1" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), d['ms_per_step'], 'inter', k['inter']['avg_ms'], 'intra', k['intra']['avg_ms'], 'deblock', k['deblock']['avg_ms'], 'alg GB', round(k['inter']['algorithmic_bytes']/1e9,2))"; }
run ""
run "--mvmax 0"
run "--mvmax 0 --coded 0"
run "--coded 0"
run "--mvmax 4"
run "--mvmax 256"
