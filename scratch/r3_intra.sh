#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 || exit 1
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}, d['ms_per_step'], d['value']); print({k:v.get('value') for k,v in d['extras'].items()})"
