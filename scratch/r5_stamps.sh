#!/bin/bash
# round 5: s_memtime stamps of one k_deblock wavefront (workgroup 100, wave 5, first unit) at 256 and 2048 pictures per launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for st in ${@:-256 2048}; do
  v=stamps
  P264AMD_TIMING_BUILD_OK=1 P264AMD_STAMPS_OUT=gpurun_out/r5_stamps_$st.txt P264AMD_BENCH_NO_GOLDEN=1 P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$v.so python bench.py --steps 4 --warmup 1 --streams $st --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $st deblock', d['kernels']['deblock']['avg_ms'])"
  python - gpurun_out/r5_stamps_$st.txt <<'PY'
import sys, statistics as st
rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l.strip()]
rows = [r for r in rows if r[0]]
print(sys.argv[1], len(rows), "iterations")
segs = [[] for _ in range(6)]
for i, r in enumerate(rows[:-1]):
    d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], rows[i + 1][0] - r[5]]
    if all(0 <= x < 10**7 for x in d):
        for k in range(6): segs[k].append(d[k])
names = ["land+vmcnt", "publish+Vpass", "flush+tile", "prefetch(+spin)", "Hpass", "tail"]
tot = 0
for k in range(6):
    if segs[k]:
        m = st.mean(segs[k]); tot += m
        print("  %-16s mean %8.0f  median %8.0f  p90 %8.0f  (s_memtime ticks)" % (names[k], m, st.median(segs[k]), sorted(segs[k])[int(len(segs[k]) * 0.9)]))
print("  per iteration %.0f ticks; whole band %d ticks" % (tot, rows[-1][5] - rows[0][0]))
PY
done 2>&1 | tee gpurun_out/r5_stamps.log
