#!/bin/bash
# round 4: s_memtime stamps of one k_deblock wavefront (workgroup 100, wave 5, its first unit): where does an iteration's time go?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in stamps stamps_nf; do
  P264AMD_STAMPS_OUT=gpurun_out/r4_$v.txt P264AMD_BENCH_NO_GOLDEN=1 P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$v.so python bench.py --steps 4 --warmup 1 --streams 1024 --no-cpu-baseline --no-extras > gpurun_out/r4_$v.json 2>/dev/null
  python - gpurun_out/r4_$v.txt <<'PY'
import sys
rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l.strip()]
rows = [r for r in rows if r[0]]
print(sys.argv[1], len(rows), "iterations")
import statistics as st
prev = None
segs = [[] for _ in range(6)]
for i, r in enumerate(rows):
    if i + 1 < len(rows):
        nxt = rows[i + 1][0]
        d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], nxt - r[5]]
        if all(0 <= x < 10**7 for x in d):
            for k in range(6): segs[k].append(d[k])
names = (sys.argv[2].split(",") if len(sys.argv) > 2 else ["land+vmcnt", "publish+Vpass", "flush+tile", "prefetch(+spin)", "Hpass", "tail"])
tot = 0
for k in range(6):
    if segs[k]:
        m = st.mean(segs[k]); tot += m
        print("  %-16s mean %8.0f  median %8.0f  p90 %8.0f  (s_memtime ticks)" % (names[k], m, st.median(segs[k]), sorted(segs[k])[int(len(segs[k]) * 0.9)]))
print("  per iteration %.0f ticks; whole band %d ticks" % (tot, rows[-1][5] - rows[0][0]))
PY
done 2>&1 | tee gpurun_out/r4_stamps.log
