#!/bin/bash
# round 4: s_memtime stamps of one k_deblock_pool wavefront (workgroup 100, wave $1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in ${1:-0} ${2:-5}; do
  v=stamps_w$w
  P264AMD_STAMPS_OUT=gpurun_out/r4_pool_$v.txt P264AMD_BENCH_NO_GOLDEN=1 P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$v.so python bench.py --steps 4 --warmup 1 --streams 1024 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deblock', d['kernels']['deblock']['avg_ms'])"
  python - gpurun_out/r4_pool_$v.txt <<'PY'
import sys, statistics as st
rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l.strip()]
rows = [r for r in rows if r[0]]
print(sys.argv[1], len(rows), "steps")
names = ["vmcnt wait", "phaseA+B(left,flush,land)", "prefetch misc", "V inner", "H", "carry"]
def seg(rs):
    out = [[] for _ in range(6)]
    for i, r in enumerate(rs[:-1]):
        d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], rs[i + 1][0] - r[5]]
        if all(0 <= x < 10**7 for x in d):
            for k in range(6): out[k].append(d[k])
    return out
for lo, hi in ((0, 34), (34, 120), (120, 160)):
    sg = seg(rows[lo:hi])
    if not sg[0]: continue
    print("  steps %d..%d: " % (lo, hi) + "  ".join("%s %.0f" % (names[k], st.mean(sg[k])) for k in range(6)) + "   total %.0f" % sum(st.mean(sg[k]) for k in range(6)))
print("  whole band %d ticks" % (rows[-1][5] - rows[0][0]))
PY
done 2>&1 | tee gpurun_out/r4_stamps2.log
