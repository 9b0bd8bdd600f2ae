#!/bin/bash
# round 6: s_memtime stamps of one k_deblock wavefront (workgroup 100, wave 4 = band 4 of a picture that has the CU to itself) at 256
# pictures per launch: the hand-over between bands through memory (lib_stampsA: -DDB_LDS_CROSS=0) against through LDS (lib_stampsB)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export P264AMD_TIMING_BUILD_OK=1
for v in stampsA stampsB; do
  P264AMD_STAMPS_OUT=gpurun_out/r6_$v.txt P264AMD_BENCH_NO_GOLDEN=1 P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$v.so python3 bench.py --steps 4 --warmup 1 --streams 256 --no-cpu-baseline --no-extras --no-live-counters 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v deblock', d['kernels']['deblock']['avg_ms'])"
  python3 - gpurun_out/r6_$v.txt <<'PY'
import sys, statistics as st
rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l.strip()]
rows = [r for r in rows if r[0]]
print(sys.argv[1], len(rows), "iterations")
names = ["land+vmcnt", "publish+Vpass", "flush+tile", "prefetch(+spin)", "wait+Hpass", "tail"]
def seg(rs):
    out = [[] for _ in range(6)]
    for i, r in enumerate(rs[:-1]):
        d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], rs[i + 1][0] - r[5]]
        if all(0 <= x < 10**7 for x in d):
            for k in range(6): out[k].append(d[k])
    return out
for lo, hi in ((0, 10), (10, 60), (60, 120), (120, 128)):
    sg = seg(rows[lo:hi])
    if not sg[0]: continue
    print("  iterations %d..%d: " % (lo, hi) + "  ".join("%s %.0f" % (names[k], st.mean(sg[k])) for k in range(6)) + "   total %.0f" % sum(st.mean(sg[k]) for k in range(6)))
print("  whole band %d ticks, first stamp %d" % (rows[-1][5] - rows[0][0], rows[0][0] % 10**9))
PY
done 2>&1 | tee gpurun_out/r6_stamps.log
