#!/bin/bash
# round 5: the pipeline end to end by parser threads around the box's CPU quota (16): is a CPU left for the submitting thread worth it?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_pipe_threads
for rep in 1 2; do for t in 13 14 15 16 17 18; do
  echo -n "threads $t: "; python -m p264decoder_amd.tools.pipe_bench --streams 128 --threads $t --pictures 72 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s, parse cpu', d['parse_cpu_seconds'], 's, submit', d['submit_seconds'], 's, wall', d['wall_seconds'])"
done; done 2>&1 | tee gpurun_out/r5_pipe_threads/log.txt
