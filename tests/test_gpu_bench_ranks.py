"""`python bench.py --gpus N` as the driver types it, N > 1, rehearsed on the box's one MI355X: bench.py itself starts the
ranks (a parent that never touches the GPU runs `python -m torch.distributed.run` as a child process), both ranks share device 0
(P264AMD_BENCH_DEVICE) and meet over gloo (P264AMD_BENCH_BACKEND) instead of RCCL - two ranks cannot share a device in one RCCL
communicator.  What is checked is the N > 1 path itself: n_gpus, every rank's streams counted in `value`, every rank's golden
stream hashed against the reference decoder, and the config-5 fan-out leg (two ranks, TCP transport with its device entry
points on) on config 5's own kind of stream."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_gpus_2_starts_its_own_ranks(lib):
    env = dict(os.environ, P264AMD_BENCH_DEVICE="0", P264AMD_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    S, K, W = 6, 2, 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", str(S), "--steps", str(K), "--warmup", str(W),
                          "--no-cpu-baseline", "--no-extras"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                             # ONE JSON line, rank 0's
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["steps"] == K and r["warmup"] == W
    assert r["golden_check"]["checked"]
    assert [x["rank"] for x in r["ranks"]] == [0, 1] and all(x["golden_checked"] and x["frames"] == S * K for x in r["ranks"])
    assert r["config"]["pictures_per_step"] == 2 * S
    assert abs(r["value"] - 2 * S * K / (r["ms_per_step"] * K * 1e-3)) < 0.01 * r["value"]      # both ranks' pictures over the slowest rank's clock
    fan = r["extras"]["fanout_config5"]                            # config 5's stream through two ranks, scatter / gather
    assert "error" not in fan and "worker_errors" not in fan, fan
    assert fan["workload"] == "main_1080p_cabac_ipb" and fan["world"] == 2 and fan["all_pictures_match_oracle"]
    assert fan["pictures_on_other_ranks"] == fan["pictures"] // 2 and fan["worker_rounds_on_the_device_road"] > 0
