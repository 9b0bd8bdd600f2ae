"""N>1 path on CPU: two processes (gloo) shard independent streams round-robin, each decodes its
own (parser -> CPU oracle here, HIP on the GPU box), nothing but the barrier / clock / result
gather crosses ranks, and the union equals the single-process result."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time, json
sys.path.insert(0, %(root)r)
from p264decoder_amd import Parser, shard
from tests import oracle_bind, synth_cases
from tests.conftest import frame_sha256
rank, local_rank, world = shard.init("gloo")
ora = oracle_bind.load()
cases = ["tiny_1x1", "row_1xN", "col_Nx1", "qp38", "nodeblock"]          # five independent "streams"
mine = shard.streams_of_rank(len(cases), rank, world)
shard.barrier()
t0 = time.perf_counter()
res = {}
for s in mine:
    parser = Parser(quiet=True)
    pics = parser.parse_stream(synth_cases.stream_bytes(cases[s]))
    store = oracle_bind.FrameStore(pics[0].mb_w, pics[0].mb_h, parser.slots)
    res[s] = [frame_sha256(*oracle_bind.reconstruct(ora, store, p)) for p in pics]
shard.barrier()
el = shard.max_over_ranks(time.perf_counter() - t0)
allres = shard.gather_objects(res)
if rank == 0:
    merged = {}
    for r in allres: merged.update(r)
    ok = all(merged[s] == synth_cases.golden(cases[s])[1] for s in range(len(cases)))
    print(json.dumps({"ok": ok, "streams": sorted(merged), "owners": [sorted(r) for r in allres], "elapsed": el}))
'''


def test_two_ranks_shard_streams_with_gloo(lib, oracle, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for attempt in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
        # (the port is free when it is picked and may be taken when the rendezvous binds it: once more on exactly that)
        if out.returncode == 0 or attempt or not any(w in out.stderr for w in ("ddress already in use", "EADDRINUSE", "failed to bind", "Connection refused")):
            break
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["ok"] and r["streams"] == [0, 1, 2, 3, 4]
    assert r["owners"] == [[0, 2, 4], [1, 3]]          # round-robin, disjoint, no stream decoded twice
    assert r["elapsed"] > 0


def test_sharding_is_a_partition():
    from p264decoder_amd import shard
    for world in (1, 2, 4, 8):
        owned = [shard.streams_of_rank(37, r, world) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(37))


def _bench(args, env_extra):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if "WORLD_SIZE" not in env_extra:
        env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """`--gpus 8` under a launcher that started one rank used to run ONE rank and print n_gpus = 1."""
    out = _bench(["--gpus", "8", "--steps", "1"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert out.returncode == 2 and "--gpus 8 but WORLD_SIZE = 1" in out.stderr and not out.stdout.strip()


def test_bench_gpus_n_without_a_launcher_starts_n_ranks():
    """No WORLD_SIZE: bench.py is its own launcher.  Without a GPU the ranks it starts refuse to run (no CPU fallback; the
    launcher ends the other rank as soon as the first one has failed, so one or two of them get to say so) and the parent reports
    the failure instead of a JSON line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check of the launcher (the GPU box runs tests/test_gpu_bench_ranks.py)")
    out = _bench(["--gpus", "2", "--steps", "1", "--streams", "2", "--no-extras"], {})
    assert out.returncode != 0 and not out.stdout.strip()
    assert 1 <= out.stderr.count("bench.py needs an MI355X") <= 2 and "the ranks failed" in out.stderr
