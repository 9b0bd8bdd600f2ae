"""round 5: does the pipeline's parse slow down with the GPU on because its output buffers (hipHostMalloc) sit on the GPU's NUMA node
while the parser threads run anywhere?  Prints the topology, then runs pipe_bench unpinned / pinned to the GPU's node / pinned to
another node."""
import glob, os, subprocess, sys

def cpus(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part: a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        elif part: out.append(int(part))
    return out

nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = cpus(open(d + "/cpulist").read())
print("nodes:", {n: "%d cpus (%d..%d)" % (len(c), c[0], c[-1]) for n, c in nodes.items()})
gpu_nodes = []
for d in glob.glob("/sys/bus/pci/devices/*"):
    try:
        if open(d + "/vendor").read().strip() == "0x1002" and open(d + "/class").read().startswith("0x03") or open(d + "/class").read().startswith("0x12"):
            if open(d + "/vendor").read().strip() == "0x1002": gpu_nodes.append((os.path.basename(d), int(open(d + "/numa_node").read())))
    except Exception: pass
print("AMD devices and their NUMA nodes:", gpu_nodes[:16])
print("affinity now: %d cpus" % len(os.sched_getaffinity(0)))
import torch  # noqa (device order as bench.py sees it)
def run(tag, cpuset):
    env = dict(os.environ, P264AMD_PIPE_DEBUG="1")
    pre = "import os; os.sched_setaffinity(0, %r); " % sorted(cpuset) if cpuset else ""
    cmd = [sys.executable, "-c", pre + "import runpy, sys; sys.argv=['pipe_bench','--streams','128','--threads','16','--pictures','24','--device','0']; runpy.run_module('p264decoder_amd.tools.pipe_bench', run_name='__main__')"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("p264pipe_run: 24") or l.startswith("{")]
    print(tag, "|", " ".join(l[:230] for l in lines))
run("unpinned", None)
allowed = os.sched_getaffinity(0)
for n, c in nodes.items():
    cs = set(c) & allowed
    if cs: run("node %d" % n, cs)
