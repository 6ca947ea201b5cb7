"""GPU: the N > 1 path of bench.py, end to end -- two ranks launched exactly as the driver launches them (python -m torch.distributed.run
--nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...), sharing the one GPU of the test box over gloo (ENVGS_DIST_BACKEND; on an 8-GPU
node the same code runs one rank per GPU over RCCL).  Exercises camera sharding, the persistent flat gradient buffers, the direct
reduce-scatter + all-gather launched from backward hooks, max-over-ranks timing and the one-JSON-line contract."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.timeout(600)
@pytest.mark.parametrize("exchange", ["direct", "allreduce", "auto"])
def test_bench_two_ranks_gloo_on_one_gpu(exchange):
    env = dict(os.environ, ENVGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--gaussians", "20000", "--env-gaussians", "8192", "--res", "128",
           "--no-cpu-baseline", "--no-render", "--exchange", exchange]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                                    # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    n_params = 20000 * (3 + 48 + 1 + 2 + 4 + 1 + 1) + 8192 * (3 + 48 + 1 + 2 + 4)
    assert d["config"]["allreduce_bytes_per_step"] >= 4 * n_params               # both flat buffers were exchanged (padded to the world size)
    assert exchange in d["config"]["parallelism"]
    ex = d["exchange"]                                                          # both exchange forms timed on the step's own flat buffers
    assert ex["world"] == 2 and ex["bytes_per_step"] == d["config"]["allreduce_bytes_per_step"] and ex["direct"]["ms"] > 0 and ex["allreduce"]["ms"] > 0
