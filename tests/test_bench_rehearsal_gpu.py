"""The driver's 8-GPU invocation of bench.py, rehearsed on ONE GPU: eight ranks under torch.distributed.run share
the device, the collectives go through gloo (BENCH_BACKEND=gloo: bench.py's rehearsal switch -- the line says so and is never a
measurement).  No 8-GPU node is reachable from the build box, so this is what can be executed of `python -m torch.distributed.run
--nproc-per-node 8 bench.py --gpus 8`: every leg of the multi-rank line must complete (ok: true), and the SAE object must carry both
partitionings of the strong-scaling step and the weak-scaling one."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(1500)
def test_eight_rank_bench_line_completes_on_one_gpu():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in [k for k in env if k.startswith("PV_")]:
        del env[k]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "64",
           "--l14-batch", "8", "--no-cpu-baseline", "--leg-timeout", "1200"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1400)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["rehearsal_backend"] == "gloo" and line["scaling"] == "weak"
    assert line["ok"] is True, line.get("leg_errors")
    assert line["metric"].startswith("images/sec run_with_cache") and line["value"] > 0 and line["config"]["workload"]
    sae = line["sae"]
    assert sae["value"] > 0 and "error" not in sae
    # the two partitionings of the strong-scaling step (features sharded: the default; tokens sharded + sharded optimizer: north_star's)
    other = sae.get("strong_scaling_data_parallel") or sae.get("strong_scaling_feature_parallel")
    assert other and other["value"] > 0 and "error" not in other
    weak = sae["weak_scaling_data_parallel"]
    assert weak["value"] > 0 and "error" not in weak
    assert sae["end_to_end"]["value"] > 0 and line["l14_336_pattern"]["value"] > 0
