"""GPU (-m gpu): the multi-GPU code paths on the ONE GPU a test box has -- launched exactly as the driver launches the scaling bench
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 ...`), backend "nccl" = RCCL:
  * `parallel.GradientReducer(force=True)`: bucketed gradient exchange from grad-ready hooks on a side HIP stream, against the
    un-reduced gradients of an identical replica (tests/_rccl_worker.py);
  * `bench.py --gpus 1` under torch.distributed.run: the process group, the barrier and the MAX all-reduce that bracket the timing.
No 1 -> 8 GPU curve can be measured here; what these tests pin is that the RCCL path runs and is numerically inert at world_size 1."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script_args, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_gradient_reducer_runs_on_rccl_with_one_rank():
    r = _torchrun([os.path.join(ROOT, "tests", "_rccl_worker.py")], 600)
    assert r.returncode == 0 and "RCCL_WORKER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_under_torch_distributed_run_with_one_gpu():
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--size", "32", "--inference-steps", "2",
                   "--cpu-baseline", "off"], 900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["output_finite"] and line["scaling"] == "weak"
    assert line["config"]["process_group"] == "nccl (RCCL), world_size 1"


def test_bench_launches_itself_the_way_the_driver_calls_it():
    """`python bench.py --gpus N` with no launcher environment: the process re-executes itself under torch.distributed.run (forced at N = 1 here by
    GM_BENCH_SELF_LAUNCH=1 -- at N > 1 it is automatic) and rank 0 of the child job prints the ONE JSON line on the parent's stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GM_BENCH_SELF_LAUNCH="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--size", "32", "--inference-steps", "2",
                        "--cpu-baseline", "off"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["output_finite"]
    assert line["config"]["process_group"] == "nccl (RCCL), world_size 1" and line["config"]["self_launched"] is True
