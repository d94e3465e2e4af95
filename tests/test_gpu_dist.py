"""The RCCL leg of the multi-GPU bench on ONE MI355X: `bench.py` with Q3TTS_BENCH_FORCE_DIST=1 initialises torch.distributed with the
"nccl" backend (= RCCL on ROCm) for a single rank and pushes the final gather (codes + PCM lengths, device tensors) through it — so
init_process_group("nccl") + all_gather + all_reduce of the N>1 path have executed on the hardware at least once.  The N-rank launch
itself is covered on CPU (tests/test_dist_cpu.py: gloo, world size 2, self-launching bench.py --gpus 2 --dry-launch); an 8-GPU node is
the driver's to run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_nccl_path_runs_on_the_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(Q3TTS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "2", "--frames", "4", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    mg = j["multi_gpu"]
    assert mg["backend"].startswith("nccl") and mg["world_size"] == 1
    assert mg["gathered_utterances"] == 2 and mg["gathered_pcm_samples"] == j["pcm_samples"] > 0
    assert "unmeasured" in mg["scaling_note"] and j["n_gpus"] == 1 and j["value"] > 0
