"""N>1 path on CPU: world_size-2 gloo.  Each rank 'synthesises' its shard with the CPU oracle (tiny
config) and the ranks exchange codes exactly as bench.py does over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in ("leaxer-qwen3-tts_amd", "oracle", "tests"):
        sys.path.insert(0, os.path.join(ROOT, p))
    import torch.distributed as dist
    import q3_oracle as qo
    import q3dist
    from util import frame_tokens
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = qo.config_tiny()
    orc = qo.Oracle(cfg, max_ctx=64, weights=qo.random_weights(cfg, 0))
    texts = [[1, 2, 3], [4], [5, 6, 7, 8, 9], [10, 11], [12, 13, 14, 15, 16, 17, 18]]
    mine = q3dist.shard_utterances([len(t) for t in texts], world, rank)
    sp = qo.Sampling(temperature=1.0, top_p=1.0, top_k=1, max_new_tokens=6)
    codes = [orc.generate(orc.build_prompt(frame_tokens(texts[i]), 0), sp, seed=5, stream=i, ignore_eos=True) for i in mine]
    allc, alln = q3dist.gather_codes(dist, codes, mine, len(texts), 6, cfg.n_groups, pcm_lens=[orc.vocoder_len(len(c)) for c in codes])
    ref = [orc.generate(orc.build_prompt(frame_tokens(t), 0), sp, seed=5, stream=i, ignore_eos=True) for i, t in enumerate(texts)]
    ok = all(a is not None and np.array_equal(a, b) for a, b in zip(allc, ref))
    ok = ok and alln == [orc.vocoder_len(len(r)) for r in ref]          # configs[3]'s gather carries every utterance's PCM length too
    ok = ok and all(np.array_equal(a, b) for a, b in zip(q3dist.gather_codes(dist, codes, mine, len(texts), 6, cfg.n_groups), ref))
    q.put((rank, mine, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_is_a_partition_and_balanced():
    sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
    import q3dist
    lengths = [3, 40, 7, 7, 19, 2, 64, 11, 5]
    for world in (1, 2, 4, 8):
        shards = [q3dist.shard_utterances(lengths, world, r) for r in range(world)]
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(lengths)))
        loads = [sum(lengths[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(lengths)


def test_two_rank_gloo_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = sorted(i for _, mine, _ in res for i in mine)
    assert seen == [0, 1, 2, 3, 4]
    assert all(ok for _, _, ok in res)


def test_bench_gpus2_self_launches_two_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the way the driver calls it) must start two ranks itself — child processes
    spawned before anything touches a GPU — rendezvous on 127.0.0.1 and run the gather; --dry-launch keeps it to that plumbing (gloo, no
    GPU work).  Rank 0's single JSON line comes through the parent, a failing rank fails the parent."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--batch", "3"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["dry_launch"] and j["n_gpus"] == 2 and j["ranks_ok"] and j["gathered_utterances"] == 6
    assert j["shard_sizes"] == [3, 3]
    # a rank count that does not match --gpus is refused (exit code 2), not silently run as one rank
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env2, capture_output=True, text=True, timeout=120)
    assert r2.returncode == 2 and "WORLD_SIZE=1" in r2.stderr


def test_bench_gpus8_dry_launch_at_configs3_shape():
    """BASELINE configs[3] at its real shape, without hardware: `bench.py --gpus 8 --dry-launch --batch 64` starts 8 child ranks (gloo,
    127.0.0.1), every rank contributes 64 ragged utterances in the real payload shape ([64][2048][16] int32 = 8.4 MB per rank), all 512
    come back from the gather on every rank with their PCM lengths, and q3dist.shard_utterances cuts a 512-utterance list into 8
    shards of 64 whose text loads differ by less than the longest text.  No scaling curve is measured anywhere (one GPU per call)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-launch", "--batch", "64", "--frames", "2048"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["dry_launch"] and j["n_gpus"] == 8 and j["ranks_ok"] and j["gathered_utterances"] == 512
    assert j["shard_sizes"] == [64] * 8
    assert max(j["shard_loads"]) - min(j["shard_loads"]) <= j["longest_text"]
    print("configs[3] dry launch: 8 ranks x 64 utterances, %.1f MB per rank, gather %.0f ms (gloo, CPU)" % (j["payload_mb_per_rank"], j["gather_ms"]))
