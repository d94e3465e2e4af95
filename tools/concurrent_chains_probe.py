"""Probe: is a b=64 decode step faster as G concurrent chains of 64/G utterances (one engine + stream each, launched from G host threads)?
The step time is nearly flat in the batch (b=8 3.8 ms, b=16 4.2, b=64 5.0): if the GPU overlaps the chains' small launches, G chains
finish 64 utterances in about one small-batch step time.  python tools/concurrent_chains_probe.py [--groups 1,2,4,8] [--frames 96]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
import q3tts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--groups", default="1,2,4,8")
ap.add_argument("--frames", type=int, default=96)
ap.add_argument("--total", type=int, default=64)
a = ap.parse_args()
cfg = q3tts.default_config("0.6b")
IM_START, ASSISTANT, TTS_BOS, TTS_EOS, IM_END = 151644, 77091, 151672, 151673, 151645
rng = np.random.default_rng(1)
toks = [np.array([IM_START, ASSISTANT, TTS_BOS] + list(rng.integers(0, 151643, 16)) + [TTS_EOS, IM_END], np.int64) for _ in range(a.total)]
sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=a.frames)
for G in [int(g) for g in a.groups.split(",")]:
    B = a.total // G
    engs = []
    for g in range(G):
        e = q3tts.Engine(cfg, device=0, max_batch=B, max_ctx=a.frames + 32)
        e.fill_synthetic(seed=0)
        for b in range(B):
            p, tr = e.build_prompt(toks[g * B + b], 0)
            e.slot_begin(b, p, tr, sp, seed=5, stream_id=g * B + b, ignore_eos=True)
        e.decode_steps(4)     # graph captured, caches warm
        engs.append(e)
    n = a.frames - 8
    bar = threading.Barrier(G + 1)
    def run(e):
        bar.wait()
        e.decode_steps(n)
        bar.wait()
    th = [threading.Thread(target=run, args=(e,)) for e in engs]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    dev = [e.last_decode_ms()[0] / n for e in engs]
    print(f"groups {G} x batch {B}: wall {dt * 1e3 / n:.3f} ms per step of {a.total} utterances; per-chain device ms/step min {min(dev):.3f} max {max(dev):.3f}", flush=True)
    for e in engs:
        e.close()
