"""Ragged-length serving workload (per-utterance F ~ U[40, 90], EOS suppressed, caps fix the lengths): N utterances
through B slots, (a) in static waves of B (each wave as long as its longest member, vocoder after the wave) and (b) as one queue with
continuous batching (finished slots re-armed, vocoder on side streams under the decode steps).  Same utterances, same results.
    python tools/ragged_bench.py [--batch 64] [--utterances 256]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
import q3tts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--utterances", type=int, default=256)
ap.add_argument("--lo", type=int, default=40)
ap.add_argument("--hi", type=int, default=90)
a = ap.parse_args()
cfg = q3tts.default_config("0.6b")
eng = q3tts.Engine(cfg, device=0, max_batch=a.batch, max_ctx=a.hi + 32)
eng.fill_synthetic(seed=0)
rng = np.random.default_rng(2)
lens = rng.integers(a.lo, a.hi + 1, a.utterances).astype(np.int32)
toks = [np.array([151644, 77091, 151672] + list(rng.integers(0, 151643, 16)) + [151673, 151645], np.int64) for _ in range(a.utterances)]
sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=int(a.hi))


def waves():
    out = []
    for u0 in range(0, a.utterances, a.batch):
        # same RNG streams as the queue run: stream = utterance index, so wave w passes its own slice with a seed-independent layout
        pcm, codes, nfr = eng.synthesize_batch(toks[u0:u0 + a.batch], sp, seed=9, ignore_eos=True, max_new_per_utt=lens[u0:u0 + a.batch])
        out += list(nfr)
    return np.array(out)


def queue():
    return eng.synthesize_batch(toks, sp, seed=9, ignore_eos=True, max_new_per_utt=lens)[2]


for name, fn in (("static waves", waves), ("continuous batching", queue)):
    fn()
    eng.counters(reset=True)
    t0 = time.perf_counter()
    nfr = fn()
    dt = time.perf_counter() - t0
    ctr = eng.counters()
    assert np.array_equal(nfr, lens), name
    print(f"{name:22s} {a.utterances} utterances x U[{a.lo},{a.hi}] frames through {a.batch} slots: {dt * 1e3:8.1f} ms  "
          f"RTF {nfr.sum() * 0.08 / dt:7.1f}x  {nfr.sum() / dt:8.0f} frames/s   [{ctr['decode_steps']} steps, {ctr['decode_ms']:.0f} ms in decode "
          f"({ctr['decode_ms'] / max(ctr['decode_steps'], 1):.2f} ms/step), vocoder window {ctr['codec_ms']:.0f} ms]")
eng.close()
