"""Determinism stress of the batched decode step: the same job N times on one engine, every run compared bit for bit with the first.
Variants by environment (read at engine creation / launch): default, Q3TTS_SEAM=0, Q3TTS_NO_SAMPLER_PLANES=1.
    python tools/determinism_stress.py [--runs 100] [--nb 24] [--frames 8]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
import q3tts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=100)
ap.add_argument("--nb", type=int, default=24)
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--variants", default="default,no-seam,no-sampler-planes")
a = ap.parse_args()
cfg = q3tts.default_config("0.6b")
rng = np.random.default_rng(64)
toks = [np.array([151644, 77091, 151672] + list(rng.integers(0, 151643, int(n))) + [151673, 151645], np.int64) for n in rng.integers(3, 20, 64)]
sp = q3tts.Sampling(max_new_tokens=a.frames, temperature=0.8, top_p=0.95, top_k=50)
ENV = {"default": {}, "no-seam": {"Q3TTS_SEAM": "0"}, "no-sampler-planes": {"Q3TTS_NO_SAMPLER_PLANES": "1"}}
for name in a.variants.split(","):
    os.environ.update(ENV[name])
    eng = q3tts.Engine(cfg, device=0, max_batch=64, max_ctx=a.frames + 40, flags=q3tts.FLAG_TEST_HOOKS)
    eng.fill_synthetic(seed=0)
    first, bad = None, []
    for r in range(a.runs):
        nb = a.nb if r % 3 else 64            # interleave a 64-row job like the test suite does
        _, codes, _ = eng.synthesize_batch(toks[:nb], sp, lang=0, seed=77, ignore_eos=True)
        if nb != a.nb:
            continue
        if first is None:
            first = codes
            continue
        d = [(u, [int(v) for v in np.argwhere(codes[u] != first[u])[0]]) for u in range(a.nb) if not np.array_equal(codes[u], first[u])]
        if d:
            bad.append((r, d[:4]))
    print(f"{name}: {a.runs} runs, nb={a.nb}, {a.frames} frames: {len(bad)} runs differ from the first", bad[:6], flush=True)
    eng.close()
    for k in ENV[name]:
        del os.environ[k]
