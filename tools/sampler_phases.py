"""Phase timestamps of k_sample from the -DQ3_SAMPLE_PROF build (Q3TTS_LIB=tools/exp/libprof.so)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("leaxer-qwen3-tts_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import q3tts  # noqa: E402
from util import tiny_pair  # noqa: E402

eng, orc, _ = tiny_pair(seed=0, max_batch=1, max_ctx=64)
orc.close()
L = C.CDLL(os.environ["Q3TTS_LIB"])
rng = np.random.default_rng(0)
logits = (rng.standard_normal(3072) * 2.5).astype(np.float32)
names = ["start", "loads", "max", "L1", "prefilter+sync", "rounds+sync", "compact+exp+sync", "top-p", "draw+sync"]
for tag, kw in (("greedy", dict(top_k=1, top_p=1.0, temperature=1.0)), ("k50_p1", dict(top_k=50, top_p=1.0, temperature=0.8)),
                ("k50_p95", dict(top_k=50, top_p=0.95, temperature=0.8))):
    sp = q3tts.Sampling(max_new_tokens=1, **kw)
    acc = np.zeros(9)
    n = 50
    for i in range(n):
        eng.sample(logits, sp, (i + 0.5) / n)
        buf = (C.c_longlong * 16)()
        L.q3_sample_prof(buf)
        t = np.array(buf[:9], dtype=np.float64)
        for k in range(1, 9):   # marks a path skipped keep their stale value: treat non-monotonic ones as zero-length
            if t[k] < t[k - 1]:
                t[k] = t[k - 1]
        acc += t - t[0]
    acc /= n
    print(tag, " ".join(f"{names[k]}={(acc[k] - acc[k - 1]) * 10:.0f}ns" for k in range(1, 9)), f"total={acc[8] * 10:.0f}ns")
eng.close()
