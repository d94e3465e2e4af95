"""rocprofv3 --kernel-trace target: k_sample in isolation with different parameter sets (200 calls each, 3072 logits)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import q3tts  # noqa: E402
from util import tiny_pair  # noqa: E402

eng, orc, _ = tiny_pair(seed=0, max_batch=1, max_ctx=64)
orc.close()
rng = np.random.default_rng(0)
logits = (rng.standard_normal(3072) * 2.5).astype(np.float32)
for name, kw in (("greedy", dict(top_k=1, top_p=1.0, temperature=1.0)), ("k50_p1", dict(top_k=50, top_p=1.0, temperature=0.8)),
                 ("k50_p95", dict(top_k=50, top_p=0.95, temperature=0.8)), ("k0_p1", dict(top_k=0, top_p=1.0, temperature=0.8))):
    sp = q3tts.Sampling(max_new_tokens=1, **kw)
    n = 200 if name != "k0_p1" else 20
    for i in range(n):
        eng.sample(logits[: 3072 - (hash(name) % 7)], sp, (i + 0.5) / n)   # distinct V per set tags the set in the trace (grid is 1 either way)
    print(name, 3072 - (hash(name) % 7))
eng.close()
