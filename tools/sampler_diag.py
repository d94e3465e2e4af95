"""Sampler parity diagnostic (GPU): k_sample vs the oracle's q3o_sample over many random rows, per setting — counts every mismatch and
prints the cases (setting, trial, vocabulary size, u, ids) so that the cause can be found.  Writes one JSON object.
    python tools/sampler_diag.py [trials] > gpurun_out/sampler_diag.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("leaxer-qwen3-tts_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import q3_oracle as qo  # noqa: E402
import q3tts  # noqa: E402
from util import tiny_pair, to_osampling  # noqa: E402

SETTINGS = [
    dict(temperature=0.8, top_p=0.95, top_k=50), dict(temperature=1.0, top_p=1.0, top_k=1), dict(temperature=0.0, top_p=1.0, top_k=0),
    dict(temperature=1.3, top_p=0.5, top_k=10), dict(temperature=0.7, top_p=0.9, top_k=0), dict(temperature=0.8, top_p=1.0, top_k=200),
    dict(temperature=0.9, top_p=0.9, top_k=64), dict(temperature=0.9, top_p=0.9, top_k=65), dict(temperature=1.0, top_p=0.8, top_k=2),
    dict(temperature=0.8, top_p=0.95, top_k=63),
]


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    eng, orc, _ = tiny_pair(seed=0, max_batch=1, max_ctx=32)
    out = {"trials_per_setting": trials, "settings": []}
    for si, kw in enumerate(SETTINGS):
        rng = np.random.default_rng(1000 + si)
        sp = q3tts.Sampling(max_new_tokens=8, **kw)
        so = to_osampling(sp)
        bad = []
        for t in range(trials):
            n = (96, 3072, 2048, 2176)[t % 4]
            lg = (rng.standard_normal(n) * 2.0).astype(np.float32)
            if t % 5 == 0:
                lg[rng.integers(0, n, 4)] = lg.max()              # exact ties at the top
            if t % 7 == 3:
                lg[rng.integers(0, n, 6)] = np.sort(lg)[-min(kw["top_k"] or 5, n - 1)]   # exact ties AT the top-k threshold
            u = float(rng.random())
            a, b = eng.sample(lg, sp, u), orc.sample(lg, so, u)
            if a != b:
                bad.append(dict(trial=t, n=n, u=u, hip=a, oracle=b, p_hip=float(lg[a]), p_orc=float(lg[b])))
        out["settings"].append(dict(params=kw, mismatches=len(bad), cases=bad[:8]))
        print("setting %d %s: %d / %d mismatches" % (si, kw, len(bad), trials), file=sys.stderr)
    eng.close()
    orc.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
