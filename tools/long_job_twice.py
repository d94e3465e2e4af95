"""Two 64 x 2048-frame jobs (bf16 KV cache) back to back in one process, wall time and engine counters of each: does the first long job
pay for lazy allocations that the second does not?  (bench.py warms the long sub-records up on 64-frame utterances.)  Measured: 13.40 and
13.37 s — no; the 12 % spread of the b64_f2048* records between boxes is not allocation.

    python tools/long_job_twice.py            (needs the GPU, ~30 s)
"""
import sys, os, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for d in ("leaxer-qwen3-tts_amd", "tests", "oracle"): sys.path.insert(0, os.path.join(R, d))
import numpy as np
import q3tts
from util import frame_tokens
cfg = q3tts.default_config("0.6b")
B, F = 64, 2048
eng = q3tts.Engine(cfg, device=0, max_batch=B, max_ctx=F + 64, flags=q3tts.FLAG_KV_BF16)
eng.fill_synthetic(seed=0)
rng = np.random.default_rng(1)
toks = [frame_tokens(rng.integers(0, 151643, 16)) for _ in range(B)]
kw = dict(temperature=0.8, top_p=0.95, top_k=50)
t = time.perf_counter(); eng.synthesize_batch(toks, q3tts.Sampling(max_new_tokens=64, **kw), lang=0, seed=1, ignore_eos=True); print("warm 64 frames: %.2f s" % (time.perf_counter() - t), flush=True)
for i in range(2):
    eng.counters(reset=True)
    t = time.perf_counter()
    pcm, codes, nfr = eng.synthesize_batch(toks, q3tts.Sampling(max_new_tokens=F, **kw), lang=0, seed=10 + i, ignore_eos=True, want_codes=True)
    dt = time.perf_counter() - t
    c = eng.counters()
    print("job %d: wall %.3f s; counters %s" % (i, dt, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in c.items()}), flush=True)
    del pcm, codes
eng.close()
