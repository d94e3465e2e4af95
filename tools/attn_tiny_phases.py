"""Phase stamps inside k_attn_tiny (the predictor's attention of the batched step; workgroup 0, wave 0), -DQ3_SAMPLE_PROF build:
    SKIP_CODEC=1 tools/build_prof_lib.sh && Q3TTS_LIB=$PWD/tools/exp/libprof.so python tools/attn_tiny_phases.py [--batch 64]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
import q3tts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
L = C.CDLL(os.environ["Q3TTS_LIB"])
cfg = q3tts.default_config("0.6b")
eng = q3tts.Engine(cfg, device=0, max_batch=a.batch, max_ctx=128)
eng.fill_synthetic(seed=0)
rng = np.random.default_rng(0)
sp = q3tts.Sampling(max_new_tokens=64)
for b in range(a.batch):
    ids = np.array([151644, 77091, 151672] + list(rng.integers(0, 151643, 16)) + [151673, 151645], np.int64)
    p, tr = eng.build_prompt(ids, 0)
    eng.slot_begin(b, p, tr, sp, seed=1, stream_id=b, ignore_eos=True)
eng.decode_steps(8)
acc = np.zeros(5); n = 0
for _ in range(16):
    eng.decode_steps(1)
    buf = (C.c_longlong * 32)()
    L.q3_kernel_prof(buf)
    t = np.array(buf[:], dtype=np.float64) * 10.0
    v = np.array([t[12], t[13], t[14], t[15], t[23]])
    acc += v - v[0]; n += 1
acc /= n
print(f"k_attn_tiny (b={a.batch}; last launch of the step = last predictor layer of sub-step 15, 15 cached tokens), ns from entry:")
for nm, x in zip(["entry", "every load issued", "cached K / V arrived", "slab sums + norms + RoPE, q in LDS", "scores + softmax + P.V + stores issued"], acc):
    print(f"  {nm:42s} {x:7.0f}")
eng.close()
