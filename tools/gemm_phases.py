"""Phase stamps inside the gate/up k_gemm2 launch of the batched decode step (workgroup (0,0), -DQ3_SAMPLE_PROF build):
    tools/build_prof_lib.sh && Q3TTS_LIB=$PWD/tools/exp/libprof.so python tools/gemm_phases.py [--batch 64]"""
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
eng = q3tts.Engine(cfg, device=0, max_batch=a.batch, max_ctx=128, flags=q3tts.FLAG_NO_GRAPH)
eng.fill_synthetic(seed=0)
rng = np.random.default_rng(0)
sp = q3tts.Sampling(max_new_tokens=64)
for b in range(a.batch):
    ids = np.array([151644, 77091, 151672] + list(rng.integers(0, 151643, 16)) + [151673, 151645], np.int64)
    p, tr = eng.build_prompt(ids, 0)
    eng.slot_begin(b, p, tr, sp, seed=1, stream_id=b, ignore_eos=True)
eng.decode_steps(4)
acc = np.zeros(16)
n = 0
for _ in range(24):
    eng.decode_steps(1)
    buf = (C.c_longlong * 32)()
    L.q3_gemm_prof(buf)
    t = np.array(buf[:16], dtype=np.float64) * 10.0
    acc += t - t[0]
    n += 1
t = acc / n
print(f"gate/up k_gemm2 (last layer's launch, workgroup (0,0)), ns from entry:")
print(f"  loads of both chunks issued   {t[1]:.0f}")
print(f"  chunk 0 landed + in LDS       {t[2]:.0f}")
print(f"  chunk 0 MFMAs issued          {t[3]:.0f}")
print(f"  chunk 1 landed + in LDS       {t[4]:.0f}")
print(f"  chunk 1 MFMAs issued          {t[5]:.0f}")
print(f"  epilogue stores issued        {t[15]:.0f}")
eng.close()
