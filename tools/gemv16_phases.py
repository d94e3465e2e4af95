"""Phase stamps inside k_gemv16's K = 3072 down projection (b = 3..11 decode step, workgroup 0; wave 0 and the last wave), -DQ3_SAMPLE_PROF build:
    SKIP_CODEC=1 tools/build_prof_lib.sh && Q3TTS_LIB=$PWD/tools/exp/libprof.so python tools/gemv16_phases.py [--batch 8]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
import q3tts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
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
names = ["entry", "loads issued", "group 0 done", "MFMAs issued", "waves met", "stored"]
acc = np.zeros((2, 6)); n = 0
for _ in range(16):
    eng.decode_steps(1)
    buf = (C.c_longlong * 32)()
    L.q3_gemm_prof(buf)
    t = np.array(buf[:], dtype=np.float64) * 10.0
    w0, wl = t[16:22], t[24:30]
    acc[0] += w0 - w0[0]; acc[1] += wl - w0[0]; n += 1
acc /= n
print(f"k_gemv16 down projection (K 3072, {a.batch} rows), workgroup 0, ns from wave 0's entry (hipGraph replay; last launch of the step)")
for i, nm in enumerate(names):
    print(f"  {nm:14s} wave 0 {acc[0, i]:7.0f}   last wave {acc[1, i]:7.0f}")
eng.close()
