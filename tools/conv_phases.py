"""Iteration timeline of k_conv_split (last launch of a codec decode = some late layer; pass --frames to pick the shapes):
    tools/build_prof_lib.sh && Q3TTS_LIB=$PWD/tools/exp/libprof.so python tools/conv_phases.py --frames 64 --stop-after-transformer"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
import q3tts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=64)
a = ap.parse_args()
L = C.CDLL(os.environ["Q3TTS_LIB"])
cfg = q3tts.default_config("0.6b")
eng = q3tts.Engine(cfg, device=0, max_batch=1, max_ctx=a.frames + 32)
eng.fill_synthetic(seed=0)
codes = np.random.default_rng(0).integers(0, 2048, (a.frames, 16)).astype(np.int64)
for _ in range(2):
    eng.codec_decode(codes)
buf = (C.c_longlong * 64)()
L.q3_conv_prof(buf)
t = np.array(buf[:], dtype=np.float64) * 10.0
print("stamped k_conv_split launch (its last occurrence in the decode): total", t[63] - t[0], "ns; loop", t[62] - t[0], "ns; tail + epilogue", t[63] - t[62])
print(f"  after the loop: block barrier {t[60] - t[62]:.0f} ns" if t[60] > 0 else "  (fused tail)")
if t[48] > 0:
    for i in range(2):
        b = 48 + 4 * i
        if t[b + 3] > t[b]:
            print(f"  fused tail, row block {i}: SnakeBeta + (hi, lo) planes to LDS {t[b + 1] - t[b]:.0f} ns, 1x1 conv MFMAs {t[b + 2] - t[b + 1]:.0f} ns, epilogue {t[b + 3] - t[b + 2]:.0f} ns")
elif t[61] > t[60] > 0:
    print(f"  epilogue of row block 0 {t[61] - t[60]:.0f} ns, of row block 1 {t[63] - t[61]:.0f} ns")
for it in range(11):
    b = 1 + it * 4
    if t[b + 3] <= t[b]:
        break
    print(f"  iter {it}: wait+sync1+stores={t[b + 1] - t[b]:.0f}ns sync2={t[b + 2] - t[b + 1]:.0f}ns issue+mfma={t[b + 3] - t[b + 2]:.0f}ns")
eng.close()
