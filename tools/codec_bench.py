"""Codec-decoder-only timing / profiling target: python tools/codec_bench.py [--frames F] [--reps N] [--fp32]
(rocprofv3 --kernel-trace --stats -- python tools/codec_bench.py ... for the per-kernel split)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
import q3tts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--fp32", action="store_true")
ap.add_argument("--stream-chunk", type=int, default=0, help="also decode the utterance in pushes of N frames: carried-state stream vs the windowed decode of the growing history")
a = ap.parse_args()
cfg = q3tts.default_config("0.6b")
eng = q3tts.Engine(cfg, device=0, max_batch=1, max_ctx=a.frames + 32, flags=(q3tts.FLAG_FP32_CODEC if a.fp32 else 0) | q3tts.FLAG_TEST_HOOKS)   # hooks: the A/B knobs (Q3TTS_CONV_*, Q3TTS_CODEC_NO_CARRY, ...) are honoured
eng.fill_synthetic(seed=0)
codes = np.random.default_rng(0).integers(0, 2048, (a.frames, 16)).astype(np.int64)
eng.codec_decode(codes)
t0 = time.perf_counter()
ms = 0.0
for _ in range(a.reps):
    eng.codec_decode(codes)
    ms += eng.last_codec_ms()
dt = time.perf_counter() - t0
print(f"frames={a.frames} device {ms / a.reps:.2f} ms/decode = {ms / a.reps / a.frames * 1e3:.2f} us/frame "
      f"({5.0e9 * a.frames / (ms / a.reps * 1e-3) / 1e12:.1f} TF/s-equivalent), wall {dt / a.reps * 1e3:.1f} ms")
if a.stream_chunk > 0:
    ch = a.stream_chunk
    sid = eng.codec_stream_begin(a.frames)
    ms_c = 0.0
    for s in range(0, a.frames, ch):
        eng.codec_stream_push(sid, codes[s:s + ch])
        ms_c += eng.last_codec_ms()
    eng.codec_stream_end(sid)
    os.environ["Q3TTS_CODEC_NO_CARRY"] = "1"
    ms_w = 0.0
    for s in range(0, a.frames, ch):      # what exact streaming cost before: every chunk decodes the window [0, b)
        eng.codec_decode(codes[: min(a.frames, s + ch)])
        ms_w += eng.last_codec_ms()
    del os.environ["Q3TTS_CODEC_NO_CARRY"]
    print(f"streaming frames={a.frames} in pushes of {ch}: carried state {ms_c:.1f} ms device ({ms_c / a.frames * 1e3:.1f} us/frame), "
          f"windowed decode of the growing history {ms_w:.1f} ms ({ms_w / a.frames * 1e3:.1f} us/frame), one-shot {ms / a.reps:.1f} ms")
eng.close()
