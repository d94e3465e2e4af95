"""Does a vocoder confined to a subset of the CUs (hipExtStreamCreateWithCUMask) leave the latency-bound decode chain alone?
Two engines on one GPU: A steps 64 slots (hipGraph replays), B decodes 256-frame utterances in a loop on a stream whose CU mask is the
32-bit pattern given (repeated over the 256 CUs).  Reports A's ms/step alone and under B, and B's decodes/s alone and under A.
    python tools/overlap_probe.py [--patterns none,ffffffff,0000ffff,000000ff]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
import q3tts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--patterns", default="none,ffffffff,0000ffff,000000ff,0000000f")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=192)
a = ap.parse_args()
cfg = q3tts.default_config("0.6b")
os.environ.pop("Q3TTS_STREAM_CU_MASK", None)
A = q3tts.Engine(cfg, device=0, max_batch=a.batch, max_ctx=1200, flags=q3tts.FLAG_TEST_HOOKS)
A.fill_synthetic(seed=0)
rng = np.random.default_rng(1)
sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=1100)


def arm():
    for b in range(a.batch):
        A.slot_release(b)
    ids = np.array([151644, 77091, 151672] + list(rng.integers(0, 151643, 8)) + [151673, 151645], np.int64)
    p, t = A.build_prompt(ids, 0)
    for b in range(a.batch):
        A.slot_begin(b, p, t, sp, seed=3, stream_id=b, ignore_eos=True)
    A.decode_steps(8)


def steps_ms(n):
    t0 = time.perf_counter()
    A.decode_steps(n)
    return (time.perf_counter() - t0) * 1e3 / n


codes = rng.integers(0, cfg.cd_codebook, (256, cfg.n_groups)).astype(np.int64)
arm()
base = min(steps_ms(a.steps // 3) for _ in range(3))
print(f"decode alone: {base:.3f} ms/step (b={a.batch})")
for pat in a.patterns.split(","):
    if pat == "none":
        os.environ.pop("Q3TTS_STREAM_CU_MASK", None)
    else:
        os.environ["Q3TTS_STREAM_CU_MASK"] = pat
    B = q3tts.Engine(cfg, device=0, max_batch=1, max_ctx=300, flags=q3tts.FLAG_TEST_HOOKS)
    B.fill_synthetic(seed=0)
    for _ in range(2):
        B.codec_decode(codes)
    t0 = time.perf_counter()
    for _ in range(5):
        B.codec_decode(codes)
    alone = 5 / (time.perf_counter() - t0)
    stop = threading.Event()
    count = [0]

    def voc():
        while not stop.is_set():
            B.codec_decode(codes)
            count[0] += 1

    th = threading.Thread(target=voc)
    th.start()
    time.sleep(0.05)
    c0, t0 = count[0], time.perf_counter()
    ms = steps_ms(a.steps)
    dt = time.perf_counter() - t0
    c1 = count[0]
    stop.set()
    th.join()
    bits = "all CUs (no mask)" if pat == "none" else f"mask {pat} ({bin(int(pat, 16)).count('1') * 8} CUs)"
    print(f"vocoder on {bits:28s}: decode {ms:.3f} ms/step ({ms / base:.2f}x), vocoder {(c1 - c0) / dt:6.1f} decodes/s under decode, {alone:6.1f} alone "
          f"({256 * alone * 0.08:.0f}x real time alone)")
    B.close()
A.close()
