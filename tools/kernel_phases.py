"""Phase timestamps inside the b=1 decode kernels (k_sample, k_gemv1, k_cp_attn_oproj) from the -DQ3_SAMPLE_PROF build:
    tools/build_prof_lib.sh && Q3TTS_LIB=$PWD/tools/exp/libprof.so python tools/kernel_phases.py
Marks are written by thread 0 of workgroup 0 with the 100 MHz wall clock; a kernel launched several times keeps its last run."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
import q3tts  # noqa: E402


def frame_tokens(text_ids):   # token framing of TTSEngine::synthesize (reference src/tts_onnx.cpp:244-259)
    return np.array([151644, 77091, 151672] + list(text_ids) + [151673, 151645], np.int64)


L = C.CDLL(os.environ["Q3TTS_LIB"])


def marks():
    buf = (C.c_longlong * 32)()
    L.q3_kernel_prof(buf)
    return np.array(buf[:], dtype=np.float64) * 10.0   # ns


def show(tag, t, idx, names):
    v = t[idx]
    print(f"{tag:28s}", " ".join(f"{n}={v[k + 1] - v[k]:.0f}ns" for k, n in enumerate(names)), f"total={v[-1] - v[0]:.0f}ns")


cfg = q3tts.default_config("0.6b")
eng = q3tts.Engine(cfg, device=0, max_batch=1, max_ctx=256, flags=q3tts.FLAG_NO_GRAPH)
eng.fill_synthetic(seed=0)
rng = np.random.default_rng(0)
logits = (rng.standard_normal(3072) * 2.5).astype(np.float32)
logits_sub = (rng.standard_normal(2048) * 2.5).astype(np.float32)   # a sub-code vocabulary: 8 slices per wave (k_sample<false, 8>), 15 of a frame's 16 samplers
for tag, kw, lg in (("k_sample greedy", dict(top_k=1, top_p=1.0, temperature=1.0), logits), ("k_sample k50 p0.95", dict(top_k=50, top_p=0.95, temperature=0.8), logits),
                    ("k_sample k50 p0.95 V=2048", dict(top_k=50, top_p=0.95, temperature=0.8), logits_sub)):
    acc = np.zeros(32)
    for i in range(40):
        eng.sample(lg, q3tts.Sampling(max_new_tokens=1, **kw), (i + 0.5) / 40)
        t = marks()
        for k in range(1, 9):
            t[k] = max(t[k], t[k - 1])
        acc += t - t[0]
    show(tag, acc / 40, list(range(9)), ["loads", "max", "L1", "prefilter", "rounds", "compact+exp", "top-p", "draw"])
# generation: the last instrumented kernels of a decode step are the talker's (k_gemv1 = codec head) and the predictor's last layer
ids = frame_tokens(rng.integers(0, 151643, 16))
p, tr = eng.build_prompt(ids, 0)
eng.slot_begin(0, p, tr, q3tts.Sampling(max_new_tokens=200), seed=1, ignore_eos=True)
eng.decode_steps(150)   # context ~160: two attention splits
acc_g, acc_c, acc_a, n = np.zeros(4), np.zeros(7), np.zeros(6), 0
for i in range(40):
    eng.decode_steps(1)
    t = marks()
    acc_g += t[8:12] - t[8]
    acc_c += t[16:23] - t[16]
    acc_a += t[24:30] - t[24]
    n += 1
g, c = acc_g / n, acc_c / n
print(f"{'k_gemv1 (last = codec head)':28s} issue_loads={g[1]:.0f}ns norm+wait={g[2] - g[1]:.0f}ns dot+reduce={g[3] - g[2]:.0f}ns total={g[3]:.0f}ns")
print(f"{'k_cp_attn_oproj (last layer)':28s} issue_loads={c[1]:.0f}ns norm+rope={c[2] - c[1]:.0f}ns lds_fragments={c[3] - c[2]:.0f}ns scores+merge(wave 0)={c[6] - c[3]:.0f}ns "
      f"block_barrier={c[4] - c[6]:.0f}ns gemv={c[5] - c[4]:.0f}ns total={c[5]:.0f}ns")
a = acc_a / n
print(f"{'k_attn talker (split 0)':28s} round1_issue={a[1]:.0f}ns wait+range={a[2] - a[1]:.0f}ns issue2+norm/rope+sync={a[3] - a[2]:.0f}ns softmax_loop={a[4] - a[3]:.0f}ns combine={a[5] - a[4]:.0f}ns total={a[5]:.0f}ns")
print("cost of one mark:", marks()[31] - marks()[30], "ns")
eng.close()
