"""Stamps inside the split-K seam of k_gemm3 (batched decode step, -DQ3_SAMPLE_PROF build): every K-slice workgroup of column tile 0 /
row block 0 of the step's last o_proj / gate-up / down launch stamps entry, body done, slab drained, ticket + wait, chunk claimed, chunk
stored, exit.    SKIP_CODEC=1 tools/build_prof_lib.sh && Q3TTS_LIB=$PWD/tools/exp/libprof.so python tools/seam_phases.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
import q3tts  # noqa: E402

L = C.CDLL(os.environ["Q3TTS_LIB"])
B = 64
cfg = q3tts.default_config("0.6b")
eng = q3tts.Engine(cfg, device=0, max_batch=B, max_ctx=128)
eng.fill_synthetic(seed=0)
rng = np.random.default_rng(0)
sp = q3tts.Sampling(max_new_tokens=64)
for b in range(B):
    ids = np.array([151644, 77091, 151672] + list(rng.integers(0, 151643, 16)) + [151673, 151645], np.int64)
    p, tr = eng.build_prompt(ids, 0)
    eng.slot_begin(b, p, tr, sp, seed=1, stream_id=b, ignore_eos=True)
eng.decode_steps(8)
names = ["body", "drain", "ticket+wait", "claim", "reduce", "exit"]
for rep in range(3):
    eng.decode_steps(1)
    buf = (C.c_longlong * (3 * 16 * 8))()
    L.q3_seam_prof(buf)
    t = np.array(buf[:], dtype=np.float64).reshape(3, 16, 8) * 10.0
    for kind, (nm, ks) in enumerate((("o_proj", 8), ("gate/up", 4), ("down", 12))):
        t0 = t[kind, :ks, 0].min()
        print(f"{nm} (tile 0, row block 0), ns from the first workgroup's entry; columns: entry body drain ticket+wait claim reduce none-left")
        for s in range(ks):
            r = t[kind, s]
            print("  slice %2d: " % s + " ".join("%7.0f" % (v - t0) if v >= t0 else "      -" for v in r[:6]) + ("   poll rounds %d" % int(r[6] / 10.0) if 0 < r[6] < 1e6 else ""))
eng.close()
