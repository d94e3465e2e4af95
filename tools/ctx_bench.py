"""Decode step of B armed slots AT a chosen talker context (the regime where run_decode's attention, /root/reference/src/tts_onnx.cpp:667-732,
streams the KV cache: 64 rows x 1041 tokens = 15.3 GB fp32 / 7.6 GB bf16 per step), without generating up to it: the slots are armed,
warmed for two frames and then moved ahead with q3tts_measure_skip_frames (synthetic KV rows; the codes emitted afterwards mean nothing).

    python tools/ctx_bench.py [--batch 64] [--ctx 1024] [--kv fp32|bf16] [--steps 16] [--no-graph] [--stages] [--base-ctx 24]

Prints one JSON line: graph-replay (or eager) step time at the context, the per-stage split (--stages: 8 eager steps with HIP events at
the stage boundaries) at the context and at --base-ctx, and from their difference the attention's share and the rate it streams KV at.
Profiling targets:  rocprofv3 --kernel-trace --stats -- python tools/ctx_bench.py --no-graph --steps 4     (by-kernel table)
                    Q3TTS_NULL_STREAM=1 rocprofv3 --pmc FETCH_SIZE -- tools/pmc_bisect <steps> <batch> <ctx> <bf16>   (traffic)"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
import q3tts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--ctx", type=int, default=1024)
ap.add_argument("--base-ctx", type=int, default=24)
ap.add_argument("--kv", choices=["fp32", "bf16"], default="fp32")
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--stages", action="store_true")
ap.add_argument("--max-ctx", type=int, default=0, help="engine capacity (default: ctx + steps + 96; the bench's 2048-frame engine has 2112)")
a = ap.parse_args()

cfg = q3tts.default_config("0.6b")
B = a.batch
flags = q3tts.FLAG_TEST_HOOKS | (q3tts.FLAG_KV_BF16 if a.kv == "bf16" else 0) | (q3tts.FLAG_NO_GRAPH if a.no_graph else 0)
max_ctx = a.max_ctx or (a.ctx + 3 * a.steps + 96)
eng = q3tts.Engine(cfg, device=0, max_batch=B, max_ctx=max_ctx, flags=flags)
eng.fill_synthetic(seed=0)
ids = np.array([151644, 77091, 151672] + list(np.random.default_rng(1).integers(0, 151643, 16)) + [151673, 151645], np.int64)
prompt, trailing = eng.build_prompt(ids, 0)
sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=max_ctx - 16)
for b in range(B):
    eng.slot_begin(b, prompt, trailing, sp, seed=3, stream_id=b, ignore_eos=True)
eng.decode_steps(2)
S = prompt.shape[0]


def stages(n=8):
    out = (q3tts.C.c_double * 4)()
    eng._ck(eng.L.q3tts_stage_profile(eng.h, n, out))
    return {"sampler_ms": round(out[0], 4), "predictor_ms": round(out[1], 4), "talker_ms": round(out[2], 4), "step_ms": round(out[3], 4)}


rec = {"batch": B, "kv": a.kv, "ctx": a.ctx, "max_ctx": max_ctx, "graph": not a.no_graph}
base = None
if a.stages:
    if a.base_ctx > S + 2:
        eng.measure_skip_frames(a.base_ctx - (S + 2))
    base = stages()
    rec["stages_base_ctx"] = dict(base, ctx=a.base_ctx)
nf, _ = eng.slot_status(0)
here = S + nf
if a.ctx > here:
    eng.measure_skip_frames(a.ctx - here)
eng.decode_steps(2)
eng.decode_steps(a.steps)
ms, n = eng.last_decode_ms()
rec["step_ms"] = round(ms / n, 4)
esz = 2 if a.kv == "bf16" else 4
ctx_mid = a.ctx + 2 + a.steps / 2.0
kv_bytes = B * ctx_mid * cfg.n_layers * 2 * cfg.n_kv_heads * cfg.head_dim * esz
rec["kv_bytes_per_step"] = int(kv_bytes)
if a.stages:
    st = stages()
    rec["stages"] = st
    d_attn = st["talker_ms"] - base["talker_ms"]
    d_bytes = B * (ctx_mid + a.steps / 2.0 + 4 - a.base_ctx) * cfg.n_layers * 2 * cfg.n_kv_heads * cfg.head_dim * esz
    rec["talker_ms_over_base"] = round(d_attn, 4)
    rec["kv_stream_TBps_of_the_increment"] = round(d_bytes / (d_attn * 1e-3) / 1e12, 3) if d_attn > 0 else None
print(json.dumps(rec))
eng.close()
