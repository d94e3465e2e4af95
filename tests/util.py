"""Shared helpers for the parity tests: same seeded weights into the HIP engine and the oracle."""
import os

import numpy as np

import q3_oracle as qo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# token framing of TTSEngine::synthesize (reference src/tts_onnx.cpp:244-259)
IM_START, ASSISTANT, TTS_BOS, TTS_EOS, IM_END = 151644, 77091, 151672, 151673, 151645


def frame_tokens(text_ids):
    return np.array([IM_START, ASSISTANT, TTS_BOS] + list(text_ids) + [TTS_EOS, IM_END], np.int64)


def load_gold(name):
    z = np.load(os.path.join(GOLD, name))
    w = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    d = {k: z[k] for k in z.files if not k.startswith("w:")}
    return w, d


def to_q3cfg(ocfg):
    import q3tts
    return q3tts.Config.from_dict(ocfg.to_dict())


def to_ocfg(cfg):
    return qo.Config.from_dict(cfg.to_dict())


def tiny_pair(seed=0, max_batch=4, max_ctx=128, extra=None, flags=0):
    """(engine, oracle, weights) on config_tiny with identical bf16-representable weights."""
    import q3tts
    ocfg = qo.config_tiny()
    w = qo.random_weights(ocfg, seed)
    if extra:
        w.update(extra)
    eng = q3tts.Engine(to_q3cfg(ocfg), device=0, max_batch=max_batch, max_ctx=max_ctx, flags=flags)
    eng.load(w)
    orc = qo.Oracle(ocfg, max_ctx=max_ctx, weights=w)
    return eng, orc, w


def to_osampling(sp):
    return qo.Sampling(sp.temperature, sp.top_p, sp.top_k, sp.repetition_penalty, sp.max_new_tokens)
