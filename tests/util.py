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


def calibrate_codec(w, ocfg, target_rms=0.2):
    """Scale the codec decoder's last conv so the PCM sits well inside the clamp range: a saturated
    output would hide errors (and blow up absolute ones)."""
    probe = dict(w)
    probe["cd.dec.conv_out.w"] = w["cd.dec.conv_out.w"] * np.float32(2.0 ** -20)
    probe["cd.dec.conv_out.b"] = w["cd.dec.conv_out.b"] * np.float32(2.0 ** -20)
    o = qo.Oracle(ocfg, max_ctx=16, weights=probe)
    codes = np.random.default_rng(0).integers(0, ocfg.cd_codebook, (6, ocfg.n_groups)).astype(np.int64)
    rms = float(np.sqrt(np.mean(o.vocoder(codes) ** 2)))
    o.close()
    k = np.float32(target_rms / max(rms, 1e-30) * 2.0 ** -20)
    w["cd.dec.conv_out.w"] = qo.bf16_round(w["cd.dec.conv_out.w"] * k)
    w["cd.dec.conv_out.b"] = qo.bf16_round(w["cd.dec.conv_out.b"] * k)
    return w


def tiny_pair(seed=0, max_batch=4, max_ctx=128, extra=None, flags=0, ocfg=None):
    """(engine, oracle, weights) on config_tiny (or `ocfg`) with identical bf16-representable weights."""
    import q3tts
    ocfg = ocfg or qo.config_tiny()
    w = calibrate_codec(qo.random_weights(ocfg, seed), ocfg)
    if extra:
        w.update(extra)
    eng = q3tts.Engine(to_q3cfg(ocfg), device=0, max_batch=max_batch, max_ctx=max_ctx, flags=flags)
    eng.load(w)
    orc = qo.Oracle(ocfg, max_ctx=max_ctx, weights=w)
    return eng, orc, w


def to_osampling(sp):
    return qo.Sampling(sp.temperature, sp.top_p, sp.top_k, sp.repetition_penalty, sp.max_new_tokens)


class Hip:
    """Device buffers for the tests of the "_dev" entry points: plain hipMalloc / hipMemcpy through the HIP runtime the library itself
    uses (any allocator would do: the entry points take raw device addresses)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.rt.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.rt.hipStreamDestroy.argtypes = [C.c_void_p]
        self.bufs = []

    def alloc(self, nbytes):
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), max(int(nbytes), 16)) == 0
        self.bufs.append(p)
        return p.value

    def put(self, arr):
        a = np.ascontiguousarray(arr)
        p = self.alloc(a.nbytes)
        assert self.rt.hipMemcpy(p, a.ctypes.data_as(self.C.c_void_p), a.nbytes, 1) == 0
        return p

    def write(self, ptr, arr):
        a = np.ascontiguousarray(arr)
        assert self.rt.hipMemcpy(ptr, a.ctypes.data_as(self.C.c_void_p), a.nbytes, 1) == 0

    def get(self, ptr, shape, dtype):
        out = np.empty(shape, dtype)
        assert self.rt.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), ptr, out.nbytes, 2) == 0
        return out

    def stream(self):
        s = self.C.c_void_p()
        assert self.rt.hipStreamCreate(self.C.byref(s)) == 0
        return s

    def free(self):
        for p in self.bufs:
            self.rt.hipFree(p)
        self.bufs = []
