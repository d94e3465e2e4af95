"""ctypes binding for the CPU oracle (oracle/q3_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)
import build as _build  # noqa: E402

_CFG_FIELDS = [
    ("hidden", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
    ("head_dim", C.c_int32), ("ffn", C.c_int32), ("vocab", C.c_int32),
    ("rope_theta", C.c_float), ("rms_eps", C.c_float),
    ("cp_layers", C.c_int32), ("cp_heads", C.c_int32), ("cp_kv_heads", C.c_int32), ("cp_head_dim", C.c_int32),
    ("cp_ffn", C.c_int32), ("n_groups", C.c_int32), ("sub_vocab", C.c_int32),
    ("cp_rope_theta", C.c_float), ("cp_rms_eps", C.c_float),
    ("text_vocab", C.c_int32), ("text_hidden", C.c_int32),
    ("cd_codebook", C.c_int32), ("cd_hidden", C.c_int32), ("cd_layers", C.c_int32), ("cd_heads", C.c_int32),
    ("cd_head_dim", C.c_int32), ("cd_ffn", C.c_int32), ("cd_window", C.c_int32),
    ("cd_rope_theta", C.c_float), ("cd_rms_eps", C.c_float),
    ("cd_n_up", C.c_int32), ("cd_up_ratios", C.c_int32 * 4),
    ("cd_decoder_dim", C.c_int32), ("cd_n_blocks", C.c_int32), ("cd_up_rates", C.c_int32 * 8),
    ("cd_tconv_trim", C.c_int32),
    ("codec_eos", C.c_int32), ("suppress_begin", C.c_int32), ("suppress_end", C.c_int32),
    ("spk_enc_dim", C.c_int32), ("spk_mel", C.c_int32), ("spk_channels", C.c_int32), ("spk_scale", C.c_int32),
    ("spk_se", C.c_int32), ("spk_att", C.c_int32),
    ("cp_hidden", C.c_int32),
]


class Config(C.Structure):
    _fields_ = _CFG_FIELDS

    def to_dict(self):
        d = {}
        for n, _ in _CFG_FIELDS:
            v = getattr(self, n)
            d[n] = list(v) if hasattr(v, "__len__") else v
        return d

    @classmethod
    def from_dict(cls, d):
        c = cls()
        for n, t in _CFG_FIELDS:
            v = d[n] if n in d or not (n.startswith("spk_") or n == "cp_hidden") else 0   # configs saved before these fields existed
            if hasattr(t, "_length_"):
                arr = t()
                for i, x in enumerate(v):
                    arr[i] = x
                setattr(c, n, arr)
            else:
                setattr(c, n, v)
        return c


class Sampling(C.Structure):
    # src/tts_onnx.h:99-105
    _fields_ = [("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int32),
                ("repetition_penalty", C.c_float), ("max_new_tokens", C.c_int32)]

    def __init__(self, temperature=0.8, top_p=0.95, top_k=50, repetition_penalty=1.0, max_new_tokens=2048):
        super().__init__(temperature, top_p, top_k, repetition_penalty, max_new_tokens)


def config_06b():
    """Qwen3-TTS-0.6B dims: src/tts_onnx.h:31-37 + [HINT] dims from SURVEY.md section 8."""
    return Config.from_dict(dict(
        hidden=1024, n_layers=28, n_heads=16, n_kv_heads=8, head_dim=128, ffn=3072, vocab=3072,
        rope_theta=1e6, rms_eps=1e-6,
        cp_layers=5, cp_heads=16, cp_kv_heads=8, cp_head_dim=128, cp_ffn=3072, n_groups=16, sub_vocab=2048,
        cp_rope_theta=1e6, cp_rms_eps=1e-6,
        text_vocab=151936, text_hidden=2048,
        cd_codebook=2048, cd_hidden=1024, cd_layers=8, cd_heads=16, cd_head_dim=64, cd_ffn=3072, cd_window=72,
        cd_rope_theta=10000.0, cd_rms_eps=1e-5,
        cd_n_up=2, cd_up_ratios=[2, 2, 0, 0], cd_decoder_dim=1536, cd_n_blocks=4,
        cd_up_rates=[8, 5, 4, 3, 0, 0, 0, 0], cd_tconv_trim=0,
        codec_eos=2150, suppress_begin=2048, suppress_end=3072,
        spk_enc_dim=1024, spk_mel=128, spk_channels=512, spk_scale=8, spk_se=128, spk_att=128))


def config_17b():
    """Qwen3-TTS-1.7B dims [HINT: the public 1.7B checkpoints' config.json]: talker twice as wide, the code predictor keeps the 0.6B
    width behind a 2048 -> 1024 projection; the speaker embedding is a talker-width row.  Beyond what the reference runs
    (README.md:125 lists 1.7B as planned; tts_onnx.h:31-37 hard-codes the 0.6B talker dims)."""
    d = config_06b().to_dict()
    d.update(hidden=2048, ffn=6144, cp_hidden=1024, spk_enc_dim=2048)
    return Config.from_dict(d)


def config_tiny():
    """Small config with the same structure; every kernel path is exercised in seconds on CPU.
    The codec vocabulary keeps the real control-token ids (2148..2157, tts_onnx.h:50-56) so prompt
    assembly runs unchanged; code0 is confined to [0,64) + EOS so that EOS is actually reachable."""
    return Config.from_dict(dict(
        hidden=64, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=16, ffn=96, vocab=2176,
        rope_theta=1e6, rms_eps=1e-6,
        cp_layers=2, cp_heads=4, cp_kv_heads=2, cp_head_dim=16, cp_ffn=96, n_groups=16, sub_vocab=64,
        cp_rope_theta=1e6, cp_rms_eps=1e-6,
        text_vocab=152000, text_hidden=32,
        cd_codebook=64, cd_hidden=32, cd_layers=2, cd_heads=2, cd_head_dim=16, cd_ffn=48, cd_window=4,
        cd_rope_theta=10000.0, cd_rms_eps=1e-5,
        cd_n_up=2, cd_up_ratios=[2, 2, 0, 0], cd_decoder_dim=64, cd_n_blocks=4,
        cd_up_rates=[8, 5, 4, 3, 0, 0, 0, 0], cd_tconv_trim=0,
        codec_eos=2150, suppress_begin=64, suppress_end=2176,
        spk_enc_dim=64, spk_mel=128, spk_channels=32, spk_scale=8, spk_se=8, spk_att=8))


def config_medium():
    """Dims that are multiples of 128, so batches > 8 rows take the bf16-MFMA skinny-GEMM path
    (K in {128, 256}) while staying small enough for the CPU oracle."""
    d = config_tiny().to_dict()
    d.update(hidden=128, n_layers=2, n_heads=2, n_kv_heads=1, head_dim=64, ffn=256,
             cp_layers=2, cp_heads=2, cp_kv_heads=1, cp_head_dim=64, cp_ffn=256, text_hidden=64, spk_enc_dim=128)
    return Config.from_dict(d)


def config_tiny_proj():
    """config_tiny with a narrower predictor behind cp.proj (the 1.7B structure)."""
    d = config_tiny().to_dict()
    d.update(cp_hidden=48)
    return Config.from_dict(d)


def config_medium_proj():
    """config_medium with the 1.7B structure: talker 256 wide, predictor 128 wide behind cp.proj."""
    d = config_medium().to_dict()
    d.update(hidden=256, ffn=384, cp_hidden=128, spk_enc_dim=256)
    return Config.from_dict(d)


def tensor_specs(cfg):
    """(name, shape, kind) for every tensor of the model.  kind: 'w' matrix, 'norm' (ones-centred),
    'b' bias, 'scale' LayerScale/gamma, 'snake' alpha/beta."""
    c = cfg
    H = c.hidden
    out = []

    def layers(prefix, n, Hh, nq, nkv, d, ffn, qk, ls):
        for i in range(n):
            p = f"{prefix}.layers.{i}."
            out.append((p + "input_norm", (Hh,), "norm"))
            out.append((p + "q_proj", (nq * d, Hh), "w"))
            out.append((p + "k_proj", (nkv * d, Hh), "w"))
            out.append((p + "v_proj", (nkv * d, Hh), "w"))
            out.append((p + "o_proj", (Hh, nq * d), "w"))
            if qk:
                out.append((p + "q_norm", (d,), "norm"))
                out.append((p + "k_norm", (d,), "norm"))
            out.append((p + "post_norm", (Hh,), "norm"))
            out.append((p + "gate_proj", (ffn, Hh), "w"))
            out.append((p + "up_proj", (ffn, Hh), "w"))
            out.append((p + "down_proj", (Hh, ffn), "w"))
            if ls:
                out.append((p + "attn_scale", (Hh,), "scale"))
                out.append((p + "mlp_scale", (Hh,), "scale"))

    layers("talker", c.n_layers, H, c.n_heads, c.n_kv_heads, c.head_dim, c.ffn, True, False)
    out.append(("talker.norm", (H,), "norm"))
    out.append(("talker.codec_head", (c.vocab, H), "w"))
    out.append(("talker.codec_embed", (c.vocab, H), "w"))
    out.append(("text.embed", (c.text_vocab, c.text_hidden), "w"))
    out.append(("text.fc1.w", (c.text_hidden, c.text_hidden), "w"))
    out.append(("text.fc1.b", (c.text_hidden,), "b"))
    out.append(("text.fc2.w", (H, c.text_hidden), "w"))
    out.append(("text.fc2.b", (H,), "b"))
    Hc = c.cp_hidden if c.cp_hidden > 0 else H          # 1.7B: narrower predictor behind cp.proj
    layers("cp", c.cp_layers, Hc, c.cp_heads, c.cp_kv_heads, c.cp_head_dim, c.cp_ffn, True, False)
    out.append(("cp.norm", (Hc,), "norm"))
    if Hc != H:
        out.append(("cp.proj.w", (Hc, H), "w"))
        out.append(("cp.proj.b", (Hc,), "b"))
    for j in range(c.n_groups - 1):
        out.append((f"cp.head.{j}", (c.sub_vocab, Hc), "w"))
    for j in range(c.n_groups - 1):
        out.append((f"cp.embed.{j}", (c.sub_vocab, H), "w"))
    CH = c.cd_hidden
    layers("cd", c.cd_layers, CH, c.cd_heads, c.cd_heads, c.cd_head_dim, c.cd_ffn, False, True)
    out.append(("cd.norm", (CH,), "norm"))
    out.append(("cd.code_embed", (c.n_groups * c.cd_codebook, CH), "w"))
    for s in range(c.cd_n_up):
        f = c.cd_up_ratios[s]
        p = f"cd.up.{s}."
        out.append((p + "tconv.w", (CH, CH, f), "w"))
        out.append((p + "tconv.b", (CH,), "b"))
        out.append((p + "cnx.dw.w", (CH, 1, 7), "w"))
        out.append((p + "cnx.dw.b", (CH,), "b"))
        out.append((p + "cnx.ln.w", (CH,), "norm"))
        out.append((p + "cnx.ln.b", (CH,), "b"))
        out.append((p + "cnx.pw1.w", (4 * CH, CH), "w"))
        out.append((p + "cnx.pw1.b", (4 * CH,), "b"))
        out.append((p + "cnx.pw2.w", (CH, 4 * CH), "w"))
        out.append((p + "cnx.pw2.b", (CH,), "b"))
        out.append((p + "cnx.gamma", (CH,), "scale"))
    D = c.cd_decoder_dim
    out.append(("cd.dec.conv_in.w", (D, CH, 7), "w"))
    out.append(("cd.dec.conv_in.b", (D,), "b"))
    for i in range(c.cd_n_blocks):
        cin, cout, r = D >> i, D >> (i + 1), c.cd_up_rates[i]
        p = f"cd.dec.blocks.{i}."
        out.append((p + "snake.alpha", (cin,), "snake"))
        out.append((p + "snake.beta", (cin,), "snake"))
        out.append((p + "tconv.w", (cin, cout, 2 * r), "w"))
        out.append((p + "tconv.b", (cout,), "b"))
        for u in range(3):
            q = p + f"res.{u}."
            out.append((q + "act1.alpha", (cout,), "snake"))
            out.append((q + "act1.beta", (cout,), "snake"))
            out.append((q + "conv1.w", (cout, cout, 7), "w"))
            out.append((q + "conv1.b", (cout,), "b"))
            out.append((q + "act2.alpha", (cout,), "snake"))
            out.append((q + "act2.beta", (cout,), "snake"))
            out.append((q + "conv2.w", (cout, cout, 1), "w"))
            out.append((q + "conv2.b", (cout,), "b"))
    OD = D >> c.cd_n_blocks
    out.append(("cd.dec.snake_out.alpha", (OD,), "snake"))
    out.append(("cd.dec.snake_out.beta", (OD,), "snake"))
    out.append(("cd.dec.conv_out.w", (1, OD, 7), "w"))
    out.append(("cd.dec.conv_out.b", (1,), "b"))
    if c.spk_enc_dim > 0:   # ECAPA-TDNN speaker encoder, torch Conv1d layout [out][in][k]
        SC, sub = c.spk_channels, c.spk_channels // c.spk_scale

        def conv(name, cout, cin, k):
            out.append((name + ".w", (cout, cin, k), "w"))
            out.append((name + ".b", (cout,), "b"))
        conv("spk.tdnn0", SC, c.spk_mel, 5)
        for i in range(3):
            conv(f"spk.blocks.{i}.tdnn1", SC, SC, 1)
            for j in range(c.spk_scale - 1):
                conv(f"spk.blocks.{i}.res2net.{j}", sub, sub, 3)
            conv(f"spk.blocks.{i}.tdnn2", SC, SC, 1)
            conv(f"spk.blocks.{i}.se1", c.spk_se, SC, 1)
            conv(f"spk.blocks.{i}.se2", SC, c.spk_se, 1)
        conv("spk.mfa", 3 * SC, 3 * SC, 1)
        conv("spk.asp.tdnn", c.spk_att, 9 * SC, 1)
        conv("spk.asp.conv", 3 * SC, c.spk_att, 1)
        conv("spk.fc", c.spk_enc_dim, 6 * SC, 1)
    return out


def bf16_round(a):
    """Round fp32 -> nearest-even bf16, returned as fp32 (values exactly representable in bf16)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def random_weights(cfg, seed=0, text_rows=None):
    """Seeded numpy weights (bf16-representable fp32) for small configs.  Matrix std is
    1/sqrt(fan_in) so activations stay O(1) through every stack."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape, kind in tensor_specs(cfg):
        if kind == "w":
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            if name.endswith("tconv.w"):
                fan_in = shape[0] * max(1, shape[2] // 2) if shape[2] > 2 else shape[0]
            if name in ("talker.codec_embed", "text.embed", "cd.code_embed") or name.startswith("cp.embed"):
                a = rng.standard_normal(shape, dtype=np.float32)
            else:
                a = rng.standard_normal(shape, dtype=np.float32) / np.sqrt(fan_in)
        elif kind == "norm":
            a = 1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif kind == "b":
            a = 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif kind == "scale":
            a = 0.5 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
        else:  # snake
            a = 0.3 * rng.standard_normal(shape, dtype=np.float32)
        w[name] = bf16_round(a)
    return w


_lib = None


def default_threads(cfg=None):
    """OpenMP threads for the oracle: the GPU boxes expose many more hardware threads than the
    job's CPU share (16 per GPU), and tiny configs are dominated by fork/join cost."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    n = max(1, min(n, 16))
    if cfg is not None and cfg.hidden <= 128:
        n = 1
    return n


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # no spinning between the many small regions
        os.environ.setdefault("OMP_NUM_THREADS", str(default_threads()))
        so = _build.build()
        L = C.CDLL(so)
        L.q3o_create.restype = C.c_void_p
        L.q3o_create.argtypes = [C.POINTER(Config), C.c_int]
        L.q3o_destroy.argtypes = [C.c_void_p]
        L.q3o_last_error.restype = C.c_char_p
        L.q3o_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
        L.q3o_tensor_numel.restype = C.c_int64
        L.q3o_tensor_numel.argtypes = [C.c_void_p, C.c_char_p]
        L.q3o_set_threads.argtypes = [C.c_int]
        L.q3o_set_kv_bf16.argtypes = [C.c_void_p, C.c_int]
        L.q3o_set_sampler_exp_libm.argtypes = [C.c_int]
        L.q3o_text_project.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.q3o_codec_embed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.q3o_cp_embed.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.q3o_prefill.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.q3o_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.q3o_code_predictor.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.q3o_vocoder.restype = C.c_int64
        L.q3o_vocoder.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        L.q3o_vocoder_len.restype = C.c_int64
        L.q3o_vocoder_len.argtypes = [C.POINTER(Config), C.c_int]
        L.q3o_vocoder_tap.restype = C.c_int64
        L.q3o_vocoder_tap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        L.q3o_speaker_encoder.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.q3o_rng_uniform.restype = C.c_float
        L.q3o_rng_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.q3o_sample.restype = C.c_int64
        L.q3o_sample.argtypes = [C.c_void_p, C.c_int, C.POINTER(Sampling), C.c_float]
        L.q3o_softmax.argtypes = [C.c_void_p, C.c_int]
        L.q3o_set_logits_dump.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.q3o_sample_trace.argtypes = [C.c_void_p, C.c_int, C.POINTER(Sampling), C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.q3o_expf.restype = C.c_float
        L.q3o_expf.argtypes = [C.c_float]
        L.q3o_sample_margin.restype = C.c_int64
        L.q3o_sample_margin.argtypes = [C.c_void_p, C.c_int, C.POINTER(Sampling), C.c_float, C.POINTER(C.c_float)]
        L.q3o_top_k_filter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.q3o_top_p_filter.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.q3o_build_prompt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.q3o_trailing.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.q3o_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Sampling), C.c_uint64, C.c_uint32,
                                   C.c_int, C.c_int, C.c_void_p]
        L.q3o_generate_margins.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Sampling), C.c_uint64, C.c_uint32,
                                           C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.q3o_synthesize_tokens.restype = C.c_int64
        L.q3o_synthesize_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Sampling), C.c_uint64,
                                            C.c_uint32, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, cfg, max_ctx=256, weights=None, kv_bf16=False):
        self.L = lib()
        self.cfg = cfg
        self.max_ctx = max_ctx
        self.h = self.L.q3o_create(C.byref(cfg), max_ctx)
        if kv_bf16:   # the product's Q3TTS_FLAG_KV_BF16: talker K / V rounded to bf16 on append
            self.L.q3o_set_kv_bf16(self.h, 1)
        self.threads = default_threads(cfg)
        self.L.q3o_set_threads(self.threads)
        if weights is not None:
            self.load(weights)

    def close(self):
        if self.h:
            self.L.q3o_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise RuntimeError(self.L.q3o_last_error().decode())
        return rc

    def set_tensor(self, name, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        self._check(self.L.q3o_set_tensor(self.h, name.encode(), _p(a), a.size))

    def load(self, weights):
        for k, v in weights.items():
            self.set_tensor(k, v)

    def speaker_encoder(self, mel):
        """mel [n_mels][frames] (MelExtractor layout) -> [spk_enc_dim]"""
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        assert mel.ndim == 2 and mel.shape[0] == self.cfg.spk_mel
        out = np.zeros(self.cfg.spk_enc_dim, np.float32)
        self._check(self.L.q3o_speaker_encoder(self.h, _p(mel), mel.shape[1], _p(out)))
        return out

    def text_project(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty((ids.size, self.cfg.hidden), np.float32)
        self._check(self.L.q3o_text_project(self.h, _p(ids), ids.size, _p(out)))
        return out

    def codec_embed(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty((ids.size, self.cfg.hidden), np.float32)
        self._check(self.L.q3o_codec_embed(self.h, _p(ids), ids.size, _p(out)))
        return out

    def cp_embed(self, tok, step):
        out = np.empty(self.cfg.hidden, np.float32)
        self._check(self.L.q3o_cp_embed(self.h, int(tok), int(step), _p(out)))
        return out

    def prefill(self, embeds):
        e = np.ascontiguousarray(embeds, dtype=np.float32)
        S = e.shape[0]
        logits = np.empty((S, self.cfg.vocab), np.float32)
        lh = np.empty(self.cfg.hidden, np.float32)
        self._check(self.L.q3o_prefill(self.h, _p(e), S, _p(logits), _p(lh)))
        return logits, lh

    def decode(self, embed):
        e = np.ascontiguousarray(embed, dtype=np.float32)
        logits = np.empty(self.cfg.vocab, np.float32)
        lh = np.empty(self.cfg.hidden, np.float32)
        self._check(self.L.q3o_decode(self.h, _p(e), _p(logits), _p(lh)))
        return logits, lh

    def code_predictor(self, seq, step):
        s = np.ascontiguousarray(seq, dtype=np.float32)
        logits = np.empty(self.cfg.sub_vocab, np.float32)
        self._check(self.L.q3o_code_predictor(self.h, _p(s), s.shape[0], int(step), _p(logits)))
        return logits

    def vocoder_len(self, F):
        return int(self.L.q3o_vocoder_len(C.byref(self.cfg), F))

    def vocoder(self, codes):
        c = np.ascontiguousarray(codes, dtype=np.int64)
        F = c.shape[0]
        n = self.vocoder_len(F)
        pcm = np.empty(n, np.float32)
        got = self._check(self.L.q3o_vocoder(self.h, _p(c), F, _p(pcm), n))
        assert got == n, (got, n)
        return pcm

    def vocoder_tap(self, codes, stage, cap):
        c = np.ascontiguousarray(codes, dtype=np.int64)
        out = np.empty(cap, np.float32)
        n = self._check(self.L.q3o_vocoder_tap(self.h, _p(c), c.shape[0], stage, _p(out), cap))
        return out[:n]

    def sample(self, logits, sp, u):
        a = np.ascontiguousarray(logits, dtype=np.float32)
        return int(self.L.q3o_sample(_p(a), a.size, C.byref(sp), C.c_float(u)))

    def dump_logits_of(self, frame, group, n):
        """the next generate*() call copies the logits row of decision (frame, group) into the returned array (debugging aid)"""
        buf = np.zeros(n, np.float32)
        self._dump_keep = buf
        self.L.q3o_set_logits_dump(int(frame), int(group), _p(buf))
        return buf

    def sample_trace(self, logits, sp):
        """(top-p running sums in sorted order, draw running sums in index order (-1 where p == 0), total): see q3o_sample_trace"""
        a = np.ascontiguousarray(logits, dtype=np.float32)
        tc, dc, tot = np.zeros(a.size, np.float32), np.zeros(a.size, np.float32), C.c_float(0)
        self.L.q3o_sample_trace(_p(a), a.size, C.byref(sp), _p(tc), _p(dc), C.byref(tot))
        return tc, dc, float(tot.value)

    def sample_margin(self, logits, sp, u):
        """(token, decision margin): see q3o_sample_margin"""
        a = np.ascontiguousarray(logits, dtype=np.float32)
        m = C.c_float(0)
        t = int(self.L.q3o_sample_margin(_p(a), a.size, C.byref(sp), C.c_float(u), C.byref(m)))
        return t, float(m.value)

    def build_prompt(self, ids, lang=0, speaker=None):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        prompt = np.zeros((16, self.cfg.hidden), np.float32)
        S = C.c_int(0)
        sp = None
        if speaker is not None:
            sp = np.ascontiguousarray(speaker, dtype=np.float32)
        self._check(self.L.q3o_build_prompt(self.h, _p(ids), ids.size, lang, _p(sp) if sp is not None else None,
                                            _p(prompt), C.byref(S)))
        return prompt[:S.value].copy()

    def trailing(self):
        pad = np.empty(self.cfg.hidden, np.float32)
        n = self.L.q3o_trailing(self.h, None, 0, _p(pad))
        rows = np.empty((n, self.cfg.hidden), np.float32)
        self.L.q3o_trailing(self.h, _p(rows), n, _p(pad))
        return rows, pad

    def generate(self, prompt, sp, seed=0, stream=0, cp_cached=True, ignore_eos=False):
        p = np.ascontiguousarray(prompt, dtype=np.float32)
        codes = np.zeros((sp.max_new_tokens, self.cfg.n_groups), np.int64)
        F = self._check(self.L.q3o_generate(self.h, _p(p), p.shape[0], C.byref(sp), seed, stream,
                                            int(cp_cached), int(ignore_eos), _p(codes)))
        return codes[:F].copy()

    def generate_margins(self, prompt, sp, seed=0, stream=0, cp_cached=True, ignore_eos=False):
        """generate() plus margins [F][2 + n_groups]: top-2 logit margin of each frame's code0 decision, smallest top-2 margin over its
        sub-codes, then the sampler decision margin (top-k gap / top-p cut / draw edge) of each of the frame's decisions"""
        p = np.ascontiguousarray(prompt, dtype=np.float32)
        codes = np.zeros((sp.max_new_tokens, self.cfg.n_groups), np.int64)
        mg = np.zeros((sp.max_new_tokens, 2 + self.cfg.n_groups), np.float32)
        F = self._check(self.L.q3o_generate_margins(self.h, _p(p), p.shape[0], C.byref(sp), seed, stream,
                                                    int(cp_cached), int(ignore_eos), _p(codes), _p(mg)))
        return codes[:F].copy(), mg[:F].copy()


def rng_uniform(seed, stream, frame, group):
    return float(lib().q3o_rng_uniform(seed, stream, frame, group))
