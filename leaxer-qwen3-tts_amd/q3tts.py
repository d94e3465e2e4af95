"""ctypes binding of libq3tts_hip.so (C-ABI: include/q3tts.h).

Host-side mirror of the reference's TTSEngine run_* family (reference src/tts_onnx.h:196-212) plus
the batched generation entry points.  There is NO CPU fallback: if the HIP library is missing or
no MI355X is visible, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("Q3TTS_LIB") or os.path.join(_HERE, "libq3tts_hip.so")   # Q3TTS_LIB: kernel experiments (tools/)

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
            v = d[n] if n in d or not (n.startswith("spk_") or n == "cp_hidden") else 0
            if hasattr(t, "_length_"):
                arr = t()
                for i, x in enumerate(v):
                    arr[i] = x
                setattr(c, n, arr)
            else:
                setattr(c, n, v)
        return c


class Sampling(C.Structure):
    """SamplingParams, reference src/tts_onnx.h:99-105."""
    _fields_ = [("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int32),
                ("repetition_penalty", C.c_float), ("max_new_tokens", C.c_int32)]

    def __init__(self, temperature=0.8, top_p=0.95, top_k=50, repetition_penalty=1.0, max_new_tokens=2048):
        super().__init__(temperature, top_p, top_k, repetition_penalty, max_new_tokens)


FLAG_NO_GRAPH = 1
FLAG_NO_FUSED_CP = 2
FLAG_FP32_CODEC = 4
FLAG_KV_ROUND_BF16 = 16   # test aid: fp32 KV storage of the bf16-rounded rows (must equal FLAG_KV_BF16 bit for bit)
FLAG_TEST_HOOKS = 32   # the engine honours the test suite's fault-injection environment hooks
FLAG_KV_BF16 = 8   # talker KV cache in bf16 (rounded on append, fp32 math); the oracle has the same switch (Oracle(kv_bf16=True))

# every symbol include/q3tts.h declares
EXPORTS = [
    "q3tts_default_config", "q3tts_create", "q3tts_create_pooled", "q3tts_kv_pool_info", "q3tts_sched_stats", "q3tts_destroy", "q3tts_last_error", "q3tts_num_tensors",
    "q3tts_tensor_info", "q3tts_set_tensor_host", "q3tts_get_tensor_host", "q3tts_fill_synthetic", "q3tts_finalize",
    "q3tts_text_project_host", "q3tts_codec_embed_host", "q3tts_cp_embed_host", "q3tts_talker_prefill_host",
    "q3tts_talker_decode_host", "q3tts_code_predictor_host", "q3tts_codec_decode_host", "q3tts_codec_decode_batch_host", "q3tts_codec_decode_len",
    "q3tts_sample_host", "q3tts_rng_uniform", "q3tts_build_prompt_host", "q3tts_slot_begin", "q3tts_decode_steps",
    "q3tts_slot_status", "q3tts_slot_codes_host", "q3tts_slot_codec_decode_host", "q3tts_slot_release",
    "q3tts_synthesize_batch_host", "q3tts_last_decode_ms", "q3tts_last_codec_ms", "q3tts_decode_step_bytes",
    "q3tts_codec_decode_dev", "q3tts_stream", "q3tts_counters", "q3tts_stage_profile", "q3tts_prefill_profile", "q3tts_codec_plane_stats", "q3tts_measure_skip_frames", "q3tts_test_poison_workspace", "q3tts_test_group_final_conv", "q3tts_test_final_conv_partials", "q3tts_codec_stream_begin", "q3tts_codec_stream_push_host", "q3tts_codec_stream_end", "q3tts_talker_prefill_dev", "q3tts_talker_decode_dev", "q3tts_code_predictor_dev", "q3tts_sample_dev", "q3tts_config_num_tensors", "q3tts_config_tensor_info", "q3tts_read_weights_config", "q3tts_load_weights_file", "q3tts_save_weights_file",
    "q3tts_tokenizer_create", "q3tts_tokenizer_destroy", "q3tts_tokenizer_load_vocab", "q3tts_tokenizer_load_merges",
    "q3tts_tokenizer_ready", "q3tts_tokenize",
    "q3tts_synthesize_clone_batch_host", "q3tts_synthesize_schedule_host", "q3tts_read_wav_host", "q3tts_resample_host", "q3tts_mel_host",
    "q3tts_has_speaker_encoder", "q3tts_speaker_encoder_host", "q3tts_extract_speaker_embedding_host",
    "q3tts_codec_decode_chunked_host", "q3tts_slot_codec_decode_range_host", "q3tts_slot_logits_host", "q3tts_step_logits_host",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python leaxer-qwen3-tts_amd/build.py` "
                           "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    L.q3tts_default_config.argtypes = [C.c_char_p, C.POINTER(Config)]
    L.q3tts_create.restype = vp
    L.q3tts_create.argtypes = [C.POINTER(Config), i32, i32, i32, C.c_uint32]
    L.q3tts_create_pooled.restype = vp
    L.q3tts_create_pooled.argtypes = [C.POINTER(Config), i32, i32, i32, C.c_int64, C.c_uint32]
    L.q3tts_kv_pool_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.q3tts_sched_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]
    L.q3tts_destroy.argtypes = [vp]
    L.q3tts_last_error.restype = C.c_char_p
    L.q3tts_last_error.argtypes = [vp]
    L.q3tts_num_tensors.argtypes = [vp]
    L.q3tts_tensor_info.argtypes = [vp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i32)]
    L.q3tts_set_tensor_host.argtypes = [vp, C.c_char_p, vp, i64]
    L.q3tts_get_tensor_host.argtypes = [vp, C.c_char_p, vp, i64]
    L.q3tts_fill_synthetic.argtypes = [vp, C.c_uint64]
    L.q3tts_finalize.argtypes = [vp]
    L.q3tts_text_project_host.argtypes = [vp, vp, i32, vp]
    L.q3tts_codec_embed_host.argtypes = [vp, vp, i32, vp]
    L.q3tts_cp_embed_host.argtypes = [vp, i64, i32, vp]
    L.q3tts_talker_prefill_host.argtypes = [vp, i32, vp, i32, vp, vp]
    L.q3tts_talker_decode_host.argtypes = [vp, i32, vp, vp, vp]
    L.q3tts_code_predictor_host.argtypes = [vp, vp, i32, i32, vp]
    L.q3tts_codec_decode_host.argtypes = [vp, vp, i32, vp, i64, C.POINTER(i64)]
    L.q3tts_codec_decode_batch_host.argtypes = [vp, i32, vp, vp, vp, i64, vp]
    L.q3tts_codec_decode_len.restype = i64
    L.q3tts_codec_decode_len.argtypes = [C.POINTER(Config), i32]
    L.q3tts_sample_host.argtypes = [vp, vp, i32, C.POINTER(Sampling), f32, i32, C.POINTER(i64)]
    L.q3tts_rng_uniform.restype = f32
    L.q3tts_rng_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    L.q3tts_build_prompt_host.argtypes = [vp, vp, i32, i32, vp, vp, C.POINTER(i32), vp, i32, C.POINTER(i32)]
    L.q3tts_slot_begin.argtypes = [vp, i32, vp, i32, vp, i32, C.POINTER(Sampling), C.c_uint64, C.c_uint32, i32]
    L.q3tts_decode_steps.argtypes = [vp, i32]
    L.q3tts_slot_status.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.q3tts_slot_codes_host.argtypes = [vp, i32, vp, i32]
    L.q3tts_slot_codec_decode_host.argtypes = [vp, i32, vp, i64, C.POINTER(i64)]
    L.q3tts_slot_release.argtypes = [vp, i32]
    L.q3tts_slot_logits_host.argtypes = [vp, i32, vp, vp]
    L.q3tts_step_logits_host.argtypes = [vp, i32, vp, i32]
    L.q3tts_synthesize_batch_host.argtypes = [vp, i32, vp, vp, i32, C.POINTER(Sampling), C.c_uint64, i32,
                                              vp, i64, vp, vp, vp]
    L.q3tts_last_decode_ms.argtypes = [vp, C.POINTER(f32), C.POINTER(i32)]
    L.q3tts_last_codec_ms.argtypes = [vp, C.POINTER(f32)]
    L.q3tts_stage_profile.argtypes = [vp, i32, C.POINTER(C.c_double)]
    L.q3tts_codec_plane_stats.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.q3tts_counters.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(i64), i32]
    L.q3tts_decode_step_bytes.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.q3tts_read_weights_config.argtypes = [C.c_char_p, C.POINTER(Config)]
    L.q3tts_load_weights_file.argtypes = [vp, C.c_char_p]
    L.q3tts_save_weights_file.argtypes = [vp, C.c_char_p]
    L.q3tts_synthesize_clone_batch_host.argtypes = [vp, i32, vp, vp, i32, vp, C.POINTER(Sampling), C.c_uint64, i32, vp, i64, vp, vp, vp]
    L.q3tts_synthesize_schedule_host.argtypes = [vp, i32, vp, vp, i32, vp, C.POINTER(Sampling), vp, C.c_uint64, i32, vp, i64, vp, vp, vp]
    L.q3tts_read_wav_host.argtypes = [C.c_char_p, vp, i64, C.POINTER(i64), C.POINTER(C.c_int32)]
    L.q3tts_resample_host.restype = i64
    L.q3tts_resample_host.argtypes = [vp, i64, i32, i32, vp, i64]
    L.q3tts_mel_host.argtypes = [vp, i64, vp, i64, C.POINTER(C.c_int32)]
    L.q3tts_has_speaker_encoder.argtypes = [vp]
    L.q3tts_speaker_encoder_host.argtypes = [vp, vp, i32, vp]
    L.q3tts_extract_speaker_embedding_host.argtypes = [vp, C.c_char_p, vp]
    L.q3tts_codec_decode_chunked_host.argtypes = [vp, vp, i32, i32, i32, vp, i64, C.POINTER(i64)]
    L.q3tts_slot_codec_decode_range_host.argtypes = [vp, i32, i32, i32, i32, vp, i64, C.POINTER(i64)]
    L.q3tts_tokenizer_create.restype = vp
    L.q3tts_tokenizer_create.argtypes = []
    L.q3tts_tokenizer_destroy.restype = None
    L.q3tts_tokenizer_destroy.argtypes = [vp]
    L.q3tts_tokenizer_load_vocab.argtypes = [vp, C.c_char_p]
    L.q3tts_tokenizer_load_merges.argtypes = [vp, C.c_char_p]
    L.q3tts_tokenizer_ready.argtypes = [vp]
    L.q3tts_tokenize.restype = i64
    L.q3tts_tokenize.argtypes = [vp, C.c_char_p, i64, C.POINTER(C.c_int32), i64]
    _lib = L
    return L


_KINDS = ("w", "norm", "b", "scale", "snake")


def tensor_specs(cfg):
    """(name, shape, kind) of every tensor the config implies — the engine's registry, from the library, no GPU needed."""
    L = lib()
    L.q3tts_config_num_tensors.argtypes = [C.c_void_p]
    L.q3tts_config_tensor_info.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    n = L.q3tts_config_num_tensors(C.byref(cfg))
    if n <= 0:
        raise ValueError("model config out of range")
    out, name, shape, nd, kind = [], C.create_string_buffer(256), (C.c_int64 * 4)(), C.c_int32(0), C.c_int32(0)
    for i in range(n):
        if L.q3tts_config_tensor_info(C.byref(cfg), i, name, 256, shape, C.byref(nd), C.byref(kind)) != 0:
            raise RuntimeError("q3tts_config_tensor_info failed")
        out.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value)), _KINDS[kind.value]))
    return out


def default_config(name="0.6b"):
    c = Config()
    if lib().q3tts_default_config(name.encode(), C.byref(c)) != 0:
        raise ValueError(f"unknown config {name!r}")
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One engine = one GPU, `max_batch` utterance slots, device-resident weights + KV cache."""

    def __init__(self, cfg, device=0, max_batch=1, max_ctx=2304, flags=0, kv_pool_tokens=0):
        """kv_pool_tokens > 0 bounds the talker's KV page pool (q3tts_create_pooled); 0 reserves max_batch x max_ctx."""
        self.L = lib()
        self.cfg = cfg
        self.max_batch = max_batch
        self.max_ctx = max_ctx
        self.flags = flags
        self.h = self.L.q3tts_create_pooled(C.byref(cfg), device, max_batch, max_ctx, kv_pool_tokens, flags)
        if not self.h:
            raise RuntimeError("q3tts_create failed: " + self.L.q3tts_last_error(None).decode())

    def kv_pool_info(self):
        """(tokens per page, pages in the pool, pages free)"""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self.L.q3tts_kv_pool_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def sched_stats(self):
        """(admitted, preempted, peak live) of the last synthesize_batch call"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int32()
        self._ck(self.L.q3tts_sched_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def close(self):
        if getattr(self, "h", None):
            self.L.q3tts_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise RuntimeError(self.L.q3tts_last_error(self.h).decode())
        return rc

    # ---- weights ----
    def tensor_infos(self):
        out = []
        name = C.create_string_buffer(128)
        shape = (C.c_int64 * 4)()
        nd = C.c_int(0)
        for i in range(self.L.q3tts_num_tensors(self.h)):
            self._ck(self.L.q3tts_tensor_info(self.h, i, name, 128, shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(shape[k] for k in range(nd.value))))
        return out

    def set_tensor(self, name, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        self._ck(self.L.q3tts_set_tensor_host(self.h, name.encode(), _p(a), a.size))

    def get_tensor(self, name, shape):
        out = np.empty(shape, np.float32)
        self._ck(self.L.q3tts_get_tensor_host(self.h, name.encode(), _p(out), out.size))
        return out

    def load(self, weights):
        for k, v in weights.items():
            self.set_tensor(k, v)
        self.finalize()

    def fill_synthetic(self, seed=0):
        self._ck(self.L.q3tts_fill_synthetic(self.h, seed))
        self.finalize()

    def finalize(self):
        self._ck(self.L.q3tts_finalize(self.h))

    def save_weights(self, path):
        self._ck(self.L.q3tts_save_weights_file(self.h, os.fsencode(path)))

    def load_weights(self, path):
        self._ck(self.L.q3tts_load_weights_file(self.h, os.fsencode(path)))

    # ---- session-shaped ----
    def text_project(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty((ids.size, self.cfg.hidden), np.float32)
        self._ck(self.L.q3tts_text_project_host(self.h, _p(ids), ids.size, _p(out)))
        return out

    def codec_embed(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty((ids.size, self.cfg.hidden), np.float32)
        self._ck(self.L.q3tts_codec_embed_host(self.h, _p(ids), ids.size, _p(out)))
        return out

    def cp_embed(self, tok, step):
        out = np.empty(self.cfg.hidden, np.float32)
        self._ck(self.L.q3tts_cp_embed_host(self.h, int(tok), int(step), _p(out)))
        return out

    def prefill(self, embeds, slot=0):
        e = np.ascontiguousarray(embeds, dtype=np.float32)
        S = e.shape[0]
        logits = np.empty((S, self.cfg.vocab), np.float32)
        lh = np.empty(self.cfg.hidden, np.float32)
        self._ck(self.L.q3tts_talker_prefill_host(self.h, slot, _p(e), S, _p(logits), _p(lh)))
        return logits, lh

    def decode(self, embed, slot=0):
        e = np.ascontiguousarray(embed, dtype=np.float32)
        logits = np.empty(self.cfg.vocab, np.float32)
        lh = np.empty(self.cfg.hidden, np.float32)
        self._ck(self.L.q3tts_talker_decode_host(self.h, slot, _p(e), _p(logits), _p(lh)))
        return logits, lh

    def code_predictor(self, seq, step):
        s = np.ascontiguousarray(seq, dtype=np.float32)
        logits = np.empty(self.cfg.sub_vocab, np.float32)
        self._ck(self.L.q3tts_code_predictor_host(self.h, _p(s), s.shape[0], int(step), _p(logits)))
        return logits

    def codec_decode_len(self, F):
        return int(self.L.q3tts_codec_decode_len(C.byref(self.cfg), F))

    def codec_decode(self, codes):
        c = np.ascontiguousarray(codes, dtype=np.int64)
        n = max(self.codec_decode_len(c.shape[0]), 1)   # F = 0 is rejected by the library, not by numpy
        pcm = np.empty(n, np.float32)
        out_len = C.c_int64(0)
        self._ck(self.L.q3tts_codec_decode_host(self.h, _p(c), c.shape[0], _p(pcm), n, C.byref(out_len)))
        return pcm[: out_len.value]

    def codec_decode_batch(self, codes_list):
        """the vocoder phase of a job on its own: one [F_u][n_groups] code array per utterance -> one PCM array each (batched blocks of
        similar length + single utterances over the side lanes, exactly as synthesize_batch vocodes its results)"""
        n = len(codes_list)
        if n == 0:
            return []
        cs = [np.ascontiguousarray(c_, np.int64).reshape(-1, self.cfg.n_groups) for c_ in codes_list]
        offs = np.zeros(n + 1, np.int32)
        offs[1:] = np.cumsum([c_.shape[0] for c_ in cs])
        flat = np.ascontiguousarray(np.concatenate(cs)) if offs[-1] else np.zeros((1, self.cfg.n_groups), np.int64)
        cap = max(self.codec_decode_len(max(c_.shape[0] for c_ in cs)), 1)
        pcm = [np.zeros(cap, np.float32) for _ in range(n)]
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in pcm])
        pcm_len = np.zeros(n, np.int64)
        self._ck(self.L.q3tts_codec_decode_batch_host(self.h, n, _p(flat), _p(offs), C.cast(ptrs, C.c_void_p), cap, _p(pcm_len)))
        return [pcm[i][: pcm_len[i]] for i in range(n)]

    def codec_decode_chunked(self, codes, chunk_frames, left_context):
        """exact chunked decode: equals codec_decode(codes) when left_context covers the history"""
        c = np.ascontiguousarray(codes, dtype=np.int64)
        n = max(self.codec_decode_len(c.shape[0]), 1)
        pcm = np.empty(n, np.float32)
        out_len = C.c_int64(0)
        self._ck(self.L.q3tts_codec_decode_chunked_host(self.h, _p(c), c.shape[0], chunk_frames, left_context, _p(pcm), n, C.byref(out_len)))
        return pcm[: out_len.value]

    def codec_stream_begin(self, max_frames):
        """streaming vocoder with carried state: -> stream id (q3tts_codec_stream_begin)"""
        sid = C.c_int(-1)
        self.L.q3tts_codec_stream_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        self._ck(self.L.q3tts_codec_stream_begin(self.h, int(max_frames), C.byref(sid)))
        return sid.value

    def codec_stream_push(self, sid, codes):
        """the next frames' codes [n][n_groups] -> the samples they own"""
        c = np.ascontiguousarray(codes, dtype=np.int64)
        cap = self.codec_decode_len(c.shape[0]) + 4096
        pcm = np.empty(cap, np.float32)
        n = C.c_int64(0)
        self.L.q3tts_codec_stream_push_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        self._ck(self.L.q3tts_codec_stream_push_host(self.h, sid, _p(c), c.shape[0], _p(pcm), cap, C.byref(n)))
        return pcm[: n.value]

    def codec_stream_end(self, sid):
        self.L.q3tts_codec_stream_end.argtypes = [C.c_void_p, C.c_int]
        self._ck(self.L.q3tts_codec_stream_end(self.h, sid))

    def slot_codec_decode_range(self, slot, frame_begin, frame_end, left_context):
        """samples owned by frames [frame_begin, frame_end) of a slot (streaming while it generates)"""
        n = self.codec_decode_len(frame_end) - (self.codec_decode_len(frame_begin) if frame_begin > 0 else 0)
        pcm = np.empty(max(n, 1), np.float32)
        out_len = C.c_int64(0)
        self._ck(self.L.q3tts_slot_codec_decode_range_host(self.h, slot, frame_begin, frame_end, left_context, _p(pcm), n, C.byref(out_len)))
        return pcm[: out_len.value]

    def sample(self, logits, sp, u, suppress=False):
        a = np.ascontiguousarray(logits, dtype=np.float32)
        tok = C.c_int64(0)
        self._ck(self.L.q3tts_sample_host(self.h, _p(a), a.size, C.byref(sp), C.c_float(u), int(suppress), C.byref(tok)))
        return int(tok.value)

    def build_prompt(self, ids, lang=0, speaker=None, cap_rows=1024):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        prompt = np.zeros((16, self.cfg.hidden), np.float32)
        trailing = np.zeros((cap_rows, self.cfg.hidden), np.float32)
        S, nt = C.c_int(0), C.c_int(0)
        sp = np.ascontiguousarray(speaker, dtype=np.float32) if speaker is not None else None
        if sp is not None and sp.size != self.cfg.hidden:
            raise ValueError("speaker embedding has %d values, the model needs %d" % (sp.size, self.cfg.hidden))
        self._ck(self.L.q3tts_build_prompt_host(self.h, _p(ids), ids.size, lang, _p(sp) if sp is not None else None,
                                                _p(prompt), C.byref(S), _p(trailing), cap_rows, C.byref(nt)))
        return prompt[: S.value].copy(), trailing[: nt.value].copy()

    # ---- fused generation ----
    def slot_begin(self, slot, prompt, trailing, sp, seed=0, stream_id=0, ignore_eos=False):
        p = np.ascontiguousarray(prompt, dtype=np.float32)
        t = np.ascontiguousarray(trailing, dtype=np.float32)
        self._ck(self.L.q3tts_slot_begin(self.h, slot, _p(p), p.shape[0], _p(t), t.shape[0], C.byref(sp), seed,
                                         stream_id, int(ignore_eos)))

    def decode_steps(self, n):
        return self._ck(self.L.q3tts_decode_steps(self.h, n))

    def slot_status(self, slot):
        nf, fin = C.c_int(0), C.c_int(0)
        self._ck(self.L.q3tts_slot_status(self.h, slot, C.byref(nf), C.byref(fin)))
        return nf.value, bool(fin.value)

    def slot_codes(self, slot):
        nf, _ = self.slot_status(slot)
        codes = np.zeros((max(nf, 1), self.cfg.n_groups), np.int64)
        self._ck(self.L.q3tts_slot_codes_host(self.h, slot, _p(codes), nf))
        return codes[:nf]

    def slot_logits(self, slot):
        """(logits[vocab], last_hidden[hidden]) the fused path holds for the slot's next frame (run_decode's outputs)"""
        lg = np.empty(self.cfg.vocab, np.float32)
        lh = np.empty(self.cfg.hidden, np.float32)
        self._ck(self.L.q3tts_slot_logits_host(self.h, slot, _p(lg), _p(lh)))
        return lg, lh

    def step_logits(self, slot):
        """one eager decode step; returns [n_groups][max(vocab, sub_vocab)]: the logits row behind each of the frame's decisions of `slot`"""
        cols = max(self.cfg.vocab, self.cfg.sub_vocab)
        out = np.zeros((self.cfg.n_groups, cols), np.float32)
        self._ck(self.L.q3tts_step_logits_host(self.h, slot, _p(out), cols))
        return out

    def slot_codec_decode(self, slot):
        nf, _ = self.slot_status(slot)
        n = self.codec_decode_len(nf) if nf > 0 else 0
        pcm = np.empty(max(n, 1), np.float32)
        out_len = C.c_int64(0)
        self._ck(self.L.q3tts_slot_codec_decode_host(self.h, slot, _p(pcm), n, C.byref(out_len)))
        return pcm[: out_len.value]

    def slot_release(self, slot):
        self._ck(self.L.q3tts_slot_release(self.h, slot))

    def generate(self, prompt, trailing, sp, seed=0, stream_id=0, ignore_eos=False, slot=0, chunk=32):
        """generate_codes (reference src/tts_onnx.cpp:782-849) for one utterance on the fused path."""
        self.slot_begin(slot, prompt, trailing, sp, seed, stream_id, ignore_eos)
        left = sp.max_new_tokens
        while left > 0:
            n = min(chunk, left)
            active = self.decode_steps(n)
            left -= n
            if active == 0:
                break
        codes = self.slot_codes(slot)
        return codes

    # ---- voice-clone front end ----
    @property
    def has_speaker_encoder(self):
        return bool(self.L.q3tts_has_speaker_encoder(self.h))

    def speaker_encoder(self, mel):
        """run_speaker_encoder (tts_onnx.cpp:367-403): mel [128][frames] -> [spk_enc_dim]"""
        mel = np.ascontiguousarray(mel, np.float32)
        out = np.zeros(self.cfg.spk_enc_dim, np.float32)
        self._ck(self.L.q3tts_speaker_encoder_host(self.h, _p(mel), mel.shape[1], _p(out)))
        return out

    def extract_speaker_embedding(self, wav_path):
        out = np.zeros(self.cfg.spk_enc_dim, np.float32)
        self._ck(self.L.q3tts_extract_speaker_embedding_host(self.h, os.fsencode(wav_path), _p(out)))
        return out

    def synthesize_batch(self, token_lists, sp, lang=0, seed=0, ignore_eos=False, want_codes=True, speakers=None, max_new_per_utt=None):
        """synthesize_tokens (reference src/tts_onnx.cpp:405-436) for a batch of utterances; `speakers` (one
        [hidden] embedding or None per utterance) makes it synthesize_clone (:264-318).  More utterances than slots queue
        (continuous batching); max_new_per_utt caps each utterance separately."""
        n = len(token_lists)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(t, np.int64) for t in token_lists]))
        offs = np.zeros(n + 1, np.int32)
        offs[1:] = np.cumsum([len(t) for t in token_lists])
        cap = self.codec_decode_len(sp.max_new_tokens)
        pcm = [np.zeros(cap, np.float32) for _ in range(n)]
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in pcm])
        pcm_len = np.zeros(n, np.int64)
        nfr = np.zeros(n, np.int32)
        codes = np.zeros((n, sp.max_new_tokens, self.cfg.n_groups), np.int64) if want_codes else None
        spk_keep, spk_ptrs = [], None
        if speakers is not None:
            if len(speakers) != n:
                raise ValueError("speakers: one entry (embedding or None) per utterance")
            spk_keep = [None if s_ is None else np.ascontiguousarray(s_, np.float32) for s_ in speakers]
            for a in spk_keep:   # the library copies `hidden` floats from each row (the speaker row of the prompt is one talker-width embedding)
                if a is not None and a.size != self.cfg.hidden:
                    raise ValueError("speaker embedding has %d values, the model needs %d" % (a.size, self.cfg.hidden))
            spk_ptrs = C.cast((C.c_void_p * n)(*[None if a is None else a.ctypes.data for a in spk_keep]), C.c_void_p)
        caps = None if max_new_per_utt is None else np.ascontiguousarray(max_new_per_utt, np.int32)
        if caps is not None and caps.shape != (n,):
            raise ValueError("max_new_per_utt: one entry per utterance")
        self._ck(self.L.q3tts_synthesize_schedule_host(self.h, n, _p(flat), _p(offs), lang, spk_ptrs, C.byref(sp), None if caps is None else _p(caps),
                                                       seed, int(ignore_eos), C.cast(ptrs, C.c_void_p), cap, _p(pcm_len), _p(nfr),
                                                       _p(codes) if want_codes else None))
        outs = [pcm[i][: pcm_len[i]] for i in range(n)]
        cl = [codes[i, : nfr[i]] for i in range(n)] if want_codes else None
        return outs, cl, nfr

    # ---- measurement ----
    def last_decode_ms(self):
        ms, st = C.c_float(0), C.c_int(0)
        self._ck(self.L.q3tts_last_decode_ms(self.h, C.byref(ms), C.byref(st)))
        return ms.value, st.value

    def last_codec_ms(self):
        ms = C.c_float(0)
        self._ck(self.L.q3tts_last_codec_ms(self.h, C.byref(ms)))
        return ms.value

    def counters(self, reset=False):
        dms, cms = C.c_double(0), C.c_double(0)
        ds, cf = C.c_int64(0), C.c_int64(0)
        self._ck(self.L.q3tts_counters(self.h, C.byref(dms), C.byref(ds), C.byref(cms), C.byref(cf), int(reset)))
        return dict(decode_ms=dms.value, decode_steps=ds.value, codec_ms=cms.value, codec_frames=cf.value)

    def codec_decode_dev(self, codes_ptr, F, pcm_ptr, cap):
        """codes_ptr: device address of int32 [F][n_groups]; pcm_ptr: device address of float [cap] (e.g. torch tensors' data_ptr())."""
        n = C.c_int64(0)
        self.L.q3tts_codec_decode_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        self._ck(self.L.q3tts_codec_decode_dev(self.h, C.c_void_p(codes_ptr), int(F), C.c_void_p(pcm_ptr), int(cap), C.byref(n)))
        return n.value

    @property
    def stream(self):
        self.L.q3tts_stream.restype = C.c_void_p
        self.L.q3tts_stream.argtypes = [C.c_void_p]
        return self.L.q3tts_stream(self.h)

    # ---- batch-first device-pointer entry points: integer arguments are device addresses (0 = NULL) ----
    def talker_prefill_dev(self, embeds_ptr, batch, S, lens=None, logits_ptr=0, hidden_ptr=0, stream=0):
        ln = None if lens is None else np.ascontiguousarray(lens, np.int32)
        self.L.q3tts_talker_prefill_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._ck(self.L.q3tts_talker_prefill_dev(self.h, embeds_ptr, batch, S, _p(ln) if ln is not None else None, logits_ptr or None, hidden_ptr or None, stream or None))

    def talker_decode_dev(self, embeds_ptr, batch, active=None, logits_ptr=0, hidden_ptr=0, stream=0):
        m = None if active is None else np.ascontiguousarray(active, np.uint8)
        self.L.q3tts_talker_decode_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._ck(self.L.q3tts_talker_decode_dev(self.h, embeds_ptr, batch, _p(m) if m is not None else None, logits_ptr or None, hidden_ptr or None, stream or None))

    def code_predictor_dev(self, hidden_ptr, code0_ptr, batch, sp, seed, stream_id0, frame, sub_ptr, stream=0):
        self.L.q3tts_code_predictor_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        self._ck(self.L.q3tts_code_predictor_dev(self.h, hidden_ptr, code0_ptr, batch, C.byref(sp), seed, stream_id0, frame, sub_ptr, stream or None))

    def sample_dev(self, logits_ptr, batch, n, sp, u_ptr, suppress, ids_ptr, stream=0):
        self.L.q3tts_sample_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self._ck(self.L.q3tts_sample_dev(self.h, logits_ptr, batch, n, C.byref(sp), u_ptr, int(suppress), ids_ptr, stream or None))

    def poison_workspace(self):
        """test hook (FLAG_TEST_HOOKS engines): NaN bytes over the vocoder's reusable workspace (q3tts_test_poison_workspace)"""
        self.L.q3tts_test_poison_workspace.argtypes = [C.c_void_p]
        self._ck(self.L.q3tts_test_poison_workspace(self.h))

    def group_final_conv(self):
        """test hook: (sx [nb][T][C], pcm [nb][T]) of the last batched vocoder group's final conv, read back from its lane's workspace"""
        T, Cc, nb = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self.L.q3tts_test_group_final_conv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        self._ck(self.L.q3tts_test_group_final_conv(self.h, None, None, 0, C.byref(T), C.byref(Cc), C.byref(nb)))
        sx = np.empty((nb.value, T.value, Cc.value), np.float32)
        pcm = np.empty((nb.value, T.value), np.float32)
        self._ck(self.L.q3tts_test_group_final_conv(self.h, _p(sx), _p(pcm), sx.size, C.byref(T), C.byref(Cc), C.byref(nb)))
        return sx, pcm

    def final_conv_partials(self):
        """test hook: [tiles][256][8] partial sums of the last batched group's final conv (Q3TTS_COUT1_PACKED=2 only), or None"""
        f = self.L.q3tts_test_final_conv_partials
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        f.restype = C.c_int64
        n = f(self.h, None, 0)
        if n <= 0:
            return None
        out = np.empty(n, np.float32)
        f(self.h, _p(out), n)
        return out.reshape(-1, 256, 8)

    def measure_skip_frames(self, n):
        """measurement aid (FLAG_TEST_HOOKS engines): armed slots jump n frames ahead over a synthetic KV cache (q3tts_measure_skip_frames)"""
        self.L.q3tts_measure_skip_frames.argtypes = [C.c_void_p, C.c_int]
        self._ck(self.L.q3tts_measure_skip_frames(self.h, int(n)))

    def codec_plane_stats(self):
        """(two_product, three_product): codec weight tensors whose fp16 lo plane is empty / needed (q3tts_codec_plane_stats)"""
        a, b = C.c_int32(0), C.c_int32(0)
        self._ck(self.L.q3tts_codec_plane_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def prefill_profile(self, n_slots, n_rows=8, reps=4):
        """mean device ms of a batched prefill pass of n_slots free slots x n_rows synthetic prompt rows (q3tts_prefill_profile)"""
        out = C.c_double(0.0)
        self.L.q3tts_prefill_profile.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        self._ck(self.L.q3tts_prefill_profile(self.h, int(n_slots), int(n_rows), int(reps), C.byref(out)))
        return out.value

    def stage_profile(self, n_steps=32):
        """ms per step of {sampler, code predictor, talker decode, sum}: eager steps with events at the stage boundaries."""
        out = (C.c_double * 4)()
        self._ck(self.L.q3tts_stage_profile(self.h, int(n_steps), out))
        return dict(sampler_ms=out[0], code_predictor_ms=out[1], talker_decode_ms=out[2], step_ms=out[3])

    def decode_step_bytes(self):
        w, kv = C.c_double(0), C.c_double(0)
        self._ck(self.L.q3tts_decode_step_bytes(self.h, C.byref(w), C.byref(kv)))
        return w.value, kv.value


def read_wav(path):
    """io::read_wav (reference src/io/wav_reader.cpp:28-143): (mono float32 samples, sample_rate) or None.  Host-only."""
    n, sr = C.c_int64(0), C.c_int32(0)
    if lib().q3tts_read_wav_host(os.fsencode(path), None, 0, C.byref(n), C.byref(sr)) != 0:
        return None
    out = np.zeros(n.value, np.float32)
    lib().q3tts_read_wav_host(os.fsencode(path), _p(out), n.value, C.byref(n), C.byref(sr))
    return out, sr.value


def resample(audio, src_rate, dst_rate):
    """io::resample (wav_reader.cpp:145-164).  Host-only."""
    a = np.ascontiguousarray(audio, np.float32)
    n = lib().q3tts_resample_host(_p(a), a.size, src_rate, dst_rate, None, 0)
    if n < 0:
        raise ValueError("q3tts_resample_host failed")
    out = np.zeros(max(n, 1), np.float32)
    lib().q3tts_resample_host(_p(a), a.size, src_rate, dst_rate, _p(out), n)
    return out[:n]


def log_mel(audio):
    """MelExtractor::extract with the clone path's settings (tts_onnx.cpp:347-354): [128][frames].  Host-only."""
    a = np.ascontiguousarray(audio, np.float32)
    fr = C.c_int32(0)
    if lib().q3tts_mel_host(_p(a), a.size, None, 0, C.byref(fr)) != 0:
        return np.zeros((128, 0), np.float32)
    out = np.zeros((128, fr.value), np.float32)
    if lib().q3tts_mel_host(_p(a), a.size, _p(out), out.size, C.byref(fr)) != 0:
        raise RuntimeError("q3tts_mel_host failed")
    return out


class Tokenizer:
    """Byte-level BPE tokenizer of the text prompt (q3tts_tokenizer_*; reference src/io/tokenizer.h:13-22).
    Host-only: usable without a GPU."""

    def __init__(self, vocab_json=None, merges_txt=None):
        self._h = lib().q3tts_tokenizer_create()
        if not self._h:
            raise MemoryError("q3tts_tokenizer_create failed")
        if vocab_json is not None and not self.load_vocab(vocab_json):
            raise ValueError(f"cannot load vocab {vocab_json}")
        if merges_txt is not None and not self.load_merges(merges_txt):
            raise ValueError(f"cannot load merges {merges_txt}")

    def load_vocab(self, path):
        return lib().q3tts_tokenizer_load_vocab(self._h, os.fsencode(path)) == 0

    def load_merges(self, path):
        return lib().q3tts_tokenizer_load_merges(self._h, os.fsencode(path)) == 0

    @property
    def ready(self):
        return bool(lib().q3tts_tokenizer_ready(self._h))

    def encode(self, text):
        b = text if isinstance(text, (bytes, bytearray)) else text.encode("utf-8")
        b = bytes(b)
        n = lib().q3tts_tokenize(self._h, b, len(b), None, 0)
        if n < 0:
            raise RuntimeError("q3tts_tokenize failed")
        out = np.zeros(max(n, 1), np.int32)
        lib().q3tts_tokenize(self._h, b, len(b), out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out[:n]

    def close(self):
        if self._h:
            lib().q3tts_tokenizer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rng_uniform(seed, stream, frame, group):
    return float(lib().q3tts_rng_uniform(seed, stream, frame, group))
