"""Write a Q3TW0001 weight file (the format TTSEngine(model_dir) loads as <model_dir>/model.q3w).

    from tools.pack_weights import write_q3w
    write_q3w("model/model.q3w", cfg, {"talker.layers.0.q_proj": np.ndarray, ...})

Tensor names / shapes: q3tts.tensor_specs(cfg) (q3tts_config_tensor_info: the registry of csrc/q3_engine.cpp).  Matrices of
the talker / predictor / text stacks may be stored bf16 (dtype code 1), everything else fp32 (0).
Converting a real checkpoint = mapping its parameter names onto this registry.
"""
import ctypes
import struct

import numpy as np

MAGIC = b"Q3TW0001"


def _bf16_bits(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def check_complete(cfg, tensors, specs=None):
    """Raise ValueError unless `tensors` holds exactly the registry of `cfg` (every name once, every shape right): the loader rejects an
    incomplete file, so fail here, where the offending names can still be mapped back to checkpoint keys."""
    if specs is None:
        import os
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "leaxer-qwen3-tts_amd"))
        import q3tts
        specs = q3tts.tensor_specs(cfg)
    want = {name: tuple(int(x) for x in shape) for name, shape in ((s[0], s[1]) for s in specs)}
    missing = [n for n in want if n not in tensors]
    extra = [n for n in tensors if n not in want]
    bad = [(n, tuple(np.shape(tensors[n])), want[n]) for n in want if n in tensors and int(np.size(tensors[n])) != int(np.prod(want[n]))]
    if missing or extra or bad:
        raise ValueError("weight set does not match the registry: %d missing %s, %d unknown %s, %d wrong size %s"
                         % (len(missing), missing[:8], len(extra), extra[:8], len(bad), bad[:4]))


def write_q3w(path, cfg, tensors, bf16_prefixes=("talker.", "cp.", "text."), validate=True, specs=None):
    """cfg: a ctypes q3tts.Config; tensors: dict name -> array.  validate: refuse a tensor set that is not the registry of cfg."""
    if validate:
        check_complete(cfg, tensors, specs)
    with open(path, "wb") as f:
        f.write(MAGIC)
        raw = bytes(ctypes.string_at(ctypes.addressof(cfg), ctypes.sizeof(cfg)))
        f.write(struct.pack("<I", len(raw)))
        f.write(raw)
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            nb = name.encode()
            a = np.ascontiguousarray(arr, np.float32)
            as_bf16 = a.ndim >= 2 and name.startswith(bf16_prefixes)
            f.write(struct.pack("<H", len(nb)))
            f.write(nb)
            f.write(struct.pack("<BQ", 1 if as_bf16 else 0, a.size))
            f.write(_bf16_bits(a).tobytes() if as_bf16 else a.tobytes())
