"""Write .onnx-FORMAT files (ModelProto -> GraphProto -> initialisers only, no nodes) from a set of registry tensors: the self-made half of
the format round trip that tools/import_onnx.py is tested with (tests/test_import_checkpoint.py, tests/test_gpu_cli.py).

    python tools/make_onnx_fixture.py --out some_dir [--seed 0]          (tiny config, seeded bf16-representable tensors)

NOT reference fixtures, and no claim about a real export's initialiser names: the files only exercise the wire format (onnx.proto3
field numbers, see tools/import_onnx.py) the way exporters use it —
  * one file per session of the reference (/root/reference/src/tts_onnx.cpp:91-107); talker_prefill.onnx and talker_decode.onnx both
    carry the whole talker stack (the converter must accept identical duplicates);
  * Linear weights of the talker / predictor / text stacks anonymous and transposed ("onnx::MatMul_<n>", [in][out]: what a MatMul node
    consumes), everything else under its parameter name (the `transformers` state_dict keys of tests/golden/hf_state_dict_keys.json
    behind a component prefix);
  * every data carrier: raw_data (float32, bfloat16), packed float_data, bfloat16 bit patterns in int32_data, one tensor in an
    external data file; plus int64 shape constants that a converter has to ignore.
"""
import argparse
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PREFIX = {"talker": "talker.", "predictor": "talker.code_predictor.", "code2wav": "decoder.", "speaker": "speaker_encoder."}   # import_safetensors' defaults
EXTRA_KEYS = {"talker.codec_embed": "talker.model.codec_embedding.weight", "text.embed": "talker.model.text_embedding.weight",
              "text.fc1.w": "talker.text_projection.linear_fc1.weight", "text.fc1.b": "talker.text_projection.linear_fc1.bias",
              "text.fc2.w": "talker.text_projection.linear_fc2.weight", "text.fc2.b": "talker.text_projection.linear_fc2.bias"}


def pb_varint(x):
    x &= (1 << 64) - 1            # negative int64 values travel as their 64-bit two's complement (ten bytes)
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def pb_field(fno, wt, payload):
    key = pb_varint(fno << 3 | wt)
    if wt == 2:
        return key + pb_varint(len(payload)) + payload
    return key + payload


def _bf16_bits(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def tensor_proto(name, arr, carrier, ext=None):
    """carrier: raw32 | rawbf16 | floatdata | i32bf16 | external (ext = (file name, open file)) | int64 (shape constants)."""
    a = np.ascontiguousarray(arr)
    dims = b"".join(pb_field(1, 0, pb_varint(int(d))) for d in a.shape) if carrier == "floatdata" else pb_field(1, 2, b"".join(pb_varint(int(d)) for d in a.shape))
    body = dims
    if carrier == "int64":
        body += pb_field(2, 0, pb_varint(7)) + pb_field(8, 2, name.encode()) + pb_field(7, 2, b"".join(pb_varint(int(v)) for v in a.ravel()))
        return body
    a = a.astype(np.float32)
    if carrier == "raw32":
        body += pb_field(2, 0, pb_varint(1)) + pb_field(8, 2, name.encode()) + pb_field(9, 2, a.astype("<f4").tobytes())
    elif carrier == "rawbf16":
        body += pb_field(2, 0, pb_varint(16)) + pb_field(8, 2, name.encode()) + pb_field(9, 2, _bf16_bits(a).astype("<u2").tobytes())
    elif carrier == "floatdata":
        body += pb_field(2, 0, pb_varint(1)) + pb_field(4, 2, a.astype("<f4").tobytes()) + pb_field(8, 2, name.encode())
    elif carrier == "i32bf16":
        body += pb_field(2, 0, pb_varint(16)) + pb_field(5, 2, b"".join(pb_varint(int(v)) for v in _bf16_bits(a).ravel())) + pb_field(8, 2, name.encode())
    elif carrier == "external":
        fname, fh = ext
        off = fh.tell()
        raw = a.astype("<f4").tobytes()
        fh.write(raw)
        entries = b"".join(pb_field(13, 2, pb_field(1, 2, k.encode()) + pb_field(2, 2, str(v).encode())) for k, v in (("location", fname), ("offset", off), ("length", len(raw))))
        body += pb_field(2, 0, pb_varint(1)) + pb_field(8, 2, name.encode()) + entries + pb_field(14, 0, pb_varint(1))
    else:
        raise ValueError(carrier)
    return body


def model_proto(tensor_bodies):
    graph = pb_field(2, 2, b"q3tts_fixture") + b"".join(pb_field(5, 2, t) for t in tensor_bodies)
    return pb_field(1, 0, pb_varint(8)) + pb_field(2, 2, b"make_onnx_fixture") + pb_field(7, 2, graph)


def is_linear(name):
    """registry names of the matrices a MatMul node would consume (talker / predictor / text Linear layers)"""
    leaf = name.split(".")[-1]
    if name.startswith(("talker.", "cp.")) and (leaf.endswith("_proj") or name == "talker.codec_head"):
        return True
    return name.startswith("cp.head.") or name in ("text.fc1.w", "text.fc2.w", "cp.proj.w")


def session_of(name):
    if name.startswith("text."):
        return "text_project"
    if name == "talker.codec_embed":
        return "codec_embed"
    if name.startswith("cp.embed."):
        return "code_predictor_embed"
    if name.startswith("talker."):
        return "talker"
    if name.startswith("cp."):
        return "code_predictor"
    if name.startswith("cd."):
        return "tokenizer12hz_decode"
    return "speaker_encoder"


FILE_ORDER = ["talker_prefill", "talker_decode", "codec_embed", "text_project", "code_predictor", "code_predictor_embed", "tokenizer12hz_decode", "speaker_encoder"]


def write_fixture(out_dir, specs, weights, keys):
    """specs: [(registry name, shape, kind)] in registry order; weights: {registry name: array}; keys: {component: {registry name: state_dict key}}.
    Returns the .onnx paths in the order tools/import_onnx.py --by-shape-order needs (sessions in registry order)."""
    os.makedirs(out_dir, exist_ok=True)
    hf = dict(EXTRA_KEYS)
    for comp, m in keys.items():
        for reg, key in m.items():
            hf[reg] = PREFIX[comp] + key
    bodies = {s: [] for s in FILE_ORDER}
    ext_name = "tokenizer12hz_decode.onnx.data"
    ext = open(os.path.join(out_dir, ext_name), "wb")
    ext.write(b"\0" * 24)                                   # a non-zero offset for the first external tensor
    anon = 1000
    carriers = ["raw32", "rawbf16", "raw32", "floatdata", "raw32", "i32bf16"]
    ext_done = False
    for i, (name, shape, _kind) in enumerate(specs):
        a = np.asarray(weights[name], np.float32).reshape(shape)
        sess = session_of(name)
        if is_linear(name):
            anon += 7
            body = tensor_proto(f"onnx::MatMul_{anon}", np.ascontiguousarray(a.T), "rawbf16" if i % 2 else "raw32")
        else:
            carrier = carriers[i % len(carriers)]
            if a.size > 4096 and carrier in ("floatdata", "i32bf16"):
                carrier = "raw32"                           # exporters keep typed fields for small tensors
            if sess == "tokenizer12hz_decode" and not ext_done and a.ndim == 3:
                carrier, ext_done = "external", True
            if name not in hf:
                raise ValueError(f"no state_dict key for {name}")
            body = tensor_proto(hf[name], a, carrier, (ext_name, ext))
        for f in (["talker_prefill", "talker_decode"] if sess == "talker" else [sess]):
            bodies[f].append(body)
    ext.close()
    paths = []
    for f in FILE_ORDER:
        consts = [tensor_proto(f"{f}/Constant_{k}_output_0", np.array(v, np.int64), "int64") for k, v in enumerate(([1, -1, 64], [0], [2, 8]))]
        path = os.path.join(out_dir, f + ".onnx")
        with open(path, "wb") as fh:
            fh.write(model_proto(consts[:1] + bodies[f] + consts[1:]))
        paths.append(path)
    return paths


def seeded_weights(specs, seed):
    """bf16-representable seeded tensors by registry kind (norm weights around 1, everything else N(0, 0.05))"""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, kind in specs:
        a = rng.standard_normal(shape).astype(np.float32) * np.float32(0.05)
        if kind in (1, "norm"):
            a = a + np.float32(1.0)
        out[name] = (_bf16_bits(a).astype(np.uint32) << 16).view(np.float32).reshape(shape)
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", required=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--config", default=os.path.join(ROOT, "tests", "golden", "tiny_config.json"), help="JSON of q3tts_config fields (default: the tests' tiny config)")
    ap.add_argument("--keys", default=os.path.join(ROOT, "tests", "golden", "hf_state_dict_keys.json"))
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
    import q3tts
    cfg = q3tts.Config.from_dict(json.load(open(a.config)))
    specs = q3tts.tensor_specs(cfg)
    paths = write_fixture(a.out, specs, seeded_weights(specs, a.seed), json.load(open(a.keys)))
    print("\n".join(paths))


if __name__ == "__main__":
    main()
