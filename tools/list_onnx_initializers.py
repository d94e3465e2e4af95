"""List the initialisers (weight tensors) of ONNX graphs — the artefacts a user of the reference actually holds: the seven .onnx files
it loads at /root/reference/src/tts_onnx.cpp:91-107 (README.md:71-89) — and suggest, by SHAPE ONLY, which registry tensors of this engine
they could be.

    python tools/list_onnx_initializers.py talker_decode.onnx [more.onnx ...] [--model 0.6b] [--match]

UNVERIFIED against a real export: neither ONNX Runtime, the `onnx` package nor any .onnx file exists in the build image, so the
reader below is written from the published protobuf schema (onnx.proto3: ModelProto.graph = 7, GraphProto.initializer = 5, TensorProto
dims = 1, data_type = 2, float_data = 4, int64_data = 7, name = 8, raw_data = 9, external_data = 13, data_location = 14) and tested
only on files this repo's own test writes with the same wire format (tests/test_import_checkpoint.py).  Exporters mangle initialiser
names ("onnx::MatMul_1234"), so `--match` can only say "this [3072, 1024] fp32 tensor has the shape of talker.layers.N.gate_proj /
up_proj": an aid for writing the mapping by hand, not a converter.  Dependency-free (no protobuf runtime)."""
import argparse
import os
import struct
import sys

DTYPES = {1: ("float32", 4), 2: ("uint8", 1), 3: ("int8", 1), 4: ("uint16", 2), 5: ("int16", 2), 6: ("int32", 4), 7: ("int64", 8), 9: ("bool", 1),
          10: ("float16", 2), 11: ("float64", 8), 12: ("uint32", 4), 13: ("uint64", 8), 16: ("bfloat16", 2)}


def _varint(buf, pos):
    val = shift = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 70:
            raise ValueError("varint too long")


def fields(buf):
    """(field number, wire type, value) of one protobuf message; length-delimited values come back as memoryview slices (no copy)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise ValueError("length-delimited field runs past the message")
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fno, wt, val


def _packed_varints(v):
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


def parse_tensor(buf):
    t = {"name": "", "dims": [], "data_type": 0, "raw_bytes": 0, "n_typed": 0, "external": False}
    for fno, wt, v in fields(buf):
        if fno == 1:
            t["dims"] += _packed_varints(v) if wt == 2 else [v]
        elif fno == 2:
            t["data_type"] = v
        elif fno == 8:
            t["name"] = bytes(v).decode("utf-8", "replace")
        elif fno == 9:
            t["raw_bytes"] = len(v)
        elif fno in (4, 5, 7, 10, 11) and wt == 2:      # packed float / int32 / int64 / double / uint64 data
            t["n_typed"] += len(v) // {4: 4, 10: 8}.get(fno, 1) if fno in (4, 10) else len(_packed_varints(v))
        elif fno == 13:
            t["external"] = True
        elif fno == 14 and v == 1:
            t["external"] = True
    return t


def initializers(path):
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    out = []
    for fno, wt, v in fields(buf):                       # ModelProto
        if fno == 7 and wt == 2:                         # .graph
            for gno, gwt, gv in fields(v):               # GraphProto
                if gno == 5 and gwt == 2:                # .initializer
                    out.append(parse_tensor(gv))
    return out


def registry_shapes(model):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "leaxer-qwen3-tts_amd"))
    import q3tts
    cfg = q3tts.default_config(model)
    by_shape = {}
    for name, shape, _kind in q3tts.tensor_specs(cfg):
        by_shape.setdefault(tuple(shape), []).append(name)
        if len(shape) == 2:                              # exporters often store MatMul weights transposed
            by_shape.setdefault((shape[1], shape[0]), []).append(name + " (transposed)")
    return by_shape


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("files", nargs="+")
    ap.add_argument("--model", default="0.6b", choices=["0.6b", "1.7b"])
    ap.add_argument("--match", action="store_true", help="suggest registry tensors of the same shape (shape only: unverified)")
    a = ap.parse_args()
    by_shape = registry_shapes(a.model) if a.match else {}
    for path in a.files:
        ts = initializers(path)
        print(f"{path}: {len(ts)} initialisers")
        for t in ts:
            dt, esz = DTYPES.get(t["data_type"], (f"type{t['data_type']}", 0))
            n = 1
            for d in t["dims"]:
                n *= d
            where = "external file" if t["external"] else (f"{t['raw_bytes']} raw bytes" if t["raw_bytes"] else f"{t['n_typed']} typed values")
            line = f"  {t['name']:60s} {dt:9s} {str(list(t['dims'])):24s} {where}"
            if t["raw_bytes"] and esz and t["raw_bytes"] != n * esz:
                line += f"  [size mismatch: dims say {n * esz} bytes]"
            if a.match:
                cands = by_shape.get(tuple(t["dims"]), [])
                if cands:
                    uniq = sorted(set(c.split(".layers.")[0] + (".layers.N." + c.split(".layers.")[1].split(".", 1)[1] if ".layers." in c else "") for c in cands))
                    line += "  ~ shape of: " + ", ".join(uniq[:4]) + (" ..." if len(uniq) > 4 else "")
            print(line)


if __name__ == "__main__":
    main()
