"""Convert the initialisers (weight tensors) of ONNX graphs into a Q3TW0001 weight file — the artefacts a user of the reference actually
holds are the seven .onnx files it loads at /root/reference/src/tts_onnx.cpp:91-107 (README.md:71-89), not a safetensors checkpoint.

    python tools/import_onnx.py --out model_dir/model.q3w onnx_dir/*.onnx [--config 0.6b|1.7b|cfg.json] [--prefix talker=...] [--map rules.json]
                                [--by-shape-order] [--allow-missing] [--tconv-trim 0|1]
    python tools/import_onnx.py --list [--match] onnx_dir/*.onnx          (names, dtypes, shapes; --match: registry tensors of the same shape)

WHAT IS AND IS NOT VERIFIED.  Neither ONNX Runtime, the `onnx` package nor any .onnx file exists in the build image.  The reader is
written from the published protobuf schema (onnx.proto3: ModelProto.graph = 7; GraphProto.initializer = 5; TensorProto dims = 1,
data_type = 2, float_data = 4, int32_data = 5, int64_data = 7, name = 8, raw_data = 9, double_data = 10, external_data = 13,
data_location = 14) and is tested as a FORMAT ROUND TRIP: tools/make_onnx_fixture.py writes graphs in that wire format from seeded
tensors, this tool must give the tensors back bit for bit (tests/test_import_checkpoint.py) and the resulting model.q3w must synthesize
the same codes as an engine filled directly (tests/test_gpu_cli.py).  It says nothing about the initialiser NAMES of a real export.

NAME RESOLUTION, in this order:
  1. --map rules.json: [[regex, replacement], ...] on "<file stem>:<initialiser name>" (or the bare name); a replacement that starts
     with "T:" transposes the matrix.  The definitive way once a real file has been listed.
  2. the parameter-name rules of tools/import_safetensors.py (exporters keep `module.path.weight` for parameters a node uses as they
     are: norms, embeddings, conv weights), with per-component prefixes (--prefix talker=model. ...).
  3. --by-shape-order: what is still unresolved — exporters rename Linear weights to "onnx::MatMul_1234" and store them transposed —
     is assigned BY SHAPE (either orientation) IN ORDER OF APPEARANCE to the registry's tensors of that shape in registry order (layer by
     layer: q, k, v, o, gate, up, down).  A heuristic: every assignment it makes is printed, and it refuses a shape class whose
     counts do not match.  Square matrices (k_proj / v_proj of a 1024-wide model) have no orientation to read off: an anonymous
     "onnx::MatMul_*" tensor is taken as [in][out] (what MatMul consumes), a named one as [out][in].
The seven graphs repeat tensors (talker_prefill / talker_decode hold the same stack): a registry tensor may arrive more than once if
every copy is bit-identical.  Dependency-free (no protobuf runtime)."""
import argparse
import json
import os
import re
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DTYPES = {1: ("float32", 4), 2: ("uint8", 1), 3: ("int8", 1), 4: ("uint16", 2), 5: ("int16", 2), 6: ("int32", 4), 7: ("int64", 8), 9: ("bool", 1),
          10: ("float16", 2), 11: ("float64", 8), 12: ("uint32", 4), 13: ("uint64", 8), 16: ("bfloat16", 2)}
FLOAT_TYPES = (1, 10, 11, 16)


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
    """One TensorProto: metadata plus references (memoryviews, no copy) to whatever carries its data."""
    t = {"name": "", "dims": [], "data_type": 0, "raw": None, "float_data": [], "double_data": [], "int32_data": [], "external": {}, "raw_bytes": 0, "n_typed": 0}
    for fno, wt, v in fields(buf):
        if fno == 1:
            t["dims"] += _packed_varints(v) if wt == 2 else [v]
        elif fno == 2:
            t["data_type"] = v
        elif fno == 8:
            t["name"] = bytes(v).decode("utf-8", "replace")
        elif fno == 9:
            t["raw"] = v
            t["raw_bytes"] = len(v)
        elif fno == 4:                                   # float_data: packed (wire type 2) or one fixed32 per field
            t["float_data"].append(v)
            t["n_typed"] += len(v) // 4
        elif fno == 10:
            t["double_data"].append(v)
            t["n_typed"] += len(v) // 8
        elif fno == 5:                                   # int32_data: also the carrier of float16 / bfloat16 bit patterns
            vals = _packed_varints(v) if wt == 2 else [v]
            t["int32_data"] += vals
            t["n_typed"] += len(vals)
        elif fno in (7, 11) and wt == 2:
            t["n_typed"] += len(_packed_varints(v))
        elif fno == 13 and wt == 2:                      # external_data: repeated StringStringEntryProto {key = 1, value = 2}
            k = val = ""
            for eno, _ewt, ev in fields(v):
                if eno == 1:
                    k = bytes(ev).decode()
                elif eno == 2:
                    val = bytes(ev).decode()
            t["external"][k] = val
        elif fno == 14 and v == 1:
            t["external"].setdefault("location", "")
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


def _half_bits_to_f32(u16, bf16):
    if bf16:
        return (u16.astype(np.uint32) << 16).view(np.float32)
    return u16.view(np.float16).astype(np.float32)


def tensor_array(t, base_dir):
    """float32 ndarray of a floating-point initialiser (None for integer / bool tensors: shapes, indices — never weights)."""
    dt = t["data_type"]
    if dt not in FLOAT_TYPES:
        return None
    n = int(np.prod(t["dims"])) if t["dims"] else 1
    esz = DTYPES[dt][1]
    raw = None
    if t["external"]:
        loc = t["external"].get("location", "")
        if not loc:
            raise ValueError(f"{t['name']}: external data without a location")
        off, ln = int(t["external"].get("offset", "0")), int(t["external"].get("length", str(n * esz)))
        with open(os.path.join(base_dir, loc), "rb") as f:
            f.seek(off)
            raw = f.read(ln)
    elif t["raw"] is not None:
        raw = bytes(t["raw"])
    if raw is not None:
        if len(raw) != n * esz:
            raise ValueError(f"{t['name']}: {len(raw)} bytes of data for dims {list(t['dims'])} of {DTYPES[dt][0]}")
        if dt == 1:
            a = np.frombuffer(raw, "<f4").astype(np.float32)
        elif dt == 11:
            a = np.frombuffer(raw, "<f8").astype(np.float32)
        else:
            a = _half_bits_to_f32(np.frombuffer(raw, "<u2"), dt == 16)
    elif dt == 1 and t["float_data"]:
        a = np.frombuffer(b"".join(bytes(v) for v in t["float_data"]), "<f4").astype(np.float32)
    elif dt == 11 and t["double_data"]:
        a = np.frombuffer(b"".join(bytes(v) for v in t["double_data"]), "<f8").astype(np.float32)
    elif dt in (10, 16) and t["int32_data"]:
        a = _half_bits_to_f32(np.array(t["int32_data"], np.uint32).astype(np.uint16), dt == 16)
    else:
        raise ValueError(f"{t['name']}: no data field")
    if a.size != n:
        raise ValueError(f"{t['name']}: {a.size} values for dims {list(t['dims'])}")
    return a.reshape(t["dims"] if t["dims"] else ())


def _binding():
    sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
    import q3tts
    return q3tts


def registry_shapes(cfg):
    q3tts = _binding()
    by_shape = {}
    for name, shape, _kind in q3tts.tensor_specs(cfg):
        by_shape.setdefault(tuple(shape), []).append(name)
        if len(shape) == 2:                              # exporters store MatMul weights transposed
            by_shape.setdefault((shape[1], shape[0]), []).append(name + " (transposed)")
    return by_shape


def _fit(a, want, name, dst, transposed_hint):
    """`a` as the registry shape `want`: as is, squeezed / reshaped when only unit dims differ, or transposed (2-D, other orientation)."""
    if tuple(a.shape) == tuple(want):
        if transposed_hint and len(want) == 2 and want[0] == want[1]:
            return np.ascontiguousarray(a.T)
        return a
    if len(want) == 2 and a.ndim == 2 and tuple(a.shape) == (want[1], want[0]):
        return np.ascontiguousarray(a.T)
    if a.size == int(np.prod(want)) and tuple(x for x in a.shape if x != 1) == tuple(x for x in want if x != 1):
        return a.reshape(want)
    raise ValueError(f"{name} -> {dst}: shape {tuple(a.shape)} does not fit {tuple(want)}")


def import_onnx(paths, cfg, prefixes=None, extra_rules=(), by_shape_order=False, allow_missing=False, log=print, assume_square_transposed=False):
    """-> ({registry name: float32 array}, unused initialiser names, missing registry names)."""
    from tools.import_safetensors import map_names
    q3tts = _binding()
    specs = [(n, tuple(s)) for n, s, _ in q3tts.tensor_specs(cfg)]
    want = dict(specs)
    src = []                                              # (qualified name, bare name, array) in order of appearance
    for p in paths:
        stem = os.path.splitext(os.path.basename(p))[0]
        for t in initializers(p):
            a = tensor_array(t, os.path.dirname(os.path.abspath(p)))
            if a is not None:
                src.append((stem + ":" + t["name"], t["name"], a))
    out, used = {}, set()

    def take(i, dst, transposed_hint=False):
        qn, _bare, a = src[i]
        if dst not in want:
            raise ValueError(f"{qn} -> {dst}: not a tensor of this config")
        arr = _fit(a, want[dst], qn, dst, transposed_hint)
        if dst in out:
            if not np.array_equal(out[dst].view(np.uint32), np.ascontiguousarray(arr, np.float32).view(np.uint32)):
                raise ValueError(f"{qn} -> {dst}: a different tensor already maps there (the graphs disagree)")
        else:
            out[dst] = np.ascontiguousarray(arr, np.float32)
        used.add(i)

    extra = [(re.compile(pat + r"\Z"), dst) for pat, dst in extra_rules]
    for i, (qn, bare, _a) in enumerate(src):             # 1. explicit rules
        for rx, dst in extra:
            m = rx.match(qn) or rx.match(bare)
            if m:
                d = m.expand(dst)
                take(i, d[2:] if d.startswith("T:") else d, d.startswith("T:"))
                break
    named = map_names([bare for _q, bare, _a in src], prefixes)   # 2. parameter-name rules
    for i, (_qn, bare, _a) in enumerate(src):
        if i not in used and bare in named and named[bare] in want:
            take(i, named[bare])
    if by_shape_order:                                    # 3. shape classes in order of appearance
        # A class is an EXACT registry shape ([out][in] for a Linear weight).  An anonymous 2-D initialiser ("onnx::MatMul_<n>": the
        # constant operand of a MatMul node, [in][out]) belongs to the class of its TRANSPOSE, a named one to the class of its own
        # shape — so q_proj [2048][1024] and o_proj [1024][2048] never share a class (they did when classes were sorted dims), and
        # only tensors of one shape AND orientation (gate / up; the layers of a stack) are told apart by order of appearance alone.
        # A square anonymous matrix fits its class in either orientation: it is taken as [in][out] only under
        # --assume-square-transposed (said on the command line, printed per tensor), otherwise refused by name.
        def is_anon(bare):
            return not re.search(r"[A-Za-z_]\.[A-Za-z_]", bare) or bare.startswith("onnx::")

        def key(shape):
            return tuple(shape)
        left_reg = {}
        for n, s in specs:
            if n not in out:
                left_reg.setdefault(key(s), []).append(n)
        left_src = {}
        seen_bits, dup_of = {}, {}
        for i, (_qn, _bare, a) in enumerate(src):
            if i in used:
                continue
            k = key(a.shape if a.ndim else ())
            if a.ndim == 2 and is_anon(_bare):
                k = (k[1], k[0])
            if k not in left_reg and a.size > 1:          # only unit dims differ ([1][C] biases, [C][1][k] depthwise kernels): the class of that squeezed shape
                sq = tuple(x for x in k if x != 1)
                alt = [r for r in left_reg if tuple(x for x in r if x != 1) == sq]
                if len(alt) == 1:
                    k = alt[0]
            if k not in left_reg:
                continue
            h = (k, a.tobytes())                          # the same tensor repeated by another graph counts once
            if h in seen_bits:
                dup_of[i] = seen_bits[h]
                continue
            seen_bits[h] = i
            left_src.setdefault(k, []).append(i)
        for k, regs in left_reg.items():
            idx = left_src.get(k, [])
            if not idx:
                continue
            if len(idx) != len(regs):
                raise ValueError(f"--by-shape-order: {len(idx)} unresolved initialisers of shape class {k} for {len(regs)} registry tensors "
                                 f"({regs[:3]} ...): counts must match; resolve them with --map")
            for i, dst in zip(idx, regs):
                anon = is_anon(src[i][1])
                if anon and len(k) == 2 and k[0] == k[1] and not assume_square_transposed:
                    raise ValueError(f"--by-shape-order: {src[i][0]} is an anonymous SQUARE matrix {k} (for {dst}): its orientation cannot be read "
                                     f"from its shape; map it explicitly (--map, 'T:' in front of the target transposes) or pass "
                                     f"--assume-square-transposed to take every such matrix as [in][out]")
                log(f"  [shape-order] {src[i][0]} {tuple(src[i][2].shape)} -> {dst}{' (as [in][out])' if anon else ''}")
                take(i, dst, transposed_hint=anon)
        used.update(i for i, first in dup_of.items() if first in used)
    missing = [n for n, _s in specs if n not in out]
    if missing and not allow_missing:
        raise ValueError(f"{len(missing)} registry tensors without a source, e.g. {missing[:6]} (use --list --match, then --map / --prefix / --by-shape-order)")
    return out, [src[i][0] for i in range(len(src)) if i not in used], missing


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("files", nargs="+")
    ap.add_argument("--out")
    ap.add_argument("--config", default="0.6b", help='"0.6b", "1.7b" or a JSON file of q3tts_config fields')
    ap.add_argument("--prefix", action="append", default=[], help="component=prefix for the parameter-name rules (talker, predictor, code2wav, speaker)")
    ap.add_argument("--map", help='JSON list of [regex, replacement] on "<file stem>:<initialiser name>"; "T:" in front of a replacement transposes')
    ap.add_argument("--by-shape-order", action="store_true", help="assign what is still unresolved by shape, in order of appearance (printed; heuristic)")
    ap.add_argument("--assume-square-transposed", action="store_true", help="with --by-shape-order: anonymous SQUARE matrices are [in][out] like the "
                    "other anonymous MatMul operands (refused otherwise: their orientation cannot be read from the shape)")
    ap.add_argument("--allow-missing", action="store_true")
    ap.add_argument("--tconv-trim", type=int, choices=[0, 1], default=None)
    ap.add_argument("--list", action="store_true", help="print the initialisers and exit")
    ap.add_argument("--match", action="store_true", help="with --list: registry tensors of the same shape (shape only)")
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    q3tts = _binding()
    cfg = q3tts.default_config(a.config.lower()) if a.config.lower() in ("0.6b", "1.7b") else q3tts.Config.from_dict(json.load(open(a.config)))
    if a.tconv_trim is not None:
        cfg.cd_tconv_trim = a.tconv_trim
    if a.list:
        by_shape = registry_shapes(cfg) if a.match else {}
        for path in a.files:
            ts = initializers(path)
            print(f"{path}: {len(ts)} initialisers")
            for t in ts:
                dt, esz = DTYPES.get(t["data_type"], (f"type{t['data_type']}", 0))
                n = int(np.prod(t["dims"])) if t["dims"] else 1
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
        return
    if not a.out:
        ap.error("--out is required")
    prefixes = dict(p.split("=", 1) for p in a.prefix)
    extra = [tuple(r) for r in json.load(open(a.map))] if a.map else []
    tensors, unused, missing = import_onnx(a.files, cfg, prefixes, extra, a.by_shape_order, a.allow_missing, assume_square_transposed=a.assume_square_transposed)
    from tools.pack_weights import write_q3w
    write_q3w(a.out, cfg, tensors, validate=not a.allow_missing)
    print(f"wrote {a.out}: {len(tensors)} tensors; {len(unused)} initialisers unused; {len(missing)} registry tensors missing")


if __name__ == "__main__":
    main()
