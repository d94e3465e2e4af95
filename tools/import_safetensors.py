"""Import a Hugging Face style checkpoint (safetensors) into a Q3TW0001 weight file.

    python tools/import_safetensors.py --out model_dir/model.q3w main.safetensors [speech_tokenizer.safetensors ...] \\
        [--prefix talker=talker. --prefix predictor=talker.code_predictor. --prefix code2wav=decoder. --prefix speaker=speaker_encoder.]
        [--map extra_rules.json] [--config 0.6b | 1.7b | cfg.json] [--allow-missing]

The reference ships no converter: it consumes seven pre-exported .onnx graphs (src/tts_onnx.cpp:91-107).  This tool
maps parameter NAMES onto the engine's tensor registry (q3tts.tensor_specs = q3tts_config_tensor_info, host-only).  The
rule tables below are the state_dict naming of the `transformers` modules the architecture was pinned against
(Qwen3 decoder, Qwen3-Omni talker code predictor, Code2Wav, ECAPA_TimeDelayNet — tests/golden/hf_state_dict_keys.json
holds the generated key list they were derived from).  Where each component sits inside a real Qwen3-TTS checkpoint
(its key PREFIX) and the names of the few tensors outside those modules (text embedding / projection, codec
embedding) are [HINT]-level defaults: override them with --prefix / --map after listing the file (--list).

Dependency-free reader: the safetensors container is an 8-byte little-endian header length, a JSON header
{name: {dtype, shape, data_offsets}} and the raw little-endian payload; F32, F16 and BF16 are converted to float32."""
import argparse
import json
import os
import re
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_LAYER = [  # inside "<prefix>layers.{i}."
    (r"input_layernorm\.weight", "input_norm"), (r"post_attention_layernorm\.weight", "post_norm"),
    (r"self_attn\.([qkvo])_proj\.weight", r"\1_proj"), (r"self_attn\.([qk])_norm\.weight", r"\1_norm"),
    (r"mlp\.(gate|up|down)_proj\.weight", r"\1_proj"),
    (r"self_attn_layer_scale\.scale", "attn_scale"), (r"mlp_layer_scale\.scale", "mlp_scale"),
]


def _layers(src_prefix, dst_prefix):
    return [(src_prefix + r"layers\.(\d+)\." + s, dst_prefix + r"layers.\g<1>." + re.sub(r"\\1", r"\\g<2>", d)) for s, d in _LAYER]


RULES = {
    # Qwen3 decoder as the talker backbone + heads/embeddings next to it
    "talker": _layers(r"model\.", "talker.") + [
        (r"model\.norm\.weight", "talker.norm"),
        (r"(?:lm_head|codec_head)\.weight", "talker.codec_head"),
        (r"model\.(?:codec_embedding|embed_tokens)\.weight", "talker.codec_embed"),                       # [HINT]
        (r"model\.text_embedding\.weight", "text.embed"),                                                 # [HINT]
        (r"text_projection\.linear_fc([12])\.(weight|bias)", lambda m: f"text.fc{m.group(1)}.{m.group(2)[0]}"),  # [HINT]
    ],
    "predictor": _layers(r"model\.", "cp.") + [
        (r"model\.norm\.weight", "cp.norm"),
        (r"lm_head\.(\d+)\.weight", r"cp.head.\1"),
        (r"model\.codec_embedding\.(\d+)\.weight", r"cp.embed.\1"),
        (r"small_to_mtp_projection\.(weight|bias)", lambda m: f"cp.proj.{m.group(1)[0]}"),                  # [HINT] 1.7B only
    ],
    "code2wav": _layers(r"pre_transformer\.", "cd.") + [
        (r"pre_transformer\.norm\.weight", "cd.norm"),
        (r"code_embedding\.weight", "cd.code_embed"),
        (r"upsample\.(\d+)\.0\.conv\.(weight|bias)", lambda m: f"cd.up.{m.group(1)}.tconv.{m.group(2)[0]}"),
        (r"upsample\.(\d+)\.1\.dwconv\.conv\.(weight|bias)", lambda m: f"cd.up.{m.group(1)}.cnx.dw.{m.group(2)[0]}"),
        (r"upsample\.(\d+)\.1\.norm\.(weight|bias)", lambda m: f"cd.up.{m.group(1)}.cnx.ln.{m.group(2)[0]}"),
        (r"upsample\.(\d+)\.1\.pwconv([12])\.(weight|bias)", lambda m: f"cd.up.{m.group(1)}.cnx.pw{m.group(2)}.{m.group(3)[0]}"),
        (r"upsample\.(\d+)\.1\.gamma", r"cd.up.\1.cnx.gamma"),
        # decoder.0 = conv_in, decoder.1..n = blocks, decoder.n+1 = SnakeBeta, decoder.n+2 = conv_out (n from the config)
        (r"decoder\.(\d+)\.block\.0\.(alpha|beta)", lambda m: f"cd.dec.blocks.{int(m.group(1)) - 1}.snake.{m.group(2)}"),
        (r"decoder\.(\d+)\.block\.1\.conv\.(weight|bias)", lambda m: f"cd.dec.blocks.{int(m.group(1)) - 1}.tconv.{m.group(2)[0]}"),
        (r"decoder\.(\d+)\.block\.(\d+)\.act([12])\.(alpha|beta)",
         lambda m: f"cd.dec.blocks.{int(m.group(1)) - 1}.res.{int(m.group(2)) - 2}.act{m.group(3)}.{m.group(4)}"),
        (r"decoder\.(\d+)\.block\.(\d+)\.conv([12])\.conv\.(weight|bias)",
         lambda m: f"cd.dec.blocks.{int(m.group(1)) - 1}.res.{int(m.group(2)) - 2}.conv{m.group(3)}.{m.group(4)[0]}"),
        (r"decoder\.(\d+)\.conv\.(weight|bias)", lambda m: ("cd.dec.conv_in." if m.group(1) == "0" else "cd.dec.conv_out.") + m.group(2)[0]),
        (r"decoder\.(\d+)\.(alpha|beta)", r"cd.dec.snake_out.\2"),
    ],
    "speaker": [
        (r"blocks\.0\.conv\.(weight|bias)", lambda m: f"spk.tdnn0.{m.group(1)[0]}"),
        (r"blocks\.(\d+)\.tdnn([12])\.conv\.(weight|bias)", lambda m: f"spk.blocks.{int(m.group(1)) - 1}.tdnn{m.group(2)}.{m.group(3)[0]}"),
        (r"blocks\.(\d+)\.res2net_block\.blocks\.(\d+)\.conv\.(weight|bias)",
         lambda m: f"spk.blocks.{int(m.group(1)) - 1}.res2net.{m.group(2)}.{m.group(3)[0]}"),
        (r"blocks\.(\d+)\.se_block\.conv([12])\.(weight|bias)", lambda m: f"spk.blocks.{int(m.group(1)) - 1}.se{m.group(2)}.{m.group(3)[0]}"),
        (r"mfa\.conv\.(weight|bias)", lambda m: f"spk.mfa.{m.group(1)[0]}"),
        (r"asp\.tdnn\.conv\.(weight|bias)", lambda m: f"spk.asp.tdnn.{m.group(1)[0]}"),
        (r"asp\.conv\.(weight|bias)", lambda m: f"spk.asp.conv.{m.group(1)[0]}"),
        (r"fc\.(weight|bias)", lambda m: f"spk.fc.{m.group(1)[0]}"),
    ],
}
DEFAULT_PREFIX = {"talker": "talker.", "predictor": "talker.code_predictor.", "code2wav": "decoder.", "speaker": "speaker_encoder."}  # [HINT]
_DTYPES = {"F32": ("<f4", 4), "F16": ("<f2", 2), "BF16": (None, 2), "F64": ("<f8", 8)}


def read_safetensors(path):
    """{name: float32 ndarray}; integer tensors are skipped."""
    out = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(n))
        base = 8 + n
        for name, meta in header.items():
            if name == "__metadata__" or meta["dtype"] not in _DTYPES:
                continue
            lo, hi = meta["data_offsets"]
            f.seek(base + lo)
            raw = f.read(hi - lo)
            np_dt, _ = _DTYPES[meta["dtype"]]
            if np_dt is None:   # BF16: the high half of a float32
                a = (np.frombuffer(raw, "<u2").astype(np.uint32) << 16).view(np.float32)
            else:
                a = np.frombuffer(raw, np_dt).astype(np.float32)
            out[name] = a.reshape(meta["shape"])
    return out


def write_safetensors(path, tensors, bf16=()):
    """Minimal writer (tests, re-export): float32, or BF16 for the names in `bf16`."""
    header, blobs, off = {}, [], 0
    for name, a in tensors.items():
        a = np.ascontiguousarray(a, np.float32)
        if name in bf16:
            u = a.view(np.uint32).astype(np.uint64)
            raw = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype("<u2").tobytes()
            dt = "BF16"
        else:
            raw, dt = a.astype("<f4").tobytes(), "F32"
        header[name] = {"dtype": dt, "shape": list(a.shape), "data_offsets": [off, off + len(raw)]}
        off += len(raw)
        blobs.append(raw)
    hj = json.dumps(header).encode()
    hj += b" " * (-len(hj) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for b in blobs:
            f.write(b)


def map_names(names, prefixes=None, extra_rules=()):
    """{checkpoint name: registry name} for every name a rule covers.  Longer prefixes win (the predictor usually
    lives inside the talker's namespace)."""
    prefixes = dict(DEFAULT_PREFIX, **(prefixes or {}))
    order = sorted(prefixes.items(), key=lambda kv: -len(kv[1]))
    compiled = {c: [(re.compile(pat + r"\Z"), dst) for pat, dst in RULES[c]] for c in RULES}
    extra = [(re.compile(pat + r"\Z"), dst) for pat, dst in extra_rules]
    out = {}
    for name in names:
        hit = None
        for rx, dst in extra:
            m = rx.match(name)
            if m:
                hit = m.expand(dst) if isinstance(dst, str) else dst(m)
                break
        if hit is None:
            for comp, pre in order:
                if not name.startswith(pre):
                    continue
                rest = name[len(pre):]
                for rx, dst in compiled[comp]:
                    m = rx.match(rest)
                    if m:
                        hit = m.expand(dst) if isinstance(dst, str) else dst(m)
                        break
                break   # a name belongs to the longest matching prefix only
        if hit is not None:
            out[name] = hit
    return out


def explain_names(names, prefixes=None, extra_rules=()):
    """For --list: per checkpoint name, (registry name | None, note).  An unmapped name gets the rules that WOULD claim it under another
    component prefix (name = <implied prefix> + <rule match>): the usual reason for a missing component is a wrong --prefix.  A mapped
    name that another component's rules would also claim under some prefix is flagged ambiguous."""
    prefixes = dict(DEFAULT_PREFIX, **(prefixes or {}))
    mapping = map_names(names, prefixes, extra_rules)
    compiled = {c: [(re.compile(pat + r"\Z"), dst, pat) for pat, dst in RULES[c]] for c in RULES}
    out = {}
    targets = {}
    for name in names:
        claims = []          # (component, implied prefix, registry name, rule)
        parts = name.split(".")
        for cut in range(len(parts)):
            pre, rest = ".".join(parts[:cut]) + ("." if cut else ""), ".".join(parts[cut:])
            for comp, rules in compiled.items():
                for rx, dst, pat in rules:
                    m = rx.match(rest)
                    if m:
                        claims.append((comp, pre, m.expand(dst) if isinstance(dst, str) else dst(m), pat))
                        break
        dst = mapping.get(name)
        if dst is not None:
            targets.setdefault(dst, []).append(name)
            others = sorted({(c, pre) for c, pre, d, _ in claims if d != dst})
            note = "" if not others else "ambiguous: also claimed by " + ", ".join(f"{c} rules under prefix '{pre}'" for c, pre in others[:3])
        elif claims:
            note = "unmapped; would map with " + "; ".join(f"--prefix {c}={pre} -> {d} (rule {pat})" for c, pre, d, pat in claims[:3])
        else:
            owner = [c for c, pre in prefixes.items() if name.startswith(pre)]
            note = "unmapped; no rule of any component matches" + (f" (inside the {owner[0]} prefix)" if owner else "") + ": add a --map entry"
        out[name] = (dst, note)
    for dst, srcs in targets.items():
        if len(srcs) > 1:
            for nme in srcs:
                out[nme] = (dst, "DUPLICATE target: " + ", ".join(srcs))
    return out


def _binding():
    """The product's ctypes binding: configs and the tensor registry come from libq3tts_hip.so (host-only calls, no GPU needed)."""
    sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
    import q3tts
    return q3tts


def import_checkpoint(paths, cfg, prefixes=None, extra_rules=(), allow_missing=False):
    """-> {registry name: float32 array} with shapes checked against tensor_specs(cfg)."""
    q3tts = _binding()
    specs = {n: tuple(s) for n, s, _ in q3tts.tensor_specs(cfg)}
    src = {}
    for p in paths:
        src.update(read_safetensors(p))
    mapping = map_names(src.keys(), prefixes, extra_rules)
    out = {}
    for name, dst in mapping.items():
        if dst not in specs:
            raise ValueError(f"{name} -> {dst}: not a tensor of this config")
        a = src[name]
        if tuple(a.shape) != specs[dst]:
            if a.size == int(np.prod(specs[dst])) and a.squeeze().shape == tuple(x for x in specs[dst] if x != 1):
                a = a.reshape(specs[dst])
            else:
                raise ValueError(f"{name} -> {dst}: shape {tuple(a.shape)} != {specs[dst]}")
        if dst in out:
            raise ValueError(f"two checkpoint tensors map to {dst}")
        out[dst] = a
    missing = [n for n in specs if n not in out]
    if missing and not allow_missing:
        raise ValueError(f"{len(missing)} tensors not found in the checkpoint, e.g. {missing[:6]} (use --allow-missing, --prefix or --map)")
    return out, sorted(set(src) - set(mapping)), missing


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("files", nargs="+")
    ap.add_argument("--out")
    ap.add_argument("--config", default="0.6b", help='"0.6b", "1.7b" or a JSON file of q3tts_config fields')
    ap.add_argument("--prefix", action="append", default=[], help="component=prefix (talker, predictor, code2wav, speaker)")
    ap.add_argument("--map", help='JSON list of [regex, replacement] applied to the full checkpoint name first')
    ap.add_argument("--allow-missing", action="store_true", help="write the file although registry tensors have no source (for inspection: "
                    "q3tts_load_weights_file refuses a file that does not carry every tensor, naming the missing ones)")
    ap.add_argument("--tconv-trim", type=int, choices=[0, 1], default=None, help="cd_tconv_trim stored in the file's config: 0 = transposed convs "
                    "trimmed on both sides (transformers Code2Wav: 1920 F - 555 samples), 1 = right side only (1920 F samples); unverifiable "
                    "without the real graph's `lengths` output (reference src/tts_onnx.cpp:759-776)")
    ap.add_argument("--list", action="store_true", help="print the checkpoint's tensor names with their mapping and exit")
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    q3tts = _binding()
    cfg = q3tts.default_config(a.config.lower()) if a.config.lower() in ("0.6b", "1.7b") else q3tts.Config.from_dict(json.load(open(a.config)))
    prefixes = dict(p.split("=", 1) for p in a.prefix)
    extra = [tuple(r) for r in json.load(open(a.map))] if a.map else []
    if a.tconv_trim is not None:
        cfg.cd_tconv_trim = a.tconv_trim
    if a.list:
        names = {}
        for p in a.files:
            names.update({k: v.shape for k, v in read_safetensors(p).items()})
        ex = explain_names(names, prefixes, extra)
        specs = {n: tuple(sh) for n, sh, _ in q3tts.tensor_specs(cfg)}
        for k in sorted(names):
            dst, note = ex[k]
            shape_note = ""
            if dst is not None and dst in specs and int(np.prod(names[k])) != int(np.prod(specs[dst])):
                shape_note = f"  SHAPE MISMATCH: registry {specs[dst]}"
            if dst is not None and dst not in specs:
                shape_note = "  NOT A TENSOR OF THIS CONFIG"
            print(f"{k:80s} {str(tuple(names[k])):24s} -> {dst or '(unmapped)'}{shape_note}" + (f"   [{note}]" if note else ""))
        got = {d for d, _ in ex.values() if d}
        missing = [n for n in specs if n not in got]
        print(f"\n{len(names)} checkpoint tensors, {sum(d is not None for d, _ in ex.values())} mapped, "
              f"{sum(d is None for d, _ in ex.values())} unmapped; {len(missing)} of {len(specs)} registry tensors without a source")
        by_comp = {}
        for n in missing:
            by_comp.setdefault(n.split(".")[0], []).append(n)
        for c, ns in sorted(by_comp.items()):
            print(f"  missing {c}.*: {len(ns)} (e.g. {', '.join(ns[:4])})")
        return
    tensors, unused, missing = import_checkpoint(a.files, cfg, prefixes, extra, a.allow_missing)
    from tools.pack_weights import write_q3w
    if not a.out:
        ap.error("--out is required")
    write_q3w(a.out, cfg, tensors, validate=not a.allow_missing)
    print(f"wrote {a.out}: {len(tensors)} tensors; {len(unused)} checkpoint tensors unused; {len(missing)} registry tensors missing")


if __name__ == "__main__":
    main()
