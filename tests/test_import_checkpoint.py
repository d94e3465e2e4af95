"""SURVEY.md 8f-4 — tools/import_safetensors.py: the name rules reproduce, from state_dict-style key names, exactly
the tensors the goldens were generated with.  tests/golden/hf_state_dict_keys.json lists the key every parameter of
the `transformers` modules has (written by make_hf_goldens.py); the test writes those keys into safetensors files
(BF16 for the matrices), imports them and compares with the goldens' weights bit for bit."""
import json
import os
import sys

import numpy as np

import q3_oracle as qo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def gold_weights(name):
    z = np.load(os.path.join(GOLD, name))
    return {k[2:]: z[k] for k in z.files if k.startswith("w:")}


def test_rules_cover_every_transformers_key(tmp_path):
    from tools.import_safetensors import import_checkpoint, map_names, read_safetensors, write_safetensors
    keys = json.load(open(os.path.join(GOLD, "hf_state_dict_keys.json")))
    cfg = qo.config_tiny()
    prefixes = {"talker": "talker.", "predictor": "talker.code_predictor.", "code2wav": "speech.decoder.", "speaker": "spk_enc."}
    files, want = [], {}
    for comp, gold in (("talker", "hf_talker.npz"), ("predictor", "hf_predictor.npz"), ("code2wav", "hf_code2wav.npz"), ("speaker", "hf_speaker.npz")):
        w = gold_weights(gold)
        assert set(w) == set(keys[comp]), comp
        tensors = {prefixes[comp] + keys[comp][n]: a for n, a in w.items()}
        path = str(tmp_path / f"{comp}.safetensors")
        write_safetensors(path, tensors, bf16={k for k, a in tensors.items() if a.ndim >= 2})   # goldens are bf16-representable
        back = read_safetensors(path)
        assert all(np.array_equal(back[k], tensors[k]) for k in tensors)
        files.append(path)
        want.update(w)
    # every key maps, and to the right registry name
    flat = {prefixes[c] + k: n for c in keys for n, k in keys[c].items()}
    assert map_names(flat.keys(), prefixes) == flat
    got, unused, missing = import_checkpoint(files, cfg, prefixes, allow_missing=True)
    assert not unused
    assert set(got) == set(want)
    for n in want:
        assert got[n].shape == want[n].shape and np.array_equal(got[n], want[n]), n
    # what the transformers modules do not hold (the text side and the talker's codec embedding) is reported, not invented
    assert set(missing) == {"talker.codec_embed", "text.embed", "text.fc1.w", "text.fc1.b", "text.fc2.w", "text.fc2.b"}
    # [HINT] names for those are accepted when present
    extra = {"talker.model.codec_embedding.weight": np.zeros((cfg.vocab, cfg.hidden), np.float32),
             "talker.model.text_embedding.weight": np.zeros((cfg.text_vocab, cfg.text_hidden), np.float32),
             "talker.text_projection.linear_fc1.weight": np.zeros((cfg.text_hidden, cfg.text_hidden), np.float32),
             "talker.text_projection.linear_fc1.bias": np.zeros(cfg.text_hidden, np.float32),
             "talker.text_projection.linear_fc2.weight": np.zeros((cfg.hidden, cfg.text_hidden), np.float32),
             "talker.text_projection.linear_fc2.bias": np.zeros(cfg.hidden, np.float32)}
    p2 = str(tmp_path / "extra.safetensors")
    write_safetensors(p2, extra)
    got2, _, missing2 = import_checkpoint(files + [p2], cfg, prefixes)
    assert not missing2 and len(got2) == len(qo.tensor_specs(cfg))


def test_imported_weights_drive_the_oracle(tmp_path):
    """End to end on the CPU side: safetensors -> import -> Q3TW file bytes -> oracle reproduces the transformers golden."""
    from tools.import_safetensors import import_checkpoint, write_safetensors
    keys = json.load(open(os.path.join(GOLD, "hf_state_dict_keys.json")))
    w = gold_weights("hf_code2wav.npz")
    path = str(tmp_path / "c2w.safetensors")
    write_safetensors(path, {"decoder." + keys["code2wav"][n]: a for n, a in w.items()})
    cfg = qo.config_tiny()
    got, _, _ = import_checkpoint([path], cfg, allow_missing=True)
    z = np.load(os.path.join(GOLD, "hf_code2wav.npz"))
    o = qo.Oracle(cfg, max_ctx=16)
    o.load(got)
    pcm = o.vocoder(z["codes_3"])
    o.close()
    assert float(np.sqrt(np.mean((pcm - z["pcm_3"]) ** 2))) < 2e-5


def test_shape_mismatch_and_missing_are_errors(tmp_path):
    import pytest
    from tools.import_safetensors import import_checkpoint, write_safetensors
    cfg = qo.config_tiny()
    p = str(tmp_path / "bad.safetensors")
    write_safetensors(p, {"talker.model.norm.weight": np.zeros(cfg.hidden + 1, np.float32)})
    with pytest.raises(ValueError, match="shape"):
        import_checkpoint([p], cfg, allow_missing=True)
    write_safetensors(p, {"talker.model.norm.weight": np.zeros(cfg.hidden, np.float32)})
    with pytest.raises(ValueError, match="not found"):
        import_checkpoint([p], cfg)


def test_registry_from_the_library_equals_the_oracles():
    """q3tts_config_tensor_info (host-only) lists the tensors the engine allocates; same names, shapes, kinds and order as the oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
    import q3tts
    for mk in (qo.config_tiny, qo.config_tiny_proj, qo.config_medium, qo.config_medium_proj, qo.config_06b, qo.config_17b):
        oc = mk()
        got = q3tts.tensor_specs(q3tts.Config.from_dict(oc.to_dict()))
        want = [(n, tuple(s), k) for n, s, k in qo.tensor_specs(oc)]
        assert got == want, mk.__name__
    bad = q3tts.Config.from_dict(qo.config_tiny().to_dict())
    bad.cd_n_blocks = 99
    import pytest
    with pytest.raises(ValueError, match="out of range"):
        q3tts.tensor_specs(bad)


def test_importer_does_not_touch_the_oracle(tmp_path):
    """The converter is product code: it must work from the library alone (oracle/ is test infrastructure)."""
    import subprocess
    code = ("import sys, json; sys.path.insert(0, %r); import tools.import_safetensors as t; q = t._binding(); "
            "c = q.default_config('1.7b'); n = len(q.tensor_specs(c)); "
            "assert not any('oracle' in m for m in sys.modules), [m for m in sys.modules if 'oracle' in m]; print(n)") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert int(r.stdout.strip()) == 710


def test_list_explains_unmapped_ambiguous_and_duplicate_keys(tmp_path):
    """--list must make a wrong prefix visible: a component stored under another prefix is unmapped WITH the --prefix that would claim
    it, a key no rule knows says so, two keys landing on one registry tensor are flagged, and the registry tensors left without a
    source are counted per component.  None of this is verified against a real Qwen3-TTS checkpoint (none in the image)."""
    import subprocess
    from tools.import_safetensors import explain_names, write_safetensors
    keys = json.load(open(os.path.join(GOLD, "hf_state_dict_keys.json")))
    names = ["talker." + k for k in keys["talker"].values()]
    names += ["mtp." + k for k in keys["predictor"].values()]                      # predictor under an unexpected prefix
    names += ["talker.model.rotary_emb.inv_freq", "talker.lm_head.weight", "talker.codec_head.weight"]
    ex = explain_names(names)
    assert ex["talker.model.norm.weight"] == ("talker.norm", "") or ex["talker.model.norm.weight"][0] == "talker.norm"
    d, note = ex["mtp.model.layers.0.mlp.up_proj.weight"]
    assert d is None and "--prefix predictor=mtp." in note and "cp.layers.0.up_proj" in note
    d, note = ex["talker.model.rotary_emb.inv_freq"]
    assert d is None and "no rule" in note and "talker prefix" in note
    assert ex["talker.lm_head.weight"][0] == ex["talker.codec_head.weight"][0] == "talker.codec_head"
    assert "DUPLICATE" in ex["talker.lm_head.weight"][1] and "DUPLICATE" in ex["talker.codec_head.weight"][1]
    # with the right prefix everything maps
    ex2 = explain_names(names, {"predictor": "mtp."})
    assert ex2["mtp.model.layers.0.mlp.up_proj.weight"][0] == "cp.layers.0.up_proj"
    # the CLI prints the same and the per-component summary of what is still missing
    path = str(tmp_path / "x.safetensors")
    write_safetensors(path, {n: np.zeros(2, np.float32) for n in names[:4] + names[-3:]})
    cfgp = str(tmp_path / "cfg.json")
    json.dump(qo.config_tiny().to_dict(), open(cfgp, "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "import_safetensors.py"), "--list", "--config", cfgp, path],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "registry tensors without a source" in r.stdout and "missing cd.*" in r.stdout and "DUPLICATE" in r.stdout


def test_pack_refuses_an_incomplete_tensor_set(tmp_path):
    """tools/pack_weights.write_q3w validates against the registry (the loader would refuse the file anyway, naming what is missing)."""
    import pytest
    sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
    import q3tts
    from tools.pack_weights import check_complete, write_q3w
    oc = qo.config_tiny()
    cfg = q3tts.Config.from_dict(oc.to_dict())
    w = qo.random_weights(oc, 0)
    check_complete(cfg, w)
    short = {k: v for k, v in w.items() if k != "cp.norm"}
    with pytest.raises(ValueError, match="1 missing"):
        write_q3w(str(tmp_path / "a.q3w"), cfg, short)
    with pytest.raises(ValueError, match="unknown"):
        write_q3w(str(tmp_path / "b.q3w"), cfg, dict(w, bogus=np.zeros(3, np.float32)))
    with pytest.raises(ValueError, match="wrong size"):
        write_q3w(str(tmp_path / "c.q3w"), cfg, dict(w, **{"cp.norm": np.zeros(3, np.float32)}))
    write_q3w(str(tmp_path / "d.q3w"), cfg, w)
    cfg.cd_tconv_trim = 1                                  # the trim convention travels in the file's config
    write_q3w(str(tmp_path / "e.q3w"), cfg, w)
    back = q3tts.Config()
    assert q3tts.lib().q3tts_read_weights_config(os.fsencode(str(tmp_path / "e.q3w")), back) == 0 and back.cd_tconv_trim == 1


def _pb_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb_field(no, wt, payload):
    return _pb_varint(no << 3 | wt) + (_pb_varint(len(payload)) + payload if wt == 2 else payload)


def test_onnx_initializer_lister_reads_its_wire_format(tmp_path):
    """tools/import_onnx.py's reader (the artefacts the reference's users hold are .onnx graphs, /root/reference/src/tts_onnx.cpp:91-107):
    the dependency-free protobuf reader finds every initialiser of a ModelProto.  NOT a reference fixture — no .onnx file, ONNX Runtime
    or `onnx` package exists in the image — the file is written here, field by field, from the published schema (ModelProto.graph = 7,
    GraphProto.initializer = 5, TensorProto dims = 1 / data_type = 2 / name = 8 / raw_data = 9 / float_data = 4 / external_data = 13),
    with both dims encodings (packed and one varint per field) and a node field in front that must be skipped."""
    from tools.import_onnx import initializers
    w = np.arange(12, dtype=np.float32).reshape(3, 4)
    t_raw = (_pb_field(1, 2, _pb_varint(3) + _pb_varint(4)) + _pb_field(2, 0, _pb_varint(1)) + _pb_field(8, 2, b"onnx::MatMul_1234") + _pb_field(9, 2, w.tobytes()))
    t_typed = (_pb_field(1, 0, _pb_varint(5)) + _pb_field(2, 0, _pb_varint(1)) + _pb_field(4, 2, np.ones(5, np.float32).tobytes()) + _pb_field(8, 2, b"talker.norm.weight"))
    t_ext = (_pb_field(1, 2, _pb_varint(2048) + _pb_varint(1024)) + _pb_field(2, 0, _pb_varint(16)) + _pb_field(8, 2, b"big")
             + _pb_field(13, 2, _pb_field(1, 2, b"location") + _pb_field(2, 2, b"weights.bin")) + _pb_field(14, 0, _pb_varint(1)))
    node = _pb_field(1, 2, _pb_field(4, 2, b"MatMul"))                       # GraphProto.node = 1: skipped
    graph = node + _pb_field(2, 2, b"main_graph") + b"".join(_pb_field(5, 2, t) for t in (t_raw, t_typed, t_ext))
    model = _pb_field(1, 0, _pb_varint(8)) + _pb_field(2, 2, b"pytorch") + _pb_field(7, 2, graph) + _pb_field(8, 2, _pb_field(2, 0, _pb_varint(17)))
    path = tmp_path / "toy.onnx"
    path.write_bytes(model)
    ts = initializers(str(path))
    assert [t["name"] for t in ts] == ["onnx::MatMul_1234", "talker.norm.weight", "big"]
    assert ts[0]["dims"] == [3, 4] and ts[0]["data_type"] == 1 and ts[0]["raw_bytes"] == 48 and not ts[0]["external"]
    from tools.import_onnx import tensor_array
    assert np.array_equal(tensor_array(ts[0], str(tmp_path)), w) and np.array_equal(tensor_array(ts[1], str(tmp_path)), np.ones(5, np.float32))
    assert ts[1]["dims"] == [5] and ts[1]["n_typed"] == 5
    assert ts[2]["dims"] == [2048, 1024] and ts[2]["data_type"] == 16 and ts[2]["external"]
    # truncated files are reported, not mis-read
    (tmp_path / "cut.onnx").write_bytes(model[: len(model) - 7])
    try:
        initializers(str(tmp_path / "cut.onnx"))
        assert False, "a truncated file must raise"
    except ValueError:
        pass


def _onnx_fixture(tmp_path, cfg, seed=4):
    from tools.make_onnx_fixture import write_fixture
    w = qo.random_weights(cfg, seed)
    keys = json.load(open(os.path.join(GOLD, "hf_state_dict_keys.json")))
    paths = write_fixture(str(tmp_path / "onnx"), qo.tensor_specs(cfg), w, keys)
    return w, paths


def test_onnx_converter_round_trips_the_self_made_graphs_bit_for_bit(tmp_path):
    """tools/import_onnx.py --by-shape-order on the eight .onnx-format files tools/make_onnx_fixture.py writes from the tiny config's seeded
    tensors (one per session of /root/reference/src/tts_onnx.cpp:91-107; Linear weights anonymous and transposed, the rest under
    state_dict names; raw / typed / bf16 / external data carriers; the talker stack twice; int64 constants): every registry tensor comes
    back bit for bit, through the Q3TW0001 file too.  A FORMAT round trip — no claim about the names inside a real export."""
    from tools.import_onnx import import_onnx
    from tools.pack_weights import write_q3w
    sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
    import q3tts
    cfg = qo.config_tiny()
    w, paths = _onnx_fixture(tmp_path, cfg)
    assert [os.path.basename(p) for p in paths] == ["talker_prefill.onnx", "talker_decode.onnx", "codec_embed.onnx", "text_project.onnx", "code_predictor.onnx",
                                                    "code_predictor_embed.onnx", "tokenizer12hz_decode.onnx", "speaker_encoder.onnx"]
    qcfg = q3tts.Config.from_dict(cfg.to_dict())
    lines = []
    # anonymous SQUARE matrices (k_proj / v_proj at these dims) have no orientation in their shape: refused unless the caller says so
    import pytest
    with pytest.raises(ValueError, match="anonymous SQUARE matrix"):
        import_onnx(paths, qcfg, by_shape_order=True, log=lambda s: None)
    got, unused, missing = import_onnx(paths, qcfg, by_shape_order=True, assume_square_transposed=True, log=lines.append)
    assert not missing and not unused and set(got) == set(w)
    for n in w:
        assert got[n].shape == tuple(w[n].shape) and np.array_equal(got[n].view(np.uint32), np.ascontiguousarray(w[n], np.float32).view(np.uint32)), n
    assert any("talker.layers.0.k_proj" in ln and "[in][out]" in ln for ln in lines)      # every heuristic assignment is reported
    out = str(tmp_path / "model.q3w")
    write_q3w(out, qcfg, got)          # validates against the registry: every tensor once, every shape right
    assert os.path.getsize(out) > sum(int(np.prod(s)) for _n, s, _k in qo.tensor_specs(cfg)) * 2


def test_onnx_converter_refuses_what_it_cannot_resolve(tmp_path):
    """Without --by-shape-order the anonymous MatMul weights stay unresolved (named in the error); an explicit --map entry resolves one and
    can transpose it; a shape class whose counts differ is refused; two graphs that disagree about a tensor are an error."""
    import pytest
    from tools.import_onnx import import_onnx
    from tools.make_onnx_fixture import model_proto, tensor_proto
    sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))
    import q3tts
    cfg = qo.config_tiny()
    w, paths = _onnx_fixture(tmp_path, cfg)
    qcfg = q3tts.Config.from_dict(cfg.to_dict())
    with pytest.raises(ValueError, match="without a source"):
        import_onnx(paths, qcfg)
    got, unused, missing = import_onnx(paths, qcfg, allow_missing=True, log=lambda s: None)
    assert "talker.layers.0.q_proj" in missing and "talker.norm" in got and any("onnx::MatMul" in u for u in unused)
    first = [u for u in unused if u.startswith("talker_prefill:onnx::MatMul")][0]
    import re
    got2, _, _ = import_onnx(paths, qcfg, extra_rules=[(re.escape(first), "T:talker.layers.0.q_proj")], allow_missing=True, log=lambda s: None)
    assert np.array_equal(got2["talker.layers.0.q_proj"], w["talker.layers.0.q_proj"])
    # one more anonymous matrix of a class that is already full: counts no longer match
    extra = str(tmp_path / "onnx" / "zz_extra.onnx")
    open(extra, "wb").write(model_proto([tensor_proto("onnx::MatMul_9", np.zeros((cfg.hidden, cfg.ffn), np.float32) + 3, "raw32")]))
    with pytest.raises(ValueError, match="counts must match"):
        import_onnx(paths + [extra], qcfg, by_shape_order=True, assume_square_transposed=True, log=lambda s: None)
    # a second copy of a named tensor with different contents
    bad = str(tmp_path / "onnx" / "zz_bad.onnx")
    open(bad, "wb").write(model_proto([tensor_proto("talker.model.norm.weight", np.full(cfg.hidden, 2.0, np.float32), "raw32")]))
    with pytest.raises(ValueError, match="graphs disagree"):
        import_onnx(paths + [bad], qcfg, by_shape_order=True, assume_square_transposed=True, log=lambda s: None)
