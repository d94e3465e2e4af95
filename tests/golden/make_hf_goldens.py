"""Generate golden vectors that pin the oracle's [HINT] network architecture.

The reference (leaxer-ai/leaxer-qwen3-tts) treats its networks as opaque .onnx files and holds no
golden vectors for them.  The architecture those graphs implement is the
public Qwen3 decoder / Qwen3-Omni talker code predictor / Code2Wav; this script instantiates the
`transformers` implementation of each (installed in the build container, never shipped) at tiny
seeded dims and stores weights + inputs + outputs under the oracle's tensor names.

    python tests/golden/make_hf_goldens.py      # rewrites tests/golden/hf_*.npz

Run in the build container only; the .npz fixtures are committed and are what the tests read.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from q3_oracle import bf16_round, config_tiny  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)
CFG = config_tiny()


def rnd_(p, kind, g):
    """Fill parameter in place with bf16-representable seeded values."""
    shape = tuple(p.shape)
    if kind == "w":
        fan = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        a = torch.randn(shape, generator=g) / np.sqrt(max(fan, 1))
    elif kind == "norm":
        a = 1.0 + 0.1 * torch.randn(shape, generator=g)
    elif kind == "scale":
        a = 0.5 + 0.1 * torch.randn(shape, generator=g)
    elif kind == "snake":
        a = 0.3 * torch.randn(shape, generator=g)
    else:
        a = 0.1 * torch.randn(shape, generator=g)
    p.copy_(torch.from_numpy(bf16_round(a.numpy())))


def layer_map(prefix, hf_layer, out, qk=True, ls=False):
    out[prefix + "input_norm"] = (hf_layer.input_layernorm.weight, "norm")
    out[prefix + "post_norm"] = (hf_layer.post_attention_layernorm.weight, "norm")
    at = hf_layer.self_attn
    out[prefix + "q_proj"] = (at.q_proj.weight, "w")
    out[prefix + "k_proj"] = (at.k_proj.weight, "w")
    out[prefix + "v_proj"] = (at.v_proj.weight, "w")
    out[prefix + "o_proj"] = (at.o_proj.weight, "w")
    if qk:
        out[prefix + "q_norm"] = (at.q_norm.weight, "norm")
        out[prefix + "k_norm"] = (at.k_norm.weight, "norm")
    out[prefix + "gate_proj"] = (hf_layer.mlp.gate_proj.weight, "w")
    out[prefix + "up_proj"] = (hf_layer.mlp.up_proj.weight, "w")
    out[prefix + "down_proj"] = (hf_layer.mlp.down_proj.weight, "w")
    if ls:
        out[prefix + "attn_scale"] = (hf_layer.self_attn_layer_scale.scale, "scale")
        out[prefix + "mlp_scale"] = (hf_layer.mlp_layer_scale.scale, "scale")


KEYS = {}


def record_keys(component, module, mapping):
    """state_dict key of every mapped parameter: the naming a checkpoint of this module uses (tools/import_safetensors.py)"""
    by_id = {id(p): k for k, p in module.named_parameters()}
    KEYS[component] = {name: by_id[id(p)] for name, (p, _) in mapping.items()}


def fill(mapping, seed):
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, (p, kind) in mapping.items():
        rnd_(p, kind, g)
        w[name] = p.detach().numpy().astype(np.float32).copy()
    return w


def talker():
    from transformers.models.qwen3.configuration_qwen3 import Qwen3Config
    from transformers.models.qwen3.modeling_qwen3 import Qwen3ForCausalLM
    c = CFG
    cfg = Qwen3Config(vocab_size=c.vocab, hidden_size=c.hidden, intermediate_size=c.ffn, num_hidden_layers=c.n_layers,
                      num_attention_heads=c.n_heads, num_key_value_heads=c.n_kv_heads, head_dim=c.head_dim,
                      rms_norm_eps=c.rms_eps, tie_word_embeddings=False, attention_bias=False,
                      rope_parameters={"rope_type": "default", "rope_theta": float(c.rope_theta)},
                      max_position_embeddings=512)
    cfg._attn_implementation = "eager"
    m = Qwen3ForCausalLM(cfg).eval()
    mp = {}
    for i, lyr in enumerate(m.model.layers):
        layer_map(f"talker.layers.{i}.", lyr, mp)
    mp["talker.norm"] = (m.model.norm.weight, "norm")
    mp["talker.codec_head"] = (m.lm_head.weight, "w")
    record_keys("talker", m, mp)
    w = fill(mp, 11)
    g = torch.Generator().manual_seed(12)
    S, n_dec = 9, 4
    x = torch.randn(1, S, c.hidden, generator=g)
    out = m(inputs_embeds=x, use_cache=True, output_hidden_states=True)
    res = {"prefill_in": x[0].numpy(), "prefill_logits": out.logits[0].numpy(),
           "prefill_last_hidden": out.hidden_states[-1][0, -1].numpy()}
    past = out.past_key_values
    dec_in, dec_logits, dec_hidden = [], [], []
    for _ in range(n_dec):
        e = torch.randn(1, 1, c.hidden, generator=g)
        o = m(inputs_embeds=e, past_key_values=past, use_cache=True, output_hidden_states=True)
        past = o.past_key_values
        dec_in.append(e[0, 0].numpy()); dec_logits.append(o.logits[0, 0].numpy()); dec_hidden.append(o.hidden_states[-1][0, 0].numpy())
    res.update(decode_in=np.stack(dec_in), decode_logits=np.stack(dec_logits), decode_last_hidden=np.stack(dec_hidden))
    np.savez_compressed(os.path.join(HERE, "hf_talker.npz"), **{"w:" + k: v for k, v in w.items()}, **res)
    print("hf_talker.npz", {k: v.shape for k, v in res.items()})


def predictor():
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeTalkerCodePredictorConfig
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import (
        Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration as CP)
    c = CFG
    cfg = Qwen3OmniMoeTalkerCodePredictorConfig(
        vocab_size=c.sub_vocab, hidden_size=c.hidden, intermediate_size=c.cp_ffn, num_hidden_layers=c.cp_layers,
        num_attention_heads=c.cp_heads, num_key_value_heads=c.cp_kv_heads, head_dim=c.cp_head_dim,
        rms_norm_eps=c.cp_rms_eps, num_code_groups=c.n_groups,
        rope_parameters={"rope_type": "default", "rope_theta": float(c.cp_rope_theta)})
    cfg._attn_implementation = "eager"
    m = CP(cfg).eval()
    mp = {}
    for i, lyr in enumerate(m.model.layers):
        layer_map(f"cp.layers.{i}.", lyr, mp)
    mp["cp.norm"] = (m.model.norm.weight, "norm")
    for j in range(c.n_groups - 1):
        mp[f"cp.head.{j}"] = (m.lm_head[j].weight, "w")
        mp[f"cp.embed.{j}"] = (m.model.codec_embedding[j].weight, "w")
    record_keys("predictor", m, mp)
    w = fill(mp, 21)
    g = torch.Generator().manual_seed(22)
    seq = torch.randn(1, c.n_groups + 1, c.hidden, generator=g)
    logits = []
    for j in range(c.n_groups - 1):  # reference call pattern: rows [0, j+2), head j (tts_onnx.cpp:862-868)
        o = m(inputs_embeds=seq[:, : j + 2], use_cache=False)
        logits.append(o.logits[0, -1].numpy())
    res = {"seq": seq[0].numpy(), "logits": np.stack(logits)}
    np.savez_compressed(os.path.join(HERE, "hf_predictor.npz"), **{"w:" + k: v for k, v in w.items()}, **res)
    print("hf_predictor.npz", {k: v.shape for k, v in res.items()})


def code2wav():
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeCode2WavConfig
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import Qwen3OmniMoeCode2Wav
    c = CFG
    cfg = Qwen3OmniMoeCode2WavConfig(
        codebook_size=c.cd_codebook, hidden_size=c.cd_hidden, num_attention_heads=c.cd_heads,
        num_key_value_heads=c.cd_heads, intermediate_size=c.cd_ffn, num_hidden_layers=c.cd_layers,
        num_quantizers=c.n_groups, decoder_dim=c.cd_decoder_dim, sliding_window=c.cd_window,
        rms_norm_eps=c.cd_rms_eps, upsample_rates=tuple(c.cd_up_rates[: c.cd_n_blocks]),
        upsampling_ratios=tuple(c.cd_up_ratios[: c.cd_n_up]),
        rope_parameters={"rope_type": "default", "rope_theta": float(c.cd_rope_theta)})
    cfg._attn_implementation = "eager"
    m = Qwen3OmniMoeCode2Wav(cfg).eval()
    mp = {}
    for i, lyr in enumerate(m.pre_transformer.layers):
        layer_map(f"cd.layers.{i}.", lyr, mp, qk=False, ls=True)
    mp["cd.norm"] = (m.pre_transformer.norm.weight, "norm")
    mp["cd.code_embed"] = (m.code_embedding.weight, "w")
    for s, (tc, cnx) in enumerate(m.upsample):
        p = f"cd.up.{s}."
        mp[p + "tconv.w"] = (tc.conv.weight, "w"); mp[p + "tconv.b"] = (tc.conv.bias, "b")
        mp[p + "cnx.dw.w"] = (cnx.dwconv.conv.weight, "w"); mp[p + "cnx.dw.b"] = (cnx.dwconv.conv.bias, "b")
        mp[p + "cnx.ln.w"] = (cnx.norm.weight, "norm"); mp[p + "cnx.ln.b"] = (cnx.norm.bias, "b")
        mp[p + "cnx.pw1.w"] = (cnx.pwconv1.weight, "w"); mp[p + "cnx.pw1.b"] = (cnx.pwconv1.bias, "b")
        mp[p + "cnx.pw2.w"] = (cnx.pwconv2.weight, "w"); mp[p + "cnx.pw2.b"] = (cnx.pwconv2.bias, "b")
        mp[p + "cnx.gamma"] = (cnx.gamma, "scale")
    dec = m.decoder
    mp["cd.dec.conv_in.w"] = (dec[0].conv.weight, "w"); mp["cd.dec.conv_in.b"] = (dec[0].conv.bias, "b")
    for i in range(c.cd_n_blocks):
        blk = dec[1 + i].block
        p = f"cd.dec.blocks.{i}."
        mp[p + "snake.alpha"] = (blk[0].alpha, "snake"); mp[p + "snake.beta"] = (blk[0].beta, "snake")
        mp[p + "tconv.w"] = (blk[1].conv.weight, "w"); mp[p + "tconv.b"] = (blk[1].conv.bias, "b")
        for u in range(3):
            r = blk[2 + u]
            q = p + f"res.{u}."
            mp[q + "act1.alpha"] = (r.act1.alpha, "snake"); mp[q + "act1.beta"] = (r.act1.beta, "snake")
            mp[q + "conv1.w"] = (r.conv1.conv.weight, "w"); mp[q + "conv1.b"] = (r.conv1.conv.bias, "b")
            mp[q + "act2.alpha"] = (r.act2.alpha, "snake"); mp[q + "act2.beta"] = (r.act2.beta, "snake")
            mp[q + "conv2.w"] = (r.conv2.conv.weight, "w"); mp[q + "conv2.b"] = (r.conv2.conv.bias, "b")
    n = 1 + c.cd_n_blocks
    mp["cd.dec.snake_out.alpha"] = (dec[n].alpha, "snake"); mp["cd.dec.snake_out.beta"] = (dec[n].beta, "snake")
    mp["cd.dec.conv_out.w"] = (dec[n + 1].conv.weight, "w"); mp["cd.dec.conv_out.b"] = (dec[n + 1].conv.bias, "b")
    record_keys("code2wav", m, mp)
    w = fill(mp, 31)
    g = torch.Generator().manual_seed(32)

    def unclamped(codes):
        hidden = m.code_embedding(codes + m.code_offset).mean(1)
        hidden = m.pre_transformer(inputs_embeds=hidden).last_hidden_state.permute(0, 2, 1)
        for blocks in m.upsample:
            for block in blocks:
                hidden = block(hidden)
        for block in m.decoder:
            hidden = block(hidden)
        return hidden

    # calibrate the last conv so the PCM sits well inside the clamp range (a saturated output hides errors)
    cal = unclamped(torch.randint(0, c.cd_codebook, (1, c.n_groups, 7), generator=g))
    scale = 0.2 / float(cal.pow(2).mean().sqrt())
    for key, prm in (("cd.dec.conv_out.w", dec[n + 1].conv.weight), ("cd.dec.conv_out.b", dec[n + 1].conv.bias)):
        prm.copy_(torch.from_numpy(bf16_round((prm * scale).numpy())))
        w[key] = prm.detach().numpy().astype(np.float32).copy()
    res = {}
    for F in (1, 3, 7):
        codes = torch.randint(0, c.cd_codebook, (1, c.n_groups, F), generator=g)
        wav = m(codes)
        res[f"codes_{F}"] = codes[0].T.contiguous().numpy().astype(np.int64)  # [F][G] frame-major (tts_onnx.cpp:421-427)
        res[f"pcm_{F}"] = wav[0, 0].numpy()
    np.savez_compressed(os.path.join(HERE, "hf_code2wav.npz"), **{"w:" + k: v for k, v in w.items()}, **res)
    print("hf_code2wav.npz", {k: v.shape for k, v in res.items()}, "pcm rms", float(np.sqrt((res["pcm_7"] ** 2).mean())))


def speaker():
    """ECAPA-TDNN speaker encoder of the clone path (the reference runs it as an opaque speaker_encoder.onnx,
    src/tts_onnx.cpp:367-403): transformers' ECAPA_TimeDelayNet at the tiny config's dims."""
    from types import SimpleNamespace
    from transformers.models.qwen2_5_omni.modeling_qwen2_5_omni import ECAPA_TimeDelayNet
    c = CFG
    SC = c.spk_channels
    hc = SimpleNamespace(mel_dim=c.spk_mel, enc_dim=c.spk_enc_dim, enc_channels=[SC, SC, SC, SC, 3 * SC],
                         enc_kernel_sizes=[5, 3, 3, 3, 1], enc_dilations=[1, 2, 3, 4, 1], enc_attention_channels=c.spk_att,
                         enc_res2net_scale=c.spk_scale, enc_se_channels=c.spk_se)
    net = ECAPA_TimeDelayNet(hc).eval()
    mp = {}

    def conv(name, mod):
        mp[name + ".w"] = (mod.weight, "w")
        mp[name + ".b"] = (mod.bias, "b")
    conv("spk.tdnn0", net.blocks[0].conv)
    for i in range(3):
        b = net.blocks[i + 1]
        conv(f"spk.blocks.{i}.tdnn1", b.tdnn1.conv)
        for j in range(c.spk_scale - 1):
            conv(f"spk.blocks.{i}.res2net.{j}", b.res2net_block.blocks[j].conv)
        conv(f"spk.blocks.{i}.tdnn2", b.tdnn2.conv)
        conv(f"spk.blocks.{i}.se1", b.se_block.conv1)
        conv(f"spk.blocks.{i}.se2", b.se_block.conv2)
    conv("spk.mfa", net.mfa.conv)
    conv("spk.asp.tdnn", net.asp.tdnn.conv)
    conv("spk.asp.conv", net.asp.conv)
    conv("spk.fc", net.fc)
    record_keys("speaker", net, mp)
    w = fill(mp, 31)
    g = torch.Generator().manual_seed(32)
    res = {}
    for T in (5, 9, 40):
        mel = torch.from_numpy(bf16_round((2.0 * torch.randn(c.spk_mel, T, generator=g) - 4.0).numpy()))   # log-mel-like range
        out = net(mel.t().unsqueeze(0))          # the session input layout [1, frames, n_mels]
        res[f"mel_{T}"] = mel.numpy()
        res[f"embed_{T}"] = out[0].numpy()
    np.savez_compressed(os.path.join(HERE, "hf_speaker.npz"), **{"w:" + k: v for k, v in w.items()}, **res)
    print("hf_speaker.npz", {k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (talker, predictor, code2wav, speaker):
        if not only or fn.__name__ in only:
            fn()
    if not only:
        import json
        json.dump(KEYS, open(os.path.join(HERE, "hf_state_dict_keys.json"), "w"), indent=0, sort_keys=True)
        print("hf_state_dict_keys.json", {k: len(v) for k, v in KEYS.items()})
