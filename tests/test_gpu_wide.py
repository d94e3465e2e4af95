"""The 1.7B structure (BASELINE configs[4]): talker wider than the code predictor, predictor inputs through cp.proj.
Beyond the reference (README.md:125 "planned"; tts_onnx.h:31-37 hard-codes the 0.6B dims), so the oracle is the [HINT] restatement
alone; every check is HIP path vs oracle on identical weights, integer outputs bit-exact."""
import numpy as np
import pytest

import q3_oracle as qo
from util import frame_tokens, tiny_pair, to_ocfg, to_osampling

pytestmark = pytest.mark.gpu


def test_projected_predictor_session_ops():
    eng, orc, _ = tiny_pair(seed=11, max_batch=2, max_ctx=64, ocfg=qo.config_tiny_proj())
    rng = np.random.default_rng(1)
    for n, step in ((1, 0), (2, 0), (5, 3), (16, 14)):
        seq = rng.standard_normal((n, eng.cfg.hidden)).astype(np.float32)
        got, ref = eng.code_predictor(seq, step), orc.code_predictor(seq, step)
        assert got.shape == ref.shape == (eng.cfg.sub_vocab,)
        assert np.abs(got - ref).max() < 2e-5, (n, step)
    e = eng.cp_embed(7, 2)
    assert e.shape == (eng.cfg.hidden,) and np.array_equal(e, orc.cp_embed(7, 2))     # predictor embeddings are talker-wide
    eng.close()
    orc.close()


@pytest.mark.parametrize("flags", [0, 1])   # hipGraph replay / eager launches
def test_projected_generation_bit_exact(flags):
    import q3tts
    eng, orc, _ = tiny_pair(seed=12, max_batch=3, max_ctx=96, ocfg=qo.config_tiny_proj(), flags=flags)
    rng = np.random.default_rng(2)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in (1, 7, 19)]
    for sp in (q3tts.Sampling(temperature=1.0, top_p=1.0, top_k=1, max_new_tokens=24),
               q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=24)):
        pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=1, seed=9, ignore_eos=False)
        for u, t in enumerate(toks):
            ref = orc.generate(orc.build_prompt(t, 1), to_osampling(sp), seed=9, stream=u, cp_cached=True, ignore_eos=False)
            assert nfr[u] == len(ref) and np.array_equal(codes[u], ref), (u, nfr[u], len(ref))
            if len(ref):
                assert float(np.sqrt(np.mean((pcm[u] - orc.vocoder(ref)) ** 2))) < 1e-4
    eng.close()
    orc.close()


def test_projected_batch12_matrix_core_path():
    """12 utterances: predictor pass 0 has 24 rows, so the projection runs in row chunks and the layers on the MFMA path."""
    import q3tts
    eng, orc, _ = tiny_pair(seed=13, max_batch=12, max_ctx=128, ocfg=qo.config_medium_proj())
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=24)
    rng = np.random.default_rng(4)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(1, 20, 12)]
    pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=3, ignore_eos=True)
    for u, t in enumerate(toks):
        ref = orc.generate(orc.build_prompt(t, 0), to_osampling(sp), seed=3, stream=u, cp_cached=True, ignore_eos=True)
        assert nfr[u] == len(ref) == 24 and np.array_equal(codes[u], ref), u
    eng.close()
    orc.close()


@pytest.mark.parametrize("nb", [20, 70])
def test_projected_batches_whose_gemms_have_fewer_k_slices_than_row_chunks(nb):
    """config_medium_proj (talker 256 wide, o_proj K = 128, down K = 384) passes every shape test of the batched step's in-launch split-K
    reduction except one: its o_proj runs ONE K slice and its down projection three, while a 17..64-row block has two 16-row chunks to
    reduce and a 65..128-row block four — and chunk c is reduced by slice c.  Such launches must keep the finish kernels (gemm_seam_ok);
    20 and 70 utterances, every utterance of the batch against its own oracle run."""
    import q3tts
    eng, orc, _ = tiny_pair(seed=17, max_batch=nb, max_ctx=64, ocfg=qo.config_medium_proj())
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=10)
    rng = np.random.default_rng(nb)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(1, 12, nb)]
    pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=8, ignore_eos=True)
    for u, t in enumerate(toks):
        ref = orc.generate(orc.build_prompt(t, 0), to_osampling(sp), seed=8, stream=u, cp_cached=True, ignore_eos=True)
        assert nfr[u] == 10 and np.array_equal(codes[u], ref), u
    eng.close()
    orc.close()


def test_full_1p7b_dims_greedy_b1_b2_b16():
    """1.7B dims, seeded weights: greedy codec ids bit-exact vs the oracle through the three row regimes of the talker
    (1 row: single-pass GEMV incl. K = 6144; 2 rows: chunked GEMV; 16 rows: matrix-core GEMM), plus the talker-wide speaker row."""
    import q3tts
    cfg = q3tts.default_config("1.7b")
    eng = q3tts.Engine(cfg, device=0, max_batch=16, max_ctx=96)
    eng.fill_synthetic(seed=0)
    orc = qo.Oracle(to_ocfg(cfg), max_ctx=48)
    for name, shape in eng.tensor_infos():
        if not name.startswith("cd."):
            orc.set_tensor(name, eng.get_tensor(name, shape))
    sp = q3tts.Sampling(temperature=1.0, top_p=1.0, top_k=1, max_new_tokens=3)
    rng = np.random.default_rng(8)
    toks = [frame_tokens(rng.integers(0, 151643, 16)) for _ in range(16)]
    mel = rng.standard_normal((cfg.spk_mel, 40)).astype(np.float32)
    spk = eng.speaker_encoder(mel)
    ref_spk = orc.speaker_encoder(mel)
    assert spk.shape == (2048,) and np.abs(spk - ref_spk).max() < 1e-4 * max(1.0, float(np.abs(ref_spk).max()))
    refs = [orc.generate(orc.build_prompt(t, 0), to_osampling(sp), seed=2, stream=b, cp_cached=True, ignore_eos=True) for b, t in enumerate(toks[:3])]
    for nb in (1, 2, 16):
        for b in range(16):
            eng.slot_release(b)
        for b in range(nb):
            p, tr = eng.build_prompt(toks[b], 0)
            eng.slot_begin(b, p, tr, sp, seed=2, stream_id=b, ignore_eos=True)
        assert eng.decode_steps(3) == 0
        for b in range(min(nb, 3)):
            assert np.array_equal(eng.slot_codes(b), refs[b]), (nb, b)
    # clone path: the speaker row joins the prompt (tts_onnx.cpp:481-490) at talker width
    p, tr = eng.build_prompt(toks[0], 0, speaker=spk)
    po = orc.build_prompt(toks[0], 0, speaker=ref_spk)
    assert p.shape == po.shape and np.abs(p - po).max() < 1e-3
    eng.close()
    orc.close()


def test_configs4_clone_batch8_at_1p7b_dims():
    """BASELINE configs[4] at test length: 1.7B dims, --ref voice-clone path, 8 utterances in one batch (each with its own speaker row),
    greedy, 4 frames: codec ids bit-exact vs the oracle, PCM of the first utterance within 1e-4 RMS."""
    import q3tts
    cfg = q3tts.default_config("1.7b")
    eng = q3tts.Engine(cfg, device=0, max_batch=8, max_ctx=96)
    eng.fill_synthetic(seed=1)
    orc = qo.Oracle(to_ocfg(cfg), max_ctx=48)
    for name, shape in eng.tensor_infos():
        orc.set_tensor(name, eng.get_tensor(name, shape))
    rng = np.random.default_rng(9)
    sp = q3tts.Sampling(temperature=1.0, top_p=1.0, top_k=1, max_new_tokens=4)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(4, 20, 8)]
    spks = [qo.bf16_round(eng.speaker_encoder((2.0 * rng.standard_normal((cfg.spk_mel, 30 + 7 * u)) - 4.0).astype(np.float32))) for u in range(8)]
    pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=1, seed=6, ignore_eos=True, speakers=spks)
    for u, t in enumerate(toks):
        prompt = orc.build_prompt(t, 1, speaker=spks[u])
        ref = orc.generate(prompt, to_osampling(sp), seed=6, stream=u, cp_cached=True, ignore_eos=True)
        assert nfr[u] == 4 and np.array_equal(codes[u], ref), u
    ref_pcm = orc.vocoder(codes[0])
    assert pcm[0].shape == ref_pcm.shape and float(np.sqrt(np.mean((pcm[0] - ref_pcm) ** 2))) < 1e-4
    eng.close()
    orc.close()
