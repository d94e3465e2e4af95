"""Why do the fused path and the oracle part at (frame, group) in sampled mode?  Rebuilds that one decision on both sides: logits row,
uniform, sampler outcome, and the cross combinations.  python tools/diverge_diag.py SEED FRAME GROUP"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("leaxer-qwen3-tts_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import q3_oracle as qo  # noqa: E402
import q3tts  # noqa: E402
from util import frame_tokens, to_ocfg, to_osampling  # noqa: E402

seed, fr, grp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = q3tts.default_config("0.6b")
eng = q3tts.Engine(cfg, device=0, max_batch=2, max_ctx=512)
eng.fill_synthetic(seed=0)
orc = qo.Oracle(to_ocfg(cfg), max_ctx=192)
for name, shape in eng.tensor_infos():
    if not name.startswith(("cd.", "spk.")):
        orc.set_tensor(name, eng.get_tensor(name, shape))
ids = frame_tokens(np.random.default_rng(4).integers(0, 151643, 16))
kw = dict(temperature=0.8, top_p=0.95, top_k=50)
sp = q3tts.Sampling(max_new_tokens=fr + 1, **kw)
p, t = eng.build_prompt(ids, 0)
po = orc.build_prompt(ids, 0)
V = cfg.vocab if grp == 0 else cfg.sub_vocab
Lo = orc.dump_logits_of(fr, grp, V)
ref, mg = orc.generate_margins(po, to_osampling(sp), seed=seed, stream=2, cp_cached=True, ignore_eos=True)
codes = eng.generate(p, t, sp, seed=seed, stream_id=2, ignore_eos=True)
print("frame", fr, "hip", codes[fr].tolist())
print("frame", fr, "orc", ref[fr].tolist())
print("decision margins of the frame:", np.array2string(mg[fr, 2:], precision=3))
# HIP side of the same decision through the session-shaped ops
spf = q3tts.Sampling(max_new_tokens=fr, **kw)
eng.generate(p, t, spf, seed=seed, stream_id=2, ignore_eos=True)          # slot 0 now waits at frame `fr`
lg0, lh = eng.slot_logits(0)
if grp == 0:
    Lh = lg0
else:
    rows = [lh, eng.codec_embed([int(ref[fr, 0])])[0]] + [eng.cp_embed(int(ref[fr, j + 1]), j) for j in range(grp - 1)]
    Lh = eng.code_predictor(np.stack(rows), grp - 1)
print("logits row: max |hip - oracle| = %.3g, oracle std %.3g, range [%.3g, %.3g]" % (float(np.abs(Lh - Lo).max()), float(Lo.std()), float(Lo.min()), float(Lo.max())))
u = qo.rng_uniform(seed, 2, fr, grp)
u2 = q3tts.rng_uniform(seed, 2, fr, grp)
so = to_osampling(sp)
print("u oracle %.9g  u lib %.9g" % (u, u2))
print("sample: hip(Lh) %d  hip(Lo) %d  orc(Lh) %d  orc(Lo) %d   (fused %d, oracle run %d)" %
      (eng.sample(Lh, sp, u), eng.sample(Lo, sp, u), orc.sample(Lh, so, u), orc.sample(Lo, so, u), codes[fr, grp], ref[fr, grp]))
tc, dc, tot = orc.sample_trace(Lo, so)
print("oracle kept after top-k/top-p:", int((dc > 0).sum()), "total", tot, " top-p sums around the cut:", tc[max(0, np.searchsorted(tc, 0.95) - 2): np.searchsorted(tc, 0.95) + 2])
srt = np.sort(Lo)[::-1]
print("top-k neighbourhood (ranks 48..53 of the oracle logits / temp):", srt[47:53] / 0.8)
# the fused path's own logits of every decision of that frame (one eager step with the heads' rows kept)
eng.slot_begin(0, p, t, sp, seed=seed, stream_id=2, ignore_eos=True)     # room for frame `fr`: max_new_tokens = fr + 1
eng.decode_steps(fr)
rows = eng.step_logits(0)
fused_codes = eng.slot_codes(0)[fr]
print("fused codes of the frame:", fused_codes.tolist())
for g in range(cfg.n_groups):
    Vg = cfg.vocab if g == 0 else cfg.sub_vocab
    Lg = orc.dump_logits_of(fr, g, Vg)
    orc.generate_margins(po, to_osampling(sp), seed=seed, stream=2, cp_cached=True, ignore_eos=True)
    d = np.abs(rows[g, :Vg] - Lg)
    d = d[np.isfinite(d)]
    print("group %2d: max |fused - oracle| logits %.3g (argmax diff at %d)  fused code %d oracle code %d" % (g, float(d.max()), int(np.argmax(np.abs(np.where(np.isfinite(Lg), rows[g, :Vg] - Lg, 0)))), fused_codes[g], ref[fr, g]))
    if fused_codes[g] != ref[fr, g]:
        ug = qo.rng_uniform(seed, 2, fr, g)
        print("   standalone on the FUSED row: hip %d orc %d; on the oracle row: hip %d orc %d" %
              (eng.sample(rows[g, :Vg], sp, ug, suppress=(g == 0)), orc.sample(rows[g, :Vg], so, ug), eng.sample(Lg, sp, ug), orc.sample(Lg, so, ug)))
        break
