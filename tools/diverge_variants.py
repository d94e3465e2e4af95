"""First divergence (vs the oracle) of a 60-frame sampled run under engine variants: which code path parts from the oracle?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("leaxer-qwen3-tts_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import q3_oracle as qo  # noqa: E402
import q3tts  # noqa: E402
from util import frame_tokens, to_ocfg, to_osampling  # noqa: E402

seed, F = int(sys.argv[1]), int(sys.argv[2])
cfg = q3tts.default_config("0.6b")
ids = frame_tokens(np.random.default_rng(4).integers(0, 151643, 16))
sp = q3tts.Sampling(max_new_tokens=F, temperature=0.8, top_p=0.95, top_k=50)
ref = None
for tag, flags, nb in (("default", 0, 1), ("no_graph", q3tts.FLAG_NO_GRAPH, 1), ("no_fused_cp", q3tts.FLAG_NO_FUSED_CP, 1),
                       ("no_graph+no_fused_cp", q3tts.FLAG_NO_GRAPH | q3tts.FLAG_NO_FUSED_CP, 1), ("two_slots", 0, 2), ("three_slots", 0, 3)):
    eng = q3tts.Engine(cfg, device=0, max_batch=max(nb, 1), max_ctx=256, flags=flags)
    eng.fill_synthetic(seed=0)
    if ref is None:
        orc = qo.Oracle(to_ocfg(cfg), max_ctx=192)
        for name, shape in eng.tensor_infos():
            if not name.startswith(("cd.", "spk.")):
                orc.set_tensor(name, eng.get_tensor(name, shape))
        ref = orc.generate(orc.build_prompt(ids, 0), to_osampling(sp), seed=seed, stream=2, cp_cached=True, ignore_eos=True)
    p, t = eng.build_prompt(ids, 0)
    for b in range(nb):
        eng.slot_begin(b, p, t, sp, seed=seed, stream_id=2, ignore_eos=True)      # every slot runs the SAME utterance and RNG stream
    eng.decode_steps(F)
    for b in range(nb):
        codes = eng.slot_codes(b)
        bad = np.argwhere(codes != ref)
        print("%-22s slot %d: first divergence %s" % (tag, b, "none in %d frames" % F if bad.size == 0 else "frame %d group %d (hip %d, oracle %d)" % (bad[0][0], bad[0][1], codes[bad[0][0], bad[0][1]], ref[bad[0][0], bad[0][1]])))
    eng.close()
