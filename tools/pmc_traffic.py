"""HBM traffic per decode step from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section):
bytes = FETCH_SIZE*1024*2 (gfx950 reports exactly half of a wide coalesced read stream) + WRITE_SIZE*1024.

    pmc_traffic.py --batch B --fetch f12.csv --write w12.csv --frames 12 [--fetch2 f24.csv --write2 w24.csv --frames2 24] [--merge-into profiles/decode_step_traffic.json]

Only the decode step's kernels are counted (k_gemv*/k_gemm*/k_attn*/k_cp_attn_oproj/k_sample/k_finish*): finalize-time kernels
(k_repack_conv, k_split_planes, k_fill_synth ...) and the codec decoder are not part of a step.  With a second pair of passes at a
different step count the per-step figure is the DIFFERENCE of the two runs divided by the difference of their step counts, which also
removes the one-time prefill launches (same kernel names as the step's) from the figure.
--merge-into updates the JSON bench.py reads, stamping it with the digest of the kernel sources the passes were taken on."""
import argparse
import csv
import hashlib
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECODE = ("k_gemv", "k_gemm", "k_attn", "k_cp_attn_oproj", "k_sample", "k_finish", "k_rmsnorm_split")


def total(path, counter):
    tot = defaultdict(float)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void q3::", "").replace("q3::", "")
            if name.startswith(DECODE):
                tot[name] += float(r["Counter_Value"])
    return tot


def src_digest():
    h = hashlib.sha1()
    d = os.path.join(ROOT, "leaxer-qwen3-tts_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, required=True)
ap.add_argument("--ctx", type=int, default=0, help="talker context the passes were taken at (0: the first steps after the prompt, context ~10-20)")
ap.add_argument("--kv", choices=["fp32", "bf16"], default="fp32")
ap.add_argument("--model", default="0.6b", help='"1.7b": BASELINE configs[4] dims (keys get an "m1.7b_" prefix)')
ap.add_argument("--fetch", required=True)
ap.add_argument("--write", required=True)
ap.add_argument("--frames", type=int, required=True)
ap.add_argument("--fetch2")
ap.add_argument("--write2")
ap.add_argument("--frames2", type=int)
ap.add_argument("--merge-into")
a = ap.parse_args()

f1, w1 = total(a.fetch, "FETCH_SIZE"), total(a.write, "WRITE_SIZE")
if a.fetch2:
    f2, w2 = total(a.fetch2, "FETCH_SIZE"), total(a.write2, "WRITE_SIZE")
    n = a.frames2 - a.frames
    fetch = {k: f2.get(k, 0.0) - f1.get(k, 0.0) for k in set(f1) | set(f2)}
    write = {k: w2.get(k, 0.0) - w1.get(k, 0.0) for k in set(w1) | set(w2)}
    method = f"difference of a {a.frames2}-step and a {a.frames}-step run (one-time prefill launches cancel)"
else:
    fetch, write, n = f1, w1, a.frames
    method = f"one {a.frames}-step run, includes the one prefill"
rd = sum(fetch.values()) * 1024 * 2
wr = sum(write.values()) * 1024
out = {"batch": a.batch, "ctx": a.ctx, "kv": a.kv, "steps_counted": n, "read_bytes_per_step": rd / n, "write_bytes_per_step": wr / n, "hbm_bytes_per_step": (rd + wr) / n,
       "method": method + "; decode-step kernels only; FETCH_SIZE doubled per the gfx950 correction",
       "src_digest": src_digest(),
       "by_kernel_read_MB_per_step": {k: round(v * 2048 / n / 1e6, 2) for k, v in sorted(fetch.items(), key=lambda kv: -kv[1])[:10]}}
print(json.dumps(out))
if a.merge_into:
    j = {}
    if os.path.exists(a.merge_into):
        try:
            j = json.load(open(a.merge_into))
        except Exception:
            j = {}
    if j.get("src_digest") != out["src_digest"]:
        j = {}                       # figures of another build are not carried over
    j["src_digest"] = out["src_digest"]
    # keyed by (batch, context, kv dtype): "b64" = the short-context fp32 figure of earlier rounds, "b64_ctx1024_bf16" a long-context one
    key = ("" if a.model == "0.6b" else f"m{a.model}_") + f"b{a.batch}" + (f"_ctx{a.ctx}" if a.ctx > 0 else "") + ("_bf16" if a.kv == "bf16" else "")
    j[key] = int(out["hbm_bytes_per_step"])
    j[key + "_detail"] = {"read": int(out["read_bytes_per_step"]), "write": int(out["write_bytes_per_step"]), "ctx": a.ctx, "kv": a.kv, "model": a.model, "method": out["method"]}
    json.dump(j, open(a.merge_into, "w"), indent=1)
