"""bench.py — headline metric of BASELINE.json: real-time factor (24 kHz audio seconds / wall seconds)
and codec-frames/sec of Qwen3-TTS-0.6B synthesis on MI355X, "0.6B @ b1/b64".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--frames F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the whole hot path over one batch of B utterances per GPU: prompt assembly,
talker prefill, F frames of [sampler + 15 code-predictor passes + talker decode] replayed from a
hipGraph, then the 12 Hz codec decode to 24 kHz PCM.  Default workload = BASELINE.json configs[1]:
0.6B, batch 1, sampled (temp 0.8 / top-k 50 / top-p 0.95), max-tokens 2048, 16-token prompt; weights
are seeded synthetic (no checkpoint in the image), EOS is suppressed so every utterance runs the full
2048 frames (random weights never learnt to stop).  The default line also carries the rest of the
metric ("0.6B @ b1/b64", north_star "batch 1/8/64") as sub-records run in the same process after the
timed region of `value`: "b64" = configs[2] (64 utterances in one batch x 256 frames, 5 timed steps),
"b8", "b64_f2048" (64 x 2048 frames: the length configs[1] and the reference default state), and the
same long workloads with the talker KV cache in bf16 ("b64_f2048_kv_bf16", "b1_f2048_kv_bf16"), each
with its own `roofline`; "b8_1p7b_clone" = configs[4] (1.7B dims, batch 8, the --ref voice-clone front end — wav -> resample ->
log-mel -> speaker encoder -> speaker row — inside the timed region); "capacity" = three 128-slot engines stepping concurrently
(labelled: not a BASELINE config).  `stages` carries per-stage device time incl. `prefill` (bytes and fraction of HBM).

Multi-GPU (configs[3]): utterances are independent, each rank runs its own batch on its own GPU
(weak scaling); RCCL carries only the final gather of codes + PCM lengths.  `--gpus N` with
WORLD_SIZE unset starts the N ranks itself: N child processes (rank i on GPU i, rendezvous on
127.0.0.1), spawned before the parent touches the GPU; under torch.distributed.run the ranks are
already there.  Either way world_size must equal --gpus.

Rank 0 prints ONE JSON line.  `roofline` is for the decode step (the hipGraph replayed per frame):
algorithmic bytes / device time measured with HIP events on the engine's stream.
`cpu_baseline` is the reference CLI on ONNX Runtime's CPU EP when an operator supplies it (probe below),
else the CPU oracle (a port of the reference's call pattern, fp32) timed on a bounded sample — on all allowed cores (capped at 16
OpenMP threads) and, `threads4`, on the 4 threads the reference pins ONNX Runtime to (tts_onnx.cpp:140).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in ("leaxer-qwen3-tts_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import numpy as np  # noqa: E402

FRAME_SECONDS = 0.08  # 12.5 Hz codec frames
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
IM_START, ASSISTANT, TTS_BOS, TTS_EOS, IM_END = 151644, 77091, 151672, 151673, 151645
HOOKS = False   # --hooks: engines honour the A/B environment knobs (measurement runs only)


def layer_bytes(Hw, nq, nkv, d, ffn):
    return 2.0 * (Hw * (nq + 2 * nkv) * d + Hw * nq * d + 3 * Hw * ffn)


def weight_step_bytes(cfg):
    """talker weights once + predictor weights once per pass (bf16)."""
    H = cfg.hidden
    Hc = cfg.cp_hidden or H        # 1.7B: narrower predictor behind cp.proj
    w = cfg.n_layers * layer_bytes(H, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.ffn) + 2.0 * H * cfg.vocab
    w += (cfg.n_groups - 1) * (cfg.cp_layers * layer_bytes(Hc, cfg.cp_heads, cfg.cp_kv_heads, cfg.cp_head_dim, cfg.cp_ffn)
                               + 2.0 * Hc * cfg.sub_vocab + (2.0 * H * Hc if Hc != H else 0.0))
    return w


def kv_step_bytes(cfg, batch, avg_ctx, bytes_per_elem):
    return batch * avg_ctx * cfg.n_layers * 2.0 * cfg.n_kv_heads * cfg.head_dim * bytes_per_elem


def algorithmic_step_bytes(cfg, batch, avg_ctx):
    """weights + the bf16-equivalent KV bytes the attention must read (B x T x 2 x L x n_kv x d x 2 B): the roofline numerator"""
    return weight_step_bytes(cfg) + kv_step_bytes(cfg, batch, avg_ctx, 2.0)


CODEC_GFLOP_PER_FRAME = 5.0     # SURVEY.md section 8d / DESIGN.md section 4
MFMA_16BIT_DENSE_TFLOPS = 2500.0


def kernel_sources_digest():
    """sha1 over the HIP sources: a PMC traffic figure under profiles/ is only quoted for the build it was measured on"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "leaxer-qwen3-tts_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(batch, model, ctx, kv_bf16):
    """HBM bytes per decode step from the rocprofv3 --pmc passes (tools/pmc_traffic.py writes profiles/decode_step_traffic.json with the
    digest of the kernel sources it ran, one entry per (batch, talker context, KV dtype)): quoted only when it was measured on THIS build
    at THIS batch and KV dtype and at a context within 15 % (or 16 tokens) of the record's own mean context — the KV stream is
    B x context x 229 KB (fp32) per step, so a short-context figure says nothing about a long-context record — else null with the reason."""
    tf = os.path.join(ROOT, "profiles", "decode_step_traffic.json")
    pre = "" if model == "0.6b" else f"m{model}_"     # tools/pmc_traffic.py --model 1.7b: keys "m1.7b_b8_ctx136"
    if not os.path.exists(tf):
        return None, "profiles/decode_step_traffic.json absent (run tools/pmc_traffic.py under gpurun)"
    try:
        j = json.load(open(tf))
    except Exception as ex:
        return None, f"unreadable traffic file: {ex}"
    if j.get("src_digest") != kernel_sources_digest():
        return None, "PMC passes under profiles/ were taken on a different build of the kernels (src_digest mismatch); re-run tools/pmc_traffic.py"
    best = None
    for key, det in j.items():
        if not key.endswith("_detail") or not key.startswith(f"{pre}b{batch}") or det.get("model", "0.6b") != model:
            continue
        base = key[: -len("_detail")]
        if base[len(pre):].split("_")[0] != f"b{batch}" or det.get("kv", "fp32") != ("bf16" if kv_bf16 else "fp32"):
            continue
        c = det.get("ctx", 0) or 14          # context 0 = the first steps after an 8-row prompt: contexts 10-20
        if abs(c - ctx) <= max(16.0, 0.15 * ctx) and (best is None or abs(c - ctx) < abs(best[1] - ctx)):
            best = (j[base], c)
    if best is None:
        return None, (f"no PMC pass at batch {batch}, context ~{ctx:.0f}, {'bf16' if kv_bf16 else 'fp32'} KV (rocprofv3 --pmc runs separately from the bench: "
                      "counters cannot be read inside the timed process; tools/run_round_profile.sh <tag> pmc)")
    return best[0], (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over eager decode steps of this build at context {best[1]} "
                     "(tools/pmc_traffic.py; slots moved there with q3tts_measure_skip_frames); FETCH_SIZE x2 (gfx950)")


def stage_report(eng, cfg, toks, sp, B, F, ctr, kv_bf16=False):
    """north_star: achieved fraction of the HBM / MFMA roofline per stage.  Arms the B slots again, advances them to mid-utterance with
    graph replays, then runs 16 EAGER steps with HIP events at the stage boundaries (q3tts_stage_profile); eager launches carry a
    little more launch gap than the graph replay the headline times.  Codec numbers come from the timed region's counters."""
    H, Hc = cfg.hidden, cfg.cp_hidden or cfg.hidden
    for b in range(B):
        eng.slot_release(b)
    t0 = time.perf_counter()
    for b in range(B):
        p, tr = eng.build_prompt(toks[b], 0)
        eng.slot_begin(b, p, tr, sp, seed=5, stream_id=b, ignore_eos=True)
    prefill_ms = (time.perf_counter() - t0) * 1e3 / B
    mid = max(1, min(F // 2, F - 20))
    eng.decode_steps(mid)
    st = eng.stage_profile(16)
    ctx = 9 + mid + 8
    talker_w = cfg.n_layers * layer_bytes(H, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.ffn) + 2.0 * H * cfg.vocab
    talker_bytes = talker_w + kv_step_bytes(cfg, B, ctx, 2.0)
    pred_bytes = (cfg.n_groups - 1) * (cfg.cp_layers * layer_bytes(Hc, cfg.cp_heads, cfg.cp_kv_heads, cfg.cp_head_dim, cfg.cp_ffn)
                                       + 2.0 * Hc * cfg.sub_vocab + (2.0 * H * Hc if Hc != H else 0.0))
    for b in range(B):
        eng.slot_release(b)
    # run_prefill (tts_onnx.cpp:615-665): B prompts x 8 rows (Auto language: S = 8) as one batched pass; the weights cross HBM once per
    # group of up to 128 rows, SURVEY.md 8d budgets them once (887 MB at 0.6B) + the prompt's KV rows written
    try:
        prefill_dev_ms = eng.prefill_profile(B, 8, reps=4)
    except Exception:
        prefill_dev_ms = None
    prefill_groups = -(-B * 8 // 128)

    def hbm(ms, nbytes):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"ms_per_step": round(ms, 4), "algorithmic_bytes": int(nbytes), "GB/s": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}
    codec_ms = ctr["codec_ms"] / max(ctr["codec_frames"], 1)
    tf = CODEC_GFLOP_PER_FRAME * 1e9 / (codec_ms * 1e-3) / 1e12 if codec_ms > 0 else 0.0
    two, three = eng.codec_plane_stats()
    return {
        "talker_decode": dict(hbm(st["talker_decode_ms"], talker_bytes), context=ctx, kv_dtype="bf16" if kv_bf16 else "fp32",
                              kv_bytes_actual=int(kv_step_bytes(cfg, B, ctx, 2.0 if kv_bf16 else 4.0))),
        "code_predictor": hbm(st["code_predictor_ms"], pred_bytes),
        "sampler": {"ms_per_step": round(st["sampler_ms"], 4), "launches_per_step": cfg.n_groups, "bound": "latency"},
        "codec_decode": {"ms_per_frame": round(codec_ms, 5), "GFLOP_per_frame": CODEC_GFLOP_PER_FRAME, "TFLOP/s": round(tf, 1),
                         "frac_mfma_16bit_dense": round(tf / MFMA_16BIT_DENSE_TFLOPS, 4),
                         "weight_tensors_2_products": two, "weight_tensors_3_products": three,
                         "note": "fp32-grade products from fp16 (hi, lo) split operands: a weight tensor that is exact in fp16 (every bf16-origin tensor, "
                                 "incl. these synthetic weights) takes 2 matrix-core products per fp32 product (exact), any other tensor 3; "
                                 "matrix-core issue is that multiple of the algorithmic rate"},
        "prompt_and_prefill": {"ms_per_utterance_wall": round(prefill_ms, 3), "note": "host prompt assembly (text_project calls) + talker prefill, one slot at a time"},
        "prefill": (dict(hbm(prefill_dev_ms, talker_w + kv_step_bytes(cfg, B, 8, 4.0)), rows=B * 8, weight_passes=prefill_groups,
                         bytes_streamed=int(prefill_groups * talker_w),
                         note="device time of one batched prefill pass over B x 8 prompt rows already in HBM (q3tts_prefill_profile, HIP events on the "
                              "engine's stream; eager launches: prefill is not graph-captured); algorithmic bytes = the talker's weights once + the "
                              "prompt's fp32 KV rows; rows beyond 128 walk the weights again per 128-row group (bytes_streamed)")
                    if prefill_dev_ms else {"error": "prefill_profile failed"}),
        "eager_step_ms": round(st["step_ms"], 4),
    }


def roofline_record(cfg, B, F, step_ms, model, kv_bf16=False):
    abytes = algorithmic_step_bytes(cfg, B, 8 + F / 2.0)
    achieved = abytes / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
    traffic, note = measured_traffic(B, model, 8 + F / 2.0, kv_bf16)
    return {"bound": "hbm", "kernel": "decode step = one hipGraph replay per frame ("
            + ("q3::k_gemv1 (QKV / o_proj / gate-up / down / heads), k_cp_attn_oproj x75, k_attn x28, k_sample x16" if B <= 2 else
               ("q3::k_gemv16 family" if B <= 16 else "q3::k_gemm3 (split-K slabs reduced in-launch: seam) + k_attn / k_attn_tiny + k_sample")) + ")",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": note,
            "algorithmic_bytes_per_launch": int(abytes), "launch_ms": round(step_ms, 4),
            "kv_bytes_bf16_equivalent": int(kv_step_bytes(cfg, B, 8 + F / 2.0, 2.0)),
            "kv_bytes_actual": int(kv_step_bytes(cfg, B, 8 + F / 2.0, 2.0 if kv_bf16 else 4.0)),
            "kv_dtype": "bf16" if kv_bf16 else "fp32",
            "kv_note": ("bf16 KV cache (Q3TTS_FLAG_KV_BF16): K / V rounded to bf16 on append, fp32 attention math; the oracle rounds at the same point. "
                        "NOT bit-exact against the oracle: logits within 4e-3, ids equal up to the first decision whose top-2 gap is under the 2e-2 bound "
                        "(a floor of 8 bit-exact frames is asserted); the 16-bit storage path is bit-exact against fp32 storage of the same rounded rows "
                        "(tests/test_gpu_full.py, tests/test_gpu_b64.py)") if kv_bf16 else
                       ("the default KV cache is fp32 (bit-exact code parity with the fp32 oracle): the hardware moves kv_bytes_actual per step, "
                        "the numerator counts the bf16-equivalent SURVEY.md 8d budgets, so frac under-reports the bytes moved")}


# ------------------------------------------------------------------------------------------------
# CPU baseline: the unmodified reference on ONNX Runtime's CPU EP when an operator supplies it, else the oracle port
# ------------------------------------------------------------------------------------------------
def probe_reference_ort():
    """the intended baseline is the reference CLI on ONNX Runtime CPU EP (4 intra-op threads, hard-coded at
    /root/reference/src/tts_onnx.cpp:140-141).  Neither ORT nor the .onnx graphs exist in the image, so this looks for operator-supplied
    pieces: Q3TTS_REF_CLI (or `leaxer-qwen3-tts` on PATH), Q3TTS_REF_MODEL_DIR holding talker_decode.onnx, and a loadable
    libonnxruntime (ONNXRUNTIME_DIR/lib or the loader path).  Returns (dict of found paths | None, reason)."""
    import ctypes.util
    cli = os.environ.get("Q3TTS_REF_CLI") or shutil.which("leaxer-qwen3-tts") or shutil.which("leaxer-tts-onnx")
    mdir = os.environ.get("Q3TTS_REF_MODEL_DIR")
    ort = None
    for cand in ([os.path.join(os.environ["ONNXRUNTIME_DIR"], "lib", "libonnxruntime.so")] if os.environ.get("ONNXRUNTIME_DIR") else []):
        if os.path.exists(cand):
            ort = cand
    if ort is None:
        ort = ctypes.util.find_library("onnxruntime")
    missing = []
    if not cli or not os.path.exists(cli):
        missing.append("reference CLI (Q3TTS_REF_CLI)")
    if not mdir or not os.path.exists(os.path.join(mdir, "talker_decode.onnx")):
        missing.append("talker_decode.onnx (Q3TTS_REF_MODEL_DIR)")
    if not ort:
        missing.append("libonnxruntime (ONNXRUNTIME_DIR)")
    if missing:
        return None, "reference ORT-CPU baseline unavailable: missing " + ", ".join(missing)
    return {"cli": cli, "model_dir": mdir, "ort": ort}, "found"


def reference_ort_baseline(found, frames):
    """Run the reference CLI once (greedy = --top-k 1; the prompt is its own tokenizer's business) and time it."""
    import tempfile
    import wave
    out = os.path.join(tempfile.mkdtemp(prefix="q3ref_"), "ref.wav")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.dirname(found["ort"]) + ":" + env.get("LD_LIBRARY_PATH", "")
    text = "The quick brown fox jumps over the lazy dog near the quiet river bank today."
    cmd = [found["cli"], "-m", found["model_dir"], "-p", text, "-o", out, "--top-k", "1", "--max-tokens", str(frames)]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
    dt = time.perf_counter() - t0
    if r.returncode != 0 or not os.path.exists(out):
        raise RuntimeError("reference CLI failed: " + (r.stderr or r.stdout)[-300:])
    with wave.open(out, "rb") as w:
        seconds = w.getnframes() / float(w.getframerate())
    return {"value": round(seconds / dt, 5), "unit": "x real-time (audio s / wall s)", "cores": 4, "threads": 4, "nproc": os.cpu_count(),
            "kind": "reference", "frames_per_s": round(seconds / FRAME_SECONDS / dt, 3),
            "sample": f"unmodified reference CLI on ONNX Runtime CPU EP (4 intra-op threads, tts_onnx.cpp:140), --top-k 1, --max-tokens {frames}: "
                      f"{seconds:.2f} s of audio in {dt:.1f} s wall (includes model load)"}


def cpu_baseline(eng, cfg, ids, sp_kwargs, frames):
    """Times the CPU oracle (oracle/, fp32, OpenMP) on a bounded sample of the same workload: prompt
    assembly + prefill + `frames` frames in the REFERENCE's call pattern (predictor re-run without a
    KV cache, tts_onnx.cpp:862-868) + vocoder of those frames.  The oracle is only the checker /
    baseline here; nothing measured as `value` touches it."""
    found, why = probe_reference_ort()
    if found is not None:
        try:
            return reference_ort_baseline(found, frames)
        except Exception as ex:
            why = f"reference ORT-CPU baseline found but failed ({ex}); falling back to the port"
    import q3_oracle as qo
    ocfg = qo.Config.from_dict(cfg.to_dict())
    orc = qo.Oracle(ocfg, max_ctx=frames + 32)
    for name, shape in eng.tensor_infos():
        orc.set_tensor(name, eng.get_tensor(name, shape))
    threads = orc.threads
    sp = qo.Sampling(max_new_tokens=frames, **sp_kwargs)
    t0 = time.perf_counter()
    prompt = orc.build_prompt(ids, 0)
    codes = orc.generate(prompt, sp, seed=3, stream=0, cp_cached=False, ignore_eos=True)
    pcm = orc.vocoder(codes)
    dt = time.perf_counter() - t0
    # the reference pins ONNX Runtime to 4 intra-op threads (tts_onnx.cpp:140); the same port on 4 threads, on a
    # shorter sample (a quarter of the frames: the run stays bounded)
    four = None
    try:
        f4 = max(8, frames // 4)
        orc.L.q3o_set_threads(4)
        sp4 = qo.Sampling(max_new_tokens=f4, **sp_kwargs)
        t4 = time.perf_counter()
        codes4 = orc.generate(orc.build_prompt(ids, 0), sp4, seed=3, stream=0, cp_cached=False, ignore_eos=True)
        orc.vocoder(codes4)
        d4 = time.perf_counter() - t4
        four = {"value": round(len(codes4) * FRAME_SECONDS / d4, 5), "unit": "x real-time (audio s / wall s)", "cores": 4, "threads": 4, "kind": "port",
                "frames_per_s": round(len(codes4) / d4, 3),
                "sample": f"the same port pinned to 4 OpenMP threads (the reference's SetIntraOpNumThreads(4), tts_onnx.cpp:140): prefill + {len(codes4)} frames + vocoder, {d4:.1f} s wall"}
    except Exception as ex:
        four = {"error": str(ex)}
    orc.close()
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count()
    return {"value": round(len(codes) * FRAME_SECONDS / dt, 5), "unit": "x real-time (audio s / wall s)", "cores": threads,
            "threads": threads, "nproc": os.cpu_count(), "cpus_allowed": affinity,
            "kind": "port", "frames_per_s": round(len(codes) / dt, 3), "threads4": four,
            "sample": f"1 utterance, 16-token prompt, prefill + {len(codes)} frames (reference call pattern, no predictor KV cache) "
                      f"+ vocoder of {len(codes)} frames ({len(pcm)} samples), fp32 oracle on {threads} OpenMP threads "
                      f"(the box reports {os.cpu_count()} logical CPUs, {affinity} allowed), {dt:.1f} s wall; {why}"}


# ------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` starts its own N ranks
# ------------------------------------------------------------------------------------------------
def self_launch(args, argv):
    """Spawn N ranks of this script (rank i -> GPU i) BEFORE this process touches the GPU (it never does: no exec, no HIP call here),
    wait for all of them, pass rank 0's JSON line through, exit non-zero if any rank failed."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   Q3TTS_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # poll every child: if any rank dies (bad GPU index, import error) before or inside the rendezvous, the others would sit in
    # init_process_group / a barrier until the store or NCCL timeout — terminate them instead and fail at once
    import threading
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[i] = p.wait()
            break
        time.sleep(0.2)
    rd.join(timeout=10)
    out0 = buf[0] if buf else ""
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        sys.exit(1)
    sys.exit(0)


def run_workload(q3tts, cfg, local_rank, B, F, sp_kwargs, steps, warmup, rank, no_graph, dist=None, torch=None, world=1, backend=None, kv_bf16=False,
                 warm_frames=0):
    """W untimed warmup steps, then exactly K timed steps bracketed by barrier + synchronize; returns the engine and the raw measurements."""
    eng = q3tts.Engine(cfg, device=local_rank, max_batch=B, max_ctx=F + 32,
                       flags=(q3tts.FLAG_NO_GRAPH if no_graph else 0) | (q3tts.FLAG_KV_BF16 if kv_bf16 else 0) | (q3tts.FLAG_TEST_HOOKS if HOOKS else 0))
    eng.fill_synthetic(seed=0)
    sp = q3tts.Sampling(max_new_tokens=F, **sp_kwargs)
    rng = np.random.default_rng(1 + rank)
    toks = [np.array([IM_START, ASSISTANT, TTS_BOS] + list(rng.integers(0, 151643, 16)) + [TTS_EOS, IM_END], np.int64) for _ in range(B)]
    gathered = {"utterances": 0, "pcm_samples": 0}

    def barrier():
        if dist is not None:
            dist.barrier()
            if torch is not None and backend == "nccl":
                torch.cuda.synchronize()

    def step(i):
        pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=100 + i, ignore_eos=True, want_codes=True)
        if dist is not None:  # the only exchange on the path: gather of the generated codes + PCM lengths (RCCL over xGMI)
            import q3dist
            dev = torch.device("cuda", local_rank) if backend == "nccl" else None
            allc, alln = q3dist.gather_codes(dist, codes, [rank * B + u for u in range(B)], world * B, F, cfg.n_groups, device=dev,
                                             pcm_lens=[len(p) for p in pcm])
            gathered["utterances"] = sum(c is not None for c in allc)
            gathered["pcm_samples"] = int(sum(alln))
        return int(nfr.sum()), sum(len(p) for p in pcm)

    if warm_frames and warm_frames < F:   # the long sub-records warm up (graph capture, lazy allocations, caches) on a short utterance of the same batch
        sp_full, sp = sp, q3tts.Sampling(max_new_tokens=warm_frames, **sp_kwargs)
        for i in range(warmup):
            step(-1 - i)
        sp = sp_full
    else:
        for i in range(warmup):
            step(-1 - i)
    eng.counters(reset=True)
    barrier()
    t0 = time.perf_counter()
    frames = samples = 0
    for i in range(steps):
        f, s = step(i)
        frames += f
        samples += s
    barrier()
    dt = time.perf_counter() - t0
    return eng, toks, sp, dt, frames, samples, eng.counters(), gathered


def write_sweep_wav(path, seconds, f0, f1, sr=24000):
    """SURVEY.md 8d's clone reference: a 24 kHz 16-bit mono sine sweep"""
    import wave
    t = np.arange(int(seconds * sr)) / sr
    x = (0.5 * np.sin(2 * np.pi * (f0 * t + (f1 - f0) * t * t / (2 * seconds))) * 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(x.tobytes())


def clone_record(q3tts, local_rank, sp_kwargs, sampling_txt, B=8, F=256, steps=3, warmup=1):
    """BASELINE.json configs[4]: Qwen3-TTS-1.7B dims, --ref voice-clone path, batch 8.  A timed step = for each of the 8 utterances
    the reference's synthesize_clone front end — read_wav -> resample to 24 kHz -> 128-bin log-mel -> speaker encoder (ECAPA-TDNN on
    the GPU) -> [hidden] embedding (tts_onnx.cpp:264-318, 331-403) — and then the batch: prompt assembly with the speaker row spliced
    before CODEC_BOS (:481-498), prefill, F frames, vocoder.  Everything inside the timed region."""
    import tempfile
    cfg = q3tts.default_config("1.7b")
    eng = q3tts.Engine(cfg, device=local_rank, max_batch=B, max_ctx=F + 40, flags=q3tts.FLAG_TEST_HOOKS if HOOKS else 0)
    eng.fill_synthetic(seed=0)
    td = tempfile.mkdtemp(prefix="q3clone_")
    wavs = []
    for u in range(B):                      # 3 s references at 16 kHz: the resampler to 24 kHz is on the path
        wavs.append(os.path.join(td, "ref%d.wav" % u))
        write_sweep_wav(wavs[-1], 3.0, 80.0 + 10 * u, 3000.0 + 200 * u, sr=16000)
    rng = np.random.default_rng(17)
    toks = [np.array([IM_START, ASSISTANT, TTS_BOS] + list(rng.integers(0, 151643, 16)) + [TTS_EOS, IM_END], np.int64) for _ in range(B)]
    sp = q3tts.Sampling(max_new_tokens=F, **sp_kwargs)
    front = []

    def step(i):
        t0 = time.perf_counter()
        spks = [eng.extract_speaker_embedding(w) for w in wavs]
        front.append(time.perf_counter() - t0)
        pcm, _, nfr = eng.synthesize_batch(toks, sp, lang=1, seed=200 + i, ignore_eos=True, want_codes=False, speakers=spks)
        return int(nfr.sum()), sum(len(p) for p in pcm)

    for i in range(warmup):
        step(-1 - i)
    eng.counters(reset=True)
    del front[:]
    t0 = time.perf_counter()
    frames = samples = 0
    for i in range(steps):
        f, s = step(i)
        frames += f
        samples += s
    dt = time.perf_counter() - t0
    ctr = eng.counters()
    sm = ctr["decode_ms"] / max(ctr["decode_steps"], 1)
    rec = {"config": {"workload": f"Qwen3-TTS-1.7B dims, --ref voice-clone path, batch={B}/GPU, 16-token prompt, {sampling_txt}, max-tokens={F} (EOS suppressed), "
                                  "3 s 16 kHz reference wav per utterance, hipGraph decode loop, synthetic seeded weights",
                      "batch_per_gpu": B, "frames_per_utterance": F},
           "value": round(frames * FRAME_SECONDS / dt, 3), "unit": "x real-time (audio s / wall s)", "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 3), "codec_frames_per_s": round(frames / dt, 2), "pcm_samples": samples,
           "clone_front_end_ms_per_utterance": round(sum(front) / max(len(front), 1) / B * 1e3, 3),
           "clone_front_end_note": "wav read + resample 16 -> 24 kHz + log-mel on the host, speaker encoder on the GPU; inside the timed region",
           "decode_ms_per_frame_step": round(sm, 4),
           "codec_decode_ms_per_frame": round(ctr["codec_ms"] / max(ctr["codec_frames"], 1), 5),
           "roofline": roofline_record(cfg, B, F, sm, "1.7b", False)}
    eng.close()
    shutil.rmtree(td, ignore_errors=True)
    return rec


def capacity_record(q3tts, cfg, local_rank, sp_kwargs, n_eng=3, B=128, F=96):
    """NOT a BASELINE config (the reference is batch 1; configs[2] is ONE 64-slot engine): what one MI355X sustains when the latency-bound
    decode chains of several engines overlap — n_eng engines x B slots stepping concurrently from n_eng host threads (one hipGraph replay
    per step and engine).  Decode chain only: the vocoder (15 us per frame) would add ~20 %."""
    import threading
    rng = np.random.default_rng(1)
    engs = []
    try:
        for g in range(n_eng):
            e = q3tts.Engine(cfg, device=local_rank, max_batch=B, max_ctx=F + 32)
            e.fill_synthetic(seed=0)
            engs.append(e)
            sp = q3tts.Sampling(max_new_tokens=F, **sp_kwargs)
            ids = np.array([IM_START, ASSISTANT, TTS_BOS] + list(rng.integers(0, 151643, 16)) + [TTS_EOS, IM_END], np.int64)
            p, tr = e.build_prompt(ids, 0)
            for b in range(B):
                e.slot_begin(b, p, tr, sp, seed=5, stream_id=g * B + b, ignore_eos=True)
            e.decode_steps(4)           # graph captured, caches warm
        n = F - 8
        bar = threading.Barrier(n_eng + 1)

        def run(e):
            bar.wait()
            e.decode_steps(n)
            bar.wait()
        th = [threading.Thread(target=run, args=(e,)) for e in engs]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        for t in th:
            t.join()
        dev = [e.last_decode_ms()[0] / n for e in engs]
    finally:
        for e in engs:
            e.close()
    utt = n_eng * B
    return {"config": {"workload": f"{n_eng} engines x {B} slots on one GPU stepping concurrently ({utt} utterances in flight), {n} decode steps each, "
                                   "decode chain only (no vocoder, no prompt assembly)", "engines": n_eng, "slots_per_engine": B},
            "label": "capacity (not a BASELINE config; the headline values are the single-engine records)",
            "wall_ms_per_step_of_all": round(dt * 1e3 / n, 4), "per_engine_device_ms_per_step": [round(d, 4) for d in dev],
            "codec_frames_per_s": round(utt * n / dt, 1), "value": round(utt * n * FRAME_SECONDS / dt, 1), "unit": "x real-time (audio s / wall s), decode chain only"}


def dry_launch(args, rank, world):
    """--dry-launch: the launch / rendezvous / gather plumbing of the N-rank bench on CPU (gloo), no GPU and no synthesis: every rank
    contributes fabricated ragged codes for its B utterances at the REAL payload shape ([B][--frames][16] int32 per rank: 8.4 MB at
    configs[3]'s 64 x 2048) and checks what the gather returns; the ranks also partition a seeded list of world x B text lengths with
    q3dist.shard_utterances (what a job with one global utterance list does) and report the shards' sizes and loads.
    tests/test_dist_cpu.py runs it with --gpus 2 and, configs[3]'s shape, --gpus 8 --batch 64."""
    import torch
    import torch.distributed as dist
    import q3dist
    assert world == args.gpus, f"world_size {world} != --gpus {args.gpus}"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, F, G = args.batch, args.frames, 16
    nfr = lambda g: 1 + (g * 37) % F                      # ragged lengths in [1, F]
    codes = [np.full((nfr(rank * B + u), G), (rank * B + u) % 2048, np.int64) for u in range(B)]
    lens = [1920 * len(c) - 555 for c in codes]
    dist.barrier()
    t0 = time.perf_counter()
    allc, alln = q3dist.gather_codes(dist, codes, [rank * B + u for u in range(B)], world * B, F, G, pcm_lens=lens)
    dist.barrier()
    dt = time.perf_counter() - t0
    ok = all(c is not None and len(c) == nfr(g) and (c == g % 2048).all() and alln[g] == 1920 * len(c) - 555 for g, c in enumerate(allc))
    text_lens = np.random.default_rng(7).integers(3, 200, world * B).tolist()
    mine = q3dist.shard_utterances(text_lens, world, rank)
    mine_t = torch.tensor([len(mine), int(sum(text_lens[i] for i in mine))], dtype=torch.int64)
    shards = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(shards, mine_t)
    covered = torch.zeros(world * B, dtype=torch.int64)
    covered[mine] = 1
    dist.all_reduce(covered)                              # a partition: every utterance on exactly one rank
    ok = ok and bool((covered == 1).all())
    if rank == 0:
        print(json.dumps({"metric": "dry launch (no GPU work)", "dry_launch": True, "n_gpus": world, "ranks_ok": bool(ok),
                          "gathered_utterances": len(allc), "gather_ms": round(dt * 1e3, 3), "backend": "gloo",
                          "payload_mb_per_rank": round(B * F * G * 4 / 1e6, 2), "shard_sizes": [int(s[0]) for s in shards],
                          "shard_loads": [int(s[1]) for s in shards], "longest_text": int(max(text_lens))}), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU per step (configs[1]: 1, configs[2]: 64)")
    ap.add_argument("--frames", type=int, default=2048, help="max-tokens per utterance")
    ap.add_argument("--greedy", action="store_true", help="top_k=1 instead of the sampled default")
    ap.add_argument("--model", default="0.6b", choices=["0.6b", "1.7b"], help="model dims (the headline is 0.6b; 1.7b = configs[4] dims)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-b64", action="store_true", help="skip the configs[2] sub-record (64 utterances x 256 frames) of the default line")
    ap.add_argument("--kv-bf16", action="store_true", help="talker KV cache in bf16 (Q3TTS_FLAG_KV_BF16) for the headline run; the default line's "
                                                            "b64_f2048 sub-record is reported in BOTH modes either way")
    ap.add_argument("--no-long", action="store_true", help="skip the b64_f2048 sub-record (64 utterances x 2048 frames, ~45 s)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (rocprofv3 kernel tracing "
                                                             "crashes inside hipGraphLaunch on this ROCm; same kernels either way)")
    ap.add_argument("--cpu-frames", type=int, default=160)
    ap.add_argument("--hooks", action="store_true", help="create the engines with Q3TTS_FLAG_TEST_HOOKS so that the A/B environment knobs "
                    "(Q3TTS_SEAM, Q3TTS_GEMM3_LA, ...) are honoured — A/B measurement runs only; the line then says so")
    ap.add_argument("--dry-launch", action="store_true", help="only the N-rank launch + gloo rendezvous + gather plumbing, on CPU")
    args = ap.parse_args()
    global HOOKS
    HOOKS = args.hooks

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])          # never returns

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch one rank per GPU (or let --gpus N start them)\n")
        sys.exit(2)
    if args.dry_launch:
        dry_launch(args, rank, world)

    dist = None
    torch = None
    backend = None
    if world > 1 or os.environ.get("Q3TTS_BENCH_FORCE_DIST") == "1":   # the env var rehearses the RCCL path with one rank
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = "nccl"
        torch.cuda.set_device(local_rank)
        import datetime
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),   # RCCL on ROCm
                                timeout=datetime.timedelta(seconds=300))

    import q3tts
    cfg = q3tts.default_config(args.model)
    MODEL = "Qwen3-TTS-" + args.model.upper()
    B, F = args.batch, args.frames
    sp_kwargs = dict(temperature=1.0, top_p=1.0, top_k=1) if args.greedy else dict(temperature=0.8, top_p=0.95, top_k=50)
    sampling_txt = "greedy top_k=1" if args.greedy else "sampled temp=0.8 top-k=50 top-p=0.95"
    eng, toks, sp, dt, frames, samples, ctr, gathered = run_workload(q3tts, cfg, local_rank, B, F, sp_kwargs, args.steps, args.warmup, rank,
                                                                     args.no_graph, dist, torch, world, backend, kv_bf16=args.kv_bf16)

    if dist is not None:
        v = torch.tensor([dt, float(frames), float(samples)], dtype=torch.float64, device="cuda")
        tmax = v.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        frames, samples = int(v[1]), int(v[2])

    first_audio = None
    if world == 1 and dist is None:
        # serving-side extra (not the headline): wall time from token ids to the first 2 s of PCM for one utterance —
        # prompt assembly + prefill + 25 frames + codec decode of those frames (a causal decoder: exactly the first
        # samples of the full utterance)
        sp25 = q3tts.Sampling(max_new_tokens=25, **sp_kwargs)
        for b in range(B):
            eng.slot_release(b)
        ts = []
        for i in range(3):
            t1 = time.perf_counter()
            eng.synthesize_batch(toks[:1], sp25, lang=0, seed=7 + i, ignore_eos=True, want_codes=False)
            ts.append((time.perf_counter() - t1) * 1e3)
        first_audio = round(min(ts), 2)

    stages = None
    if world == 1 and dist is None and not args.no_graph:
        try:
            stages = stage_report(eng, cfg, toks, sp, B, F, ctr, args.kv_bf16)
        except Exception as ex:   # a reported extra, never the measurement
            stages = {"error": str(ex)}

    if rank == 0:
        step_ms = ctr["decode_ms"] / max(ctr["decode_steps"], 1)
        out = {
            "metric": f"real-time factor (24 kHz audio sec / wall sec), {MODEL}",
            "value": round(frames * FRAME_SECONDS / dt, 3),
            "unit": "x real-time (audio s / wall s)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 weights, fp32 activations/accumulate (codec decoder: fp16 hi/lo split operands, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"{MODEL}, batch={B}/GPU, 16-token prompt, {sampling_txt}"
                                   + f", max-tokens={F} (EOS suppressed), synthetic seeded weights",
                       "batch_per_gpu": B, "frames_per_utterance": F, "parallelism": f"dp{world} (independent utterances)"},
            "codec_frames_per_s": round(frames / dt, 2),
            "pcm_samples": samples,
            "decode_ms_per_frame_step": round(step_ms, 4),
            "codec_decode_ms_per_frame": round(ctr["codec_ms"] / max(ctr["codec_frames"], 1), 5),
            "first_2s_audio_latency_ms": first_audio,
            "roofline": roofline_record(cfg, B, F, step_ms, args.model, args.kv_bf16),
            # round 5: kernels are compiled without packed fp32 instructions — with them results depended on what else the GPU was running
            # (profiles/r05_hunt/README.txt); costs 2-3 % of the b=1 step, nothing at batch 64 or in the codec
            "build": {"library": os.path.basename(q3tts.LIB_PATH),
                      "packed_fp32_instructions": "none in the default build (tests/test_kernel_resources.py checks the code objects)"},
        }
        if HOOKS:
            out["ab_knobs"] = {k: v for k, v in os.environ.items() if k.startswith("Q3TTS_")}   # an A/B run: which knobs were set
        if dist is not None:
            out["multi_gpu"] = {"backend": "nccl (RCCL)", "world_size": world, "gathered_utterances": gathered["utterances"],
                                "gathered_pcm_samples": gathered["pcm_samples"],
                                "scaling_note": "weak scaling over independent utterances; efficiency is the driver's to compute"
                                if world > 1 else "unmeasured: one rank rehearsing the RCCL path (Q3TTS_BENCH_FORCE_DIST=1)"}
        if stages is not None:
            out["stages"] = stages
        eng.close()
        eng = None
        if world == 1 and dist is None and B == 1 and args.model == "0.6b" and not args.no_b64 and not args.no_graph:
            # the rest of BASELINE.json's metric ("0.6B @ b1/b64", north_star: "batch 1/8/64"), same process, after the timed region of `value`:
            #   b64       configs[2], 64 utterances in one batch x 256 frames, 5 timed steps
            #   b64_f2048 the same batch at the length configs[1] and the reference default (tts_onnx.h:65) state: 64 x 2048 frames, 1 timed step
            #   b8        8 utterances x 256 frames
            def sub_record(B2, F2, steps2, warm2, with_stages, kvb=False):
                e2, toks2, sp2, dt2, fr2, smp2, ctr2, _ = run_workload(q3tts, cfg, local_rank, B2, F2, sp_kwargs, steps2, warm2, rank, False, kv_bf16=kvb,
                                                                       warm_frames=0)   # full-length warmup: a short one left the vocoder's arenas (~30 GB at 64 x 2048 frames) to be allocated inside the timed job — 610 vs 664x between runs of one build
                sm2 = ctr2["decode_ms"] / max(ctr2["decode_steps"], 1)
                rec = {"config": {"workload": f"{MODEL}, batch={B2}/GPU, 16-token prompt, {sampling_txt}, max-tokens={F2} (EOS suppressed), "
                                              "hipGraph decode loop, synthetic seeded weights", "batch_per_gpu": B2, "frames_per_utterance": F2},
                       "value": round(fr2 * FRAME_SECONDS / dt2, 3), "unit": "x real-time (audio s / wall s)", "steps": steps2, "warmup": warm2,
                       "warmup_frames": F2,
                       "ms_per_step": round(dt2 / steps2 * 1e3, 3), "codec_frames_per_s": round(fr2 / dt2, 2),
                       "decode_ms_per_frame_step": round(sm2, 4),
                       "codec_decode_ms_per_frame": round(ctr2["codec_ms"] / max(ctr2["codec_frames"], 1), 5),
                       "roofline": roofline_record(cfg, B2, F2, sm2, args.model, kvb)}
                if with_stages:
                    try:
                        rec["stages"] = stage_report(e2, cfg, toks2, sp2, B2, F2, ctr2, kvb)
                    except Exception as ex:
                        rec["stages"] = {"error": str(ex)}
                e2.close()
                return rec
            #   *_kv_bf16 the same workloads with the talker KV cache in bf16 (the mode SURVEY.md 8d's KV budget describes)
            for key, (B2, F2, steps2, warm2, with_stages, kvb) in (("b64", (64, 256, 5, 1, True, False)), ("b8", (8, 256, 5, 1, True, False)),
                                                                   ("b64_f2048", (64, 2048, 1, 1, False, False)),
                                                                   ("b64_f2048_kv_bf16", (64, 2048, 1, 1, False, True)),
                                                                   ("b1_f2048_kv_bf16", (1, 2048, 2, 1, False, True))):
                if key.startswith("b64_f2048") and args.no_long:
                    continue
                try:
                    out[key] = sub_record(B2, F2, steps2, warm2, with_stages, kvb)
                except Exception as ex:
                    out[key] = {"error": str(ex)}
        if world == 1 and dist is None and B == 1 and args.model == "0.6b" and not args.no_b64 and not args.no_graph:
            try:      # BASELINE configs[4]: 1.7B dims, --ref voice-clone path, batch 8
                out["b8_1p7b_clone"] = clone_record(q3tts, local_rank, sp_kwargs, sampling_txt)
            except Exception as ex:
                out["b8_1p7b_clone"] = {"error": str(ex)}
            if not args.no_long:
                try:
                    out["capacity"] = capacity_record(q3tts, cfg, local_rank, sp_kwargs)
                except Exception as ex:
                    out["capacity"] = {"error": str(ex)}
        if world == 1 and dist is None and not args.no_cpu_baseline:
            try:
                e3 = q3tts.Engine(cfg, device=local_rank, max_batch=1, max_ctx=64)   # weights only: the oracle copies the same seeded tensors
                e3.fill_synthetic(seed=0)
                out["cpu_baseline"] = cpu_baseline(e3, cfg, toks[0], sp_kwargs, args.cpu_frames)
                e3.close()
            except Exception as ex:  # the baseline is a reported extra, never the measurement
                out["cpu_baseline"] = {"value": None, "unit": "x real-time (audio s / wall s)", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {ex}"}
        print(json.dumps(out), flush=True)
    if eng is not None:
        eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
