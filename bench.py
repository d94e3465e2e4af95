"""bench.py — headline metric of BASELINE.json: real-time factor (24 kHz audio seconds / wall seconds)
and codec-frames/sec of Qwen3-TTS-0.6B synthesis on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--frames F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the whole hot path over one batch of B utterances per GPU: prompt assembly,
talker prefill, F frames of [sampler + 15 code-predictor passes + talker decode] replayed from a
hipGraph, then the 12 Hz codec decode to 24 kHz PCM.  Default workload = BASELINE.json configs[1]:
0.6B, batch 1, sampled (temp 0.8 / top-k 50 / top-p 0.95), max-tokens 2048, 16-token prompt; weights
are seeded synthetic (no checkpoint in the image), EOS is suppressed so every utterance runs the full
2048 frames (random weights never learnt to stop).  Multi-GPU: utterances are independent, each rank
runs its own batch on its own GPU (weak scaling); RCCL carries only the final gather of codes.

Rank 0 prints ONE JSON line.  `roofline` is for the decode step (the hipGraph replayed per frame):
algorithmic bytes / device time measured with HIP events on the engine's stream.
`cpu_baseline` times the CPU oracle (a port of the reference's call pattern, fp32) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in ("leaxer-qwen3-tts_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import numpy as np  # noqa: E402

FRAME_SECONDS = 0.08  # 12.5 Hz codec frames
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_step_bytes(cfg, batch, avg_ctx):
    """talker weights once + predictor weights once per pass (bf16) + bf16-equivalent
    KV bytes the attention must read (B x T x 2 x L x n_kv x d x 2 B)."""
    H = cfg.hidden
    Hc = cfg.cp_hidden or H        # 1.7B: narrower predictor behind cp.proj

    def layer(Hw, nq, nkv, d, ffn):
        return 2.0 * (Hw * (nq + 2 * nkv) * d + Hw * nq * d + 3 * Hw * ffn)
    w = cfg.n_layers * layer(H, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.ffn) + 2.0 * H * cfg.vocab
    w += (cfg.n_groups - 1) * (cfg.cp_layers * layer(Hc, cfg.cp_heads, cfg.cp_kv_heads, cfg.cp_head_dim, cfg.cp_ffn)
                               + 2.0 * Hc * cfg.sub_vocab + (2.0 * H * Hc if Hc != H else 0.0))
    kv = batch * avg_ctx * cfg.n_layers * 2.0 * cfg.n_kv_heads * cfg.head_dim * 2.0
    return w + kv


CODEC_GFLOP_PER_FRAME = 5.0     # SURVEY.md section 8d / DESIGN.md section 4
MFMA_16BIT_DENSE_TFLOPS = 2500.0


def stage_report(eng, q3tts, cfg, toks, sp, B, F, ctr):
    """north_star: achieved fraction of the HBM / MFMA roofline per stage.  Arms the B slots again, advances them to mid-utterance with
    graph replays, then runs 16 EAGER steps with HIP events at the stage boundaries (q3tts_stage_profile); eager launches carry a
    little more launch gap than the graph replay the headline times.  Codec numbers come from the timed region's counters."""
    H, Hc = cfg.hidden, cfg.cp_hidden or cfg.hidden

    def layer(Hw, nq, nkv, d, ffn):
        return 2.0 * (Hw * (nq + 2 * nkv) * d + Hw * nq * d + 3 * Hw * ffn)
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
    talker_bytes = cfg.n_layers * layer(H, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.ffn) + 2.0 * H * cfg.vocab \
        + B * ctx * cfg.n_layers * 2.0 * cfg.n_kv_heads * cfg.head_dim * 2.0
    pred_bytes = (cfg.n_groups - 1) * (cfg.cp_layers * layer(Hc, cfg.cp_heads, cfg.cp_kv_heads, cfg.cp_head_dim, cfg.cp_ffn)
                                       + 2.0 * Hc * cfg.sub_vocab + (2.0 * H * Hc if Hc != H else 0.0))
    for b in range(B):
        eng.slot_release(b)

    def hbm(ms, nbytes):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"ms_per_step": round(ms, 4), "algorithmic_bytes": int(nbytes), "GB/s": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}
    codec_ms = ctr["codec_ms"] / max(ctr["codec_frames"], 1)
    tf = CODEC_GFLOP_PER_FRAME * 1e9 / (codec_ms * 1e-3) / 1e12 if codec_ms > 0 else 0.0
    return {
        "talker_decode": dict(hbm(st["talker_decode_ms"], talker_bytes), context=ctx),
        "code_predictor": hbm(st["code_predictor_ms"], pred_bytes),
        "sampler": {"ms_per_step": round(st["sampler_ms"], 4), "launches_per_step": cfg.n_groups, "bound": "latency"},
        "codec_decode": {"ms_per_frame": round(codec_ms, 5), "GFLOP_per_frame": CODEC_GFLOP_PER_FRAME, "TFLOP/s": round(tf, 1),
                         "frac_mfma_16bit_dense": round(tf / MFMA_16BIT_DENSE_TFLOPS, 4),
                         "note": "fp32 products as 3 fp16 MFMA passes: matrix-core issue is 3x the algorithmic rate"},
        "prompt_and_prefill": {"ms_per_utterance_wall": round(prefill_ms, 3), "note": "host prompt assembly (text_project calls) + talker prefill"},
        "eager_step_ms": round(st["step_ms"], 4),
    }


def cpu_baseline(eng, cfg, ids, sp_kwargs, frames):
    """Times the CPU oracle (oracle/, fp32, OpenMP) on a bounded sample of the same workload: prompt
    assembly + prefill + `frames` frames in the REFERENCE's call pattern (predictor re-run without a
    KV cache, tts_onnx.cpp:862-868) + vocoder of those frames.  The oracle is only the checker /
    baseline here; nothing measured as `value` touches it."""
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
    orc.close()
    return {"value": round(len(codes) * FRAME_SECONDS / dt, 5), "unit": "x real-time (audio s / wall s)", "cores": threads,
            "kind": "port", "frames_per_s": round(len(codes) / dt, 3),
            "sample": f"1 utterance, 16-token prompt, prefill + {len(codes)} frames (reference call pattern, no predictor KV cache) "
                      f"+ vocoder of {len(codes)} frames ({len(pcm)} samples), fp32 oracle, {dt:.1f} s wall; "
                      "reference ORT-CPU baseline unavailable (no onnxruntime / models in image)"}


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
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (rocprofv3 kernel tracing "
                                                             "crashes inside hipGraphLaunch on this ROCm; same kernels either way)")
    ap.add_argument("--cpu-frames", type=int, default=160)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    if world > 1 or os.environ.get("Q3TTS_BENCH_FORCE_DIST") == "1":   # the env var rehearses the RCCL path with one rank
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # RCCL on ROCm

    import q3tts
    cfg = q3tts.default_config(args.model)
    MODEL = "Qwen3-TTS-" + args.model.upper()
    B, F = args.batch, args.frames
    eng = q3tts.Engine(cfg, device=local_rank, max_batch=B, max_ctx=F + 32, flags=q3tts.FLAG_NO_GRAPH if args.no_graph else 0)
    eng.fill_synthetic(seed=0)
    sp_kwargs = dict(temperature=1.0, top_p=1.0, top_k=1) if args.greedy else dict(temperature=0.8, top_p=0.95, top_k=50)
    sp = q3tts.Sampling(max_new_tokens=F, **sp_kwargs)
    rng = np.random.default_rng(1 + rank)
    IM_START, ASSISTANT, TTS_BOS, TTS_EOS, IM_END = 151644, 77091, 151672, 151673, 151645
    toks = [np.array([IM_START, ASSISTANT, TTS_BOS] + list(rng.integers(0, 151643, 16)) + [TTS_EOS, IM_END], np.int64)
            for _ in range(B)]

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def step(i):
        pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=100 + i, ignore_eos=True, want_codes=True)
        if dist is not None:  # the only exchange on the path: gather of the generated codes (RCCL over xGMI)
            import q3dist
            q3dist.gather_codes(dist, codes, [rank * B + u for u in range(B)], world * B, F, cfg.n_groups,
                                device=torch.device("cuda", local_rank))
        return int(nfr.sum()), sum(len(p) for p in pcm)

    for i in range(args.warmup):
        step(-1 - i)
    eng.counters(reset=True)
    barrier()
    t0 = time.perf_counter()
    frames = samples = 0
    for i in range(args.steps):
        f, s = step(i)
        frames += f
        samples += s
    barrier()
    dt = time.perf_counter() - t0
    ctr = eng.counters()

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
            stages = stage_report(eng, q3tts, cfg, toks, sp, B, F, ctr)
        except Exception as ex:   # a reported extra, never the measurement
            stages = {"error": str(ex)}

    if rank == 0:
        step_ms = ctr["decode_ms"] / max(ctr["decode_steps"], 1)
        abytes = algorithmic_step_bytes(cfg, B, 8 + F / 2.0)
        achieved = abytes / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "decode_step_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(f"b{B}") if args.model == "0.6b" else None
            except Exception:
                traffic = None
        out = {
            "metric": f"real-time factor (24 kHz audio sec / wall sec), {MODEL}",
            "value": round(frames * FRAME_SECONDS / dt, 3),
            "unit": "x real-time (audio s / wall s)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 weights, fp32 activations/accumulate (codec decoder: fp16 hi/lo split operands, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"{MODEL}, batch={B}/GPU, 16-token prompt, "
                                   + ("greedy top_k=1" if args.greedy else "sampled temp=0.8 top-k=50 top-p=0.95")
                                   + f", max-tokens={F} (EOS suppressed), synthetic seeded weights",
                       "batch_per_gpu": B, "frames_per_utterance": F, "parallelism": f"dp{world} (independent utterances)"},
            "codec_frames_per_s": round(frames / dt, 2),
            "pcm_samples": samples,
            "decode_ms_per_frame_step": round(step_ms, 4),
            "codec_decode_ms_per_frame": round(ctr["codec_ms"] / max(ctr["codec_frames"], 1), 5),
            "first_2s_audio_latency_ms": first_audio,
            "roofline": {"bound": "hbm", "kernel": "decode step = one hipGraph replay per frame ("
                                   + ("q3::k_gemv1 (QKV / o_proj / gate-up / down / heads), k_cp_attn_oproj x75, k_attn x28, k_sample x16" if B <= 4 else
                                      "q3::k_gemm2 + k_finish + k_attn + k_attn_combine + k_sample") + ")",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(abytes), "launch_ms": round(step_ms, 4)},
        }
        if stages is not None:
            out["stages"] = stages
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(eng, cfg, toks[0], sp_kwargs, args.cpu_frames)
            except Exception as ex:  # the baseline is a reported extra, never the measurement
                out["cpu_baseline"] = {"value": None, "unit": "x real-time (audio s / wall s)", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {ex}"}
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
