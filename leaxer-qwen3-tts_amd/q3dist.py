"""Batch sharding of independent utterances over ranks (one process per GPU).

The hot path has no per-step exchange: every utterance is an independent autoregressive stream
(the reference is literally batch 1, src/tts_onnx.cpp:411).  The only communication is the
scatter of work (a pure function of (n_utt, world, rank) here, so nothing is sent) and the final
gather of the variable-length codec frames, one all_gather of a padded int32 tensor (RCCL over
xGMI when the backend is "nccl", gloo in the CPU tests).
"""
import numpy as np


def shard_utterances(lengths, world, rank):
    """Indices of the utterances `rank` synthesises.  Longest-first round-robin ("LPT") over the text
    lengths so that every rank gets a similar amount of decode work; deterministic on every rank."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return sorted(order[rank::world])


def gather_codes(dist, codes_list, index_list, n_total, max_frames, n_groups, device=None, pcm_lens=None):
    """All-gather per-utterance code arrays (and, with pcm_lens, each utterance's PCM sample count: the PCM itself stays on the rank
    that produced it).  codes_list[i] is [F_i, n_groups] for global utterance index_list[i].  Returns a list of n_total arrays on
    every rank — plus the list of n_total sample counts when pcm_lens is given."""
    import torch
    world = dist.get_world_size()
    per = max(1, -(-n_total // world))
    if len(codes_list) > per:
        raise ValueError("gather_codes: a rank holds more utterances than ceil(n_total / world)")
    buf = np.full((per, max_frames, n_groups), -1, np.int32)
    meta = np.full((per, 4), -1, np.int32)  # (global index, n_frames, pcm samples low 31 bits, pcm samples high bits)
    for slot, (c, gi) in enumerate(zip(codes_list, index_list)):
        f = min(len(c), max_frames)
        buf[slot, :f] = c[:f]
        n = int(pcm_lens[slot]) if pcm_lens is not None else 0
        meta[slot] = (gi, f, n & 0x7FFFFFFF, n >> 31)
    tb, tm = torch.from_numpy(buf), torch.from_numpy(meta)
    if device is not None:
        tb, tm = tb.to(device), tm.to(device)
    ob = [torch.empty_like(tb) for _ in range(world)]
    om = [torch.empty_like(tm) for _ in range(world)]
    dist.all_gather(ob, tb)
    dist.all_gather(om, tm)
    out = [None] * n_total
    lens = [0] * n_total
    for r in range(world):
        b, m = ob[r].cpu().numpy(), om[r].cpu().numpy()
        for slot in range(per):
            gi, f = int(m[slot, 0]), int(m[slot, 1])
            if gi >= 0:
                out[gi] = b[slot, :f].astype(np.int64)
                lens[gi] = int(m[slot, 2]) | (int(m[slot, 3]) << 31)
    return (out, lens) if pcm_lens is not None else out
