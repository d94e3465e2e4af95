"""CPU-side invariant on the BUILT product library: no gfx950 kernel of libq3tts_hip.so has a private (scratch) segment.

A spill is a memory round trip per access on a latency-bound chain, and it was round 4's suspect for the one wrong batched-vocoder
result (an intermediate build whose dilation-9 fused unit spilled 8 bytes while nine lane streams decoded concurrently,
DESIGN.md section 8).  Round 5 found four spilling kernels in the shipped library (k_attn_win 68 B, two never-launched k_gemv1
instantiations, k_cp_attn_oproj<2, 4> 52 B) and removed them; this test keeps it that way.  Needs no GPU: it reads the code object's
metadata notes (tools/kernel_resources.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_product_kernel_uses_scratch():
    import q3tts
    from kernel_resources import kernel_table
    rows = kernel_table(q3tts.LIB_PATH)
    assert len(rows) > 100, "expected the product library's few hundred kernel instantiations, got %d" % len(rows)
    bad = [(name, scratch) for name, vgpr, agpr, sgpr, scratch, lds in rows if scratch != 0]
    assert not bad, "kernels with a scratch segment (bytes per lane): %s" % bad
    # launch_bounds sanity: no kernel asks for more LDS than a workgroup may have
    assert all(lds <= 160 * 1024 for *_, lds in rows)   # gfx950: 160 KB of LDS per workgroup


def test_last_conv_kernel_has_no_packed_fp32_instruction():
    """k_conv_cout1_reg (the vocoder's C_out = 1 conv) must be scalar fp32: with v_pk_fma_f32 it returned wrong partial sums in 12-28 %
    of the jobs of tools/vocoder_stress.py (round 5); the packed form survives only as the A/B reproducer."""
    import q3tts
    from kernel_resources import kernel_disassembly
    ks = kernel_disassembly(q3tts.LIB_PATH, "k_conv_cout1_reg<")
    prod = [n for n in ks if "false, false" in n]
    repro = [n for n in ks if "true, false" in n]
    assert len(prod) == 1 and len(repro) == 1, sorted(ks)
    assert "v_pk_" not in ks[prod[0]] and ks[prod[0]].count("v_fmac_f32") + ks[prod[0]].count("v_fma_f32") >= 96 * 8
    assert "v_pk_fma_f32" in ks[repro[0]]      # the reproducer still is what it claims to be
