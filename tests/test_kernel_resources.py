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
