"""Register / scratch / LDS use of every gfx950 kernel in a built code object (the product .so or a .o):

    python tools/kernel_resources.py leaxer-qwen3-tts_amd/libq3tts_hip.so [name-substring]

`kernel_table(path)` is what tests/test_kernel_resources.py imports: the product library must not hold a kernel with a private
(scratch) segment — a spill costs a memory round trip per access, and scratch on nine concurrent vocoder lane streams was round 4's
suspect for the one wrong batched-vocoder result."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def kernel_table(obj):
    """[(demangled name, vgpr, agpr, sgpr, scratch bytes, lds bytes)] for EVERY gfx950 code object bundled in `obj` (a .so linked from
    several .hip sources holds one offload bundle per source, back to back in .hip_fatbin)."""
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "k.fat")
        subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        if not starts:
            raise RuntimeError("no offload bundle in " + obj)
        for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
            one, co = os.path.join(tmp, "b%d.fat" % n), os.path.join(tmp, "b%d.co" % n)
            open(one, "wb").write(blob[a:b])
            out = subprocess.run([LLVM + "clang-offload-bundler", "--list", "--type=o", "--input=" + one], capture_output=True, text=True, check=True)
            tgts = [l for l in out.stdout.split() if "gfx950" in l]
            if not tgts:
                continue
            subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + one, "--targets=" + tgts[0], "--output=" + co], check=True)
            md = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in md.split("  - .agpr_count")[1:]:
                g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
                rows.append((re.search(r"\.name:\s+(\S+)", blk).group(1), g("vgpr_count"), int(re.match(r":\s+(\d+)", blk).group(1)), g("sgpr_count"),
                             g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    return [(re.sub(r"\(.*", "", d.replace("void q3::", "")),) + r[1:] for r, d in zip(rows, dem)]


def kernel_disassembly(obj, name_substr):
    """{demangled name: instruction text} of every gfx950 kernel of `obj` whose demangled name contains name_substr (one llvm-objdump
    pass per offload bundle, split at the symbol headers)."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "k.fat")
        subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
            one, co = os.path.join(tmp, "b%d.fat" % n), os.path.join(tmp, "b%d.co" % n)
            open(one, "wb").write(blob[a:b])
            lst = subprocess.run([LLVM + "clang-offload-bundler", "--list", "--type=o", "--input=" + one], capture_output=True, text=True, check=True)
            tgts = [l for l in lst.stdout.split() if "gfx950" in l]
            if not tgts:
                continue
            subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + one, "--targets=" + tgts[0], "--output=" + co], check=True)
            md = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            kernels = set(re.findall(r"\.name:\s+(\S+)", md))          # kernel entry points (device functions are not in the notes)
            txt = subprocess.run([LLVM + "llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
            parts = re.split(r"^[0-9a-f]+ <([^>]+)>:\n", txt, flags=re.M)
            syms, bodies = parts[1::2], parts[2::2]
            keep = [(m_, t_) for m_, t_ in zip(syms, bodies) if m_ in kernels]
            dem = subprocess.run(["c++filt"], input="\n".join(m_ for m_, _ in keep), capture_output=True, text=True).stdout.split("\n")
            for (m_, t_), d_ in zip(keep, dem):
                if name_substr in d_:
                    out[re.sub(r"\(.*", "", d_.replace("void q3::", ""))] = t_
    return out


if __name__ == "__main__":
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, vgpr, agpr, sgpr, scratch, lds in kernel_table(sys.argv[1]):
        if pat in name:
            print(f"{name[:90]:90s} vgpr {vgpr:>4} agpr {agpr:>4} sgpr {sgpr:>4} scratch {scratch:>5} lds {lds:>6}")
