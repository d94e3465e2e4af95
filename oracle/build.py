"""Build the CPU oracle (test infrastructure) with gcc.  Output: oracle/_build/libq3oracle.so.

The library is compiled with -march=native, so it is rebuilt whenever the host CPU's flag set
differs from the one it was built on (the GPU box is a different machine than the build box).
"""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
SRC = [os.path.join(HERE, "q3_oracle.c")]


def _cpu_tag():
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    h = hashlib.sha1()
    h.update(flags.encode())
    for s in SRC + [os.path.join(HERE, "q3_oracle.h")]:
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    so = os.path.join(OUT_DIR, "libq3oracle.so")
    tag_file = so + ".tag"
    tag = _cpu_tag()
    if not force and os.path.exists(so) and os.path.exists(tag_file) and open(tag_file).read() == tag:
        return so
    cmd = ["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared", "-fPIC",
           "-std=gnu11", "-Wall", "-Wno-unused-function", "-o", so + ".tmp"] + SRC + ["-lm"]
    subprocess.run(cmd, check=True)
    os.replace(so + ".tmp", so)
    with open(tag_file, "w") as f:
        f.write(tag)
    return so


if __name__ == "__main__":
    print(build(force=True))
