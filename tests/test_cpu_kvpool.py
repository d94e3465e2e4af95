"""Host logic of the talker's KV page pool and of the scheduler's page policy (leaxer-qwen3-tts_amd/csrc/q3_kvpool.h) without a GPU: a g++
harness (tests/cpp/kvpool_harness.cpp) executes commands against the header the engine and the scheduler use.  The reference has no
counterpart (it grows one KVCache per utterance, src/tts_onnx.h:108-115); what is checked is conservation and isolation of pages, the
admission rule, and that running dry preempts the youngest and never the oldest."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("kvpool") / "kvpool_harness")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "leaxer-qwen3-tts_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "kvpool_harness.cpp"), "-o", exe], check=True)

    def run(script):
        r = subprocess.run([exe], input=script, capture_output=True, text=True, check=True)
        return r.stdout.strip().splitlines()
    return run


def parse(line):
    m = re.match(r"(\w+) total (\d+) free (\d+) identity (\d) owned (.*) table(.*)", line)
    owned = [[int(x) for x in g.split(",")] if g else [] for g in re.findall(r"\[([\d,]*)\]", m.group(5))]
    return {"total": int(m.group(2)), "free": int(m.group(3)), "identity": int(m.group(4)), "owned": owned, "table": [int(x) for x in m.group(6).split()]}


def check_invariants(st, pps):
    pages = [p for o in st["owned"] for p in o]
    assert len(pages) == len(set(pages)), "a page has two owners"
    assert st["free"] + len(pages) == st["total"], "pages leaked or invented"
    if not st["identity"]:
        assert all(1 <= p <= st["total"] for p in pages)            # page 0 is the scratch page
        for s, o in enumerate(st["owned"]):
            row = st["table"][s * pps:(s + 1) * pps]
            assert row[:len(o)] == o and all(v == 0 for v in row[len(o):]), "table row does not mirror ownership"
    else:
        assert st["table"] == list(range(len(st["table"])))
        for s, o in enumerate(st["owned"]):
            assert o == [s * pps + i for i in range(len(o))]


def test_bounded_pool_reserve_release_and_exhaustion(harness):
    out = harness("\n".join([
        "init 3 5 6 4",            # 3 slots of up to 5 pages, 4 pages in the pool
        "reserve 0 109 1",         # 2 pages
        "reserve 1 64 1",          # exactly one page
        "reserve 2 65 1",          # needs 2, 1 free: refused, nothing changes
        "reserve 0 10 0",          # not exact: keeps its 2 pages
        "reserve 0 10 1",          # exact: gives one back
        "reserve 2 65 1",          # now fits
        "reserve 1 0 1", "reserve 0 0 1", "reserve 2 0 1",
    ]))
    states = [parse(ln) for ln in out if not ln.startswith("rc")]
    rcs = [ln for ln in out if ln.startswith("rc")]
    for st in states:
        check_invariants(st, 5)
    assert states[0]["identity"] == 0 and states[0]["total"] == 4
    assert states[1]["owned"][0] == [1, 2] and states[2]["owned"][1] == [3]
    assert rcs[2].startswith("rc -1 err KV page pool exhausted: slot 2 needs 2 more pages of 64 tokens, 1 of 4 free") and states[3] == states[2]
    assert rcs[3] == "rc 0" and states[4]["owned"][0] == [1, 2]
    assert rcs[4] == "rc 1" and states[5]["owned"][0] == [1] and states[5]["free"] == 2
    assert sorted(states[6]["owned"][2]) == [2, 4]                   # the page slot 0 gave back is reused
    assert states[-1]["free"] == 4 and all(o == [] for o in states[-1]["owned"])


def test_full_size_pool_is_the_identity_and_only_counts(harness):
    out = harness("init 2 3 6 0\nreserve 1 130 1\nreserve 0 1 1\nreserve 1 200 0\nreserve 1 0 1\n")
    states = [parse(ln) for ln in out if not ln.startswith("rc")]
    for st in states:
        check_invariants(st, 3)
    assert states[0]["identity"] == 1 and states[0]["total"] == 6
    assert states[1]["owned"][1] == [3, 4, 5] and states[1]["free"] == 3
    rcs = [ln for ln in out if ln.startswith("rc")]
    assert rcs[:2] == ["rc 0", "rc 0"] and rcs[3] == "rc 0"          # the identity table never changes: nothing to upload
    assert rcs[2].startswith("rc -1 err KV reservation exceeds the slot's page run") and states[3] == states[2]   # 200 tokens = 4 pages > 3 per slot
    assert states[-1]["owned"][1] == []


def test_admission_rule(harness):
    out = harness("\n".join([
        "init 4 5 6 6",
        "admit 1 0 4 4 3 3 3 3",       # lengths known, idle engine: 3 + 3 fit 6 pages
        "admit 1 0 4 1 7",             # alone it is admitted even if it could not fit (the caller rejects that case up front)
        "admit 0 0 4 4 1 1 1 1",       # on demand, idle: head-room 0, 1, 2 -> 1+0, 1+1 <= 5 left, 1+2 <= 4 left; fourth: 1+3 > 3 left
        "reserve 0 64 1", "reserve 1 64 1",
        "admit 0 2 2 2 1 1",           # two running, 4 pages free: 1 + 2 <= 4, then 1 + 3 > 3
        "admit 1 2 2 2 2 3",           # lengths known: 2 fits, then 3 > 2 left
    ]))
    adm = [int(ln.split()[1]) for ln in out if ln.startswith("admit")]
    assert adm == [2, 1, 3, 1, 1]


def test_growth_preempts_the_youngest_and_never_the_oldest(harness):
    out = harness("\n".join([
        "init 4 5 6 6",
        "reserve 0 64 1", "reserve 1 64 1", "reserve 2 64 1", "reserve 3 64 1",       # four running, one page each, 2 free
        "grow 4 0 129 1 129 2 129 3 129",      # oldest first; each wants 3 pages: 0 takes both free pages, then 3 and 2 go so that 1 can grow
        "grow 2 0 320 1 320",                  # 5 pages each from a 6-page pool: 1 is preempted, 0 (the oldest) gets its 5
    ]))
    pre = [ln for ln in out if ln.startswith("preempted")]
    states = [parse(ln) for ln in out if ln.startswith("grow")]
    for st in states:
        check_invariants(st, 5)
    assert pre[0].split(" changed")[0] == "preempted 3 2"
    assert [len(o) for o in states[0]["owned"]] == [3, 3, 0, 0] and states[0]["free"] == 0
    assert pre[1].split(" changed")[0] == "preempted 1"
    assert [len(o) for o in states[1]["owned"]] == [5, 0, 0, 0] and states[1]["free"] == 1
    for ln in pre:                             # every slot whose row the device must see again is reported
        ch = ln.split(" changed")[1].split()
        assert len(ch) == len(set(ch))
