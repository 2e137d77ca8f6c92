"""The LDS hand-back idiom behind round 3's wrong rounds (profiles/r04a_lds_handback_rootcause.txt): one lane stores a __shared__ word, the whole
wave reads it right after.  tests/lds_idiom/lds_idiom.hip holds the idiom three ways, built with the product's flags:
  0 plain (a data race between lanes for the compiler), 1 with the product's LANE0_PUBLISHED() wavefront fence, 2 read through readfirstlane (UNI32).
CPU: the generated ISA — where the ds_read sits relative to the loop — is checked, so the property holds for the toolchain that builds the library.
GPU: the kernels run; the fenced and readfirstlane forms give every lane the stored value, the plain form is shown to be stale on the other lanes."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "lds_idiom")


def _built(name):
    path = os.path.join(DIR, name)
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(DIR, "lds_idiom.hip")):
        subprocess.check_call(["make", "-s", "-C", DIR])
    return path


def _kernel_text(variant):
    asm = open(_built("lds_idiom.s")).read()
    m = re.search(rf"^_Z7k_idiomILi{variant}EEvPKiiPx:.*?s_endpgm", asm, re.S | re.M)
    assert m, f"variant {variant} not in the assembly"
    return m.group(0)


def _read_is_inside_the_loop(text):
    """the loop's blocks carry 'in Loop:' / 'Inner Loop Header' comments; the wave's ds_read must sit in one of them"""
    in_loop = False
    for line in text.split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", line):
            in_loop = "Loop" in line
        if "ds_read_b32" in line and in_loop:
            return True
    return False


def test_fence_keeps_the_read_behind_the_store():
    assert _read_is_inside_the_loop(_kernel_text(1)), "LANE0_PUBLISHED() no longer keeps the wave's read behind lane 0's store"


def test_plain_form_is_what_the_compiler_breaks():
    # not a property the product relies on: it records that this toolchain DOES hoist the plain form's read out of the loop (the root cause on file)
    plain = _kernel_text(0)
    if _read_is_inside_the_loop(plain):
        pytest.skip("this compiler keeps the plain form's read in the loop: the hazard is not visible in the ISA any more (the fences stay)")
    assert plain.index("ds_read_b32") < plain.index("Inner Loop Header")


def test_product_sources_fence_every_partial_lane_lds_store():
    """every `if (lane == 0) <LDS word> = ...` of the round kernel's sources is followed by a fence, a barrier or an atomic publish before the wave can
    read the word again: a source-level audit that fails when a new site forgets it"""
    csrc = os.path.join(HERE, "..", "armada_amd", "csrc")
    offenders = []
    for fn in ("round_fast.h", "round_run.h", "armada_sched.hip"):
        lines = open(os.path.join(csrc, fn)).read().split("\n")
        for i, line in enumerate(lines):
            if not re.search(r"if \((FLANE|lane) == 0( && [^)]*)?\)", line) or "define " in line:
                continue
            j, depth, stmt = i, 0, ""
            while j < len(lines):   # the statement (or block) this condition guards, up to its end
                stmt += lines[j] + "\n"
                depth += lines[j].count("{") - lines[j].count("}")
                if depth <= 0 and ";" in lines[j]:
                    break
                j += 1
            guarded = stmt[stmt.index("== 0"):]
            guarded = guarded[guarded.index(")") :]   # behind the condition: the assignment TARGETS are what counts (an LDS value used as an HBM index is not a store to LDS)
            if not re.search(r"[{;)]\s*(FL\.|RS\.|g_fl\.|g_rs\.|RREC\(|RJOB\(|RQ\()[^;=<>!]*(=|\+\+|\+=|-=)[^=]", guarded):
                continue   # no plain LDS store under the condition (HBM stores / atomics only)
            if re.search(r"statSeg", guarded) and not re.search(r"(FL\.|g_fl\.)[^;=]*=[^=]", guarded):
                continue   # profiling counters: read-modify-write by lane 0, read back by the host only
            after = "\n".join(lines[i:j + 4])
            if not re.search(r"LANE0_PUBLISHED\(\)|LDS_ORDER\(\)|__syncthreads\(\)", after):
                offenders.append(f"{fn}:{i + 1}: {line.strip()[:140]}")
    assert not offenders, "partial-lane LDS stores without a fence:\n" + "\n".join(offenders)


def _run(variant, rounds=50):
    lib = ctypes.CDLL(_built("liblds_idiom.so"))
    lib.lds_idiom_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(5)
    slots = rng.integers(0, 1 << 20, size=1024).astype(np.int32)
    out = np.zeros(64, dtype=np.int64)
    rc = lib.lds_idiom_run(variant, slots.ctypes.data, rounds, out.ctypes.data)
    assert rc == 0
    want = sum(int(slots[[(r * 64 + l) & 1023 for l in range(64)]].max()) for r in range(rounds))
    return out, want


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [1, 2])
def test_fenced_and_readfirstlane_forms_are_exact_on_the_device(variant):
    out, want = _run(variant)
    assert (out == want).all()


@pytest.mark.gpu
def test_plain_form_is_stale_on_the_other_lanes_on_the_device():
    out, want = _run(0)
    assert out[0] == want, "lane 0 stores and reads its own value"
    if (out == want).all():
        pytest.skip("the plain form happens to be compiled correctly by this toolchain")
    assert (out[1:] != want).all()   # what round 3's slot variants computed with: the value from before the store
