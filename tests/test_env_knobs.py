"""The library's environment surface: at most 20 variables, each documented in include/sdpa_hip.h; every
other switch is a name inside $SDPA_DEBUG and is listed in csrc/sdpa_debug.h."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")


def sources():
    for pat in ("csrc/*", "host/*", "*.py"):
        for f in glob.glob(os.path.join(PKG, pat)):
            if f.endswith((".hip", ".h", ".c", ".cpp", ".inc", ".py")):
                yield f, open(f).read()


def test_at_most_twenty_environment_variables_all_documented():
    read = set()
    for f, s in sources():
        read |= set(re.findall(r'"(SDPA_[A-Z0-9_]+)"', s))
    header = open(os.path.join(ROOT, "include", "sdpa_hip.h")).read()
    doc = header[header.index("Environment (the WHOLE list"):header.index("#ifndef SDPA_HIP_H")]
    documented = set(re.findall(r"\b(SDPA_[A-Z0-9_]+)\b", doc)) - {"SDPA_F_BF16", "SDPA_F_PLAN_QROWS", "SDPA_F_MERGE_ALLREDUCE"}
    assert read <= documented, sorted(read - documented)
    assert documented <= read, sorted(documented - read)
    assert len(read) <= 20, sorted(read)


def test_every_debug_name_read_is_listed_in_sdpa_debug_h():
    names = set()
    for f, s in sources():
        names |= set(re.findall(r'sdpa_debug_(?:int|pos|is|find)\("([a-z0-9_]+)"', s))
    head = open(os.path.join(PKG, "csrc", "sdpa_debug.h")).read().split("#ifndef SDPA_DEBUG_H")[0]
    missing = [n for n in sorted(names) if not re.search(r"\b%s\b" % n, head)]
    assert not missing, missing
    assert len(names) >= 20
