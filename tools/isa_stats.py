#!/usr/bin/env python3
"""Compile one kernel translation unit to gfx950 assembly and print, per kernel matching a regex, the register
budget hipcc reports and the instruction histogram of its main loop (tests/test_kernel_isa.py's definition).
    python tools/isa_stats.py sdpa_fwd_f32.hip 'fused_pipelined_kernelILi128ELi128E' [-DFLAG ...]
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_kernel_isa as T  # noqa: E402


def main():
    src, pat = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    out = os.path.join(tempfile.mkdtemp(dir="/tmp"), src + ".s")
    subprocess.check_call([T.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fvisibility=hidden", "-Wno-unused-result",
                           "-Wno-inline-asm", "--cuda-device-only", "-S", os.path.join(T.CSRC, src), "-o", out] + extra)
    text = open(out).read()
    lines = text.split("\n")
    names = [m.group(1) for l in lines if (m := re.match(r"^(_ZN4sdpa\S*" + pat + r"\S*):", l))]
    for name in names:
        k = T.kernel_lines(lines, re.escape(name[len("_ZN4sdpa"):]))
        lo, hi, c = T.main_loop_span(k)
        meta = {}
        blk = text[text.find(".amdhsa_kernel " + name):]
        blk = blk[:blk.find(".end_amdhsa_kernel")]
        for key in ("next_free_vgpr", "next_free_sgpr", "accum_offset", "private_segment_fixed_size"):
            m = re.search(r"\.amdhsa_" + key + r"\s+(\d+)", blk)
            meta[key] = int(m.group(1)) if m else None
        body = "\n".join(re.sub(r"\.LBB\d+_\d+", "L", l) for l in k[lo:hi + 1] if l.startswith("\t") and not l.startswith("\t;"))
        print(name)
        print("  regs:", meta, " kernel lines:", len(k), " loop lines:", hi - lo + 1, " loop sha:", hashlib.sha1(body.encode()).hexdigest()[:12])
        print("  loop mix:", {kk: v for kk, v in sorted(c.items()) if v and (kk.startswith(("v_mfma", "scratch", "global_load", "ds_", "v_exp", "s_barrier", "v_accvgpr", "s_waitcnt", "buffer")))})


if __name__ == "__main__":
    main()
