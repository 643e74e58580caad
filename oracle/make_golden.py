"""Generate tests/golden/*.bin with the REFERENCE's own serial attention().

Run in the build container (needs /root/reference):
    make -C oracle ref && python oracle/make_golden.py

Each file is in the reference's on-disk format (attention.c:92-121, :139-140) and
its answer block is the output of the reference's unmodified `attention()`
(attention.c:20-75, compiled into oracle/_ref/libref_serial.so).  The files are
committed so the GPU box -- which has no /root/reference -- checks against
reference-produced numbers.  Inputs come from oracle.make_inputs (seeded numpy).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle as O  # noqa: E402

CASES = [
    # name,            m,   n,  dk,  dv, dist, seed
    ("tiny_D1",        12,  20,   8,   8, "D1", 1),
    ("ragged_D2",      33,  70,  72,  40, "D2", 2),
    ("fewkeys_D2",     64,   5,  16,  16, "D2", 3),
    ("peaky_D3",       40, 300,  32,  24, "D3", 4),
    ("adversarial_D4", 48, 200,  64,  64, "D4", 5),
    ("cfg1_small_D1",  96, 128,  64,  64, "D1", 6),
    ("d128_D1",        40, 160, 128, 128, "D1", 7),
]

# mid-size cases (round 4): multi-tile, multi-split launches checked against REFERENCE bytes.
# Only the answer block is committed; the inputs are the seeded draw (oracle.load_golden checks their sha256).
SEEDED_CASES = [
    ("mid_d128_D3",   256, 8192, 128, 128, "D3", 8),     # 2 query blocks x 256 K/V tiles: in-GPU splits / stream-K pieces
    ("mid_d512_D3",   256, 8192, 512, 512, "D3", 9),     # the dk-split fp32 kernel and the bf16 tandem kernel
]


def main():
    if not O.RefSerial.available():
        raise SystemExit("oracle/_ref/libref_serial.so missing: run `make -C oracle ref`")
    ref = O.RefSerial()
    out_dir = os.path.join(os.path.dirname(O.HERE), "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    index = []
    for name, m, n, dk, dv, dist, seed in CASES:
        Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed)
        ans = ref.attention(Q, K, V)
        path = os.path.join(out_dir, name + ".bin")
        O.write_case(path, Q, K, V, ans)
        index.append(dict(name=name, m=m, n=n, dk=dk, dv=dv, dist=dist, seed=seed,
                          file=name + ".bin", answer="reference attention.c:20-75"))
        print(name, os.path.getsize(path), "bytes")
    import hashlib
    for name, m, n, dk, dv, dist, seed in SEEDED_CASES:
        Q, K, V = O.make_inputs(m, n, dk, dv, dist, seed)
        ans = ref.attention(Q, K, V)
        path = os.path.join(out_dir, name + ".ans.f64")
        ans.astype("<f8").tofile(path)
        sha = {k: hashlib.sha256(a.astype("<f8").tobytes()).hexdigest() for k, a in (("Q", Q), ("K", K), ("V", V))}
        index.append(dict(name=name, m=m, n=n, dk=dk, dv=dv, dist=dist, seed=seed, file=None,
                          answer_file=name + ".ans.f64", inputs="oracle.make_inputs(m, n, dk, dv, dist, seed)",
                          inputs_sha256=sha, answer="reference attention.c:20-75"))
        print(name, os.path.getsize(path), "bytes (answer block only)")
    with open(os.path.join(out_dir, "INDEX.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    main()
