"""CPU: the planner of the C host (sdpa_plan_describe -- the schedule sdpa_attention_f64 runs: Q batches,
row pieces, per-rank K/V or query-row ranges, streamed K/V chunks, in-launch splits and their slots).
Pure host arithmetic, so it is checked here without a GPU: partitions are the reference's
owner_count / owner_disp (attention-mpi.c:19-27), chunks tile a shard exactly on boundaries every
operand image's tiling divides, slots are dense, and the environment knobs do what INTEGRATION.md says."""
import pytest

from conftest import set_knobs

SDPA_F_NO_PIPELINE, SDPA_F_BF16, SDPA_F_PLAN_QROWS, SDPA_F_MERGE_ALLREDUCE = 1, 2, 4, 8

SHAPES = [(32768, 65536, 128, 128), (8192, 8192, 128, 128), (512, 512, 64, 64), (131072, 65536, 128, 128),
          (32768, 262144, 128, 128), (32768, 65536, 512, 512), (1, 1, 1, 1), (300, 5000, 100, 200), (77, 3, 64, 64),
          (5, 100000, 256, 256), (40000, 7, 33, 1000)]


@pytest.fixture(autouse=True)
def clean_env(monkeypatch):
    for k in ("SDPA_STREAMED", "SDPA_DEBUG", "SDPA_QBATCH", "SDPA_PLAN", "SDPA_MERGE", "SDPA_PRECISION"):
        monkeypatch.delenv(k, raising=False)


def check_rank(pl, rp, n_keys_expected, rows_expected):
    assert rp["key_cnt"] == n_keys_expected and rp["row_cnt"] == rows_expected
    ch = rp["chunks"]
    if rp["key_cnt"] == 0:
        assert ch == [] and rp["n_slots"] == 0
        return
    # chunks tile [0, key_cnt) in order; every inner boundary is a multiple of 1024 keys
    assert ch[0][0] == 0 and sum(c[1] for c in ch) == rp["key_cnt"]
    for a, b in zip(ch, ch[1:]):
        assert b[0] == a[0] + a[1] and b[0] % 1024 == 0
    # slots are dense and in launch order; every launch has at least one split
    slot = 0
    for k0, keys, splits, slot0 in ch:
        assert keys > 0 and 1 <= splits <= 64 and slot0 == slot
        slot += splits
    assert rp["n_slots"] == slot
    # in-launch splits never cut below 4 tiles of 32 keys (8 for the bf16 duo kernel)
    for k0, keys, splits, slot0 in ch:
        assert splits == 1 or keys // splits >= 4 * 32 - 32
    pr = rp["piece_rows"]
    rows0 = min(pl["q_batch"], rp["row_cnt"])
    assert (pr == rows0) or (pr % 128 == 0 and pl["piece_min_rows"] <= pr < rows0) or rows0 == 0
    check_stream(pl, rp)


def check_stream(pl, rp):
    """the streamed form of the first batch (round 5): ONE launch, `splits` equal K/V ranges of `tiles_per_split` tiles;
    group c brings tiles [end_tile[c-1], end_tile[c]) of EVERY split, as row ranges that tile the shard exactly once"""
    st = rp["stream"]
    if not st["on"]:
        assert st["entries"] == []
        return
    assert 1 <= st["splits"] <= 8 and rp["key_cnt"] >= 8192
    if pl["bf16"]:       # the bf16 form (second half of round 5): the tandem kernel's shapes only, and a whole-chip launch only
        assert pl["compute_cus"] == 256
    tps, ends = st["tiles_per_split"], st["end_tile"]
    ntiles = -(-rp["key_cnt"] // 32)
    assert tps == -(-ntiles // st["splits"])
    assert 1 <= len(ends) <= 16 and ends == sorted(set(ends)) and ends[-1] == tps and ends[0] >= 1
    if st["interleaved"]:
        # round 6: group c is the c-th CONTIGUOUS key range of the shard (one pitched copy per operand puts its `splits` slices at
        # tiles [ends[c-1], ends[c]) of the splits' ranges); fp32, more than one split, a whole number of tiles per split; a group
        # is at least 2048 keys
        assert not pl["bf16"] and st["splits"] > 1 and len(ends) >= 2 and st["splits"] * tps * 32 == rp["key_cnt"]
        at = 0
        for c, (a, b) in enumerate(zip([0] + ends, ends)):
            keys = st["splits"] * (b - a) * 32
            assert keys >= 2048 and st["entries"][c] == [at, keys, c]
            at += keys
        assert at == rp["key_cnt"] and len(st["entries"]) == len(ends)
        return
    # a split's share of a group is at least 2048 keys (64 tiles) unless it is the split's whole range (one group)
    for a, b in zip([0] + ends, ends):
        assert b - a >= 64 or len(ends) == 1
    # the row ranges: group major, inside a group ascending; group c covers exactly tiles [ends[c-1], ends[c]) of every
    # split (adjacent ranges of a group are one copy); all of them tile the shard once
    want = {c: set() for c in range(len(ends))}
    for c, (a, b) in enumerate(zip([0] + ends, ends)):
        for sx in range(st["splits"]):
            want[c].update(t for t in range(sx * tps + a, sx * tps + b) if t < ntiles)
    got = {c: set() for c in range(len(ends))}
    last = (0, -1)
    for k0, keys, group in st["entries"]:
        assert keys > 0 and k0 % 32 == 0 and (group, k0) > last and group < len(ends)
        last = (group, k0)
        tiles = set(range(k0 // 32, -(-(k0 + keys) // 32)))
        assert not (tiles & got[group])
        got[group] |= tiles
        assert k0 + keys <= rp["key_cnt"] and ((k0 + keys) % 32 == 0 or k0 + keys == rp["key_cnt"])
    assert got == want
    assert sum(k for _, k, _ in st["entries"]) == rp["key_cnt"]
    if len(ends) == 1:
        assert st["entries"] == [[0, rp["key_cnt"], 0]]


@pytest.mark.parametrize("m,n,dk,dv", SHAPES)
@pytest.mark.parametrize("ranks", [1, 2, 3, 8, 16])
@pytest.mark.parametrize("flags", [0, SDPA_F_BF16])
def test_kv_sharded_plan(m, n, dk, dv, ranks, flags, pkg, orc):
    if flags & SDPA_F_BF16 and (dk > 512 or dv > 1024):
        pytest.skip("outside the bf16 path")
    pl = pkg.plan(m, n, dk, dv, flags, ranks)
    assert pl["ranks"] == ranks and pl["qrows"] == 0 and pl["bf16"] == (1 if flags & SDPA_F_BF16 else 0)
    assert pl["collectives"] == (1 if ranks > 1 else 0)
    assert pl["q_batch"] == min(32768, m) and pl["q_batches"] == -(-m // pl["q_batch"])
    off = 0
    for g, rp in enumerate(pl["r"]):
        assert rp["key_off"] == orc.owner_disp(n, ranks, g) == off
        check_rank(pl, rp, orc.owner_count(n, ranks, g), m)
        assert rp["row_off"] == 0
        off += rp["key_cnt"]
    assert off == n


@pytest.mark.parametrize("m,n,dk,dv", SHAPES)
@pytest.mark.parametrize("ranks", [2, 5, 8])
def test_query_row_sharded_plan(m, n, dk, dv, ranks, pkg, orc):
    pl = pkg.plan(m, n, dk, dv, SDPA_F_PLAN_QROWS, ranks)
    assert pl["qrows"] == 1 and pl["collectives"] == 0
    off = 0
    for g, rp in enumerate(pl["r"]):
        assert rp["key_off"] == 0 and rp["row_off"] == orc.owner_disp(m, ranks, g) == off
        check_rank(pl, rp, n, orc.owner_count(m, ranks, g))
        off += rp["row_cnt"]
    assert off == m
    assert pl["q_batch"] == min(32768, max(1, orc.owner_count(m, ranks, 0)))
    # one rank cannot shard query rows: the flag is ignored
    assert pkg.plan(m, n, dk, dv, SDPA_F_PLAN_QROWS, 1)["qrows"] == 0


def test_metric_shape_schedule_is_the_documented_one(pkg):
    """chunks of 4096, 4096, 8192, 16384, ... keys, 4 row pieces of 8192 rows; the largest chunk by how the
    feed time compares with the kernel time (round 4): 65536 keys when kernel bound, 8192 when feed bound"""
    pl = pkg.plan(32768, 65536, 128, 128)
    rp = pl["r"][0]
    assert [c[1] for c in rp["chunks"]] == [4096, 4096, 8192, 16384, 32768]
    assert [c[1] for c in pkg.plan(32768, 262144, 128, 128)["r"][0]["chunks"]] == [4096, 4096, 8192, 16384, 32768, 65536, 65536, 65536]
    assert [c[1] for c in pkg.plan(32768, 65536, 512, 512, SDPA_F_BF16)["r"][0]["chunks"]] == [4096, 4096] + [8192] * 7
    assert pl["row_pieces"] == 4 and rp["piece_rows"] == 8192 and pl["q_batches"] == 1
    # config 4: four Q batches over the same K/V schedule
    assert pkg.plan(131072, 65536, 128, 128)["q_batches"] == 4
    # config 3 on 8 ranks: each rank's 32768 keys
    p8 = pkg.plan(32768, 262144, 128, 128, 0, 8)
    assert all([c[1] for c in r["chunks"]] == [4096, 4096, 8192, 16384] for r in p8["r"])


def test_compute_units_are_left_to_the_comm_streams_only_where_it_pays(pkg, monkeypatch):
    """16 of 256 compute units' worth of workgroup slots stay free for the comm streams when the call merges over several
    ranks, has a next batch to hide a tail under, and a batch's kernels are short enough for the hidden tail (~0.25 ms)
    to outweigh the 7.6-7.9 % they cost (make_plan; profiles/r04/rank_share_*.json)"""
    monkeypatch.delenv("SDPA_COMM_CUS", raising=False)
    assert pkg.plan(131072, 65536, 128, 128, 0, 8)["compute_cus"] == 240      # config 4 on 8 ranks: 4 batches of ~1 ms
    assert pkg.plan(131072, 65536, 128, 128, 0, 4)["compute_cus"] == 240      # ~2.1 ms
    assert pkg.plan(131072, 65536, 128, 128, 0, 2)["compute_cus"] == 256      # ~4.2 ms of kernel per batch: not worth 0.3 ms
    assert pkg.plan(32768, 65536, 128, 128, 0, 8)["compute_cus"] == 256       # one batch: nothing to hide a tail under
    assert pkg.plan(131072, 65536, 128, 128, 0, 1)["compute_cus"] == 256      # one rank: no collective
    assert pkg.plan(131072, 65536, 128, 128, SDPA_F_PLAN_QROWS, 8)["compute_cus"] == 256
    monkeypatch.setenv("SDPA_COMM_CUS", "32")
    assert pkg.plan(131072, 65536, 128, 128, 0, 2)["compute_cus"] == 224      # the caller decides
    monkeypatch.setenv("SDPA_COMM_CUS", "0")
    assert pkg.plan(131072, 65536, 128, 128, 0, 8)["compute_cus"] == 256


def test_knobs(pkg, monkeypatch):
    base = pkg.plan(100000, 100000, 128, 128)
    assert base["q_batch"] == 32768 and base["q_batches"] == 4
    monkeypatch.setenv("SDPA_QBATCH", "10000")
    assert pkg.plan(100000, 100000, 128, 128)["q_batches"] == 10
    monkeypatch.delenv("SDPA_QBATCH")
    set_knobs(monkeypatch, SDPA_KV_CHUNK_MIN=8192, SDPA_KV_CHUNK_MAX=8192)          # (test knobs: $SDPA_DEBUG=kv_chunk_min=...,kv_chunk_max=...)
    ch = pkg.plan(32768, 65536, 128, 128)["r"][0]["chunks"]
    assert [c[1] for c in ch] == [8192] * 8
    set_knobs(monkeypatch, SDPA_KV_CHUNK_MIN=5000, SDPA_KV_CHUNK_MAX=5000)          # rounded down to a multiple of 1024
    assert [c[1] for c in pkg.plan(32768, 65536, 128, 128)["r"][0]["chunks"]][:3] == [4096] * 3
    set_knobs(monkeypatch, SDPA_KV_CHUNK_MIN=None, SDPA_KV_CHUNK_MAX=None, SDPA_ROW_PIECES=1)
    assert pkg.plan(32768, 65536, 128, 128)["r"][0]["piece_rows"] == 32768
    set_knobs(monkeypatch, SDPA_ROW_PIECES=None)
    # SDPA_F_NO_PIPELINE: one batch, one chunk, no pieces (the round-1 structure)
    pl = pkg.plan(100000, 100000, 128, 128, SDPA_F_NO_PIPELINE)
    assert pl["q_batches"] == 1 and pl["q_batch"] == 100000 and len(pl["r"][0]["chunks"]) == 1 and pl["row_pieces"] == 1
    # the environment selects plan / merge / precision like the flags do
    monkeypatch.setenv("SDPA_PLAN", "qrows"); monkeypatch.setenv("SDPA_PRECISION", "bf16")
    pl = pkg.plan(1000, 1000, 64, 64, 0, 4)
    assert pl["qrows"] == 1 and pl["bf16"] == 1
    monkeypatch.delenv("SDPA_PLAN")
    monkeypatch.setenv("SDPA_MERGE", "allreduce")
    assert pkg.plan(1000, 1000, 64, 64, 0, 4)["merge_allreduce"] == 1
    assert pkg.plan(1000, 1000, 64, 64, SDPA_F_MERGE_ALLREDUCE, 4)["merge_allreduce"] == 1


def test_describe_rejects_bad_arguments(pkg):
    import ctypes
    lib = pkg.load()
    buf = ctypes.create_string_buffer(8)
    assert lib.sdpa_plan_describe(10, 10, 4, 4, 0, 1, buf, len(buf)) < 0          # buffer too small
    big = ctypes.create_string_buffer(1 << 16)
    assert lib.sdpa_plan_describe(10, 10, 4, 4, 0, 0, big, len(big)) < 0          # ranks out of range
    assert lib.sdpa_plan_describe(10, 10, 4, 4, 0, 17, big, len(big)) < 0
    assert lib.sdpa_plan_describe(10, 10, 0, 4, 0, 1, big, len(big)) < 0
    assert lib.sdpa_plan_describe(0, 0, 4, 4, 0, 2, big, len(big)) == 0            # empty problem: empty plan


def test_streamed_first_batch_plan_on_the_baseline_shapes(pkg, monkeypatch):
    """Where the first Q batch runs as ONE persistent launch that follows its K/V groups: every
    BASELINE fp32 shape with d <= 128 -- with the split count of the device-level launch on the resident shard, which is
    what makes the two bit-identical -- and where it does not: bf16 with dv <= 256, fp32 with d > 128, few query blocks (many splits), a launch
    stream-K would cut differently, SDPA_F_NO_PIPELINE, $SDPA_STREAMED=0."""
    lib = pkg.load()
    for (m, n, d, ranks) in [(32768, 65536, 128, 1), (8192, 8192, 128, 1), (32768, 262144, 128, 1), (131072, 65536, 128, 1),
                             (32768, 262144, 128, 8), (32768, 65536, 64, 1)]:
        pl = pkg.plan(m, n, d, d, 0, ranks)
        for rp in pl["r"]:
            st = rp["stream"]
            assert st["on"] == 1, (m, n, d, ranks)
            assert st["splits"] == lib.sdpa_dev_kv_splits(min(m, 32768), rp["key_cnt"], d, d)
            check_stream(pl, rp)
    hd = pkg.plan(32768, 65536, 128, 128, 0, 1)["r"][0]["stream"]
    assert hd["interleaved"] == 1 and hd["splits"] == 2 and hd["end_tile"] == [32, 64, 128, 256, 512, 1024] and len(hd["entries"]) == 6
    # config 2: 8 splits of 1024 keys.  Rounds 5: ONE group (16 row ranges of 256 KiB per group cost more than they hid); interleaved
    # groups are one pitched copy each.  The call is FEED bound (0.36 ms of inputs for 0.26 ms of kernel): four EQUAL groups of 2048
    # keys (what counts is how little work is left when the last one lands), and the Q rows ride in front of group 0's ready word
    c2 = pkg.plan(8192, 8192, 128, 128, 0, 1)["r"][0]["stream"]
    assert c2["splits"] == 8 and c2["interleaved"] == 1 and c2["q_with_group0"] == 1 and c2["end_tile"] == [8, 16, 24, 32]
    assert c2["entries"] == [[0, 2048, 0], [2048, 2048, 1], [4096, 2048, 2], [6144, 2048, 3]]
    assert hd["q_with_group0"] == 0                           # kernel bound: the Q row pieces keep their own words (early starts)
    # a ragged shard (not a whole number of tiles per split) keeps the row ranges of round 5
    rg = pkg.plan(32768, 65536 + 40, 128, 128, 0, 1)["r"][0]["stream"]
    assert rg["on"] == 1 and rg["interleaved"] == 0
    # bf16 (second half of round 5): the tandem kernel's shapes (dv > 256) have a persistent form -- config 5 in bf16 is ONE launch over
    # 9 groups, with the device-level launch's split count; narrower value matrices (duo / pipe kernels) keep the launch per chunk
    c5 = pkg.plan(32768, 65536, 512, 512, SDPA_F_BF16, 1)
    st = c5["r"][0]["stream"]
    assert c5["bf16"] == 1 and st["on"] == 1 and st["splits"] == lib.sdpa_dev_kv_splits_bf16(32768, 65536, 512, 512) == 1
    assert st["end_tile"] == [128, 256, 512, 768, 1024, 1280, 1536, 1792, 2048] and len(st["entries"]) == 9
    check_stream(c5, c5["r"][0])
    for (m, n, dk, dv) in [(8192, 16384, 512, 512), (20000, 16384, 256, 1024), (4096, 20011, 300, 400), (32768, 262144, 512, 512)]:
        pl = pkg.plan(m, n, dk, dv, SDPA_F_BF16, 1)
        st = pl["r"][0]["stream"]
        assert st["on"] == 1 and st["splits"] == lib.sdpa_dev_kv_splits_bf16(min(m, 32768), n, dk, dv), (m, n, dk, dv)
        check_stream(pl, pl["r"][0])
    for (m, n, d, flags) in [(32768, 65536, 256, SDPA_F_BF16), (32768, 65536, 128, SDPA_F_BF16), (32768, 65536, 512, 0), (32768, 65536, 256, 0),
                             (512, 65536, 128, 0), (32768, 4096, 128, 0), (32768, 65536, 128, SDPA_F_NO_PIPELINE)]:
        assert pkg.plan(m, n, d, d, flags, 1)["r"][0]["stream"]["on"] == 0, (m, n, d, flags)
    monkeypatch.setenv("SDPA_STREAMED", "0")
    assert pkg.plan(32768, 65536, 128, 128, 0, 1)["r"][0]["stream"]["on"] == 0


# ---- the feed model (round 6): where the fp64 -> operand converts run, for P ranks sharing ONE host -------------
CONFIG3, METRIC, CONFIG4 = (32768, 262144, 128, 128), (32768, 65536, 128, 128), (131072, 65536, 128, 128)


@pytest.mark.parametrize("shape,ranks,page_locked", [
    # ONE pool serves every rank; each rank has its own PCIe link and its own kernels.  Host converts (and with them the streamed
    # first batch) while the pool keeps up, device converts + launch per chunk once t_host > 1.1 max(t_kernel, t_link)
    (CONFIG3, 1, "host"), (CONFIG3, 2, "host"), (CONFIG3, 8, "host"),       # P = 8: 570 MB through the pool in 3.4 ms (measured 2.9-3.6) under 4.2 ms of kernel per rank
    (METRIC, 1, "host"), (METRIC, 2, "host"), (METRIC, 8, "host"),          # P = 8: 1.0 ms of pool (measured 0.9-1.1) beside 1.06 ms of kernel, 0.9 ms of link
    (CONFIG4, 1, "host"), (CONFIG4, 2, "host"), (CONFIG4, 8, "device"),     # P = 8: four 1-ms batches leave 16 CUs to the comm streams --
                                                                            # stream-K grids have no streamed form: device converts, as before
])
def test_feed_model_choice_for_the_baseline_shapes(shape, ranks, page_locked, pkg, monkeypatch):
    """pinned for the GPU boxes' host as the library sees it: a 16-core CPU quota ($SDPA_DEBUG=host_cores=16 stands in for cgroup cpu.max here),
    32 pool threads, ~170 GB/s of fp64 source (measured: profiles/r06/feed_model_p8.log).  Pageable caller arrays keep the pool at every P (device converts would pull them
    through the runtime's bounce buffers on the enqueueing threads); page-locked ones follow the model."""
    set_knobs(monkeypatch, SDPA_HOST_CORES=16)
    for k in ("SDPA_HOST_CVT", "SDPA_HOST_CVT_THREADS", "SDPA_STREAMED"):
        monkeypatch.delenv(k, raising=False)
    m, n, dk, dv = shape
    f = pkg.plan(m, n, dk, dv, 0, ranks)["feed"]
    assert f["cores"] == 16 and f["pool_threads"] == 32 and abs(f["pool_GBps"] - 169.6) < 0.5, f
    assert f["pageable"] == "host" and f["page_locked"] == page_locked, f
    # the three times are the documented arithmetic
    total = (n * (dk + dv) + m * dk) * 8.0
    per_rank = (n / ranks * (dk + dv) + m * dk) * 8.0
    assert abs(f["t_host_ms"] - total / 169.6e9 * 1e3) < 0.01 * f["t_host_ms"] + 0.002, f
    assert abs(f["t_link_ms"] - per_rank / 55e9 * 1e3) < 0.01 * f["t_link_ms"] + 0.002, f
    assert abs(f["t_kernel_ms"] - 2.0 * m * (n / ranks) * (dk + dv) / 1.3e14 * 1e3) < 0.01 * f["t_kernel_ms"] + 0.002, f


def test_feed_model_follows_the_hosts_real_core_count(pkg, monkeypatch):
    """the pool is sized by the CPUs the process may really use (affinity mask cut down to the cgroup quota), two threads per core, at
    most 128; a small host keeps the device converts; a big one lifts the pool's rate and with it the P = 8 decision"""
    for k in ("SDPA_HOST_CVT", "SDPA_HOST_CVT_THREADS", "SDPA_STREAMED"):
        monkeypatch.delenv(k, raising=False)
    m, n, dk, dv = CONFIG3
    set_knobs(monkeypatch, SDPA_HOST_CORES=8)
    f = pkg.plan(m, n, dk, dv, 0, 1)["feed"]
    assert f["pool_threads"] == 16 and f["pageable"] == "device" and f["page_locked"] == "device", f
    set_knobs(monkeypatch, SDPA_HOST_CORES=128)
    f = pkg.plan(m, n, dk, dv, 0, 8)["feed"]
    assert f["pool_threads"] == 128 and f["pool_GBps"] == 240.0 and f["page_locked"] == "host", f      # 2.4 ms of pool under 4.2 ms of kernel
    # a pool that cannot keep up with eight ranks: page-locked arrays go to the device converts (every rank pulls its fp64 shard over
    # its own link, launch per chunk), pageable ones stay with the pool
    set_knobs(monkeypatch, SDPA_HOST_CORES=16)
    monkeypatch.setenv("SDPA_HOST_CVT_THREADS", "16")
    f = pkg.plan(m, n, dk, dv, 0, 8)["feed"]
    assert f["t_host_ms"] > 1.1 * max(f["t_kernel_ms"], f["t_link_ms"]) and f["page_locked"] == "device" and f["pageable"] == "host", f
    assert pkg.plan(m, n, dk, dv, 0, 2)["feed"]["page_locked"] == "host"                               # ... but keeps up with two
    monkeypatch.delenv("SDPA_HOST_CVT_THREADS")
    monkeypatch.setenv("SDPA_HOST_CVT_THREADS", "12")
    assert pkg.plan(m, n, dk, dv, 0, 8)["feed"]["pool_threads"] == 12
    set_knobs(monkeypatch, SDPA_HOST_CORES=None)
    monkeypatch.delenv("SDPA_HOST_CVT_THREADS")
    f = pkg.plan(m, n, dk, dv, 0, 1)["feed"]                                                          # whatever this machine is: consistent
    import os
    assert 1 <= f["cores"] <= (os.cpu_count() or 1) and f["pool_threads"] >= 1, f
    # the knob still overrides the model
    monkeypatch.setenv("SDPA_HOST_CVT", "1")
    assert pkg.plan(m, n, dk, dv, 0, 8)["feed"]["page_locked"] == "host"
    monkeypatch.setenv("SDPA_HOST_CVT", "0")
    assert pkg.plan(m, n, dk, dv, 0, 8)["feed"]["pageable"] == "device"
