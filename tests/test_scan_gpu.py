"""GPU parity tests for the scan half: the HIP path, called through the C ABI (pixelbox_amd.capi ->
libpixelbox_hip.so), against the CPU oracle (oracle/) and the committed golden vectors.

Bar: top-k image_id lists identical, distances BIT-exact (u32 view), counts identical -- for both the
int-dot filter path (with certificate) and the exhaustive exact path."""
import os

import numpy as np
import pytest

from oracle import capi as oracle
from pixelbox_amd import capi, synth

pytestmark = pytest.mark.gpu

F32 = np.float32
AUTO, EXACT, SINGLE, MULTI = 0, 1, 2, 3


def make_index(rows, ids=None, capacity=None, path=AUTO):
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, d = rows.shape
    ids = np.arange(n, dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)
    ix = capi.Index(d, capacity or max(n, 1))
    ix.set_option(capi.PB_OPT_SEARCH_PATH, path)
    if n:
        ix.load(ids, rows)
    return ix


def check_against_oracle(ix, rows, ids, queries, k=100, max_dist=1e3):
    queries = np.asarray(queries, dtype=np.uint8).reshape(-1, rows.shape[1])
    got_ids, got_d, got_c = ix.search(queries, k, max_dist)
    for qi, q in enumerate(queries):
        want_ids, want_d = oracle.scan_topk(q, rows, ids, k, max_dist)
        c = int(got_c[qi])
        assert c == len(want_ids), (qi, c, len(want_ids))
        assert np.array_equal(got_ids[qi, :c], want_ids), qi
        assert np.array_equal(got_d[qi, :c].view(np.uint32), want_d.view(np.uint32)), qi


def _uniform_inputs(g):
    n, d = int(g["n"]), int(g["d"])
    rows = synth.fill_synthetic(int(g["seed_rows"]), 0, n * d).reshape(n, d)
    query = synth.fill_synthetic(int(g["seed_query"]), 0, d)
    return query, rows, g["ids"]


# ---- committed golden vectors (sqlite3 + reference SQL) -------------------------------------------------
@pytest.mark.parametrize("path", [AUTO, EXACT])
@pytest.mark.parametrize("case", ["md1e3", "md5", "md2e6"])
def test_golden_uniform_4k(golden_dir, case, path):
    g = np.load(os.path.join(golden_dir, "scan_uniform_4k.npz"))
    query, rows, ids = _uniform_inputs(g)
    ix = make_index(rows, ids, path=path)
    got_ids, got_d = ix.search_one(query, 100, float(g[f"{case}_max_dist"]))
    assert np.array_equal(got_ids, g[f"{case}_ids"])
    assert np.array_equal(got_d.view(np.uint32), g[f"{case}_dist"].view(np.uint32))


@pytest.mark.parametrize("path", [AUTO, EXACT])
def test_golden_plateau_tail(golden_dir, path):
    g = np.load(os.path.join(golden_dir, "scan_uniform_150_plateau.npz"))
    query, rows, ids = _uniform_inputs(g)
    ix = make_index(rows, ids, path=path)
    got_ids, got_d = ix.search_one(query, 100, float(g["max_dist"]))
    assert np.array_equal(got_ids, g["out_ids"])
    assert np.array_equal(got_d.view(np.uint32), g["out_dist"].view(np.uint32))


@pytest.mark.parametrize("path", [AUTO, EXACT])
@pytest.mark.parametrize("case", ["md1e3", "md1e-3", "md2e6"])
def test_golden_clustered_2k(golden_dir, case, path):
    g = np.load(os.path.join(golden_dir, "scan_clustered_2k.npz"))
    ix = make_index(g["rows"], g["ids"], path=path)
    got_ids, got_d = ix.search_one(g["query"], 100, float(g[f"{case}_max_dist"]))
    assert np.array_equal(got_ids, g[f"{case}_ids"])
    assert np.array_equal(got_d.view(np.uint32), g[f"{case}_dist"].view(np.uint32))


def test_reference_kats_through_the_abi():
    # engine.rs:703-708 with dim = 2 (exact path: the filter pass needs dim >= 16)
    rows = np.array([[255, 0], [0, 255]], dtype=np.uint8)
    ix = make_index(rows)
    ids, d = ix.search_one(np.array([255, 0], dtype=np.uint8), 100, 2e6)
    assert ids.tolist() == [0, 1]
    assert d[0] == F32(-1.1920928955078125e-07) and d[0] < F32(1e-6)
    assert d[1] == F32(999999.0) and d[1] > F32(2.0)
    ids, d = ix.search_one(np.array([0, 255], dtype=np.uint8), 100, 1e3)
    assert ids.tolist() == [1] and d[0] < F32(1e-6)


# ---- seeded random tables vs the oracle ---------------------------------------------------------------
@pytest.mark.parametrize("path", [AUTO, EXACT])
@pytest.mark.parametrize("n", [1, 3, 63, 64, 65, 1000, 4097, 50000])
def test_uniform_sizes(n, path):
    rng = np.random.default_rng(100 + n)
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    ids = np.cumsum(rng.integers(1, 5, size=n)).astype(np.int64)
    queries = rng.integers(0, 256, size=(3, 256), dtype=np.uint8)
    queries[1] = rows[n // 2]
    ix = make_index(rows, ids, path=path)
    check_against_oracle(ix, rows, ids, queries)


@pytest.mark.parametrize("n", [31, 32 * 2048 + 5, 300007, 2100001])
def test_one_query_calls_hand_out_the_table_tail_by_tickets(n, monkeypatch):
    # a one-query call's filter launch gives the last eighth of the table out dynamically -- in chunks a workgroup requests
    # from per-region counters and shares among its waves through LDS (k_scan_filter STEAL, PB_FORCE_STEAL: tables under 2M
    # rows take static shares by default), or, in round 2's form, by one ticket per wave (DYN, PB_FORCE_TAIL_TICKETS):
    # neighbours planted at both ends of the static share, across the regions and in the last rows must all be found, three
    # times in a row (the counters are cleared between launches), and the static form gives the same answer
    rng = np.random.default_rng(4242 + n)
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    ids = np.arange(n, dtype=np.int64) * 3 + 1
    q = rng.integers(0, 256, size=256, dtype=np.uint8)
    spots = np.unique(np.concatenate([np.arange(min(n, 8)), n - 1 - np.arange(min(n, 40)),
                                      (n * 7 // 8 + np.arange(-20, 20) * 32) % n, rng.integers(0, n, size=60)]))
    for j, r in enumerate(spots):
        rows[r] = q
        rows[r, j % 256] ^= np.uint8(1 + j % 7)
    for force in ("PB_FORCE_STEAL", "PB_FORCE_TAIL_TICKETS"):
        monkeypatch.setenv(force, "1")
        ix = make_index(rows, ids)
        for _ in range(3):
            check_against_oracle(ix, rows, ids, q[None, :])
        monkeypatch.delenv(force)
    check_against_oracle(make_index(rows, ids), rows, ids, q[None, :])  # the default for this size
    monkeypatch.setenv("PB_STATIC_TAIL", "1")
    check_against_oracle(make_index(rows, ids), rows, ids, q[None, :])


def test_one_query_calls_repeat_exactly_in_every_form(monkeypatch):
    # The dynamic forms of the one-query launch depend on timing (which workgroup gets which chunk, in which order requests
    # are answered), so one passing call proves little: a first version of the chunked form let requests overtake each other
    # and 2-3 % of calls over this table silently missed a row of the pool -- certified, wrong.  150 calls per form, four
    # queries (one with ~150 planted near-duplicates), every answer compared with the oracle's.
    n = 2100001
    rng = np.random.default_rng(4242 + n)
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    ids = np.arange(n, dtype=np.int64) * 3 + 1
    q = rng.integers(0, 256, size=256, dtype=np.uint8)
    spots = np.unique(np.concatenate([np.arange(8), n - 1 - np.arange(40), (n * 7 // 8 + np.arange(-20, 20) * 32) % n,
                                      rng.integers(0, n, size=60)]))
    for j, r in enumerate(spots):
        rows[r] = q
        rows[r, j % 256] ^= np.uint8(1 + j % 7)
    qs = np.concatenate([q[None, :], rng.integers(0, 256, size=(3, 256), dtype=np.uint8)])
    want = [oracle.scan_topk(x, rows, ids, 100, 1e3) for x in qs]
    # PB_FUSE (round 6): the ONE-launch form -- the filter launch's last-arriving workgroup selects, re-scores and certifies
    # (k_scan_filter FUSE: write-through lists, an arrival counter, sc1 loads) -- over the chunked and over the static partition
    for form in (None, "PB_FORCE_TAIL_TICKETS", "PB_STATIC_TAIL", "PB_FUSE", "PB_FUSE+PB_STATIC_TAIL"):  # None: the default for this size, the chunked form
        if form:
            for f in form.split("+"):
                monkeypatch.setenv(f, "1")
        ix = make_index(rows, ids)
        for rep in range(150):
            qi = rep % len(qs)
            gi, gd, gc = ix.search(qs[qi:qi + 1], 100, 1e3)
            c = int(gc[0])
            assert c == len(want[qi][0]), (form, rep, qi)
            assert np.array_equal(gi[0, :c], want[qi][0]), (form, rep, qi)
            assert np.array_equal(gd[0, :c].view(np.uint32), want[qi][1].view(np.uint32)), (form, rep, qi)
        # (a late granule is legitimate on a GPU shared with another process and is handled -- test_one_query_calls_that_give_up_polling_
        # still_answer covers the give-up path; here only that polling is what normally completes a call)
        assert ix.stats().stamp_timeouts <= 15
        if form:
            for f in form.split("+"):
                monkeypatch.delenv(f)


@pytest.mark.parametrize("n_dup", [300, 700])
def test_one_launch_form_with_more_candidates_than_its_last_workgroup_has_threads(monkeypatch, n_dup):
    # PB_FUSE: the selection runs on the 512 threads of the filter launch's last workgroup, one candidate per thread; a query with
    # more candidates (700 near-duplicates within the margin of the 100th) reports status 2 and the host finishes the call with
    # the selection kernel of its own -- same answer either way, and small tables (fewer workgroups than CUs) take the form too
    monkeypatch.setenv("PB_FUSE", "1")
    for n in (200003, 5000):
        rng = np.random.default_rng(99 + n + n_dup)
        rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
        ids = np.arange(n, dtype=np.int64) * 5 + 2
        q = rng.integers(0, 256, size=256, dtype=np.uint8)
        for j, r in enumerate(rng.choice(n, n_dup, replace=False)):
            rows[r] = q
            rows[r, j % 256] ^= np.uint8(1)  # all at (nearly) the same distance: hundreds of candidates around the 100th
        ix = make_index(rows, ids)
        for _ in range(3):
            check_against_oracle(ix, rows, ids, q[None, :])


def test_one_query_calls_that_give_up_polling_still_answer(monkeypatch):
    # a one-query call polls pinned memory for its result granules; on a busy GPU they may be late: after the time-out the
    # call waits for the stream and reads the same granules, three time-outs in a row rest the polling for 256 calls
    # (ordinary result arrays), then it is tried again.  PB_POLL_TIMEOUT_US=0 makes every polled call time out.
    monkeypatch.setenv("PB_POLL_TIMEOUT_US", "0")
    rng = np.random.default_rng(77)
    rows = rng.integers(0, 256, size=(70001, 256), dtype=np.uint8)
    ids = np.arange(70001, dtype=np.int64) * 2 + 5
    qs = rng.integers(0, 256, size=(4, 256), dtype=np.uint8)
    ix = make_index(rows, ids)
    for rep in range(270):
        if rep < 12 or rep % 40 == 0 or rep > 258:
            check_against_oracle(ix, rows, ids, qs[rep % 4][None, :])
        else:
            ix.search(qs[rep % 4][None, :], 100, 1e3)
    st = ix.stats()
    assert 3 <= st.stamp_timeouts <= 12, st.stamp_timeouts  # 3, a rest of 256 calls, then again


@pytest.mark.parametrize("k", [1, 7, 100, 256])
def test_k_values(k):
    rng = np.random.default_rng(5)
    rows = rng.integers(0, 256, size=(20000, 256), dtype=np.uint8)
    ids = np.arange(20000, dtype=np.int64) + 10
    q = rng.integers(0, 256, size=(2, 256), dtype=np.uint8)
    for path in (AUTO, EXACT):
        ix = make_index(rows, ids, path=path)
        check_against_oracle(ix, rows, ids, q, k=k)


@pytest.mark.parametrize("d", [16, 32, 64, 128, 512, 1024, 8, 24, 100, 1])
def test_dims(d):
    rng = np.random.default_rng(d)
    n = 6000
    rows = rng.integers(0, 256, size=(n, d), dtype=np.uint8)
    ids = np.arange(n, dtype=np.int64)
    q = rng.integers(0, 256, size=(2, d), dtype=np.uint8)
    ix = make_index(rows, ids)
    check_against_oracle(ix, rows, ids, q)
    check_against_oracle(ix, rows, ids, q, max_dist=2e6)


def test_many_queries_chunking():
    rng = np.random.default_rng(9)
    rows = rng.integers(0, 256, size=(30000, 256), dtype=np.uint8)
    ids = np.arange(30000, dtype=np.int64)
    q = rng.integers(0, 256, size=(70, 256), dtype=np.uint8)  # > one launch group of 64
    ix = make_index(rows, ids)
    check_against_oracle(ix, rows, ids, q)
    st = ix.stats()
    assert st.queries == 70 and st.fast_path + st.fallback == 70
    assert st.fast_path >= 60  # uniform data: the certificate should pass


@pytest.mark.parametrize("max_dist", [float("nan"), -1.0, -1e-7, 0.0, 1e-6, 2.5, 3.0, 10.0, 999998.9, 999999.0, 999999.1, 1e30, float("inf")])
def test_max_dist_edges(max_dist):
    rng = np.random.default_rng(21)
    rows = rng.integers(0, 256, size=(5000, 256), dtype=np.uint8)
    rows[100] = rows[4000]
    ids = np.arange(5000, dtype=np.int64)
    q = np.stack([rows[4000], rng.integers(0, 256, size=256, dtype=np.uint8)])
    for path in (AUTO, EXACT):
        ix = make_index(rows, ids, path=path)
        check_against_oracle(ix, rows, ids, q, max_dist=max_dist)


# ---- adversarial tables: ties, duplicates, tiny norms --------------------------------------------------
def test_all_rows_identical_forces_exhaustive_pass():
    rows = np.tile(np.arange(256, dtype=np.uint8), (3000, 1))
    ids = np.arange(3000, dtype=np.int64) * 2
    ix = make_index(rows, ids)
    check_against_oracle(ix, rows, ids, rows[:1])
    check_against_oracle(ix, rows, ids, 255 - rows[:1], max_dist=2e6)  # everything on the 999999 plateau
    assert ix.stats().fallback >= 1


def test_many_duplicates_of_the_query():
    rng = np.random.default_rng(33)
    rows = rng.integers(0, 256, size=(40000, 256), dtype=np.uint8)
    q = rng.integers(0, 256, size=256, dtype=np.uint8)
    dup = rng.choice(40000, size=700, replace=False)
    rows[dup] = q  # 700 exact duplicates > any candidate buffer
    near = rng.choice(40000, size=300, replace=False)
    for j in near:
        r = q.copy()
        r[rng.integers(0, 256)] ^= 1
        rows[j] = r
    ids = np.arange(40000, dtype=np.int64)
    ix = make_index(rows, ids)
    check_against_oracle(ix, rows, ids, q.reshape(1, -1))
    check_against_oracle(ix, rows, ids, q.reshape(1, -1), k=256)


def test_contiguous_burst_of_near_duplicates():
    rng = np.random.default_rng(34)
    rows = rng.integers(0, 256, size=(100000, 256), dtype=np.uint8)
    q = rng.integers(0, 256, size=256, dtype=np.uint8)
    for j in range(5000, 5090):  # 90 adjacent near-duplicates (a photo burst in one folder)
        r = q.copy()
        idx = rng.choice(256, size=8, replace=False)
        r[idx] = rng.integers(0, 256, size=8)
        rows[j] = r
    ids = np.arange(100000, dtype=np.int64)
    ix = make_index(rows, ids)
    check_against_oracle(ix, rows, ids, q.reshape(1, -1))


def test_near_grey_rows_smallest_norms():
    # bytes 127/128 de-quantise to -0.0039215684 / +0.003921628: the worst case for the filter's error budget
    rng = np.random.default_rng(35)
    rows = rng.integers(127, 129, size=(20000, 256), dtype=np.uint8)
    ids = np.arange(20000, dtype=np.int64)
    q = rng.integers(127, 129, size=(2, 256), dtype=np.uint8)
    ix = make_index(rows, ids)
    check_against_oracle(ix, rows, ids, q)
    rows2 = rng.integers(0, 256, size=(20000, 256), dtype=np.uint8)
    rows2[::3] = rng.integers(120, 136, size=(len(rows2[::3]), 256), dtype=np.uint8)
    ix2 = make_index(rows2, ids)
    check_against_oracle(ix2, rows2, ids, q)
    check_against_oracle(ix2, rows2, ids, rows2[:2])


def test_clustered_embedding_like_table():
    rng = np.random.default_rng(36)
    centers = np.tanh(rng.standard_normal((20, 256)).astype(np.float32))
    which = rng.integers(0, 20, size=60000)
    f = np.tanh(np.arctanh(np.clip(centers[which], -0.999, 0.999)) + rng.standard_normal((60000, 256)).astype(np.float32) * 0.05)
    rows = oracle.quantize(f.astype(np.float32))
    ids = np.arange(60000, dtype=np.int64) * 7
    q = rows[[5, 77, 4242]]
    ix = make_index(rows, ids)
    check_against_oracle(ix, rows, ids, q)


# ---- store semantics (INSERT OR IGNORE, engine.rs:251-256) ---------------------------------------------
def test_append_insert_or_ignore_and_out_of_order_ids():
    rng = np.random.default_rng(40)
    d = 256
    rows = rng.integers(0, 256, size=(500, d), dtype=np.uint8)
    ix = capi.Index(d, 1000)
    assert len(ix) == 0
    ids0, d0 = ix.search_one(rows[0])
    assert len(ids0) == 0  # empty table
    assert ix.append(np.arange(0, 200) * 2, rows[:200]) == 200  # even ids 0..398
    assert ix.append([10, 12, 400, 402], rows[200:204]) == 2  # 10, 12 exist -> ignored
    assert ix.append([5, 7, 399], rows[204:207]) == 3  # odd ids: inserted in place
    assert ix.append([5], rows[300:301]) == 0  # first write wins
    truth = {}
    for i in range(200):
        truth[i * 2] = rows[i]
    truth[400], truth[402] = rows[202], rows[203]
    truth[5], truth[7], truth[399] = rows[204], rows[205], rows[206]
    t_ids = np.array(sorted(truth), dtype=np.int64)
    t_rows = np.stack([truth[i] for i in t_ids])
    assert len(ix) == len(t_ids)
    r_ids, r_rows = ix.read(0, len(ix))
    assert np.array_equal(r_ids, t_ids) and np.array_equal(r_rows, t_rows)
    check_against_oracle(ix, t_rows, t_ids, rows[[0, 204, 206]], max_dist=2e6)
    with pytest.raises(capi.PixelboxError):
        ix.append(np.arange(1000, 2000), rng.integers(0, 256, size=(1000, d), dtype=np.uint8))  # over capacity


def test_search_sees_rows_appended_after_a_search():
    rng = np.random.default_rng(41)
    rows = rng.integers(0, 256, size=(3000, 256), dtype=np.uint8)
    ids = np.arange(3000, dtype=np.int64)
    ix = capi.Index(256, 4000)
    ix.append(ids[:2000], rows[:2000])
    check_against_oracle(ix, rows[:2000], ids[:2000], rows[2500:2501])
    ix.append(ids[2000:], rows[2000:])
    check_against_oracle(ix, rows, ids, rows[2500:2501])


def test_bad_arguments_fail_loudly():
    with pytest.raises(capi.PixelboxError):
        capi.Index(0, 10)
    with pytest.raises(capi.PixelboxError):
        capi.Index(2048, 10)
    ix = capi.Index(256, 10)
    with pytest.raises(capi.PixelboxError):
        ix.search(np.zeros((1, 256), dtype=np.uint8), k=0)
    with pytest.raises(capi.PixelboxError):
        ix.search(np.zeros((1, 256), dtype=np.uint8), k=capi.PB_MAX_K + 1)
    with pytest.raises(capi.PixelboxError):
        ix.load(np.array([3, 2], dtype=np.int64), np.zeros((2, 256), dtype=np.uint8))


def test_append_device_rows():
    import torch

    rng = np.random.default_rng(52)
    rows = rng.integers(0, 256, size=(3000, 256), dtype=np.uint8)
    ids = np.arange(3000, dtype=np.int64) * 3 + 7
    ix = capi.Index(256, 4000)
    ix.append(ids[:1000], rows[:1000])
    d = torch.from_numpy(rows[1000:]).cuda()
    ix.append_device(ids[1000:2500], d.data_ptr())
    ix.append_device(ids[2500:], d[1500:].data_ptr())
    assert len(ix) == 3000
    got_ids, got_rows = ix.read(0, 3000)
    assert np.array_equal(got_ids, ids) and np.array_equal(got_rows, rows)
    check_against_oracle(ix, rows, ids, rows[[5, 1500, 2999]])
    with pytest.raises(capi.PixelboxError):
        ix.append_device(ids[10:12], d.data_ptr())  # not beyond the stored ids
    with pytest.raises(capi.PixelboxError):
        ix.append_device(np.array([20000, 19999], dtype=np.int64), d.data_ptr())  # not ascending
    assert len(ix) == 3000


def test_packed_device_results_and_merge_roundtrip():
    import torch

    rng = np.random.default_rng(50)
    rows = rng.integers(0, 256, size=(20000, 256), dtype=np.uint8)
    rows[7] = rows[19000]  # a tie, and a query with negative self-distance bits
    ids = np.arange(20000, dtype=np.int64) * 2
    q = np.stack([rows[19000], rng.integers(0, 256, size=256, dtype=np.uint8), 255 - rows[3]])
    k = 100
    # two shards of the same table; packed device messages; merge == oracle on the whole table
    halves = [(rows[:10000], ids[:10000]), (rows[10000:], ids[10000:])]
    msgs = []
    for r, i in halves:
        ix = make_index(r, i)
        t = torch.zeros((len(q), 2 * k + 1), dtype=torch.int64, device="cuda")
        ix.search_packed(q, k, 1e3, t.data_ptr())
        torch.cuda.synchronize()
        host = t.cpu().numpy()
        l_ids, l_d, l_c = ix.search(q, k, 1e3)
        assert np.array_equal(host[:, 2 * k], l_c)
        for qi in range(len(q)):
            c = int(l_c[qi])
            assert np.array_equal(host[qi, :c], l_ids[qi, :c])
            assert np.array_equal(host[qi, k : k + c].astype(np.uint32), l_d[qi, :c].view(np.uint32))
        msgs.append(host)
    g_ids, g_d, g_c = capi.topk_merge_packed(np.stack(msgs), k)
    for qi in range(len(q)):
        want_ids, want_d = oracle.scan_topk(q[qi], rows, ids, k, 1e3)
        assert g_c[qi] == len(want_ids)
        assert np.array_equal(g_ids[qi, : g_c[qi]], want_ids)
        assert np.array_equal(g_d[qi, : g_c[qi]].view(np.uint32), want_d.view(np.uint32))


def test_packed_burst_matches_host_results_including_fallback_queries():
    # the all-gather message of a burst (> 64 queries) comes from the concurrent-query path; queries whose
    # certificate failed are patched from the exhaustive pass before packing
    import torch

    rng = np.random.default_rng(51)
    n = 262144 + 5
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    v = rng.integers(0, 256, size=256, dtype=np.uint8)
    rows[1000:7000] = v  # 6000 duplicates: that query's candidate list overflows -> exhaustive pass
    ids = np.arange(n, dtype=np.int64) + 3
    q = rng.integers(0, 256, size=(150, 256), dtype=np.uint8)
    q[77] = v
    k = 50
    ix = make_index(rows, ids)
    t = torch.zeros((len(q), 2 * k + 1), dtype=torch.int64, device="cuda")
    ix.search_packed(q, k, 1e3, t.data_ptr())
    torch.cuda.synchronize()
    st = ix.stats()
    assert st.fallback + st.second_chance >= 1  # the duplicates' query did not certify at first
    host = t.cpu().numpy()
    ref = make_index(rows, ids, path=SINGLE)
    l_ids, l_d, l_c = ref.search(q, k, 1e3)
    assert np.array_equal(host[:, 2 * k], l_c)
    for qi in range(len(q)):
        c = int(l_c[qi])
        assert np.array_equal(host[qi, :c], l_ids[qi, :c])
        assert np.array_equal(host[qi, k : k + c].astype(np.uint32), l_d[qi, :c].view(np.uint32))
    assert np.array_equal(host[77, :k], np.arange(1003, 1003 + k))


# ---- concurrent-query pass (i8 MFMA, one pass over the table for up to 64 queries) ----------------------
@pytest.mark.parametrize("nq", [1, 15, 16, 17, 33, 64, 65, 100])
def test_multi_query_pass_vs_oracle(nq):
    rng = np.random.default_rng(60 + nq)
    n = 70000
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    ids = np.arange(n, dtype=np.int64) * 2 + 5
    q = rng.integers(0, 256, size=(nq, 256), dtype=np.uint8)
    q[0] = rows[123]
    ix = make_index(rows, ids, path=MULTI)
    check_against_oracle(ix, rows, ids, q)
    st = ix.stats()
    assert st.queries == nq and st.fast_path >= nq - 2  # uniform data: the certificates should pass


@pytest.mark.parametrize("nq", [65, 513, 600])
def test_multi_query_burst_vs_oracle(nq):
    # nq > 64 on a table of >= 262144 rows: the burst form -- one collect launch whose workgroups share each row
    # tile among 512 queries.  All results are compared with the one-pass-per-query path (itself oracle-checked
    # above and below), every 9th query and the planted ones with the oracle.
    rng = np.random.default_rng(160 + nq)
    n = 262144 + 77
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    ids = np.arange(n, dtype=np.int64) * 3 + 1
    q = rng.integers(0, 256, size=(nq, 256), dtype=np.uint8)
    q[0], q[1], q[nq - 1] = rows[123], rows[n - 1], rows[n - 70]
    ix = make_index(rows, ids, path=MULTI)
    got = ix.search(q, 100, 1e3)
    st = ix.stats()
    assert st.queries == nq and st.fast_path >= nq - 3
    ref = make_index(rows, ids, path=SINGLE)
    want = ref.search(q, 100, 1e3)
    assert all(np.array_equal(x, y) for x, y in zip(got, want))
    sel = sorted(set(range(0, nq, 9)) | {0, 1, nq - 1})
    check_against_oracle(ix, rows, ids, q[sel])


def test_burst_certifies_on_a_table_whose_neighbouring_rows_have_very_different_norms():
    # the burst's collect pass first tests a quad of four consecutive rows against the quad's loosest factors (smallest
    # norm, largest byte sum) and only then row by row: with near-grey and full-range rows interleaved the first stage
    # alone would pass whole quads, flood the survivor queue and send every query to the exhaustive pass (still the
    # right answers, at 20x the cost) -- the path counters must say that did not happen
    rng = np.random.default_rng(613)
    n, nq = 400_000, 600
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    grey = rng.integers(118, 138, size=(n // 2, 256), dtype=np.uint8)
    rows[1::2] = grey[: len(rows[1::2])]
    ids = np.arange(n, dtype=np.int64) + 1
    q = rng.integers(0, 256, size=(nq, 256), dtype=np.uint8)
    q[::7] = rows[rng.integers(0, n, size=len(q[::7]))]
    ix = make_index(rows, ids, path=MULTI)
    got = ix.search(q, 100, 1e3)
    st = ix.stats()
    assert st.queries == nq and st.fast_path >= int(0.9 * nq), (st.fast_path, st.second_chance, st.fallback)
    sel = [0, 7, 14, 300, 599]
    check_against_oracle(ix, rows, ids, q[sel])
    ref = make_index(rows, ids, path=EXACT).search(q[sel], 100, 1e3)
    for j, i in enumerate(sel):
        c = int(ref[2][j])
        assert np.array_equal(got[0][i, :c], ref[0][j, :c]) and np.array_equal(got[1][i, :c].view(np.uint32), ref[1][j, :c].view(np.uint32))


def test_burst_with_queries_anti_correlated_to_the_whole_table():
    # every row lies on the bright side of 128 and every query on the dark side: all cosines are negative, the sample pass of
    # the burst finds no positive threshold (k_mq_pick_tau clamps it at a small positive value instead of returning <= 0,
    # where the collect pass's group bound -- (4 max acc + max cr) / min W, an upper bound only for a non-negative numerator
    # -- could reject a quad holding a row that passes its own test: ADVICE r2).  engine.rs:587 maps every cosine <= 1e-6
    # to the distance 1 / 1e-6 - 1: with the default threshold nothing qualifies (count 0), with max_dist = 2e6 EVERY row
    # does, all at the same distance, and ORDER BY dist, image_id LIMIT 100 is the 100 smallest ids.  Whatever path each
    # query takes, the results must be the reference's.
    rng = np.random.default_rng(977)
    n, nq = 300_000, 130
    rows = (140 + rng.integers(0, 110, size=(n, 256))).astype(np.uint8)
    ids = np.arange(n, dtype=np.int64) * 3 + 5  # ascending, as pb_index_load requires (the table is read ORDER BY image_id)
    q = (116 - rng.integers(0, 110, size=(nq, 256))).astype(np.uint8)
    ix = make_index(rows, ids, path=MULTI)
    ex = make_index(rows, ids, path=EXACT)
    for max_dist in (1e3, 2e6):
        got = ix.search(q, 100, max_dist)
        want = ex.search(q, 100, max_dist)
        assert np.array_equal(got[2], want[2])
        for i in range(nq):
            c = int(got[2][i])
            assert np.array_equal(got[0][i, :c], want[0][i, :c]) and np.array_equal(got[1][i, :c].view(np.uint32), want[1][i, :c].view(np.uint32)), i
        if max_dist < 1e6:
            assert (got[2] == 0).all()
        else:
            assert (got[2] == 100).all() and (got[1][:, 0] > 9e5).all()
            assert np.array_equal(got[0][0], ids[:100])
        check_against_oracle(ix, rows, ids, q[[0, 64, 129]], max_dist=max_dist)
    assert ix.stats().queries == 2 * (nq + 3)


def test_multi_query_pass_tail_rows_and_auto_switch():
    rng = np.random.default_rng(61)
    n = 65536 + 13  # not a multiple of the 16-row tile
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    rows[-1] = rows[5]
    ids = np.arange(n, dtype=np.int64)
    q = np.concatenate([rows[[5, n - 1, n - 2]], rng.integers(0, 256, size=(9, 256), dtype=np.uint8)])
    ix = make_index(rows, ids)  # AUTO: 12 queries >= 8 -> shared pass
    check_against_oracle(ix, rows, ids, q)
    check_against_oracle(ix, rows, ids, q, k=7, max_dist=4.0)
    ix2 = make_index(rows, ids, path=SINGLE)
    a = ix.search(q, 100, 1e3)
    b = ix2.search(q, 100, 1e3)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_multi_query_burst_equals_per_chunk_passes_and_handles_tails():
    rng = np.random.default_rng(63)
    n = 262144 + 45  # 128-row steps: a partial last step with whole 16-row tiles past the end
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    ids = np.arange(n, dtype=np.int64) + 9
    q = rng.integers(0, 256, size=(130, 256), dtype=np.uint8)
    q[:4] = rows[[0, n - 1, n - 14, n - 33]]
    a = make_index(rows, ids, path=MULTI)
    b = make_index(rows, ids, path=MULTI)
    b.set_option(capi.PB_OPT_MQ_PER_CHUNK, 1)
    ra, rb = a.search(q, 100, 1e3), b.search(q, 100, 1e3)
    assert all(np.array_equal(x, y) for x, y in zip(ra, rb))
    check_against_oracle(a, rows, ids, q[:6])
    check_against_oracle(a, rows, ids, q[124:], k=5, max_dist=3.5)


def _tight_cluster_table(rng, n, n_cluster):
    """n uniform rows plus n_cluster rows that differ from one centre vector in a handful of bytes by +-1: thousands
    of rows whose cosines to the centre (and to each other) agree to ~1e-5 -- more near-ties around the 100th
    neighbour than any candidate list holds."""
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    centre = rng.integers(20, 236, size=256, dtype=np.uint8)
    where = rng.choice(n, size=n_cluster, replace=False)
    c = np.repeat(centre[None, :], n_cluster, axis=0).astype(np.int16)
    for i in range(n_cluster):
        idx = rng.choice(256, size=rng.integers(0, 4), replace=False)
        c[i, idx] += rng.choice([-1, 1], size=len(idx))
    rows[where] = c.astype(np.uint8)
    return rows, centre, where


@pytest.mark.parametrize("path", [SINGLE, MULTI])
def test_second_chance_answers_tight_clusters_exactly(path):
    # the first attempt finds 100 results but cannot certify them (9000 rows within ~1e-5 of the 100th cosine);
    # the second chance lists every row within the margin of that cosine and re-scores all of them
    rng = np.random.default_rng(70 + path)
    n = 120000
    rows, centre, where = _tight_cluster_table(rng, n, 9000)
    ids = np.arange(n, dtype=np.int64) * 2 + 1
    q = np.concatenate([centre[None, :], rows[where[:5]], rng.integers(0, 256, size=(6, 256), dtype=np.uint8)])
    ix = make_index(rows, ids, path=path)
    ix.set_option(capi.PB_OPT_SECOND_CHANCE, 1)  # always (on a table this small the cost model prefers the exhaustive pass)
    check_against_oracle(ix, rows, ids, q)
    st = ix.stats()
    assert st.second_chance >= 3 and st.queries == st.fast_path + st.second_chance + st.fallback
    # left to the cost model (default) the same queries take the exhaustive pass here: 120k rows cost it ~10 us a query
    ix3 = make_index(rows, ids, path=path)
    check_against_oracle(ix3, rows, ids, q[:3])
    assert ix3.stats().second_chance == 0 and ix3.stats().fallback >= 2
    # and with the second chance disabled the same queries take the exhaustive pass (same answers)
    os.environ["PB_NO_SECOND_CHANCE"] = "1"
    try:
        ix2 = make_index(rows, ids, path=path)
        ix2.set_option(capi.PB_OPT_SECOND_CHANCE, 1)  # the environment switch wins
        check_against_oracle(ix2, rows, ids, q[:3])
        assert ix2.stats().second_chance == 0 and ix2.stats().fallback >= 2
    finally:
        del os.environ["PB_NO_SECOND_CHANCE"]


@pytest.mark.parametrize("k", [100, 256])
def test_a_bursts_uncertified_queries_take_the_exhaustive_pass_in_one_queue(k):
    # a collection like the end-to-end leg's: a few distinct rows repeated tens of thousands of times.  Queries at those
    # rows tie with more rows than any list holds; the cost model sends all of them (more than one 64-query chunk, a
    # ragged last group of four) to the exhaustive pass at once, results written straight into the host arrays
    rng = np.random.default_rng(91 + k)
    n = 300000
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    protos = rng.integers(0, 256, size=(5, 256), dtype=np.uint8)
    dup = rng.choice(n, size=200000, replace=False)
    rows[dup] = protos[rng.integers(0, 5, size=len(dup))]
    ids = np.arange(n, dtype=np.int64) * 2 + 7
    nq = 150
    q = rng.integers(0, 256, size=(nq, 256), dtype=np.uint8)
    hot = rng.choice(nq, size=71, replace=False)  # 71 = 64 + 4 + 3
    q[hot] = protos[rng.integers(0, 5, size=len(hot))]
    q[hot[:20], 3] ^= 1  # near-duplicates of a prototype as well
    ix = make_index(rows, ids, path=MULTI)
    got = ix.search(q, k, 1e3)
    st = ix.stats()
    assert st.fallback >= 60 and st.second_chance == 0 and st.queries == nq == st.fast_path + st.fallback
    ref = make_index(rows, ids, path=EXACT)
    want = ref.search(q, k, 1e3)
    cnt = want[2]
    assert np.array_equal(got[2], cnt)
    for i in range(nq):
        c = int(cnt[i])
        assert np.array_equal(got[0][i, :c], want[0][i, :c]) and np.array_equal(got[1][i, :c].view(np.uint32), want[1][i, :c].view(np.uint32))
    check_against_oracle(ix, rows, ids, q[[hot[0], hot[25], hot[70]]], k=k)
    # the device-result entry point patches the same queries through the host arrays before it emits
    import torch

    d_ids = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    d_dist = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    d_cnt = torch.zeros((nq,), dtype=torch.int32, device="cuda")
    ix.search_device(q, k, 1e3, d_ids.data_ptr(), d_dist.data_ptr(), d_cnt.data_ptr())
    assert np.array_equal(d_cnt.cpu().numpy().astype(np.uint32), cnt)
    gi, gd = d_ids.cpu().numpy(), d_dist.cpu().numpy()
    for i in range(nq):
        c = int(cnt[i])
        assert np.array_equal(gi[i, :c], want[0][i, :c]) and np.array_equal(gd[i, :c].view(np.uint32), want[1][i, :c].view(np.uint32))


def test_cost_model_keeps_the_second_chance_for_large_tables_and_full_chunks():
    # 2.2M rows, 64 queries inside a tight cluster: here one more MFMA sweep for the chunk is cheaper than 64 exhaustive
    # passes, and it succeeds -- the default policy must still take it (PB_OPT_SECOND_CHANCE = 0)
    rng = np.random.default_rng(97)
    n = 2200000
    rows, centre, where = _tight_cluster_table(rng, n, 9000)
    ids = np.arange(n, dtype=np.int64) + 1
    q = rows[where[:64]].copy()
    ix = make_index(rows, ids, path=MULTI)
    got = ix.search(q, 100, 1e3)
    st = ix.stats()
    assert st.second_chance >= 32 and st.queries == 64 == st.fast_path + st.second_chance + st.fallback
    ix2 = make_index(rows, ids, path=MULTI)
    ix2.set_option(capi.PB_OPT_SECOND_CHANCE, 2)
    want = ix2.search(q, 100, 1e3)
    assert ix2.stats().second_chance == 0
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    check_against_oracle(ix, rows, ids, q[:2])


def test_second_chance_in_a_burst_and_list_overflow():
    rng = np.random.default_rng(73)
    n = 262144 + 3
    rows, centre, where = _tight_cluster_table(rng, n, 9000)
    v = rng.integers(0, 256, size=256, dtype=np.uint8)
    dup = rng.choice(np.setdiff1d(np.arange(n), where), size=70000, replace=False)
    rows[dup] = v  # 70000 exact duplicates: more than the second chance's 65536-entry list -> exhaustive pass
    ids = np.arange(n, dtype=np.int64) + 1
    q = rng.integers(0, 256, size=(90, 256), dtype=np.uint8)
    q[3], q[40], q[41], q[77] = centre, rows[where[0]], rows[where[1]], v
    ix = make_index(rows, ids, path=MULTI)
    ix.set_option(capi.PB_OPT_SECOND_CHANCE, 1)
    check_against_oracle(ix, rows, ids, q[[3, 40, 41, 77, 0, 89]])
    ix.stats(reset=True)
    got = ix.search(q, 100, 1e3)
    st = ix.stats()
    assert st.second_chance >= 3 and st.fallback >= 1
    ref = make_index(rows, ids, path=EXACT)
    want = ref.search(q, 100, 1e3)
    assert all(np.array_equal(x, y) for x, y in zip(got, want))


def test_multi_query_burst_survivor_queue_overflow():
    # forty identical queries and 3000 consecutive exact duplicates of them: a 128-row step then holds 1280 survivor
    # quads, more than the burst kernel's LDS queue; the affected lists are flagged and those queries re-run
    # exhaustively
    rng = np.random.default_rng(64)
    n = 262144 + 19
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    v = rng.integers(0, 256, size=256, dtype=np.uint8)
    rows[5000:8000] = v
    ids = np.arange(n, dtype=np.int64) + 1
    q = rng.integers(0, 256, size=(80, 256), dtype=np.uint8)
    q[:40] = v
    ix = make_index(rows, ids, path=MULTI)
    got_ids, got_d, got_c = ix.search(q, 100, 1e3)
    st = ix.stats()
    assert st.fallback + st.second_chance >= 1  # flagged lists do not certify; the duplicates fit the second chance
    assert all(np.array_equal(got_ids[i, :100], np.arange(5001, 5101)) for i in range(40))
    check_against_oracle(ix, rows, ids, q[[0, 39, 40, 41, 60, 79]])


def test_multi_query_pass_adversarial_falls_back():
    rng = np.random.default_rng(62)
    n = 80000
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    q = rng.integers(0, 256, size=(20, 256), dtype=np.uint8)
    rows[rng.choice(n, size=6000, replace=False)] = q[3]  # 6000 exact duplicates of one query: candidate list overflows
    rows[1000:1300] = 255 - q[4]
    ids = np.arange(n, dtype=np.int64)
    ix = make_index(rows, ids, path=MULTI)
    check_against_oracle(ix, rows, ids, q)
    check_against_oracle(ix, rows, ids, q[3:6], max_dist=2e6)
    st = ix.stats()
    assert st.fallback + st.second_chance >= 1
    big = np.concatenate([q, rng.integers(0, 256, size=(70, 256), dtype=np.uint8)])  # the same through the burst form
    check_against_oracle(ix, rows, ids, big)
    # clustered, embedding-like table
    centers = np.tanh(rng.standard_normal((30, 256)).astype(np.float32))
    which = rng.integers(0, 30, size=n)
    f = np.tanh(np.arctanh(np.clip(centers[which], -0.999, 0.999)) + rng.standard_normal((n, 256)).astype(np.float32) * 0.05)
    rows2 = oracle.quantize(f.astype(np.float32))
    ix2 = make_index(rows2, ids, path=MULTI)
    check_against_oracle(ix2, rows2, ids, rows2[rng.choice(n, size=24, replace=False)])


# ---- synthetic generator and BASELINE-size checks -------------------------------------------------------
def test_device_synthetic_fill_matches_the_stream():
    ix = capi.Index(256, 5000)
    ix.fill_synthetic(synth.SEED_INDEX, 0, 3000, 1)
    ix.fill_synthetic(synth.SEED_INDEX, 3000, 2000, 3001)
    ids, rows = ix.read(0, 5000)
    assert np.array_equal(ids, np.arange(1, 5001))
    assert np.array_equal(rows.reshape(-1), oracle.fill_synthetic(synth.SEED_INDEX, 0, 5000 * 256))


def test_config2_1m_rows_vs_oracle():
    # BASELINE.json configs[1]: 1M x 256 u8 index, batch-1 query, k = 100, max_dist = 1e3
    n, d = 1_000_000, 256
    ix = capi.Index(d, n)
    ix.fill_synthetic(synth.SEED_INDEX, 0, n, 1)
    rows = synth.fill_synthetic(synth.SEED_INDEX, 0, n * d).reshape(n, d)
    ids = np.arange(1, n + 1, dtype=np.int64)
    q = synth.fill_synthetic(synth.SEED_QUERY, 0, d)
    check_against_oracle(ix, rows, ids, q.reshape(1, -1))
    assert ix.stats().fast_path == 1
    ix.set_option(capi.PB_OPT_SEARCH_PATH, EXACT)
    check_against_oracle(ix, rows, ids, q.reshape(1, -1))


def test_full_size_10m_properties():
    # 10M x 256 (BASELINE metric size): too slow for the oracle on every row, so check size-independent
    # properties: (a) filter path == exhaustive exact path, bit for bit; (b) planted exact duplicates of the
    # query come first, in image_id order, with the reference's self-distance; (c) the result restricted to
    # a 200k-row window equals the oracle on that window's candidates.
    n, d = 10_000_000, 256
    ix = capi.Index(d, n + 8)
    ix.fill_synthetic(synth.SEED_INDEX, 0, n, 1)
    q = synth.fill_synthetic(synth.SEED_QUERY, 0, d)
    ids_a, d_a = ix.search_one(q)
    assert ix.stats().fast_path == 1
    ix.set_option(capi.PB_OPT_SEARCH_PATH, EXACT)
    ids_b, d_b = ix.search_one(q)
    assert np.array_equal(ids_a, ids_b) and np.array_equal(d_a.view(np.uint32), d_b.view(np.uint32))
    assert len(ids_a) == 100 and np.all(np.diff(d_a) >= 0)
    # every reported distance is the oracle's distance for that row (rows regenerated from the stream)
    for i in (0, 1, 50, 99):
        r = int(ids_a[i]) - 1
        row = synth.fill_synthetic(synth.SEED_INDEX, r * d, d)
        assert oracle.cosine_distance(q, row).view(np.uint32) == d_a[i].view(np.uint32)
    # one full oracle top-100 at the metric's size: 10M rows through the CPU restatement, a million rows at a time
    # (~3 s of CPU), merged with the reference's (dist, image_id) order
    cand_ids, cand_d = [], []
    step = 1_000_000
    for lo in range(0, n, step):
        rows = synth.fill_synthetic(synth.SEED_INDEX, lo * d, step * d).reshape(step, d)
        oi, od = oracle.scan_topk(q, rows, np.arange(lo + 1, lo + step + 1, dtype=np.int64), 100, 1e3)
        cand_ids.append(oi)
        cand_d.append(od)
    cand_ids, cand_d = np.concatenate(cand_ids), np.concatenate(cand_d)
    order = np.lexsort((cand_ids, cand_d))[:100]
    assert np.array_equal(ids_a, cand_ids[order])
    assert np.array_equal(d_a.view(np.uint32), cand_d[order].view(np.uint32))
    # the metric's size uses the launch form whose partition is timing-dependent (chunked stealing, lead / shift geometry chosen
    # for this size): 150 one-query calls, every one certified by the filter pass (rows_seen: every row exactly once) and equal
    # to the oracle's answer above, plus three other queries 50 times each against their own first answer (= the exhaustive pass)
    ix.set_option(capi.PB_OPT_SEARCH_PATH, AUTO)
    before = ix.stats()
    for rep in range(150):
        ids_r, d_r = ix.search_one(q)
        assert np.array_equal(ids_r, ids_a) and np.array_equal(d_r.view(np.uint32), d_a.view(np.uint32)), rep
    after = ix.stats()
    assert after.fast_path - before.fast_path == 150 and after.fallback == before.fallback
    for s in (1, 2, 3):
        q2 = synth.fill_synthetic(synth.SEED_QUERY, s * d, d)
        ix.set_option(capi.PB_OPT_SEARCH_PATH, EXACT)
        ids_e, d_e = ix.search_one(q2)
        ix.set_option(capi.PB_OPT_SEARCH_PATH, AUTO)
        for rep in range(50):
            ids_r, d_r = ix.search_one(q2)
            assert np.array_equal(ids_r, ids_e) and np.array_equal(d_r.view(np.uint32), d_e.view(np.uint32)), (s, rep)
    # (b) plant duplicates beyond the synthetic ids
    ix.append([n + 5, n + 6, n + 7], np.tile(q, (3, 1)))
    ids_c, d_c = ix.search_one(q)
    assert ids_c[:3].tolist() == [n + 5, n + 6, n + 7]
    assert np.all(d_c[:3] == oracle.cosine_distance(q, q))
    assert np.array_equal(ids_c[3:], ids_a[:97])


def test_concurrent_callers_like_the_reference_threads():
    # the reference calls mlhash from 4 crawler threads + the UI thread against one model (engine.rs:22,180,356)
    # and searches on one thread while another inserts (engine.rs:184-203,374): handles must serialise safely
    import threading

    from pixelbox_amd import weights as W

    rng = np.random.default_rng(70)
    d = 256
    rows = rng.integers(0, 256, size=(20000, d), dtype=np.uint8)
    ids = np.arange(20000, dtype=np.int64)
    ix = capi.Index(d, 30000)
    ix.load(ids[:10000], rows[:10000])
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 64, 64, 16)
    emb = capi.Embedder(blob, max_batch=4)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, 8, 64, 64)
    want_u8, _ = emb.embed(imgs)
    errors = []

    def hasher(t):
        try:
            for i in range(10):
                j = (t + i) % 8
                assert np.array_equal(emb.mlhash(imgs[j]), want_u8[j])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def inserter():
        try:
            for lo in range(10000, 20000, 500):
                ix.append(ids[lo : lo + 500], rows[lo : lo + 500])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def searcher():
        try:
            for i in range(15):
                got_ids, got_d = ix.search_one(rows[i])
                n_now = len(ix)
                # every result must be a valid row of some prefix >= 10000 of the table, best hit = the row itself
                assert got_ids[0] == i and got_d[0] <= 1e-6 and got_ids.max() < max(n_now, 10000)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=hasher, args=(t,)) for t in range(4)] + [threading.Thread(target=inserter), threading.Thread(target=searcher)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(ix) == 20000
    check_against_oracle(ix, rows, ids, rows[[3, 15000]])


# ---- the reference's other two blob distances over a phashes-like table -------------------
@pytest.mark.parametrize("metric", [capi.PB_METRIC_BYTE, capi.PB_METRIC_HAMMING])
@pytest.mark.parametrize("d", [32, 2, 33, 256])
def test_byte_and_hamming_scans(metric, d):
    rng = np.random.default_rng(90 + d + metric)
    n = 30000
    rows = rng.integers(0, 256, size=(n, d), dtype=np.uint8)
    rows[100] = rows[20000]
    q = np.stack([rows[20000], rng.integers(0, 256, size=d, dtype=np.uint8), 255 - rows[3]])
    ids = np.arange(n, dtype=np.int64) * 3
    ix = capi.Index(d, n, metric=metric)
    ix.load(ids, rows)
    got_ids, got_d, got_c = ix.search(q, 100, 0.45)
    for qi in range(len(q)):
        want_ids, want_d = oracle.scan_topk_metric(metric, q[qi], rows, ids, 100, 0.45)
        c = int(got_c[qi])
        assert c == len(want_ids)
        assert np.array_equal(got_ids[qi, :c], want_ids)
        assert np.array_equal(got_d[qi, :c].view(np.uint32), want_d.view(np.uint32))
    # the reference's hamming KATs through the ABI (engine.rs:693-701)
    if metric == capi.PB_METRIC_HAMMING and d == 2:
        ix2 = capi.Index(2, 4, metric=metric)
        ix2.load(np.arange(2), np.array([[0x0F, 0x0F], [0b01010101, 0b10101010]], dtype=np.uint8))
        i2, d2 = ix2.search_one(np.array([0xFF, 0x0F], dtype=np.uint8), 100, 10.0)
        assert i2.tolist() == [0, 1] and d2[0] == F32(0.25)


@pytest.mark.parametrize("metric", [capi.PB_METRIC_BYTE, capi.PB_METRIC_HAMMING])
def test_byte_and_hamming_coalesced_pass_and_its_fallback(metric):
    # dims the streaming skeleton accepts (powers of two, 16..1024) take the coalesced exact-key pass; its result
    # stands when the k-th key lies below every key a workgroup dropped, else the exhaustive pass answers
    rng = np.random.default_rng(95 + metric)
    n = 200000
    rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    rows[150000] = rows[7]
    ids = np.arange(n, dtype=np.int64) * 2 + 1
    q = np.stack([rows[7], rng.integers(0, 256, size=256, dtype=np.uint8), rows[199999]])
    ix = capi.Index(256, n, metric=metric)
    ix.load(ids, rows)
    for k, md in ((100, 0.6), (7, 0.3), (256, 10.0)):
        got_ids, got_d, got_c = ix.search(q, k, md)
        for qi in range(len(q)):
            want_ids, want_d = oracle.scan_topk_metric(metric, q[qi], rows, ids, k, md)
            c = int(got_c[qi])
            assert c == len(want_ids) and np.array_equal(got_ids[qi, :c], want_ids)
            assert np.array_equal(got_d[qi, :c].view(np.uint32), want_d.view(np.uint32))
    st = ix.stats()
    assert st.fast_path >= 6 and st.queries == st.fast_path + st.fallback
    # 5000 identical rows in one id range: far more ties than a workgroup lists -> drop bound fails -> exhaustive
    rows2 = rows.copy()
    rows2[40000:45000] = rows2[40000]
    ix2 = capi.Index(256, n, metric=metric)
    ix2.load(ids, rows2)
    g_ids, g_d, g_c = ix2.search(rows2[40000][None, :], 100, 0.6)
    w_ids, w_d = oracle.scan_topk_metric(metric, rows2[40000], rows2, ids, 100, 0.6)
    assert np.array_equal(g_ids[0, : g_c[0]], w_ids) and np.array_equal(g_d[0, : g_c[0]].view(np.uint32), w_d.view(np.uint32))


# ---- randomized sweep: table shapes, data shapes, k, max_dist, paths -----------------------------------------
def _random_table(rng, n, d, kind):
    if kind == "uniform":
        return rng.integers(0, 256, size=(n, d), dtype=np.uint8)
    if kind == "narrow":  # bytes around 128: tiny norms, the filter's worst case
        return rng.integers(120, 137, size=(n, d), dtype=np.uint8)
    if kind == "binary":  # saturated hashes: many exact ties
        return (rng.integers(0, 2, size=(n, d), dtype=np.uint8) * 255).astype(np.uint8)
    if kind == "clustered":
        centers = rng.integers(0, 256, size=(8, d), dtype=np.int16)
        which = rng.integers(0, 8, size=n)
        return np.clip(centers[which] + rng.integers(-3, 4, size=(n, d)), 0, 255).astype(np.uint8)
    if kind == "dups":
        base = rng.integers(0, 256, size=(max(1, n // 50), d), dtype=np.uint8)
        return base[rng.integers(0, len(base), size=n)]
    raise ValueError(kind)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PB_SWEEP_SEEDS", "24"))))
def test_randomized_sweep_against_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([16, 32, 64, 128, 256, 256, 256, 512, 1024, 48, 100]))  # the last two: no filter pass for that width
    n = int(rng.choice([1, 2, 63, 64, 65, 1000, 4097, 20000, 70000]))
    kind = str(rng.choice(["uniform", "narrow", "binary", "clustered", "dups"]))
    k = int(rng.choice([1, 7, 100, 255, 256]))
    max_dist = float(rng.choice([1e3, 1e3, 0.5, 3.0, 2e6, 1e-7]))
    path = int(rng.choice([AUTO, AUTO, SINGLE, MULTI, EXACT]))
    rows = _random_table(rng, n, d, kind)
    ids = np.sort(rng.choice(np.arange(10 * n + 10, dtype=np.int64), size=n, replace=False)) - 5 * n  # negative ids too
    nq = int(rng.choice([1, 3, 9, 70]))
    q = _random_table(rng, nq, d, kind)
    q[0] = rows[rng.integers(0, n)]
    ix = make_index(rows, ids, path=path)
    check_against_oracle(ix, rows, ids, q, k=k, max_dist=max_dist)
    st = ix.stats()
    assert st.queries == nq == st.fast_path + st.second_chance + st.fallback


@pytest.mark.parametrize("seed", range(int(os.environ.get("PB_SWEEP_SEEDS", "16"))))
def test_randomized_sweep_byte_and_hamming(seed):
    rng = np.random.default_rng(2000 + seed)
    metric = int(rng.choice([capi.PB_METRIC_BYTE, capi.PB_METRIC_HAMMING]))
    d = int(rng.choice([2, 16, 32, 33, 64, 256, 256, 1024]))
    n = int(rng.choice([1, 64, 1000, 30000, 120000]))
    kind = str(rng.choice(["uniform", "binary", "clustered", "dups"]))
    k = int(rng.choice([1, 100, 256]))
    max_dist = float(rng.choice([10.0, 0.45, 0.3, 1e-9]))
    rows = _random_table(rng, n, d, kind)
    ids = np.arange(n, dtype=np.int64) * 2 - n
    q = _random_table(rng, 3, d, kind)
    q[0] = rows[rng.integers(0, n)]
    ix = capi.Index(d, n, metric=metric)
    ix.load(ids, rows)
    got_ids, got_d, got_c = ix.search(q, k, max_dist)
    for qi in range(3):
        want_ids, want_d = oracle.scan_topk_metric(metric, q[qi], rows, ids, k, max_dist)
        c = int(got_c[qi])
        assert c == len(want_ids) and np.array_equal(got_ids[qi, :c], want_ids)
        assert np.array_equal(got_d[qi, :c].view(np.uint32), want_d.view(np.uint32))


def test_launch_shapes_give_identical_results():
    # PB_OPT_SCAN_LAUNCH: 0 = one launch per query, 1 = queries side by side in one grid, 2 (default) = one launch in
    # which every workgroup answers the queries one after the other.  Same lists, same bits.
    rng = np.random.default_rng(77)
    n, d = 300_000, 256
    rows = _random_table(rng, n, d, "clustered")
    ids = np.arange(n, dtype=np.int64) + 5
    q = _random_table(rng, 37, d, "clustered")
    q[3] = rows[1234]
    ix = capi.Index(d, n)
    ix.load(ids, rows)
    ix.set_option(capi.PB_OPT_SEARCH_PATH, 2)
    outs = []
    for mode in (0, 1, 2):
        ix.set_option(capi.PB_OPT_SCAN_LAUNCH, mode)
        outs.append(ix.search(q, 100, 10.0))
    for o in outs[1:]:
        assert np.array_equal(o[2], outs[0][2])
        assert np.array_equal(o[0], outs[0][0])
        assert np.array_equal(o[1].view(np.uint32), outs[0][1].view(np.uint32))
    for qi in (0, 3, 36):
        want_ids, want_d = oracle.scan_topk(q[qi], rows, ids, 100, 10.0)
        c = int(outs[2][2][qi])
        assert c == len(want_ids) and np.array_equal(outs[2][0][qi, :c], want_ids)


def test_append_device_async_on_a_shared_stream():
    # PB_OPT_APPEND_ASYNC + PB_OPT_STREAM: appends queue their copies and norms on the producer's stream and return;
    # a search afterwards sees every row (and the error margin's running minimum) as if the appends had been waited for.
    import torch

    rng = np.random.default_rng(5)
    d, n, nb = 256, 6000, 500
    rows = _random_table(rng, n, d, "clustered")
    ids = np.arange(n, dtype=np.int64) * 3 + 7
    ref = capi.Index(d, n)
    ref.load(ids, rows)
    ix = capi.Index(d, n)
    s = torch.cuda.Stream()
    ix.set_option(capi.PB_OPT_STREAM, s.cuda_stream)
    ix.set_option(capi.PB_OPT_APPEND_ASYNC, 1)
    dev = torch.from_numpy(rows).cuda()
    with torch.cuda.stream(s):
        for lo in range(0, n, nb):
            staged = dev[lo:lo + nb].clone()  # produced on the same stream, consumed by the queued copy
            ix.append_device(ids[lo:lo + nb], staged.data_ptr())
    q = _random_table(rng, 9, d, "clustered")
    q[0] = rows[4321]
    got = ix.search(q, 100, 10.0)
    want = ref.search(q, 100, 10.0)
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0])
    assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    assert len(ix) == n


def test_search_device_leaves_results_on_the_device():
    # pb_index_search_device: ids / distances / counts written by a device kernel (no host round trip for the results),
    # unused slots id = INT64_MAX, dist = +inf; chunked (> 64 queries) and burst paths, empty index
    import torch

    rng = np.random.default_rng(11)
    d, n, k = 256, 70_000, 100
    rows = _random_table(rng, n, d, "uniform")
    ix = capi.Index(d, n)
    for nq, md in ((1, 1e3), (70, 1e3), (9, 2.9)):
        q = _random_table(rng, nq, d, "uniform")
        d_ids = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
        d_dist = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
        d_cnt = torch.full((nq,), 77, dtype=torch.int32, device="cuda")
        if len(ix) == 0:
            ix.search_device(q, k, md, d_ids.data_ptr(), d_dist.data_ptr(), d_cnt.data_ptr())
            assert np.all(d_cnt.cpu().numpy() == 0) and np.all(d_ids.cpu().numpy() == np.iinfo(np.int64).max)
            assert np.all(np.isinf(d_dist.cpu().numpy()))
            ix.load(np.arange(1, n + 1, dtype=np.int64), rows)
        ix.search_device(q, k, md, d_ids.data_ptr(), d_dist.data_ptr(), d_cnt.data_ptr())
        want = ix.search(q, k, md)
        cnt = d_cnt.cpu().numpy().astype(np.uint32)
        assert np.array_equal(cnt, want[2])
        gi, gd = d_ids.cpu().numpy(), d_dist.cpu().numpy()
        for i in range(nq):
            c = int(cnt[i])
            assert np.array_equal(gi[i, :c], want[0][i, :c])
            assert np.array_equal(gd[i, :c].view(np.uint32), want[1][i, :c].view(np.uint32))
            assert np.all(gi[i, c:] == np.iinfo(np.int64).max) and np.all(np.isinf(gd[i, c:]))


@pytest.mark.parametrize("qn", [0, 1, 2, 4])
@pytest.mark.parametrize("k", [100, 256])
def test_coalesced_exhaustive_pass_vs_oracle(qn, k):
    # k_scan_exact_co / k_scan_exact_co4: 64-row tiles staged through LDS, QN queries per sweep; ragged last tile, odd query counts,
    # k beyond 128 (the larger key buffers), duplicates of the query (ties broken by image_id)
    rng = np.random.default_rng(100 + qn)
    d, n = 256, 64 * 300 + 37
    rows = _random_table(rng, n, d, "clustered")
    ids = np.arange(n, dtype=np.int64) * 2 + 5
    q = _random_table(rng, 5, d, "clustered")
    q[1] = rows[n - 1]
    rows[100:104] = q[2]
    ix = capi.Index(d, n)
    ix.load(ids, rows)
    ix.set_option(capi.PB_OPT_SEARCH_PATH, EXACT)
    ix.set_option(capi.PB_OPT_EXACT_QN, qn)
    for md in (1e3, 0.5):
        got = ix.search(q, k, md)
        for i in range(len(q)):
            wi, wd = oracle.scan_topk(q[i], rows, ids, k, md)
            c = int(got[2][i])
            assert c == len(wi)
            assert np.array_equal(got[0][i, :c], wi) and np.array_equal(got[1][i, :c].view(np.uint32), wd.view(np.uint32))
    assert ix.stats().fallback == 10
    with pytest.raises(capi.PixelboxError):
        ix.set_option(capi.PB_OPT_EXACT_QN, 3)


def test_option_values_are_range_checked():
    ix = capi.Index(256, 16)
    for opt, bad in ((capi.PB_OPT_SCAN_LAUNCH, -1), (capi.PB_OPT_SCAN_LAUNCH, 3), (4, 16), (4, -1), (4, 6), (7, -5), (7, 100000),
                     (capi.PB_OPT_SECOND_CHANCE, 3), (capi.PB_OPT_SECOND_CHANCE, -1)):
        with pytest.raises(capi.PixelboxError):
            ix.set_option(opt, bad)
    # an append that runs out of capacity stores what fits, reports how many, and fails with PB_ERR_CAPACITY
    ix.append(np.arange(1, 11, dtype=np.int64), np.zeros((10, 256), np.uint8))
    import ctypes as C

    ids20 = np.arange(11, 31, dtype=np.int64)
    rows20 = np.zeros((20, 256), np.uint8)
    stored = C.c_uint64(77)
    rc = capi.lib().pb_index_append(ix._h, ids20.ctypes.data_as(C.POINTER(C.c_int64)), rows20.ctypes.data_as(C.POINTER(C.c_uint8)), 20, C.byref(stored))
    assert rc == -4 and stored.value == 6 and len(ix) == 16  # capacity 16
    assert capi.lib().pb_index_read(ix._h, 2**63, 2**63 + 3, None, None) == -1  # first + n wraps: PB_ERR_INVALID


@pytest.mark.parametrize("n_shards", [1, 2, 3])
def test_sharded_index_through_the_c_abi(n_shards):
    # pb_sharded_*: one process, N shards (all on device 0 here: the copy exchange; N distinct devices take the RCCL
    # all-gather), per-shard packed top-k, device merge -- against the oracle over the whole table and against a
    # single index; loads split by contiguous ranges, appends go to the least-full shard with INSERT OR IGNORE over
    # ALL shards
    rng = np.random.default_rng(40 + n_shards)
    d, n = 256, 30_011
    rows = _random_table(rng, n, d, "clustered")
    ids = np.arange(n, dtype=np.int64) * 3 + 11
    sh = capi.ShardedIndexC(d, n + 600, [0] * n_shards)
    assert sh.info()["n_shards"] == n_shards
    sh.load(ids, rows)
    tot, per = sh.sizes()
    assert tot == n and per.max() - per.min() <= n_shards and int(per.sum()) == n
    q = _random_table(rng, 70, d, "clustered")
    q[0] = rows[17]
    q[1] = rows[n - 1]
    for k, md in ((100, 1e3), (7, 0.5), (256, 2e6)):
        got = sh.search(q, k, md)
        for i in (0, 1, 2, 33, 69):
            wi, wd = oracle.scan_topk(q[i], rows, ids, k, md)
            c = int(got[2][i])
            assert c == len(wi)
            assert np.array_equal(got[0][i, :c], wi) and np.array_equal(got[1][i, :c].view(np.uint32), wd.view(np.uint32))
    assert sh.info()["n_exchanges"] >= 3
    # appends: 500 new rows + 20 ids that exist on some shard + a repeat inside the call -> 500 stored
    new_rows = _random_table(rng, 500, d, "clustered")
    new_ids = np.arange(500, dtype=np.int64) + 10_000_000
    app_ids = np.concatenate([new_ids, ids[::1500][:20], new_ids[:1]])
    app_rows = np.concatenate([new_rows, 255 - rows[::1500][:20], new_rows[:1]])
    assert sh.append(app_ids, app_rows) == 500
    all_rows, all_ids = np.concatenate([rows, new_rows]), np.concatenate([ids, new_ids])
    got = sh.search(new_rows[:4], 50, 1e3)
    for i in range(4):
        wi, wd = oracle.scan_topk(new_rows[i], all_rows, all_ids, 50, 1e3)
        assert np.array_equal(got[0][i, :len(wi)], wi) and np.array_equal(got[1][i, :len(wi)].view(np.uint32), wd.view(np.uint32))
    # capacity is the total: 100 slots are left
    with pytest.raises(capi.PixelboxError) as ei:
        sh.append(np.arange(200, dtype=np.int64) + 20_000_000, np.zeros((200, d), np.uint8))
    assert ei.value.code == -4
    assert len(sh) == n + 600
    st = sh.stats()
    assert st.queries > 0


def test_configs3_at_full_size_through_eight_shards_on_one_device():
    # BASELINE configs[3] -- 10M x 256 rows over EIGHT shards -- through the product's sharded code at full size, the eight
    # shards sharing this GPU (an 8-GPU node runs the same code with RCCL's all-gather in place of the copy exchange):
    # pb_sharded_fill_synthetic splits the synthetic stream by contiguous ranges (1.25M rows per shard), 64 queries go through
    # per-shard top-k + exchange + device merge; ids and distance bits must equal the single 10M-row index for all 64, and
    # the CPU oracle over the whole table for one.  Reference: engine.rs:363-396.
    n, d, k = 10_000_000, 256, 100
    sh = capi.ShardedIndexC(d, n, [0] * 8)
    sh.fill_synthetic(synth.SEED_INDEX, n, 1)
    tot, per = sh.sizes()
    assert tot == n and list(per) == [n // 8] * 8 and sh.info()["n_shards"] == 8
    one = capi.Index(d, n)
    one.fill_synthetic(synth.SEED_INDEX, 0, n, 1)
    q = np.stack([synth.fill_synthetic(synth.SEED_QUERY + i, 0, d) for i in range(64)])
    # a stored row (distance ~ 0 at the top), the last row of a shard and the first of the next one as queries as well
    for j, r in ((1, 123_456), (2, n // 8 - 1), (3, n // 8)):
        q[j] = synth.fill_synthetic(synth.SEED_INDEX, r * d, d)
    s_ids, s_d, s_c = sh.search(q, k, 1e3)
    o_ids, o_d, o_c = one.search(q, k, 1e3)
    assert np.array_equal(s_c, o_c) and int(s_c.min()) == k
    assert np.array_equal(s_ids, o_ids) and np.array_equal(s_d.view(np.uint32), o_d.view(np.uint32))
    assert [int(s_ids[j, 0]) for j in (1, 2, 3)] == [123_457, n // 8, n // 8 + 1]  # ids = row + 1
    # one query against the oracle over the whole table (2.56 GB of rows generated in 1M-row pieces)
    best_ids, best_d = np.empty(0, np.int64), np.empty(0, np.float32)
    step = 1_000_000
    for r0 in range(0, n, step):
        rows = synth.fill_synthetic(synth.SEED_INDEX, r0 * d, step * d).reshape(step, d)
        wi, wd = oracle.scan_topk(q[0], rows, np.arange(r0 + 1, r0 + step + 1, dtype=np.int64), k, 1e3)
        allv = sorted(zip(np.concatenate([best_d, wd]).tolist(), np.concatenate([best_ids, wi]).tolist()))[:k]
        best_d = np.array([v[0] for v in allv], dtype=np.float32)
        best_ids = np.array([v[1] for v in allv], dtype=np.int64)
    assert np.array_equal(s_ids[0], best_ids) and np.array_equal(s_d[0].view(np.uint32), best_d.view(np.uint32))
    assert sh.info()["n_exchanges"] >= 1


def test_sharded_index_single_shard_through_rccl(monkeypatch):
    # PB_SHARDED_FORCE_RCCL=1: a one-device communicator (ncclCommInitAll with n = 1) runs the real all-gather path
    monkeypatch.setenv("PB_SHARDED_FORCE_RCCL", "1")
    rng = np.random.default_rng(77)
    d, n = 256, 9000
    rows = _random_table(rng, n, d, "uniform")
    ids = np.arange(1, n + 1, dtype=np.int64)
    sh = capi.ShardedIndexC(d, n, [0])
    assert sh.info()["uses_rccl"]
    sh.load(ids, rows)
    q = _random_table(rng, 5, d, "uniform")
    got = sh.search(q, 100, 1e3)
    for i in range(5):
        wi, wd = oracle.scan_topk(q[i], rows, ids, 100, 1e3)
        assert np.array_equal(got[0][i, :len(wi)], wi) and np.array_equal(got[1][i, :len(wi)].view(np.uint32), wd.view(np.uint32))


def test_device_merge_kernel_matches_the_host_merge():
    import torch

    rng = np.random.default_rng(3)
    for n_lists, nq, k in ((1, 3, 100), (8, 40, 100), (8, 5, 256), (20, 4, 200)):  # the last: beyond the LDS staging
        cnt = rng.integers(0, k + 1, size=(n_lists, nq))
        g = np.zeros((n_lists, nq, 2 * k + 1), dtype=np.int64)
        for a in range(n_lists):
            for b in range(nq):
                c = int(cnt[a, b])
                dist = np.sort(rng.choice(np.array([0.0, 0.5, 1.25, 3.0, 999999.0, -1.1920929e-07], np.float32), size=c))
                idv = rng.choice(10**6, size=c, replace=False).astype(np.int64) * n_lists + a  # unique across lists
                order = np.lexsort((idv, dist))
                g[a, b, :c] = idv[order]
                g[a, b, k:k + c] = dist[order].view(np.uint32).astype(np.int64)
                g[a, b, :k][c:] = np.iinfo(np.int64).max
                g[a, b, k:2 * k][c:] = 0x7F800000
                g[a, b, 2 * k] = c
        want = capi.topk_merge_packed(g, k)
        dg = torch.from_numpy(g).cuda()
        got = capi.topk_merge_packed_device(0, dg.data_ptr(), n_lists, nq, k)
        assert np.array_equal(got[2], want[2])
        for b in range(nq):
            c = int(want[2][b])
            assert np.array_equal(got[0][b, :c], want[0][b, :c])
            assert np.array_equal(got[1][b, :c].view(np.uint32), want[1][b, :c].view(np.uint32))


def test_bulk_insert_in_any_order_is_one_merge_pass_per_call():
    # INSERT OR IGNORE with ids in descending / shuffled order, duplicates of stored ids and repeats inside the call: the new
    # pairs are appended and merged into image_id order in ONE permutation pass per call (round 1 shifted the tail per row)
    import time

    rng = np.random.default_rng(31)
    d, n = 256, 40_000
    rows = _random_table(rng, n, d, "uniform")
    ids = rng.permutation(np.arange(1, n + 1, dtype=np.int64) * 5)
    ix = capi.Index(d, n + 10)
    half = n // 2
    assert ix.append(ids[:half][::-1], rows[:half][::-1]) == half  # any order
    t0 = time.perf_counter()
    # second call: the other half shuffled, plus 100 ids that are stored already, plus a repeat of its own first pair
    mix_ids = np.concatenate([ids[half:], ids[:100], ids[half:half + 1]])
    mix_rows = np.concatenate([rows[half:], 255 - rows[:100], 255 - rows[half:half + 1]])
    assert ix.append(mix_ids, mix_rows) == n - half
    assert time.perf_counter() - t0 < 5.0
    got_ids, got_rows = ix.read(0, n)
    order = np.argsort(ids)
    assert np.array_equal(got_ids, ids[order]) and np.array_equal(got_rows, rows[order])  # first write won everywhere
    q = rows[order][[0, n // 3, n - 1]]
    check_against_oracle(ix, rows[order], ids[order], q)
    # ascending tail appends still take the fast path, and a row in the middle goes where it belongs
    assert ix.append([5 * n + 7, 3], np.stack([rows[0], rows[1]])) == 2
    got_ids, _ = ix.read(0, n + 2)
    assert got_ids[0] == 3 and got_ids[-1] == 5 * n + 7 and np.all(np.diff(got_ids) > 0)
