"""Pins the oracle's hash restatement against the reference's own vectors.

* TestHashRouterComputesDestination (pkg/sql/colflow/routers_test.go:698-781):
  int64 keys 0..1023, 4 outputs => {273, 252, 287, 212}.
* TestHashFunctionFamily (pkg/sql/colexec/colexechash/hash_utils_test.go:24):
  a different init value moves > 13/16 of the keys to another bucket.
* memhash(8 bytes) == memhash64 (hash.go:83-86 vs :131-134).
Exact memhash values over byte strings / floats have no literal vector in the
reference: "parity unpinned" for those (pinned only through join/agg results).
"""
import struct

import numpy as np

from cockroach_amd import coldata as cd


def test_router_known_answer(oracle):
    v = cd.Vec.from_numpy(cd.INT64, np.arange(1024, dtype=np.int64))
    dest, counts = oracle.distribute([v], [0], 4)
    assert counts.tolist() == [273, 252, 287, 212]
    h = oracle.hash_cols([v], init=1, nbuckets=4, width=64)
    assert np.array_equal(h.astype(np.int32), dest)


def test_memhash64_equals_memhash_8_bytes(oracle):
    rng = np.random.default_rng(0)
    for x in rng.integers(-2**63, 2**63 - 1, size=200, dtype=np.int64).tolist() + [0, 1, -1]:
        for seed in (0, 1, 2, 0xDEADBEEF, 2**64 - 1):
            assert oracle.memhash(struct.pack("<q", x), seed) == oracle.memhash64(x, seed)


def test_memhash_length_branches_differ(oracle):
    # every length branch of hash.go:76-123 is exercised and deterministic
    data = bytes(range(1, 200))
    seen = set()
    for ln in list(range(0, 70)) + [95, 96, 97, 128, 199]:
        h1 = oracle.memhash(data[:ln], 1)
        assert h1 == oracle.memhash(data[:ln], 1)
        seen.add(h1)
    assert len(seen) == 75


def test_hash_function_family(oracle):
    # hash_utils_test.go:24-68: numKeys = 1024, numBuckets = 16 ... keys that
    # stay in the same bucket under a different init value: <= 3/16
    n, nb = 1024, 16
    v = cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64))
    h1 = oracle.hash_cols([v], init=1, nbuckets=nb, width=32)
    h2 = oracle.hash_cols([v], init=2, nbuckets=nb, width=32)
    assert (h1 == h2).sum() <= n * 3 // 16


def test_mixed_width_ints_hash_equal(oracle):
    vals = np.array([0, 1, -1, 100, -32768, 32767], dtype=np.int64)
    cols = [cd.Vec.from_numpy(t, vals.astype(cd.NP_DTYPE[t])) for t in (cd.INT16, cd.INT32, cd.INT64)]
    hs = [oracle.hash_cols([c], width=32, nbuckets=256) for c in cols]
    assert np.array_equal(hs[0], hs[1]) and np.array_equal(hs[1], hs[2])


def test_float_zero_and_nan_hash(oracle):
    f = cd.Vec.from_numpy(cd.FLOAT64, np.array([0.0, -0.0, float("nan"), -float("nan"), 1.5]))
    h = oracle.hash_cols([f], width=64)
    assert h[0] == h[1] == h[2] == h[3]          # NaN -> 0 -> c1*(c0^h)
    c0, c1 = 33054211828000289, 23344194077549503
    assert int(h[0]) == (c1 * (c0 ^ 1)) % 2**64
    assert int(h[4]) == oracle.memhash64(struct.unpack("<q", struct.pack("<d", 1.5))[0], 1)


def test_null_leaves_hash_unchanged_and_32bit_truncation(oracle):
    a = cd.Vec.from_values(cd.INT64, [5, None, 7])
    b = cd.Vec.from_values(cd.BYTES, [b"x", b"yy", None])
    h = oracle.hash_cols([a, b], width=64)
    assert int(h[1]) == oracle.memhash(b"yy", 1)
    assert int(h[2]) == oracle.memhash64(7, 1)
    # T = uint32: truncated after every column (hash_utils.eg.go:76)
    h32 = oracle.hash_cols([a, b], width=32)
    p = oracle.memhash64(5, 1) & 0xFFFFFFFF
    assert int(h32[0]) == oracle.memhash(b"x", p) & 0xFFFFFFFF


def test_selection_vector(oracle):
    v = cd.Vec.from_numpy(cd.INT64, np.arange(100, dtype=np.int64))
    sel = np.array([3, 10, 99], dtype=np.int64)
    full = oracle.hash_cols([v], width=64)
    assert np.array_equal(oracle.hash_cols([v], sel=sel, width=64), full[sel])
