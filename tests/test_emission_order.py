"""The replay of the reference's aggregation hash table (src/qlib/hash.h:225-287, 330-419: prime sizes, 60 % growth with a full
rehash in slot order, linear probing) decides the order an unsorted aggregation's rows come in.  The engine's tail replays it
cut into independent probe clusters on the host's worker pool (resql_amd/csrc/hostref.cpp); the sequential replay — the form
the oracle and the reference-made goldens pin — must give the same permutation: with and without growth, with colliding and
identical hashes, with clusters that wrap around the table's end, at sizes around the chunk and level boundaries."""
import ctypes as C

import numpy as np
import pytest

from resql_amd import engine


def _order(hashes: np.ndarray, min_size: int, parallel: bool) -> np.ndarray:
    L = engine.lib()
    h = np.ascontiguousarray(hashes, dtype=np.uint64)
    out = np.empty(h.size, dtype=np.uint32)
    rc = L.rsq_ref_emission_order(h.ctypes.data, h.size, min_size, 1 if parallel else 0, out.ctypes.data)
    assert rc == 0
    return out


CASES = [
    # (n, min_size, kind)
    (9_000, 2, "random"),               # grows from 5 slots
    (9_000, 40_000, "random"),          # no growth
    (60_000, 2, "random"),
    (60_000, 100_003, "random"),        # n right at the 60 % threshold region of 116731
    (70_039, 116_731, "random"),        # 116731 * 6 / 10 = 70038: one growth at the very last insert
    (70_038, 116_731, "random"),        # ... and none
    (250_000, 1_000, "random"),
    (250_000, 1_000, "small-range"),    # hashes in a narrow range: long clusters
    (120_000, 3, "multiples"),          # hashes that are multiples of many table sizes' factors: homes collide
    (50_000, 2, "identical-runs"),      # runs of identical hashes (Values::hash is symmetric in its keys)
    (40_000, 2, "end-heavy"),           # homes near the table's end: clusters wrap around
    (1_300_000, 1_200_000, "random"),   # the 10 B-row shard's shape: 2 M slots, one growth
]


@pytest.mark.parametrize("n,min_size,kind", CASES)
def test_cluster_parallel_replay_equals_the_sequential_one(n, min_size, kind):
    rng = np.random.default_rng(n * 31 + min_size)
    if kind == "random":
        h = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    elif kind == "small-range":
        h = rng.integers(0, 3 * n, n, dtype=np.uint64)
    elif kind == "multiples":
        h = rng.integers(0, 1 << 40, n, dtype=np.uint64) * np.uint64(5 * 11 * 23 * 47)
    elif kind == "identical-runs":
        h = np.repeat(rng.integers(0, 1 << 62, n // 50 + 1, dtype=np.uint64), 50)[:n]
        h = h[rng.permutation(n)]
    else:   # end-heavy: most hashes land in the last slots of whatever prime the table has
        primes = [5, 11, 23, 47, 97, 199, 409, 823, 1741, 3469, 6949, 14033, 28411, 57557, 116731, 236897]
        top = primes[-1]
        h = (rng.integers(0, 1 << 30, n, dtype=np.uint64) * np.uint64(top) + np.uint64(top) - rng.integers(1, 400, n).astype(np.uint64))
    seq = _order(h, min_size, False)
    par = _order(h, min_size, True)
    assert sorted(seq.tolist()) == list(range(n))
    assert np.array_equal(seq, par)


def test_small_inputs_take_the_sequential_replay():
    h = np.arange(100, dtype=np.uint64) * np.uint64(7919)
    assert np.array_equal(_order(h, 2, True), _order(h, 2, False))
    assert _order(np.empty(0, dtype=np.uint64), 2, True).size == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,min_size,kind", CASES)
def test_device_replay_equals_the_sequential_one(gpu_ctx, n, min_size, kind):
    """the same replay as kernels (resql_amd/csrc/devtail.hip: home slots -> counting sort -> running minimum for the carries ->
    one thread per probe cluster), level by level on the device"""
    rng = np.random.default_rng(n * 31 + min_size)
    if kind == "random":
        h = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    elif kind == "small-range":
        h = rng.integers(0, 3 * n, n, dtype=np.uint64)
    elif kind == "multiples":
        h = rng.integers(0, 1 << 40, n, dtype=np.uint64) * np.uint64(5 * 11 * 23 * 47)
    elif kind == "identical-runs":
        h = np.repeat(rng.integers(0, 1 << 62, n // 50 + 1, dtype=np.uint64), 50)[:n]
        h = h[rng.permutation(n)]
    else:
        top = 236897
        h = (rng.integers(0, 1 << 30, n, dtype=np.uint64) * np.uint64(top) + np.uint64(top) - rng.integers(1, 400, n).astype(np.uint64))
    seq = _order(h, min_size, False)
    out = np.empty(n, dtype=np.uint32)
    hh = np.ascontiguousarray(h, dtype=np.uint64)
    rc = gpu_ctx._L.rsq_ref_emission_order_device(gpu_ctx.h, hh.ctypes.data, n, min_size, out.ctypes.data)
    assert rc == 0, gpu_ctx._L.rsq_last_error(gpu_ctx.h)
    assert np.array_equal(seq, out)
