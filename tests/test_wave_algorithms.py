"""The wave-level algorithms of kernels/rsq_device.h restated lane by lane in Python (CPU, no GPU): what they must guarantee under
ANY set of active lanes.  A lane that is not active does not execute; a shuffle that reads from it returns garbage (modelled as a
random value), which the algorithm must never let through."""
import random


def bm_set_combined(active, w, mask, rnd):
    """rsq::bm_set_combined: returns the list of (word, bits) atomics the wave issues"""
    n = 64
    garbage = lambda: rnd.getrandbits(32)

    def shfl_up(vals, d):
        return [vals[l - d] if l - d >= 0 and active[l - d] else (vals[l] if l - d < 0 else garbage()) for l in range(n)]

    def shfl_down(vals, d):
        return [vals[l + d] if l + d < n and active[l + d] else (vals[l] if l + d >= n else garbage()) for l in range(n)]

    wp = shfl_up(w, 1)
    head = [1 if (l == 0 or not active[l - 1] or wp[l] != w[l]) else 0 for l in range(n)]
    m, f = list(mask), list(head)
    d = 1
    while d < 64:
        mo, fo = shfl_up(m, d), shfl_up(f, d)
        for l in range(n):
            if active[l] and l >= d and not f[l]:
                m[l] |= mo[l]
                f[l] |= fo[l]
        d <<= 1
    next_head = shfl_down(head, 1)
    out = []
    for l in range(n):
        if not active[l]:
            continue
        last = l == 63 or not active[l + 1] or next_head[l] != 0
        if last:
            out.append((w[l], m[l] & 0xffffffff))
    return out


def test_combined_bit_sets_set_exactly_the_requested_bits():
    rnd = random.Random(20240613)
    fewer = 0
    for trial in range(3000):
        density = rnd.choice([0.05, 0.3, 0.7, 1.0])
        active = [rnd.random() < density for _ in range(64)]
        kind = trial % 3
        if kind == 0:        # keys in ascending order, eight per word (TPC-H order keys)
            base = rnd.randrange(1 << 20)
            keys = [base + 4 * l for l in range(64)]
        elif kind == 1:      # random keys
            keys = [rnd.randrange(1 << 16) for _ in range(64)]
        else:                # runs with repeats and a few outliers
            keys = sorted(rnd.randrange(256) for _ in range(64))
        w = [k >> 5 for k in keys]
        mask = [1 << (k & 31) for k in keys]
        want = {}
        for l in range(64):
            if active[l]:
                want[w[l]] = want.get(w[l], 0) | mask[l]
        got = {}
        atomics = bm_set_combined(active, w, mask, rnd)
        for word, bits in atomics:
            assert bits != 0
            got[word] = got.get(word, 0) | bits
        assert got == want, trial
        n_active = sum(active)
        assert len(atomics) <= n_active
        if kind == 0 and n_active > 8:
            fewer += len(atomics) < n_active
    assert fewer > 500          # clustered keys do combine


def look_back(chain, i, total):
    """the wave look-back of k_rank_blocks_chained / k_scan_chained for chunk i as a generator: every `yield` is one round trip
    (64 chain words read at once); returns the exclusive base and publishes the inclusive total"""
    if i == 0:
        chain[0] = (total << 2) | 2
        return 0
    chain[i] = (total << 2) | 1
    base, hi = 0, i - 1
    while True:
        yield
        v = [chain[hi - l] if hi - l >= 0 else 2 for l in range(64)]
        ready = [(x & 3) != 0 for x in v]
        incl = [(x & 3) == 2 for x in v]
        if any(incl):
            f = incl.index(True)
            if all(ready[:f]):
                base += sum(x >> 2 for x in v[:f + 1])
                break
        elif all(ready):
            base += sum(x >> 2 for x in v)
            hi -= 64
    chain[i] = ((base + total) << 2) | 2
    return base


def test_decoupled_look_back_gives_every_chunk_its_exclusive_prefix():
    rnd = random.Random(7)
    for trial in range(60):
        n = rnd.choice([1, 2, 63, 64, 65, 130, 300, 1000])
        totals = [rnd.randrange(0, 5000) for _ in range(n)]
        chain = [0] * n
        bases = [None] * n
        # chunks start in index order (the hardware dispatches workgroups in order) but at most `resident` of them run at a time,
        # and the running ones take their round trips in random order
        resident = rnd.choice([1, 3, 16, 200])
        running, nxt, done = {}, 0, 0
        while done < n:
            while nxt < n and len(running) < resident:
                running[nxt] = look_back(chain, nxt, totals[nxt])
                nxt += 1
            i = rnd.choice(list(running))
            try:
                next(running[i])
            except StopIteration as e:
                bases[i] = e.value
                del running[i]
                done += 1
        want, acc = [], 0
        for t in totals:
            want.append(acc)
            acc += t
        assert bases == want, (trial, n, resident)
        assert [c & 3 for c in chain] == [2] * n and chain[-1] >> 2 == acc


def range_select(images, want, bins=2048):
    """the short candidate selection of aot_kernels.hip (k_topk_range_hist / _gather, k_topk_range_select): ONE histogram of the images
    stretched to their range; the bin that holds the `want`-th largest image and the bins above it are the candidates"""
    hi, lo = max(images), min(images)
    shift = 0
    if hi > lo:
        shift = 64 - (hi - lo).bit_length()                 # __builtin_clzll(hi - lo)
    digit = lambda u: (((u - lo) << shift) & ((1 << 64) - 1)) >> 53
    hist = [0] * bins
    for u in images:
        hist[digit(u)] += 1
    # the highest b with (rows in bins >= b) >= want; bin 0 (every row) when there are fewer than `want` rows
    chosen, running = 0, 0
    for b in range(bins - 1, -1, -1):
        if running < want <= running + hist[b]:
            chosen = b
        running += hist[b]
    return [i for i, u in enumerate(images) if digit(u) >= chosen]


def test_range_selection_returns_a_superset_of_the_leading_rows():
    rnd = random.Random(99)
    for trial in range(400):
        n = rnd.choice([1, 5, 100, 3000])
        kind = trial % 4
        if kind == 0:
            images = [rnd.getrandbits(64) for _ in range(n)]
        elif kind == 1:        # a narrow range high up (sums of similar magnitude)
            base = rnd.getrandbits(63) | (1 << 63)
            images = [base + rnd.randrange(1 << rnd.choice([3, 20, 40])) for _ in range(n)]
        elif kind == 2:        # heavy ties
            vals = [rnd.getrandbits(64) for _ in range(max(1, n // 50))]
            images = [rnd.choice(vals) for _ in range(n)]
        else:                  # all equal
            images = [rnd.getrandbits(64)] * n
        want = rnd.choice([1, 10, 100, 5000])
        cand = set(range_select(images, want))
        order = sorted(range(n), key=lambda i: -images[i])
        if want >= n:
            assert cand == set(range(n))
            continue
        threshold = images[order[want - 1]]
        must = {i for i in range(n) if images[i] >= threshold}        # the leading `want` rows and every tie of the last one
        assert must <= cand, trial
