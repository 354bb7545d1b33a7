"""A third, independent evaluation of the seeded join plans of tests/test_gpu_fuzz_joins.py: pandas (merge / groupby) against
the oracle.  The GPU test holds the engine to the oracle on these plans, and the reference itself is wrong on some of them (cut
probe chains, tests/test_reference_defect.py) - so the oracle's answers are checked here against something that shares no code
and no hash table with either."""
import importlib.util
import os
from collections import Counter

import pandas as pd
import pytest

from oracle import orc

_spec = importlib.util.spec_from_file_location("fuzzjoins", os.path.join(os.path.dirname(__file__), "test_gpu_fuzz_joins.py"))
fuzzjoins = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzzjoins)


def _col(t, name):
    c = t.col(name)
    if c.data.ndim == 2:
        return [bytes(r).split(b"\0")[0].decode() for r in c.data]
    return c.data


def pandas_answer(plan, spec):
    d, f = plan.tables
    dd = pd.DataFrame({c.name: _col(d, c.name) for c in d.columns})
    ff = pd.DataFrame({c.name: _col(f, c.name) for c in f.columns})
    if "dx_below" in spec:
        dd = dd[dd.dx < spec["dx_below"]]
    if "fx_from" in spec:
        ff = ff[ff.fx >= spec["fx_from"]]
    kind = spec["key_kind"]
    if kind in ("narrow", "wide", "computed"):
        dd, ff = dd.assign(k=dd.dk), ff.assign(k=ff.fk)
    elif kind == "varchar":                     # VARCHAR(8) = VARCHAR(10): the same characters, trailing spaces included
        dd, ff = dd.assign(k=dd.ds), ff.assign(k=ff.fs)
    elif kind == "char":                        # CHAR(8) = CHAR(8): equal up to trailing spaces
        dd, ff = dd.assign(k=dd.ds.str.rstrip(" ")), ff.assign(k=ff.fs.str.rstrip(" "))
    else:                                       # VARCHAR(8) build, CHAR(7) probe: equal up to trailing spaces AND the build value 7 long
        dd = dd[dd.ds.str.len() == 7]
        dd, ff = dd.assign(k=dd.ds.str.rstrip(" ")), ff.assign(k=ff.fs.str.rstrip(" "))
    j = ff.merge(dd, on="k")
    if "sum_below" in spec:
        j = j[(j.dx + j.fx) < spec["sum_below"]]
    top = spec["top"]
    if top == "rows":
        return Counter(f"{a}|{b}|{c}|" for a, b, c in zip(j.dx, j.fx, j.fg))
    if top == "global":
        return Counter([f"{j.fx.sum()}|{len(j)}|{j.dx.max()}|"]) if len(j) else Counter()
    if top == "hash":
        g = j.assign(k2=j.dx + j.fg).groupby("k2").agg(s=("fx", "sum"), c=("fx", "size")).reset_index()
        return Counter(f"{a}|{b}|{c}|" for a, b, c in zip(g.k2, g.s, g.c))
    keys = ["dg", "fg"] if spec["dense_keys"] == 2 else ["fg"]
    g = j.groupby(keys).agg(s=("fx", "sum"), c=("fx", "size"), mx=("dx", "max")).reset_index()
    return Counter("|".join(str(v) for v in row) + "|" for row in g[keys + ["s", "c", "mx"]].itertuples(index=False))


@pytest.mark.parametrize("block", range(0, 520, 65))       # seeds 7000-7119 are the ones the GPU test runs; 400 more for the oracle alone
def test_oracle_equals_pandas_on_the_join_shapes(block):
    failures = []
    for seed in range(block, block + 65):
        plan, what, spec = fuzzjoins.make(7000 + seed, with_spec=True)
        got = Counter(orc.execute(plan).text.splitlines()[1:])
        if got != pandas_answer(plan, spec):
            failures.append((seed, what))
    assert not failures, failures
