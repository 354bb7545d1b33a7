"""The environment switches the library reads are exactly the ones tests/test_gpu_knobs.py flips on the GPU (CPU check of the list)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_switch_list_is_complete():
    """the switches read by the library == the ones flipped in tests/test_gpu_knobs.py (plus RSQ_DEVICE_TAIL_MIN / RSQ_FORCE_GENERIC /
    RSQ_MULTI_GENERAL_MERGE, which other test files flip, and the macros RSQ_CQ_NV / RSQ_LC_SLOTS, which are not environment variables); at most 30"""
    text = open(os.path.join(ROOT, "tests", "test_gpu_knobs.py")).read()
    flipped = set(re.findall(r'^    \("(RSQ_[A-Z0-9_]+)", "', text, re.M))
    found = set()
    src = os.path.join(ROOT, "resql_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".cpp", ".hip")):
            found |= set(re.findall(r'"(RSQ_[A-Z0-9_]+)"', open(os.path.join(src, f)).read()))
    # (RSQ_KCACHE_USED_LOG selects nothing: __graft_entry__.build() sets it to learn which code objects the build resolves, and prunes the rest)
    elsewhere = {"RSQ_DEVICE_TAIL_MIN", "RSQ_FORCE_GENERIC", "RSQ_MULTI_GENERAL_MERGE", "RSQ_CQ_NV", "RSQ_LC_SLOTS", "RSQ_KCACHE_USED_LOG"}
    assert found - elsewhere == flipped, sorted((found - elsewhere) ^ flipped)
    assert len(found) <= 30
