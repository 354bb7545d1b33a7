"""The context's memory arenas (resql_amd/csrc/mempool.cpp) on the host: a C++ driver over malloc slabs (tests/cpp/arena_test.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arena_ranges_pending_coalescing_trim(tmp_path):
    exe = str(tmp_path / "arena_test")
    src = os.path.join(ROOT, "resql_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-I" + src,
                           os.path.join(ROOT, "tests", "cpp", "arena_test.cpp"), os.path.join(src, "mempool.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "arena_test ok" in out.stdout
