"""Where do rocprofv3's kernel duration and the engine's HIP events differ for the TPC-H Q1 kernel?  ("what's weak" 1: the
--stats average was 366 us where the events of the same box said 345 us and the driver-timed whole step 358 us.)

Runs the SAME bench command four ways on this box and prints one table:
  A  no profiler                                     -> the events' kernel_ms, ms_per_step
  B  rocprofv3 --kernel-trace (no --stats)           -> per-dispatch durations from the trace + the events under the profiler
  C  rocprofv3 --kernel-trace --stats                -> the stats file's average + per-dispatch durations + the events under the profiler
Per-dispatch figures: launches, first launch, mean with / without the first launch, median, min, max.
This script never touches the GPU itself; bench.py stands directly behind `--` of every rocprofv3 command.
usage: python3 tools/q1_profile_gap.py OUT.txt [steps]"""
import csv
import glob
import json
import os
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1]
steps = sys.argv[2] if len(sys.argv) > 2 else "50"
KERNEL = "rsq_p0_lineitem_aggregate"
bench = ["python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", steps, "--warmup", "5"]
env = dict(os.environ, TMPDIR="/tmp")


def run(tag, prefix):
    d = f"/tmp/gap_{tag}"
    shutil.rmtree(d, ignore_errors=True)
    cmd = [c.replace("@D", d) for c in prefix] + bench
    pr = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
    line = None
    for l in pr.stdout.splitlines():
        if l.startswith("{"):
            line = json.loads(l)
    return d, line, pr


def dispatches(d):
    """durations (us) of the Q1 kernel's dispatches in start order, from the kernel trace"""
    fs = glob.glob(d + "/*/*kernel_trace.csv")
    if not fs:
        return []
    rows = [r for r in csv.DictReader(open(fs[0])) if r["Kernel_Name"].startswith(KERNEL)]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]


def stats_avg(d):
    fs = glob.glob(d + "/*/*kernel_stats.csv")
    if not fs:
        return None, None
    for r in csv.DictReader(open(fs[0])):
        if r["Name"].startswith(KERNEL):
            return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    return None, None


lines = []
A = run("a", [])
B = run("b", ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", "@D", "--"])
Cc = run("c", ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", "@D", "--"])
lines.append(f"TPC-H Q1 at SF10, kernel {KERNEL}, bench.py --no-cpu-baseline --no-extras --steps {steps} --warmup 5, one box, back to back")
lines.append("run                                   | events kernel_ms | ms_per_step | dispatches | first us | mean us | mean w/o first | median | min | max | --stats AverageNs")
for tag, (d, line, pr) in (("A no profiler", A), ("B rocprofv3 --kernel-trace", B), ("C rocprofv3 --kernel-trace --stats", Cc)):
    ev = line["roofline"]["kernel_ms"] * 1e3 if line else float("nan")
    st = line["ms_per_step"] * 1e3 if line else float("nan")
    ds = dispatches(d) if tag[0] != "A" else []
    avg, calls = stats_avg(d) if tag[0] == "C" else (None, None)
    if ds:
        lines.append(f"{tag:37s} | {ev:13.1f} us | {st:8.1f} us | {len(ds):10d} | {ds[0]:8.1f} | {statistics.mean(ds):7.1f} | {statistics.mean(ds[1:]) if len(ds) > 1 else float('nan'):14.1f} | "
                     f"{statistics.median(ds):6.1f} | {min(ds):5.1f} | {max(ds):5.1f} | " + (f"{avg:.1f} us over {calls} calls" if avg else "-"))
        if len(ds) > 8:
            lines.append(f"{'':37s}   first 8 dispatches (us): " + ", ".join(f"{v:.1f}" for v in ds[:8]))
    else:
        lines.append(f"{tag:37s} | {ev:13.1f} us | {st:8.1f} us | {'-':>10s} | (no trace)" + ("" if line else f"   [no bench line; rc {pr.returncode}: {pr.stderr[-300:]}]"))
open(out_path, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
