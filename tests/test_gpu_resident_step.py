"""RSQ_PERSISTENT_STEP=1 — the one-launch step's kernel kept on the chip and started by a doorbell (engine.cpp "the resident step").
The knob is read once per process, so each setting runs tools/resident_step.py in a child process; the tool itself checks every
step's answer against the first one, across a pause longer than the kernel waits, another query on the same context, and a partial
execution of the same query."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_tool(resident: bool, sf: float):
    env = dict(os.environ)
    env.pop("RSQ_PERSISTENT_STEP", None)
    if resident:
        env["RSQ_PERSISTENT_STEP"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "resident_step.py"), "--sf", str(sf), "--steps", "300", "--warmup", "10"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout + p.stderr
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("sf", [0.05, 1.0])
def test_resident_steps_answer_like_launched_ones(sf):
    launched = run_tool(False, sf)
    resident = run_tool(True, sf)
    assert launched["resident_step"] is False and resident["resident_step"] is True
    assert launched["answers_equal"] and resident["answers_equal"]
    assert resident["answer_sha1"] == launched["answer_sha1"]
    if sf == 1.0:
        assert resident["equals_reference_answer"] is True          # tests/golden/ref_full_q1_sf1.tbl
    assert resident["device_us_last_step"] > 0
