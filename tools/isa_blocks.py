"""Basic blocks of a gfx950 assembly file (tools/isa.sh): per block the instruction mix, and the backward branches (loops).
usage: python tools/isa_blocks.py FILE.s [--loops]"""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
blocks = []          # (label, [instrs])
cur = ("entry", [])
pos = {}
for l in lines:
    s = l.strip()
    if not s or s.startswith((";", ".", "//")) and not re.match(r"^\.LBB\d+_\d+:", s):
        if re.match(r"^\.LBB\d+_\d+:", s):
            pass
        else:
            continue
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [])
        continue
    if re.match(r"^[a-z_0-9]+:", s):       # function label
        continue
    cur[1].append(s.split(";")[0].strip())
blocks.append(cur)
for i, (lab, _) in enumerate(blocks):
    pos[lab] = i


def mix(ins):
    c = {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "smem": 0, "branch": 0, "wait": 0, "other": 0}
    for x in ins:
        op = x.split()[0] if x else ""
        if op.startswith(("v_",)): c["valu"] += 1
        elif op.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
        elif op.startswith(("s_waitcnt", "s_nop", "s_barrier")): c["wait"] += 1
        elif op.startswith(("s_load", "s_buffer_load")): c["smem"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op: c["other"] += 1
    return c


loops = []
for i, (lab, ins) in enumerate(blocks):
    for x in ins:
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", x)
        if m:
            t = m.group(1) or m.group(2)
            if t in pos and pos[t] <= i:
                loops.append((pos[t], i))
loops.sort(key=lambda p: (p[0], -p[1]))
print(f"{len(blocks)} blocks, {sum(len(b[1]) for b in blocks)} instructions, {len(loops)} backward branches")
for a, b in loops:
    tot = {}
    n = 0
    for _, ins in blocks[a:b + 1]:
        for k, v in mix(ins).items():
            tot[k] = tot.get(k, 0) + v
        n += len(ins)
    print(f"loop {blocks[a][0]} .. {blocks[b][0]} ({b - a + 1} blocks, {n} instrs): " + " ".join(f"{k}={v}" for k, v in tot.items() if v))
if "--blocks" in sys.argv:
    for lab, ins in blocks:
        print(lab, len(ins), mix(ins))
