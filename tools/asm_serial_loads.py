"""dev: static scan of a gfx950 .s file for SERIALISED loads -- per kernel, how many `s_waitcnt vmcnt(0)` follow exactly ONE global / buffer load
(a load whose latency nothing overlaps), the number of loads and the largest batch issued back to back.  hipcc gives every conditional load
(`c ? *p : 0`, `if (ptr) v += *ptr`) inside an unrolled loop a block and a wait of its own; round 6 found k_ups' staging (24 dependent round trips
per thread and tile) and the IVF planner that way.  Make the loads unconditional (clamped index, aliased pointer) and select on the VALUE.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o x.s csrc/file.hip -Iinclude && python tools/asm_serial_loads.py x.s [substring]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(\S+):\s*;\s*@", s, re.M)]
rows = []
for i, (pos, name) in enumerate(starts):
    if pat not in name:
        continue
    body = s[pos: starts[i + 1][0] if i + 1 < len(starts) else len(s)]
    ins = [l.strip().split(";")[0].strip() for l in body.split("\n")]
    ins = [x for x in ins if x and not x.startswith(".") and not x.endswith(":")]
    loads = sum(1 for x in ins if x.startswith(("global_load", "buffer_load")))
    if loads < 4:
        continue
    single = since = batch = 0
    for x in ins:
        if x.startswith(("global_load", "buffer_load")):
            since += 1
            batch = max(batch, since)
        elif x.startswith("s_waitcnt") and "vmcnt(0)" in x:
            single += since == 1
            since = 0
    rows.append((single, loads, batch, name))
for single, loads, batch, name in sorted(rows, reverse=True):
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%3d single-load waits / %4d loads / largest batch %3d  %s" % (single, loads, batch, dn[:110]))
