"""dev: for every kernel in a gfx950 .s file, list basic blocks that contain scratch (spill) traffic and say whether the
block also contains MFMAs (i.e. whether the spill sits inside a K loop)."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\S+):\s*;\s*@', s, re.M)]
for i, (pos, name) in enumerate(starts):
    if pat not in name:
        continue
    body = s[pos: starts[i + 1][0] if i + 1 < len(starts) else len(s)]
    blocks, lab = {}, "entry"
    for l in body.split("\n"):
        mm = re.match(r'(\.LBB\d+_\d+):', l)
        if mm:
            lab = mm.group(1)
        b = blocks.setdefault(lab, [0, 0, 0])
        b[0] += 'v_mfma' in l
        b[1] += 'scratch_' in l
        b[2] += 1
    tot = sum(b[1] for b in blocks.values())
    print(name[:70], "mfma", sum(b[0] for b in blocks.values()), "scratch ops", tot)
    for k, b in blocks.items():
        if b[1]:
            print("   %-12s mfma %3d scratch %3d lines %4d" % (k, b[0], b[1], b[2]))
