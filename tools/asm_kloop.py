"""dev: static view of the MFMA loops of a kernel in a gfx950 .s file: for every basic block with >= MIN MFMAs, the
instructions between consecutive MFMAs (the fillers of each 32-cycle slot) and every s_waitcnt / s_nop in the block.
usage: asm_kloop.py file.s kernel-substring [min_mfma] [--full]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 16
full = '--full' in sys.argv
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\S+):\s*;\s*@', s, re.M)]
for i, (pos, name) in enumerate(starts):
    if pat not in name:
        continue
    body = s[pos: starts[i + 1][0] if i + 1 < len(starts) else len(s)]
    blocks, lab, order = {}, "entry", []
    for l in body.split("\n"):
        mm = re.match(r'(\.LBB\d+_\d+):', l)
        if mm:
            lab = mm.group(1)
        if lab not in blocks:
            blocks[lab] = []
            order.append(lab)
        t = l.strip()
        if t and not t.startswith(';') and not t.startswith('.') and not re.match(r'\S+:$', t):
            blocks[lab].append(t.split(';')[0].strip())
    print("==", name[:90])
    for lab in order:
        ins = blocks[lab]
        nm = sum(x.startswith('v_mfma') for x in ins)
        if nm < mn:
            continue
        loop = any(('s_cbranch' in x or 's_branch' in x) and lab in x for x in ins)
        gaps, cur = [], []
        for x in ins:
            if x.startswith('v_mfma'):
                gaps.append(cur)
                cur = []
            else:
                cur.append(x)
        tail = cur
        cnt = {}
        for x in ins:
            op = x.split()[0]
            cnt[op] = cnt.get(op, 0) + 1
        print("  %s%s: %d instr, %d mfma, %.2f non-mfma per mfma" % (lab, " (self-loop)" if loop else "", len(ins), nm, (len(ins) - nm) / nm))
        print("    ops:", ", ".join("%s x%d" % kv for kv in sorted(cnt.items(), key=lambda kv: -kv[1])))
        waits = [(gi, x) for gi, g in enumerate(gaps) for x in g if x.startswith('s_waitcnt') or x.startswith('s_nop')]
        print("    waits/nops (before mfma #):", "; ".join("%d:%s" % (gi, x.replace('s_waitcnt ', '')) for gi, x in waits[:60]))
        hist = {}
        for g in gaps[1:]:
            hist[len(g)] = hist.get(len(g), 0) + 1
        print("    fillers-per-gap histogram:", dict(sorted(hist.items())), "tail", len(tail))
        if full:
            for gi, g in enumerate(gaps):
                print("      [%3d] %s" % (gi, " | ".join(g)))
            print("      [tail] %s" % " | ".join(tail))
