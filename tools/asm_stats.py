"""Diagnostic: per-kernel instruction mix of a device assembly file (hipcc -S --cuda-device-only)."""
import re
import sys
s = open(sys.argv[1]).read()
PATS = [("valu", r"^\s*v_"), ("ds", r"^\s*ds_"), ("vmem", r"^\s*(buffer|global)_"), ("salu", r"^\s*s_"), ("wait", r"s_waitcnt"), ("scratch", r"scratch_")]
KEYS = [("vgpr", r"\.vgpr_count"), ("spill", r"\.vgpr_spill_count"), ("lds", r"\.group_segment_fixed_size")]
for m in re.finditer(r"^(_Z\w+):\s*;", s, flags=re.M):
    n = m.group(1)
    if "kernel" not in n:
        continue
    body = s[m.start():s.index("s_endpgm", m.start())]
    meta = re.search(r"\.name:\s+" + re.escape(n) + r"\n(.*?)\.wavefront_size", s, flags=re.S)
    out = [n[12:72].ljust(60)]
    for k, p in PATS:
        out.append("%s %d" % (k, len(re.findall(p, body, flags=re.M))))
    for k, p in KEYS:
        mm = re.search(p + r":\s+(\d+)", meta.group(1)) if meta else None
        out.append("%s %s" % (k, mm.group(1) if mm else "?"))
    print("  ".join(out))

# with a second argument (substring of a kernel name): the instruction mix of every loop of the matching kernels
if len(sys.argv) > 2:
    for m in re.finditer(r"^(_Z\w+):\s*;", s, flags=re.M):
        n = m.group(1)
        if sys.argv[2] not in n:
            continue
        lines = [l.strip() for l in s[m.start():s.index("s_endpgm", m.start())].split("\n")]
        labels = {}
        for i, l in enumerate(lines):
            mm = re.match(r"^(\.LBB\d+_\d+):", l)
            if mm:
                labels[mm.group(1)] = i
        print(n[12:90])
        for i, l in enumerate(lines):
            mm = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if not mm or mm.group(1) not in labels or labels[mm.group(1)] >= i:
                continue
            seg = [x for x in lines[labels[mm.group(1)]:i + 1] if x and x[0] not in ".;" and not x.endswith(":")]
            cnt = {}
            for x in seg:
                op = x.split()[0]
                k = ("valu" if op.startswith("v_") else "ds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_"))
                     else "wait" if op.startswith(("s_waitcnt", "s_barrier", "s_nop")) else "salu")
                cnt[k] = cnt.get(k, 0) + 1
            print("  loop %s: %d instructions  %s" % (mm.group(1), len(seg), "  ".join("%s %d" % kv for kv in sorted(cnt.items()))))
