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
