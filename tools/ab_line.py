"""Print the timing fields of the last bench.py JSON line on stdin."""
import json, sys
d = json.loads(sys.stdin.readlines()[-1])
r = d["roofline"]
print("ms/step %.4f  bwd %.4f (avg %.4f)  fwd %.4f" % (d["ms_per_step"], r["median_ms"], r["avg_ms"], r.get("fwd_median_ms", float("nan"))))
