"""Print the headline fields of a bench.py JSON line read from stdin."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(f"ms_per_step {d['ms_per_step']:.4f}  value {d['value']:.4g}  kernel avg {r['avg_ms'] * 1e3:.1f} us frac {r['frac']:.3f}  "
      f"warm {r.get('warm_avg_ms', 0) * 1e3:.1f} us frac {r.get('warm_frac', 0):.3f}  probes {r['timed_launches']}  settle {d.get('settle_steps')}")
