"""Print a few fields of a bench.py JSON line read from stdin (development aid)."""
import json, sys
d = json.loads(sys.stdin.read())
r = d.get("roofline", {})
print(sys.argv[1] if len(sys.argv) > 1 else "", "ms/step %.3f" % d["ms_per_step"], "value %.4g" % d["value"],
      "roofline %.1f %s (%.1f%%)" % (r.get("achieved", 0), r.get("unit", ""), 100 * r.get("frac", 0)))
