"""Prints the headline and the 16-bit legs of a bench line: python tests/tools/show_legs.py [file = gpurun_out/r5/b16.json]"""
import json, sys
d = json.loads(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r5/b16.json").read().strip().splitlines()[-1])
print("headline", round(d["value"], 1), "bags/s", round(d["ms_per_step"], 3), "ms")
for k in ("deform16", "deform16_tabfwd", "deform16_table"):
    x = d.get(k, {})
    t = x.get("timing", {})
    print(k, round(x.get("ms_per_step", 0), 3), "ms wall,", t.get("step_ms_events"), "host enqueue", round(t.get("host_enqueue_ms_per_step", 0), 2),
          {kk: round(v["avg_ms"], 3) for kk, v in x.get("kernel_events", {}).items()}, x.get("error"))
