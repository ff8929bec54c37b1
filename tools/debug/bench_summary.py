"""Developer tool: one-screen summary of a bench.py line (python tools/debug/bench_summary.py gpurun_out/x/bench.json)."""
import json, sys
d = json.load(open(sys.argv[1]))
print("headline %.1f M frames/s, %.3f ms; roofline %s frac %.3f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"]))
sec = d.get("secondary", {})
for k, v in sec.get("baseline_configs", {}).items():
    cb = v.get("cpu_baseline") or {}
    print("  %-44s %9.1f M/s %9.3f ms  %-20s frac %.3f  cpu1 %s  all %s  verified %s" % (k, v["value"] / 1e6, v["ms_per_step"], v["roofline"]["kernel"], v["roofline"]["frac"],
          cb.get("value"), (cb.get("all_cores") or {}).get("value"), (v.get("verified") or {}).get("items")))
for k, v in sec.items():
    if isinstance(v, dict) and "frames_per_s" in v:
        print("  %-44s %9.1f M/s" % (k, v["frames_per_s"] / 1e6))
if "single_call_ms" in sec:
    print("  single calls:", {k: v for k, v in sec["single_call_ms"].items() if isinstance(v, list)})
if "cpu_baseline" in d:
    print("  cpu:", d["cpu_baseline"]["value"], (d["cpu_baseline"].get("all_cores") or {}).get("value"))
