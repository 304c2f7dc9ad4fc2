"""print `value` (+ optional keys) of a bench.py JSON line read from stdin: python bench.py ... | python tools/benchval.py [label]"""
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", "fps %.1f" % d["value"], "inflight", d.get("frames_in_flight"),
      {k: round(v, 3) for k, v in d["stage_ms_single_stream"].items()})
