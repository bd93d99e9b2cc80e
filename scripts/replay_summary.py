"""one line per tools/isv_replay JSON record in a log (scripts/replay_bench.sh)"""
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(f"S={d['sequences']:5d} groups={d['groups']}: {d['frames_per_second']:9.1f} frames/s; mean step ms", {k: round(v, 2) for k, v in d["mean_step_ms"].items()})
    else:
        print(l.strip()[:200])
