"""One line per bench.py JSON line on stdin: config, conformations/s, ms per step and the per-kernel-class averages (ms per launch).

    python bench.py --config cfg3 --no-cpu-baseline --no-secondary | python profiles/bench_brief.py
"""
import json
import sys

for line in sys.stdin.read().strip().splitlines():
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    name = d["metric"].rsplit(", ", 1)[-1]
    print(name, round(d["value"]), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d.get("kernel_avg_ms", {}).items()})
