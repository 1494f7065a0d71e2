import json,sys
for l in sys.stdin.read().strip().splitlines():
    if not l.startswith("{"): continue
    d=json.loads(l); print(d["metric"][-6:], round(d["value"]), round(d["ms_per_step"],4), {k: round(v,4) for k,v in d.get("kernel_avg_ms",{}).items()})
