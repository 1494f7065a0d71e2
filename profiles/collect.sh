# Collects the rocprofv3 evidence bench.py and DESIGN.md cite.  Run on the GPU box from the repo root:
#   bash profiles/collect.sh <tag> [cfg2|cfg2r|cfg3|cfg4|cfg5] [extra bench flag, e.g. --separable]
#                                                          -> gpurun_out/prof_<tag>/{kernel_stats_summary.csv, pmc_summary.txt, pmc_traffic.json}
# then copy the summaries to profiles/<tag>_*.  Kernel trace and every PMC set run in separate passes (no --sys-trace).
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
export TMPDIR=/tmp
tag=${1:?usage: collect.sh <tag> [config]}
cfg=${2:-cfg2}
extra=${3:-}
out=gpurun_out/prof_$tag
mkdir -p "$out"
BENCH="python3 bench.py --config $cfg $extra --no-cpu-baseline --no-secondary --no-e2e --no-sweep --no-also --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $BENCH --steps 200 --warmup 40 > $out/bench_under_rocprof.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv produced; see $out/bench_under_rocprof.log" >&2; exit 1; }
python3 - "$f" "$out/kernel_stats_summary.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:14]:
        w.writerow([r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
print(open(sys.argv[2]).read())
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$out/pmc_$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- $BENCH --steps 3 --warmup 1 --no-profile > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] || { echo "no counter_collection.csv for [$set]; see $d.log" >&2; exit 1; }
  python3 - "$f" >> $out/pmc_summary.txt <<'PY'
import csv, sys, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:48]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    key=(k, r['Dispatch_Id'])
    if key not in seen: seen.add(key); n[k]+=1
for k,v in agg.items():
    if k.startswith('void k_') or k.startswith('k_'):
        print(k, 'dispatches', n[k], {c: f"{x:.4g}" for c,x in v.items()})
PY
done
cat $out/pmc_summary.txt
python3 bench.py --config $cfg $extra --no-cpu-baseline --signature > $out/signature.json
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{name[:12]}*/**/*counter_collection.csv", recursive=True):
        tot = collections.defaultdict(float); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
        for k, v in tot.items():
            res[k][name + "_KB_per_dispatch"] = v / max(len(disp[k]), 1)
res["_build"] = json.load(open(f"{out}/signature.json"))  # bench.py quotes these counters only while the build and the kernel selection are the same
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if "conv" in k or "node" in k or k == "_build"}, indent=1))
PY
# the raw rocprofv3 outputs stay on the box: gpurun merges at most 64 MiB back, and three configs' traces exceed that (round 6: a whole call's results
# were dropped); the summaries above are what gets committed
rm -rf "$out/trace" "$out"/pmc_*/
du -sh "$out" >&2
