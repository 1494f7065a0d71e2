# rocprofv3 kernel stats of one bench config:  TAG=x CFG=cfg2 bash profiles/kernel_stats.sh
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
export TMPDIR=/tmp JAMUN_NO_REBUILD=1
out=gpurun_out/kstat_${TAG:-x}
rm -rf "./$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --config ${CFG:-cfg2} --no-cpu-baseline --no-secondary --repeats 1 --steps 10 --warmup 2 > $out/log.txt 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), f'{float(r["AverageNs"])/1000:9.2f} us', r["Percentage"])
PY
