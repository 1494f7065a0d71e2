# A/B of two builds of the library on ONE box (boxes of the pool differ by several per cent): bash profiles/exp/ab_lib.sh <other.so> <cfg>...
# alternates the in-tree library with <other.so>, two rounds, bench.py without the CPU legs
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
other=$1; shift
head=$(mktemp /tmp/lib_head.XXXXXX.so)
cp jamun_amd/libjamun_hip.so "$head"
trap 'cp "$head" jamun_amd/libjamun_hip.so' EXIT  # (a failing step or Ctrl-C must not leave the foreign library installed)
for round in 1 2; do
  for c in "$@"; do
    for which in head other; do
      if [ $which = head ]; then cp "$head" jamun_amd/libjamun_hip.so; else cp "$other" jamun_amd/libjamun_hip.so; fi
      echo -n "$which "; python3 bench.py --config $c --no-cpu-baseline --no-secondary --no-e2e --no-sweep 2>/dev/null | python3 profiles/bench_brief.py
    done
  done
done
