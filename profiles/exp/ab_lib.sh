# A/B of two builds of the library on ONE box (boxes of the pool differ by several per cent): bash profiles/exp/ab_lib.sh <other.so> <cfg>...
# alternates the in-tree library with <other.so>, two rounds, bench.py without the CPU legs
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
other=$1; shift
cp jamun_amd/libjamun_hip.so /tmp/lib_head.so
for round in 1 2; do
  for c in "$@"; do
    for which in head other; do
      if [ $which = head ]; then cp /tmp/lib_head.so jamun_amd/libjamun_hip.so; else cp "$other" jamun_amd/libjamun_hip.so; fi
      echo -n "$which "; python3 bench.py --config $c --no-cpu-baseline --no-secondary 2>/dev/null | python3 profiles/bench_brief.py
    done
  done
done
cp /tmp/lib_head.so jamun_amd/libjamun_hip.so
