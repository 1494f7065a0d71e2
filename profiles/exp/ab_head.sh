# bitwise x-hat comparison and alternating bench of the in-tree library against scratch/libs/base_head.so
cd $GRAFT_REPO_ROOT
export JAMUN_NO_REBUILD=1
cp jamun_amd/libjamun_hip.so /tmp/new.so
cp scratch/libs/base_head.so jamun_amd/libjamun_hip.so; python3 profiles/exp/xhat_dump.py /tmp/xh_old.pt 2>&1 | grep -v amdgpu | tail -2
cp /tmp/new.so jamun_amd/libjamun_hip.so; python3 profiles/exp/xhat_dump.py /tmp/xh_new.pt 2>&1 | grep -v amdgpu | tail -2
python3 - <<'PY'
import torch
a=torch.load('/tmp/xh_old.pt'); b=torch.load('/tmp/xh_new.pt')
for k in a: print(k, 'bit-identical' if torch.equal(a[k],b[k]) else 'max diff %.3g' % (a[k]-b[k]).abs().max().item(), tuple(a[k].shape))
PY
for round in 1 2; do
  for c in "$@"; do
    cp scratch/libs/base_head.so jamun_amd/libjamun_hip.so
    echo -n "head $c "; python3 bench.py --config $c --no-cpu-baseline --no-secondary --no-e2e --no-sweep --no-also 2>/dev/null | python3 profiles/bench_brief.py
    cp /tmp/new.so jamun_amd/libjamun_hip.so
    echo -n "new  $c "; python3 bench.py --config $c --no-cpu-baseline --no-secondary --no-e2e --no-sweep --no-also 2>/dev/null | python3 profiles/bench_brief.py
  done
done
cp /tmp/new.so jamun_amd/libjamun_hip.so
