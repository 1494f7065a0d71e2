# usage: trace_tp.sh <cfg> <out>  — per-wave timelines of k_tprod_h and k_node_update_h from the -DTP_TRACE -DNH_TRACE build
cd $GRAFT_REPO_ROOT
export JAMUN_NO_REBUILD=1
cp jamun_amd/libjamun_hip.so /tmp/head.so
cp scratch/libs/tp_trace.so jamun_amd/libjamun_hip.so
CFG=$1 python3 - > $2 2>&1 <<'PY'
import ctypes as C, torch, sys, os
sys.path.insert(0,'.')
import bench
from jamun_amd import synth, native, _lib
from jamun_amd.data import WalkerBatch
from jamun_amd.model import Denoiser
dev=torch.device('cuda',0)
cfg=os.environ.get("CFG","cfg2")
batch = WalkerBatch.from_molecules(bench.workload_molecules(cfg, bench.CONFIGS[cfg]["walkers"])).to(dev)
model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint()).to(dev)
smp = model.sampler_for(batch, 0.04)
y = batch.pos + 0.04*torch.randn_like(batch.pos)
for _ in range(20): smp.score(y)
torch.cuda.synchronize()
lib=_lib.load(); buf=(C.c_uint64*8)()
for _ in range(5): smp.score(y)
torch.cuda.synchronize()
print(smp.stats(), file=sys.stderr)
_lib.check(lib.jamun_debug_stamps(buf))
PY
cp /tmp/head.so jamun_amd/libjamun_hip.so
