# per-wave timeline of workgroup 7: k_conv_dg from the -DJAMUN_STAMP build (default), k_conv_mf with XF=-DMF_TRACE:  XF=-DMF_TRACE CFG=cfg2 bash profiles/wave_timeline.sh
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
# the diagnostic build replaces the in-tree library: put the production build back on every exit path
trap 'env -u JAMUN_EXTRA_CFLAGS python3 jamun_amd/csrc/build.py > /dev/null' EXIT
export JAMUN_EXTRA_CFLAGS="${XF:--DJAMUN_STAMP}"
python3 jamun_amd/csrc/build.py > /dev/null
python3 - <<'PY'
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
for _ in range(2): smp.score(y)
lib=_lib.load(); buf=(C.c_uint64*8)()
_lib.check(lib.jamun_debug_stamps(buf))
smp.score(y)
torch.cuda.synchronize()
print(smp.stats(), file=sys.stderr)
_lib.check(lib.jamun_debug_stamps(buf))
PY
