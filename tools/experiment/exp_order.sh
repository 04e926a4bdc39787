cd $GRAFT_REPO_ROOT
for lib in libmadsim_hip_base.so libmadsim_hip.so; do
MADSIM_HIP_LIB=$PWD/madsim_amd/$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import os
from madsim_amd import runtime, workload as W
runtime.init(0)
out,_=runtime.run_batch(W.pingpong(4,8),0,256)
import torch
try:
    torch.zeros(4,device="cuda"); print(os.environ["MADSIM_HIP_LIB"].split("/")[-1],"lib first, then torch: ok", torch.cuda.device_count())
except Exception as e: print(os.environ["MADSIM_HIP_LIB"].split("/")[-1],"lib first, then torch: FAIL", e)
PY
done
env | grep -i "hip\|rocr\|hsa\|gpu_" | head
