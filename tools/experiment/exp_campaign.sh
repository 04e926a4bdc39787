# madsim_hip_run_campaign: batches in flight vs layout (compact = auto for the 4-node ping-pong, plain = state_mem 1)
cd $GRAFT_REPO_ROOT
GPU_MAX_HW_QUEUES=8 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, time
from madsim_amd import runtime, workload as W, _abi as A
runtime.init(0)
w, lim, _ = W.bench_case("pingpong")
n = 65536
for sm in (0, 1):
    lim.state_mem = sm
    for fl in (3, 4, 5, 6, 8):
        runtime.run_campaign(w, 1 << 40, 8 * n, n, fl, False, None, lim)
        best = 1e9
        for rep in range(3):
            r = runtime.run_campaign(w, (1 << 41) + rep * 100 * n, 60 * n, n, fl, False, None, lim)
            best = min(best, r.wall_s / 60 * 1e3)
        print("state_mem", sm, "in_flight", fl, "ms/batch", round(best, 4), "kernel_ms/batch", round(r.kernel_ms / 60, 3))
PY
