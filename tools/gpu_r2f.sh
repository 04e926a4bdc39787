#!/bin/bash
# Round-2 GPU call F: smoke() + a second, longer fuzz campaign on the final kernels (fresh generator seeds).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r2f; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 700 python tools/fuzz_campaign.py 600 9000000 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
