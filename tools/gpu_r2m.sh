#!/bin/bash
# Round-2 GPU call M: the whole GPU suite, smoke(), the default bench line, the rocprofv3 summary of the final ping-pong kernel
# (after the xoshiro / accept-test rewrite) and a fuzz campaign on the final kernels.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r2m; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench ms/step', d['ms_per_step'], 'verified', d['verified_seeds'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])"
bash tools/prof_workload.sh r2m/pp "" full; tail -5 $O/pp/summary.txt
timeout 260 python tools/fuzz_campaign.py ${FUZZ_S:-180} 11000000 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
