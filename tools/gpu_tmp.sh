#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp
( time python bench.py > gpurun_out/exp/default.json 2> gpurun_out/exp/default.err ) 2>&1 | grep real
python -c "import json; d=json.load(open('gpurun_out/exp/default.json')); print(d['ms_per_step'], d['value'], d['verified_seeds'], d['cpu_baseline'], d['extra']['first_fail']['time_to_first_fail_ms'], d['extra']['stream_trial_ms_per_step'])"
( time python bench.py --steps 100 --warmup 10 --measure-traffic --no-cpu-baseline > gpurun_out/exp/traffic.json 2> gpurun_out/exp/traffic.err ) 2>&1 | grep real
python -c "import json; d=json.load(open('gpurun_out/exp/traffic.json')); print(d['roofline']['traffic'], d['roofline']['traffic_detail'], d['roofline']['measured_hbm_gbps'])"; tail -3 gpurun_out/exp/traffic.err
