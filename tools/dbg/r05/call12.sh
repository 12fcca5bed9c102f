#!/bin/bash
# round 5, call 12: the final tree once more - smoke, GPU suite, the driver's bench line (roofline.traffic now finds the round's PMC summary)
O=gpurun_out/r05_call12; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
/usr/bin/time -v timeout 900 python bench.py > $O/bench_driver_defaults.json 2> $O/bench_driver_defaults.err; grep -E "Elapsed|Maximum resident" $O/bench_driver_defaults.err
python -c "
import json; d=json.loads(open('$O/bench_driver_defaults.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['value_contract'], d['value_f16'], d['cpu_baseline']['kind'], d['config3']['p50_ms'])"
