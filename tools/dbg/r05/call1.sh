#!/bin/bash
# round 5, call 1: the whole GPU suite on the round's first tree (new: config-1 golden, reference scheduler on the engine, empty-matrix DTW),
# the default bench line (config3 / f16 / contract keys), decode-step baseline at 16 / 32 / 64 streams
set -x
O=gpurun_out/r05_call1; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
timeout 900 python bench.py --steps 20 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python tools/bench_decode.py --layers 32 --batches 16,32,64 --tokens 128 > $O/decode_b16_32_64.txt 2>&1
cat $O/decode_b16_32_64.txt | tail -4
