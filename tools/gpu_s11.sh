#!/bin/bash
O=$PWD/gpurun_out/s11; mkdir -p $O
cd $PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lifting" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python bench.py --workload c4 --no-cpu --no-secondary --steps 200 --warmup 50 > $O/c4.json 2> $O/c4.err
tail -c 400 $O/c4.json
