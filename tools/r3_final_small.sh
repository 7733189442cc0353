#!/bin/bash
# small-batch evidence of the round: GPU tests, phase clocks of the resident decode, kernel traces, bench with legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sbr
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/sbr/pytest_gpu.txt
timeout 300 python tools/resident_prof.py 1 16 128 2>&1 | grep -v amdgpu.ids > gpurun_out/sbr/resident_phase_clocks.txt
bash tools/small_batch_trace.sh > gpurun_out/sbr/trace_tail.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 1500 python bench.py > gpurun_out/sbr/bench.json 2> gpurun_out/sbr/bench.err
tail -c 600 gpurun_out/sbr/bench.err
cat gpurun_out/sbr/pytest_gpu.txt
