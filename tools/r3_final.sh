#!/bin/bash
# evidence of the final build of the round: GPU tests, smoke, bench (+ legs), profiles, small-batch traces
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1
timeout 1800 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
timeout 300 python tools/resident_prof.py 1 16 128 256 2>&1 | grep -v amdgpu.ids > gpurun_out/final/resident_phase_clocks.txt
bash tools/small_batch_trace.sh > gpurun_out/final/trace_tail.txt 2>&1
cd $GRAFT_REPO_ROOT
bash tools/r3_profiles.sh > gpurun_out/final/profiles_tail.txt 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/final/pytest_gpu.txt; tail -2 gpurun_out/final/smoke.txt; tail -c 300 gpurun_out/final/bench.json
