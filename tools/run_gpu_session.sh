mkdir -p gpurun_out/r2k
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "argmax or scoring or beam" > gpurun_out/r2k/t_k.log 2>&1; echo "kernel tests rc=$?"; tail -15 gpurun_out/r2k/t_k.log
python tools/argmax_bench.py 32768 16384 8192 65536 2>&1 | tail -4
CARE_V32_MIN_ROWS=100000000 python tools/argmax_bench.py 32768 16384 8192 2>&1 | tail -3
timeout 600 python bench.py --no-legs > gpurun_out/r2k/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r2k/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['kernels']['step_vocab_argmax'])"
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/r2k/t_all.log 2>&1; echo "all gpu tests rc=$?"; tail -5 gpurun_out/r2k/t_all.log
