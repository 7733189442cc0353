mkdir -p gpurun_out/r2g
timeout 900 python bench.py > gpurun_out/r2g/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r2g/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(json.dumps(d.get('legs'), indent=1))"
tail -5 gpurun_out/r2g/bench.log | cut -c1-400
