for rg in "" 1 2; do
CARE_LN_RG=$rg timeout -k 5 200 python bench.py --no-legs --no-cpu-baseline --batch 16384 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('B=16384 RG=[$rg]', d['value'], d['ms_per_step'], k['enc_gemm'], 'dxd_ln %.1f ffn_ln %.1f' % (k['step_dxd_ln']['avg_us'], k['step_ffn_gemm_ln']['avg_us']))"
done
for rg in "" 2; do
CARE_LN_RG=$rg timeout -k 5 200 python bench.py --no-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('B=32768 RG=[$rg]', d['value'], d['ms_per_step'], k['enc_gemm'], 'dxd_ln %.1f ffn_ln %.1f' % (k['step_dxd_ln']['avg_us'], k['step_ffn_gemm_ln']['avg_us']))"
done
