for v in "" ln16 "" ln16; do
lib=""; [ -n "$v" ] && lib=care_amd/dbg/libcare_hip_$v.so
CARE_HIP_LIB=$lib timeout -k 5 200 python bench.py --no-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('variant [$v]', d['value'], d['ms_per_step'], 'dxd_ln %.1f ffn_ln %.1f enc %.1f' % (k['step_dxd_ln']['avg_us'], k['step_ffn_gemm_ln']['avg_us'], k['enc_gemm']['avg_us']))"
done
