cd $GRAFT_REPO_ROOT
run() { # label, env..., args
  local label=$1; shift
  env "$@" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
g=lambda t: k.get(t,{}).get('avg_us',0)
print('$label', d['value'], d['ms_per_step'], 'dxd_ln %.1f dxd_gemm %.1f ffn_ln %.1f ffn %.1f add_ln %.1f' % (g('step_dxd_ln'), g('step_dxd_gemm'), g('step_ffn_gemm_ln'), g('step_ffn_gemm'), g('step_add_ln')))"
}
for cfg in "msrvtt_care 16384" "msrvtt_base_ami 16384" "msrvtt_base_ami 8192"; do
  set -- $cfg
  A="python bench.py --config $1 --batch $2 --no-legs --no-cpu-baseline --steps 5"
  run "$1 B=$2 fused(default)" CARE_X=1 $A
  run "$1 B=$2 unfused as+splitk" CARE_LN_MIN_ROWS=1000000 $A
  run "$1 B=$2 unfused as+tileFFN2" CARE_LN_MIN_ROWS=1000000 CARE_FFN2_TILE_ROWS=4096 $A
  run "$1 B=$2 unfused tile all" CARE_LN_MIN_ROWS=1000000 CARE_FFN2_TILE_ROWS=4096 CARE_FORCE_TILE=1 $A
done
A="python bench.py --config msrvtt_care_beam5 --beam 5 --batch 4096 --no-legs --no-cpu-baseline --steps 5"
run "beam5 B=4096 fused(default)" CARE_X=1 $A
run "beam5 B=4096 unfused as+tileFFN2" CARE_LN_MIN_ROWS=1000000 CARE_FFN2_TILE_ROWS=4096 $A
