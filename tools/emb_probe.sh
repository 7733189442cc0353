cd $GRAFT_REPO_ROOT
export EMB_CHECK=0
python3 tools/emb_bench.py 32768 $EMB_KS > gpurun_out/emb_abl.txt 2>&1
for f in tools/lib/*.so; do CARE_HIP_LIB=$PWD/$f python3 tools/emb_bench.py 32768 $EMB_KS 2>&1 | grep -v amdgpu.ids >> gpurun_out/emb_abl.txt; done
cat gpurun_out/emb_abl.txt
