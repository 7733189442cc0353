python tools/argmax_bench.py 32768
for v in x16 x48; do CARE_HIP_LIB=care_amd/dbg/libcare_hip_$v.so python tools/argmax_bench.py 32768; done
