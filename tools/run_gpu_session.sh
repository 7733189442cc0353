timeout -k 5 60 ./tools/micro/mfma_valu_overlap
