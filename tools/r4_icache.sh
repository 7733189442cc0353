#!/bin/bash
# I-cache behaviour of the resident kernels (rocprofv3 PMC pass; kernel-trace only)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4d
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQ_WAIT_IFETCH|SQ_IFETCH" | head -40 > $OUT/counters.txt
cd $GRAFT_REPO_ROOT
cat > /tmp/run_res.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from care_amd import get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict
dev = torch.device("cuda:0")
opt = make_opt("msrvtt_care")
model = get_framework(opt).eval()
model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
model.set_compute_dtype("bf16"); model.to(dev)
eng = model.engine()
for B, bm in ((1, 1), (128, 1), (1, 5), (128, 5)):
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
    for _ in range(3):
        if bm > 1: eng.translate_beam(feats, bm, bm, use_graph=False, lean=True)
        else: eng.translate_greedy(feats, use_graph=False, lean=True)
    torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU -d $OUT/pmc1 -o pmc1 --output-format csv -- python3 /tmp/run_res.py > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_SALU SQ_INSTS_SMEM -d $OUT/pmc2 -o pmc2 --output-format csv -- python3 /tmp/run_res.py > $OUT/pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r4d"
for tag in ("pmc1", "pmc2"):
    for f in glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.OrderedDict()
        for r in rows:
            k = r["Kernel_Name"]
            if "decode_resident" not in k: continue
            key = (k[:70], r.get("Grid_Size"), r["Counter_Name"])
            agg.setdefault(key, []).append(float(r["Counter_Value"]))
        with open(out + "/" + tag + "_summary.txt", "w") as fo:
            for k, v in agg.items():
                fo.write("%s grid=%s %s n=%d mean=%.1f\n" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
PY
