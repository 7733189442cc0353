"""engine.metrics_step (teacher-forced pass, 4096 x 29 of msrvtt_base_ami) on one stream and with the encoder + static K / V
chain on a side stream (CARE_TF_OVERLAP=1), alternating, ms per pass (GPU box).   python tools/tf_overlap_probe.py [clips] [rounds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from care_amd import get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
opt = make_opt("msrvtt_base_ami")
model = get_framework(opt).eval()
model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
model.set_compute_dtype(os.environ.get("MODE", "bf16"))
model.to(dev)
eng = model.engine()
gen = torch.Generator(device=dev)
gen.manual_seed(7)
feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
ids = torch.randint(4, opt["vocab_size"], (B, eng.T), generator=gen, device=dev)
ids[:, 0] = 1
labels = torch.randint(4, opt["vocab_size"], (B, eng.T), generator=gen, device=dev)


def timed(n=10):
    for _ in range(3):
        eng.metrics_step(feats, ids, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.metrics_step(feats, ids, labels)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for r in range(rounds):
    os.environ["CARE_TF_OVERLAP"] = "0"
    one = timed()
    os.environ["CARE_TF_OVERLAP"] = "1"
    two = timed()
    print("round %d: one stream %.3f ms, two streams %.3f ms" % (r, one, two), flush=True)
