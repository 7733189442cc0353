"""The training step of bench.py's `training_step` leg on its own (for rocprofv3 --kernel-trace --stats):
    python tools/train_prof.py [clips] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from care_amd import get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_input_ids, synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
opt = make_opt("msrvtt_care")
model = get_framework(opt)
model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
model.set_compute_dtype(os.environ.get("CARE_TRAIN_DTYPE", "fp32"))
model.to(dev)
model.train()
gen = torch.Generator(device=dev)
gen.manual_seed(5)
feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
ids = synth_input_ids(7, B, opt["max_len"] - 1, opt["vocab_size"]).to(dev)
batch = {"feats": feats, "input_ids": ids}
g = None


def step():
    global g
    for prm in model.parameters():
        prm.grad = None
    out = model(batch)
    if g is None:
        g = torch.randn_like(out["logits"]) * 1e-3
    torch.autograd.backward([out["logits"]], [g])


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print("training step, %d clips: %.3f ms" % (B, (time.perf_counter() - t0) / steps * 1e3))
