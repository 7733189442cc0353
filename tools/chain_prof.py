"""The chained beam step (csrc/decode_chain.hip) run eagerly a few times: one kernel per phase of the resident beam launch, so
`rocprofv3 --pmc ... -- python3 tools/chain_prof.py [clips]` gives hardware counters PER PHASE (the resident launch itself is
one kernel).  python tools/chain_prof.py [clips] [passes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from care_amd import get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
opt = make_opt("msrvtt_care_beam5")
model = get_framework(opt).eval()
model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
model.set_compute_dtype("bf16")
model.to(dev)
eng = model.engine()
eng.resident_beam_max_rows, eng.chain_beam_max_rows = 0, 1 << 20
gen = torch.Generator(device=dev)
gen.manual_seed(5)
feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
for _ in range(n):
    eng.translate_beam(feats, 5, 5, use_graph=False, lean=True, early_exit=False)
torch.cuda.synchronize()
assert eng.last_decode.get("chain")
