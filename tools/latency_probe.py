"""Single-clip latency through the Translator seam (the reference's `translate.py --latency`): greedy and beam 5, B = 1,
host -> captions on the host, per call (GPU box)."""
import sys, time, torch
sys.path.insert(0, ".")
from care_amd import get_framework, get_translator
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict
dev = torch.device("cuda:0")
for cfg, beam in (("msrvtt_base_ami", 1), ("msrvtt_care", 1), ("msrvtt_care_beam5", 5)):
    opt = make_opt(cfg); opt["beam_size"] = beam
    model = get_framework(opt).eval()
    model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
    model.set_compute_dtype("bf16"); model.to(dev)
    tr = get_translator(opt)
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, 1)]
    for _ in range(5):
        tr.translate_batch([model], {"feats": feats})
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 50
    for _ in range(n):
        hyps, scores = tr.translate_batch([model], {"feats": feats})
    dt = (time.perf_counter() - t0) / n
    print("%-20s beam %d: %.3f ms per caption (translate_batch, results on the host), caption length %d" % (cfg, beam, dt * 1e3, len(hyps[0][0])))
