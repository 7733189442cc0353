import sys, time, os, torch
sys.path.insert(0, ".")
from care_amd.configs import make_opt, feat_shapes
from care_amd.synth import synth_feats, synth_state_dict
from care_amd import get_framework
from oracle import care_cpu
opt = make_opt("msrvtt_base_ami")
m = get_framework(opt)
P = synth_state_dict(0, [(k, tuple(v.shape)) for k, v in m.state_dict().items()])
print("cpus", os.cpu_count())
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    for B in (64, 128):
        f = synth_feats(2000, feat_shapes(opt, B))
        care_cpu.translate_batch(P, opt, f)
        t = time.perf_counter(); care_cpu.translate_batch(P, opt, f); dt = time.perf_counter() - t
        print("threads %3d B %3d: %.2f s -> %.1f captions/s" % (nt, B, dt, B / dt), flush=True)
