"""Teacher-forced forward (feedforward_step / the metrics step) at bench scale: time + per-kernel breakdown (GPU box)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from care_amd import _lib, get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "msrvtt_base_ami"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dev = torch.device("cuda:0")
    opt = make_opt(config)
    model = get_framework(opt).eval()
    model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
    model.set_compute_dtype("bf16")
    model.to(dev)
    eng = model.engine()
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
    T = eng.T
    ids = torch.randint(4, opt["vocab_size"], (B, T), generator=gen, device=dev)
    ids[:, 0] = 1
    labels = torch.randint(4, opt["vocab_size"], (B, T), generator=gen, device=dev)

    def score():
        eng._begin_pass()
        enc = eng.encode(feats)
        return eng.score_teacher_forced(ids, labels, enc["encoder_hidden_states"], enc.get("semantic_hidden_states"),
                                        sem_embs=enc.get("semantic_embs"))

    for _ in range(3):
        score()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        score()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    d, ff, V, Lk = eng.d, eng.ff, eng.V, eng.Lk
    enc_fl = sum(2 * eng.rows_of[ch] * d * opt["dim_" + ch] for ch in eng.modality)
    step_fl = 2 * d * d * 6 + 4 * d * ff + 2 * d * V + 4 * Lk * d
    total_fl = enc_fl + 4 * Lk * d * d + sum(step_fl + 4 * t * d for t in range(1, T + 1))
    print("score_teacher_forced %s B=%d: %.3f ms per pass, %.1f K clips/s, %.1f TFLOP/s (%.1f%% of 2500)" %
          (config, B, dt * 1e3, B / dt / 1e3, total_fl * B / dt / 1e12, total_fl * B / dt / 2.5e13))
    _lib.TIMING = {}
    score()
    torch.cuda.synchronize()
    timing, _lib.TIMING = _lib.TIMING, None
    rows = sorted(((sum(s.elapsed_time(e) for s, e in ev), tag, len(ev)) for tag, ev in timing.items()), reverse=True)
    for ms, tag, k in rows:
        print("   %-22s %3d launches %8.3f ms" % (tag, k, ms))
    print("   tagged total %.3f ms" % sum(r[0] for r in rows))


if __name__ == "__main__":
    main()
