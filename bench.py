#!/usr/bin/env python3
"""Benchmark of the captioning hot path (BASELINE.json metric: greedy captions/sec + decoder-step us).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype bf16|fp16|fp32|fp16x3] [--config NAME]

One "step" = one full pass of the hot path over one batch of B synthetic clips per GPU:
per-modality embedding -> (concept head) -> cross-K/V projection -> 29 greedy decoder
steps with the fused vocabulary argmax, plus the RCCL all-gather of the per-rank results
(token ids, lengths, scores: the metrics-step exchange).  Inputs are
resident in HBM before the timed region.  Workload = BASELINE.json configs[1]
(MSRVTT Transformer/base, task Base, feats ViT, modality ami, bf16 greedy, 28-frame feats).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every
launch of each tagged kernel in an extra instrumented pass on the same workload;
`cpu_baseline` times the CPU oracle (the reference algorithm as written) on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TF = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3, "fp16x3": 2500.0 / 3}   # fp16x3: three 16-bit MFMA passes per product


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32768, help="clips per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32", "fp16x3"])
    ap.add_argument("--config", default="msrvtt_base_ami")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--beam", type=int, default=1, help="beam size > 1: time the beam-search pass instead (extra, "
                    "not the BASELINE metric; no roofline/cpu legs)")
    ap.add_argument("--lanes", type=int, default=1,
                    help="batch lanes on separate HIP streams inside the captured graph (engine.lanes_for): 2 gives "
                         "+5-8%% at B >= 4096; the default 1 keeps every kernel alone on the chip so that the "
                         "per-kernel roofline and profiles/ describe the timed pass exactly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the extra legs (other BASELINE configs, beam 5, fp32, B = 128, crafted-EOS early exit, "
                         "bf16 error) that rank 0 reports in `legs` beside the headline measurement at N = 1")
    ap.add_argument("--cpu-batch", type=int, default=128)
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch threads for the CPU oracle (16 was the fastest of 8..128 on the 2x64-core host)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: every rank joins a gloo group, exchanges synthetic caption records through the same "
                         "care_amd.sharding calls as the real run and rank 0 prints a line marked dry_run (the launcher "
                         "and the N > 1 plumbing under test on a CPU-only host; nothing is measured)")
    return ap.parse_args()


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _visible_gpus() -> int:
    """GPUs this process could use, counted WITHOUT touching the HIP runtime (the relaunch below must come from a process
    that never initialised the GPU): the render nodes of /dev/dri, cut by HIP / ROCR_VISIBLE_DEVICES when set."""
    import glob
    n = len(glob.glob("/dev/dri/renderD*"))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start `torch.distributed.run --nproc-per-node N bench.py ...` as a
    CHILD process (subprocess - never os.exec*: on this pool a process that has initialised the GPU must not be
    replaced, and this one stays a plain parent that has not touched the HIP runtime at all: the GPU count comes from
    /dev/dri), relay the child's output (rank 0's JSON line) and return its exit code.  Every rank also exits with
    `no GPU` on its own when its device is missing."""
    import subprocess

    if not args.dry_run:
        have = _visible_gpus()
        if have < args.gpus and not (have == 0 and os.path.exists("/dev/kfd")):  # (no render nodes listed but a KFD: unknown - let the ranks say)
            sys.stderr.write("bench.py: --gpus {} but this host has {} GPU(s) visible\n".format(args.gpus, have))
            return 2
    # --standalone: torchrun picks the rendezvous port itself (c10d on a free port of 127.0.0.1) - no port chosen here
    # and released before torchrun binds it, so concurrent self-launches cannot collide
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, world: int, rank: int) -> None:
    """The N-rank plumbing without a GPU (gloo): per-rank synthetic records -> pack -> ONE all-gather -> unpack."""
    from care_amd.sharding import all_gather_records, pack_records, unpack_records

    if world > 1:
        dist.init_process_group("gloo")
    B, T = min(args.batch, 64), 29
    gen = torch.Generator().manual_seed(1000 + rank)
    fed = torch.randint(4, 10547, (B, T + 1), generator=gen, dtype=torch.int32)
    length = torch.full((B,), T, dtype=torch.int32)
    score = torch.full((B,), -float(rank + 1))
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        tok, ln, sc = unpack_records(all_gather_records(pack_records(fed, length, score, B)))
    dt = (time.perf_counter() - t0) / max(1, args.steps)
    ranks_seen = sorted({int(round(-float(v))) - 1 for v in sc.tolist()})
    ok = tok.shape[0] == world * B and ranks_seen == list(range(world)) and torch.equal(tok[rank * B:(rank + 1) * B], fed)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(dict(metric="captions/sec (greedy)", value=None, unit="captions/s", n_gpus=world, steps=args.steps,
                              warmup=args.warmup, dry_run=True, backend="gloo", ranks_seen=ranks_seen,
                              records_gathered=int(tok.shape[0]), all_gather_us=round(dt * 1e6, 1), exchange_ok=bool(ok))),
              flush=True)
    if not ok:
        raise SystemExit(3)


def kernel_model(tag, eng, B, dtype):
    """Algorithmic bytes and flops of ONE launch of a tagged kernel."""
    d, H, ff, V, Lk, T = eng.d, eng.H, eng.ff, eng.V, eng.Lk, eng.T
    es = 2 if dtype in ("bf16", "fp16") else 4
    if tag == "step_cross_attn" and eng.latent_for(B):
        # absorbed form: ONE bf16 copy of the clip's memory, expanded query in, latent context out
        return dict(bytes=B * (Lk * d * 2 + 2 * H * d * 2), flops=B * 4.0 * H * Lk * d, bound="hbm")
    if tag == "step_head_expand":  # q [B,d] bf16 in, [B,H,d] bf16 out
        return dict(bytes=B * (d * 2 + H * d * 2) + d * d * 2, flops=2.0 * B * d * d, bound="hbm")
    if tag == "step_head_reduce":  # [B,H,d] bf16 in, context [B,d] bf16 out
        return dict(bytes=B * (H * d * 2 + d * 2) + d * d * 2, flops=2.0 * B * d * d, bound="hbm")
    if tag == "step_cross_attn":   # K and V of every clip once, q in, context out, bias
        return dict(bytes=B * (2 * Lk * d * es + 2 * d * 4), flops=B * 4 * Lk * d, bound="hbm")
    if tag == "step_self_attn":    # average over t = 1..T keys
        return dict(bytes=B * (2 * ((T + 1) / 2) * d * es + 2 * d * 4), flops=B * 4 * ((T + 1) / 2) * d, bound="hbm")
    if tag == "step_vocab_argmax":
        return dict(bytes=B * d * 4 + V * d * es, flops=2.0 * B * d * V, bound="mfma")
    if tag == "step_ffn_gemm":
        return dict(bytes=B * (d + ff) * 4 + d * ff * es, flops=2.0 * B * d * ff, bound="mfma")
    if tag in ("step_dxd_gemm", "step_dxd_ln"):
        return dict(bytes=B * 2 * d * 4 + d * d * es, flops=2.0 * B * d * d, bound="mfma")
    if tag == "step_ffn_gemm_ln":
        return dict(bytes=B * (ff * es + 3 * d * 4) + d * ff * es, flops=2.0 * B * d * ff, bound="mfma")
    if tag == "step_qkv_gemm":
        return dict(bytes=B * (d * 4 + d * 4 + 2 * d * es) + 3 * d * d * es, flops=2.0 * B * d * 3 * d, bound="mfma")
    if tag == "step_add_ln":       # fp32 GEMM output + fp32 residual in, fp32 + bf16 mirror out
        return dict(bytes=B * d * (4 + 4 + 4 + 2), flops=8.0 * B * d, bound="hbm")
    if tag == "step_update_embed":  # word row + position row in, fp32 + bf16 activations out (+ the arg-max partials)
        return dict(bytes=B * d * (4 + 4 + 4 + 2), flops=8.0 * B * d, bound="hbm")
    if tag == "cross_kv_gemm":
        return dict(bytes=B * Lk * (d * 4 + 2 * d * es) + 2 * d * d * es, flops=2.0 * B * Lk * d * 2 * d, bound="mfma")
    if tag == "decode_resident":
        # the whole decode of a small batch in one launch: per step every decoder weight and the vocabulary matrix once
        # (bf16), per row the projected cross K/V and the self-attention cache so far; a latency-bound launch (grid
        # barriers between phases), priced against HBM like the kernels it replaces
        nl = eng.n_layers
        wbytes = nl * (3 * d * d + d * d + 2 * d * d + 2 * d * ff) * 2 + V * d * 2
        row = nl * (2 * Lk * d * 2 + 2 * ((T + 1) / 2) * d * 2)
        return dict(bytes=T * (wbytes + B * row), flops=T * B * 2.0 * (nl * (6 * d * d + 2 * d * ff) + d * V), bound="hbm")
    return None


def small_batch_roofline(eng, clips, beam, step_us):
    """A decoder step of a resident launch against a STATED floor: every byte a step must read once -
    decoder + vocabulary weights (16-bit), the clips' static K/V (shared by the beams of a clip), the rows' self-attention
    cache so far - at the 6 TB/s this part sustains, or its FLOPs at the dense 16-bit MFMA peak, whichever is longer, PLUS the
    hand-offs of the step's dependent phases at the measured floor of a device-wide counter hand-off (1.9 us: two-level
    counters, profiles/r03_barrier_bench.txt).  frac = floor / measured step; what separates them is named in DESIGN.md 5.1."""
    d, ff, V, Lk, T, nl = eng.d, eng.ff, eng.V, eng.Lk, eng.T, eng.n_layers
    rows = clips * beam
    n_att = 2 if eng.attr_att else 1
    wbytes = nl * (3 * d * d + d * d + n_att * 2 * d * d + 2 * d * ff) * 2 + V * d * 2
    kv = nl * (clips * 2 * Lk * d * 2 + rows * 2 * ((T + 1) / 2) * d * 2)
    flops = rows * 2.0 * (nl * (4 * d * d + n_att * 2 * d * d + 2 * d * ff) + d * V + nl * (2 * Lk * d + (T + 1) * d))
    handoffs = nl * (3 + 3 * n_att + 2) + 1 + (1 if beam > 1 else 0)
    hbm_us, mfma_us, ho_us = (wbytes + kv) / 6.0e12 * 1e6, flops / (MFMA_PEAK_TF["bf16"] * 1e12) * 1e6, handoffs * 1.9
    floor = max(hbm_us, mfma_us) + ho_us
    return dict(bound="latency: dependent phase hand-offs + hbm", model="max(bytes / 6 TB/s, flops / 2.5 PFLOP/s) + hand-offs x 1.9 us",
                bytes_per_step=int(wbytes + kv), flops_per_step=int(flops), hbm_us=round(hbm_us, 2), mfma_us=round(mfma_us, 2),
                handoffs_per_step=handoffs, handoff_floor_us=1.9, floor_us=round(floor, 2), achieved_us=round(step_us, 2),
                frac=round(floor / step_us, 4), unit="us per decoder step (whole pass / steps: the encoder's share included)")


def _timed(fn, iters):
    """Seconds per call of fn (already warmed up)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def _timed_median(fn, iters, repeats=3):
    """Median over `repeats` loops of _timed(fn, iters): for the legs that time HOST work call by call (the drop-in API), where
    one 30 ms hiccup of a shared host inside a 60 ms loop halves the figure (seen in 3 of 8 runs on the 4-tenant boxes of round 6)."""
    return sorted(_timed(fn, iters) for _ in range(repeats))[repeats // 2]


def extra_legs(dev, main_dtype, legs):
    """The other operating points of BASELINE.json / VERDICT, measured in the same process after the
    headline run (rank 0, N = 1 only): each is `captions/s` of whole passes over synthetic clips
    resident in HBM, hipGraph replay, measured over >= 5 passes after 3 warm-up calls."""
    from care_amd import get_framework
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_input_ids, synth_state_dict

    import gc

    def build(config, dtype, row_scale=None, seed=0, **over):
        # engines of finished legs (GBs of workspaces, captured graphs, model <-> engine cycles) go before the next one
        gc.collect()
        torch.cuda.empty_cache()
        opt = make_opt(config, **over)
        model = get_framework(opt).eval()
        P = synth_state_dict(seed, [(k, tuple(v.shape)) for k, v in model.state_dict().items()], row_scale=row_scale or {})
        model.load_state_dict(P, strict=True)
        model.set_compute_dtype(dtype)
        model.to(dev)
        eng_model[0] = model
        return opt, model.engine()

    eng_model = [None]

    def feats_for(opt, B, seed=2000):
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        return [torch.randn(shape, generator=gen, device=dev, dtype=torch.float32) for shape in feat_shapes(opt, B)]

    def greedy_leg(config, dtype, B, iters=5, **kw):
        opt, eng = build(config, dtype, **kw)
        feats = feats_for(opt, B)
        run = lambda: eng.translate_greedy(feats, use_graph=True, lean=True)
        for _ in range(3):
            run()
        dt = _timed(run, iters)
        leg = dict(config=config, dtype=dtype, clips_per_step=B, captions_per_s=round(B / dt, 1),
                   ms_per_pass=round(dt * 1e3, 3), decoder_step_us=round(dt * 1e6 / eng.T, 2),
                   # the decoder steps the LAST timed pass ran (a replayed pass that ended early would be fast for the wrong reason:
                   # the captured-memset bug of round 6 did exactly that to passes over recycled buffers)
                   steps_run=int(eng.last_decode.get("steps", eng.T)))
        if eng.last_decode.get("resident"):
            leg["roofline"] = small_batch_roofline(eng, B, 1, dt * 1e6 / eng.T)
        return leg, eng, feats, opt

    # BASELINE configs[2]: the concept-guided (CARE) path
    legs["msrvtt_care_greedy"] = greedy_leg("msrvtt_care", main_dtype, 32768)[0]   # (*measured* round 6: 452 K at 16384 clips, 499 K at 32768)
    # the headline workload in the OTHER 16-bit mode: fp16 (the same kernels compiled for IEEE half, libcare_hip_f16.so) -
    # bf16's bytes and MFMA rate, 8 x smaller error (fp16_hidden_state_error below)
    if main_dtype == "bf16":
        legs["msrvtt_base_ami_fp16"] = greedy_leg("msrvtt_base_ami", "fp16", 32768, iters=4)[0]
    # the headline workload with the batch cut into TWO lanes on two HIP streams inside the one captured graph (engine.lanes:
    # HBM-bound and MFMA-bound kernels of the two halves overlap; off in the headline so that its per-kernel roofline describes
    # a kernel on its own)
    opt, eng = build("msrvtt_base_ami", main_dtype)
    eng.lanes = 2
    feats = feats_for(opt, 32768)
    run = lambda: eng.translate_greedy(feats, use_graph=True, lean=True, early_exit=False)
    for _ in range(3):
        run()
    dt = _timed(run, 4)
    legs["msrvtt_base_ami_two_lanes"] = dict(config="msrvtt_base_ami", dtype=main_dtype, clips_per_step=32768, lanes=2,
                                             captions_per_s=round(32768 / dt, 1), ms_per_pass=round(dt * 1e3, 3))
    del feats
    # fp32 parity mode (the only mode inside north_star's 1e-5 tolerance)
    legs["msrvtt_base_ami_fp32"] = greedy_leg("msrvtt_base_ami", "fp32", 4096)[0]
    # the mode between the two: fp32 storage, every GEMM as three fp16 MFMA passes over hi/lo pieces (fp32-grade results)
    legs["msrvtt_base_ami_fp16x3"] = greedy_leg("msrvtt_base_ami", "fp16x3", 4096)[0]
    legs["msrvtt_base_ami_fp16x3_B16384"] = greedy_leg("msrvtt_base_ami", "fp16x3", 16384, iters=3)[0]
    # the reference's own operating point: translate.py batch 128 (translate.py:137); a step here is a latency
    legs["msrvtt_base_ami_B128"] = greedy_leg("msrvtt_base_ami", main_dtype, 128, iters=20)[0]
    legs["msrvtt_base_ami_B1"] = greedy_leg("msrvtt_base_ami", main_dtype, 1, iters=20)[0]
    # the concept-guided model at translate.py's batch (greedy)
    legs["msrvtt_care_B128"] = greedy_leg("msrvtt_care", main_dtype, 128, iters=20)[0]
    # BASELINE configs[3]: d_model = 1024 - at the batch that fills the chip, and at the 32 clips per GPU that config names
    # ("batch = 256 sharded over 8 x MI355X": one resident launch per decode since round 4, csrc/decode_resident.hip D = 1024)
    legs["vatex_care_large"] = greedy_leg("vatex_care_large", main_dtype, 4096)[0]
    legs["vatex_care_large_B16384"] = greedy_leg("vatex_care_large", main_dtype, 16384, iters=3)[0]
    legs["vatex_care_large_B32"] = greedy_leg("vatex_care_large", main_dtype, 32, iters=20)[0]
    # d_model = 768 (archs.yaml "median", GELU) at the batch that fills the chip
    legs["care_median_gelu_B16384"] = greedy_leg("care_median_gelu", main_dtype, 16384, iters=3)[0]
    # BASELINE configs[4]: CARE, beam 5 (opts.py beam_size 5): a large batch, the reference's batch of 128
    # (translate.py:137,144) and its latency mode (translate.py:208-209: one clip) - the last two as ONE resident launch
    # per search (csrc/decode_resident_beam.hip)
    for B in (4096, 128, 1):
        eng = None
        opt, eng = build("msrvtt_care_beam5", main_dtype)
        feats = feats_for(opt, B)
        run = lambda: eng.translate_beam(feats, 5, 5, use_graph=True, lean=True)
        for _ in range(3):
            run()
        dt = _timed(run, 5 if B > 128 else 20)
        legs["msrvtt_care_beam5_B%d" % B] = dict(config="msrvtt_care_beam5", dtype=main_dtype, clips_per_step=B, beam_size=5,
                                                  rows_per_decoder_step=5 * B, captions_per_s=round(B / dt, 1),
                                                  ms_per_pass=round(dt * 1e3, 3), decoder_step_us=round(dt * 1e6 / eng.T, 2),
                                                  resident_launch=bool(eng.last_decode.get("resident")),
                                                  steps_run=int(eng.last_decode.get("steps", eng.T)))
        if eng.last_decode.get("resident"):
            legs["msrvtt_care_beam5_B%d" % B]["roofline"] = small_batch_roofline(eng, B, 5, dt * 1e6 / eng.T)
        if B == 1:
            legs["msrvtt_care_beam5_B1"]["ms_per_caption"] = round(dt * 1e3, 3)
    # ---- batch sweep (B in {1, 64, 128, 256, 1024, 4096}; translate.py:136 lets a user pick any
    # batch): greedy d = 512 and beam 5 across the hand-overs between the forms of the decode - the resident launch (greedy
    # <= engine.resident_max_rows clips, beam <= engine.resident_beam_max_rows rows), the multi-launch small forms, the
    # large-batch forms.  captions/s must grow with B through every hand-over (`shape` below says whether it did); engine.py's crossover constants cite this table (profiles/r05_batch_sweep.json).
    sweep = {"greedy": {}, "beam5": {}}
    opt, eng = build("msrvtt_base_ami", main_dtype)
    for B in (1, 64, 128, 256, 512, 1024, 2048, 4096):
        feats = feats_for(opt, B)
        run = lambda: eng.translate_greedy(feats, use_graph=True, lean=True)
        for _ in range(3):
            run()
        dt = _timed(run, 20 if B <= 512 else 8)
        sweep["greedy"][str(B)] = dict(captions_per_s=round(B / dt, 1), decoder_step_us=round(dt * 1e6 / eng.T, 2),
                                       form="resident" if eng.last_decode.get("resident") else "multi-launch")
    opt, eng = build("msrvtt_care_beam5", main_dtype)
    for B in (1, 32, 64, 128, 256, 512, 1024, 4096):
        feats = feats_for(opt, B)
        run = lambda: eng.translate_beam(feats, 5, 5, use_graph=True, lean=True)
        for _ in range(3):
            run()
        dt = _timed(run, 20 if B <= 128 else 6)
        sweep["beam5"][str(B)] = dict(captions_per_s=round(B / dt, 1), decoder_step_us=round(dt * 1e6 / eng.T, 2), rows=5 * B,
                                      form="resident" if eng.last_decode.get("resident") else
                                      "chain" if eng.last_decode.get("chain") else "multi-launch")
    # the sweep's own check (a wall-clock property: it lives here, not in the parity suite): captions/s grows through every
    # hand-over between forms, and no point sits more than 15 % below the line through its neighbours
    shape = {}
    for kind, pts in sweep.items():
        pts = [(int(b), v["captions_per_s"]) for b, v in pts.items()]
        grow = min(r1 / r0 for (_, r0), (_, r1) in zip(pts, pts[1:]))
        dip = min(r1 / (r0 + (r2 - r0) * (b1 - b0) / (b2 - b0)) for (b0, r0), (b1, r1), (b2, r2) in zip(pts, pts[1:], pts[2:]))
        shape[kind] = dict(min_ratio_to_previous_point=round(grow, 3), min_ratio_to_neighbours_line=round(dip, 3),
                           monotone=bool(grow > 0.97), no_cliff=bool(dip > 0.85))
    legs["batch_sweep"] = dict(config_greedy="msrvtt_base_ami", config_beam5="msrvtt_care_beam5", dtype=main_dtype, shape=shape, **sweep)
    # BASELINE configs[3] with translate.py's default decode: d_model 1024, beam 5, the 32 clips per GPU of a 256-clip batch
    # over 8 GPUs (160 rows) - one resident launch since round 5 (csrc/decode_resident_beam.hip, D = 1024)
    opt, eng = build("vatex_care_large", main_dtype)
    feats = feats_for(opt, 32)
    run = lambda: eng.translate_beam(feats, 5, 5, use_graph=True, lean=True)
    for _ in range(3):
        run()
    dt = _timed(run, 20)
    legs["vatex_care_large_beam5_B32"] = dict(config="vatex_care_large", dtype=main_dtype, clips_per_step=32, beam_size=5,
                                              rows_per_decoder_step=160, captions_per_s=round(32 / dt, 1), ms_per_pass=round(dt * 1e3, 3),
                                              decoder_step_us=round(dt * 1e6 / eng.T, 2), resident_launch=bool(eng.last_decode.get("resident")))
    if eng.last_decode.get("resident"):
        legs["vatex_care_large_beam5_B32"]["roofline"] = small_batch_roofline(eng, 32, 5, dt * 1e6 / eng.T)
    # a model that ENDS its captions (EOS row of the vocabulary projection x 5: mixed lengths, mean ~8 like trained
    # captions; random-init weights never emit EOS): early termination + compaction against the fixed 29 steps
    boost = {"cls_head.tgt_word_prj.weight": {3: 5.0}}
    leg, eng, feats, opt = greedy_leg("msrvtt_base_ami", main_dtype, 32768, row_scale=boost)
    _, fed, length, _ = eng.translate_greedy(feats, use_graph=True, lean=True)
    mean_len = float(length.float().mean())
    stats = dict(eng.last_decode)
    fixed = lambda: eng.translate_greedy(feats, use_graph=True, lean=True, early_exit=False)
    for _ in range(3):
        fixed()
    dt_fixed = _timed(fixed, 5)
    leg.update(mean_caption_length=round(mean_len, 2), steps_run=stats["steps"], compactions=stats["compactions"],
               row_steps=stats["row_steps"], row_steps_fixed=32768 * eng.T,
               fixed_29_steps_captions_per_s=round(32768 / dt_fixed, 1),
               speedup_vs_fixed_29=round(dt_fixed / (leg["ms_per_pass"] * 1e-3), 2))
    legs["early_exit_eos_model"] = leg
    # the same for beam 5 (configs[4] on a model that ends its captions): finished clips leave between segments.  A clip
    # is done once beam_size hypotheses have ended (Beam.py:38-43): with the x5 row almost none gets there in 29 steps,
    # so this leg boosts the EOS row x20 (*measured* row-steps: x5 594 K of 594 K, x8 527 K, x12 440 K, x20 351 K)
    opt, eng = build("msrvtt_care_beam5", main_dtype, row_scale={"cls_head.tgt_word_prj.weight": {3: 20.0}})
    feats = feats_for(opt, 4096)
    runs = {}
    for name, ee in (("early_exit", True), ("fixed_29_steps", False)):
        run = lambda: eng.translate_beam(feats, 5, 5, use_graph=True, lean=True, early_exit=ee)
        for _ in range(3):
            run()
        runs[name] = _timed(run, 5)
        if ee:
            st = dict(eng.last_decode)
    legs["early_exit_eos_model_beam5"] = dict(config="msrvtt_care_beam5", dtype=main_dtype, clips_per_step=4096, beam_size=5,
                                              captions_per_s=round(4096 / runs["early_exit"], 1),
                                              fixed_29_steps_captions_per_s=round(4096 / runs["fixed_29_steps"], 1),
                                              speedup_vs_fixed_29=round(runs["fixed_29_steps"] / runs["early_exit"], 2),
                                              steps_run=st["steps"], compactions=st["compactions"], row_steps=st["row_steps"],
                                              row_steps_fixed=4096 * 5 * eng.T)
    # ---- teacher-forced forward (models/Framework.py:215-237; the eval metrics step of Wrapper.py:182-184), B = 4096:
    # encode + the decoder over all 29 positions + fused scoring (word accuracy / perplexity inputs; the [B*29, V]
    # logits never exist), and the same through the module API with the fp32 logits materialised (5 GB)
    opt, eng = build("msrvtt_base_ami", main_dtype)
    for Btf, leg_name in ((4096, "feedforward_step"), (16384, "feedforward_step_B16384")):
        feats = feats_for(opt, Btf)
        gen = torch.Generator(device=dev)
        gen.manual_seed(7)
        ids = torch.randint(4, opt["vocab_size"], (Btf, eng.T), generator=gen, device=dev)
        ids[:, 0] = 1
        labels = torch.randint(4, opt["vocab_size"], (Btf, eng.T), generator=gen, device=dev)

        def tf_score():
            return eng.metrics_step(feats, ids, labels)

        for _ in range(3):
            tf_score()
        dt = _timed(tf_score, 10 if Btf <= 4096 else 4)
        d_, ff_, V_, Lk_, T_ = eng.d, eng.ff, eng.V, eng.Lk, eng.T
        fl = (sum(2 * eng.rows_of[ch] * d_ * opt["dim_" + ch] for ch in eng.modality) + 4 * Lk_ * d_ * d_ +
              sum(2 * d_ * d_ * 6 + 4 * d_ * ff_ + 2 * d_ * V_ + 4 * Lk_ * d_ + 4 * t * d_ for t in range(1, T_ + 1)))
        _lib_mod = __import__("care_amd")._lib
        _lib_mod.TIMING = {}
        tf_score()
        torch.cuda.synchronize()
        timing, _lib_mod.TIMING = _lib_mod.TIMING, None
        tfk = {t: round(sum(s.elapsed_time(e) for s, e in ev), 3) for t, ev in timing.items()}
        legs[leg_name] = dict(config="msrvtt_base_ami", dtype=main_dtype, clips_per_step=Btf, positions=T_,
                              what="engine.metrics_step: encode (lean) + teacher-forced decoder over all positions + fused scoring (no logits "
                                   "in memory)",
                              clips_per_s=round(Btf / dt, 1), ms_per_pass=round(dt * 1e3, 3),
                              gflop_per_clip=round(fl / 1e9, 4), tflops=round(fl * Btf / dt / 1e12, 1),
                              frac_of_bf16_mfma_peak=round(fl * Btf / dt / 1e12 / MFMA_PEAK_TF["bf16"], 4),
                              fast_path=bool(eng.tf_fast_ok(T_, False)),
                              kernel_ms=dict(sorted(tfk.items(), key=lambda kv: -kv[1])))
        if Btf == 4096:
            # the same pass on ONE stream (CARE_TF_OVERLAP=0; the default runs the encoder + static K/V chain on a side stream
            # beside the decoder's embedding + self-attention block)
            os.environ["CARE_TF_OVERLAP"] = "0"
            for _ in range(2):
                tf_score()
            dt1 = _timed(tf_score, 10)
            del os.environ["CARE_TF_OVERLAP"]
            legs[leg_name]["one_stream"] = dict(ms_per_pass=round(dt1 * 1e3, 3), frac_of_bf16_mfma_peak=round(fl * Btf / dt1 / 1e12 / MFMA_PEAK_TF["bf16"], 4))
        if Btf != 4096:
            del feats, ids, labels
    Btf = 4096
    feats = feats_for(opt, Btf)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    ids = torch.randint(4, opt["vocab_size"], (Btf, eng.T), generator=gen, device=dev)
    ids[:, 0] = 1
    T_, V_ = eng.T, eng.V
    model_tf = eng_model[0]
    batch = {"feats": feats, "input_ids": ids}
    api = lambda: model_tf.feedforward_step(batch, output_auxiliary=False)
    for _ in range(2):
        api()
    dt_api = _timed(api, 5)
    legs["feedforward_step"]["module_api_with_fp32_logits"] = dict(
        ms_per_pass=round(dt_api * 1e3, 3), clips_per_s=round(Btf / dt_api, 1), tflops=round(fl * Btf / dt_api / 1e12, 1),
        logits_bytes=Btf * T_ * V_ * 4)
    del model_tf, batch, api

    # ---- training mode (train.py's step without loss / optimiser: Wrapper.py:423-435): forward + backward of the whole
    # path under torch.autograd, every forward and backward a HIP kernel (care_amd/training.py; fp32 storage, exact-f32
    # MFMA), at the reference's training batch (opts.py --batch_size 64), dropout at the reference's rates
    opt, eng = build("msrvtt_care", "fp32")
    model_tr = eng_model[0]
    model_tr.train()
    Btr = 64
    f_tr = feats_for(opt, Btr)
    ids_tr = synth_input_ids(7, Btr, opt["max_len"] - 1, opt["vocab_size"]).to(dev)
    batch = {"feats": f_tr, "input_ids": ids_tr}
    g_tr = None

    def train_step():
        nonlocal g_tr
        for prm in model_tr.parameters():
            prm.grad = None
        out = model_tr(batch)
        if g_tr is None:
            g_tr = torch.randn_like(out["logits"]) * 1e-3
        torch.autograd.backward([out["logits"]], [g_tr])

    fl_tr = 3.0 * fl  # forward + the two backward products of every GEMM, per clip (same shapes as the teacher-forced forward)
    from care_amd import training as _training
    for Btr, name in ((64, "training_step"), (512, "training_step_B512"), (512, "training_step_B512_f32")):
        # the default ("auto": per product the exact-f32 MFMA, or - from a few GFLOP on - split products of pre-scaled fp16
        # pieces at the 16-bit matrix rate; care_amd/training.py TRAIN_GEMM), and the 512-clip step with every product exact
        _training.set_train_gemm("f32" if name.endswith("_f32") else "auto")
        f_tr = feats_for(opt, Btr)
        ids_tr = synth_input_ids(7, Btr, opt["max_len"] - 1, opt["vocab_size"]).to(dev)
        batch = {"feats": f_tr, "input_ids": ids_tr}
        g_tr = None
        for _ in range(2):
            train_step()
        dt_tr = _timed(train_step, 5)
        legs[name] = dict(config="msrvtt_care", dtype="f32", clips_per_step=Btr, gemm=_training.TRAIN_GEMM,
                          what="model.train(); forward + backward through care_amd/training.py (autograd Functions over HIP kernels), "
                               "gradient of a fixed cotangent on the logits; no loss, no optimiser",
                          ms_per_step=round(dt_tr * 1e3, 3), clips_per_s=round(Btr / dt_tr, 1),
                          tflops=round(fl_tr * Btr / dt_tr / 1e12, 2))
    _training.set_train_gemm("auto")
    model_tr.eval()
    del model_tr, batch, f_tr, ids_tr, g_tr

    # ---- host-fed: the reference moves `feats` host -> device per batch (translate.py:34-38); here pinned host batches
    # (what DataLoader(pin_memory=True) yields) through FeaturePrefetcher - H2D on a side stream, overlapped with the
    # previous batch's decode.  fp32 features, and bf16 ones a loader rounded on the host (bit-identical products for a
    # model without a concept head: its embedder multiplies bf16-rounded features anyway).
    from care_amd.data import FeaturePrefetcher

    for Bh in (4096, 16384):
        opt, eng = build("msrvtt_base_ami", main_dtype)
        shapes = feat_shapes(opt, Bh)
        for name, hdt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            host = [[torch.randn(sh, device=dev).to(hdt).cpu().pin_memory() for sh in shapes] for _ in range(2)]
            nb = 8 if Bh <= 4096 else 5
            def run_host():
                for f in FeaturePrefetcher((host[i % 2] for i in range(nb)), dev):
                    eng.translate_greedy(f, use_graph=True, lean=True)
            run_host()
            run_host()
            dt = _timed(run_host, 2) / nb
            nbytes = sum(t.numel() * t.element_size() for t in host[0])
            res = feats_for(opt, Bh)
            if hdt == torch.bfloat16:
                res = [t.to(hdt) for t in res]
            runr = lambda: eng.translate_greedy(res, use_graph=True, lean=True)
            for _ in range(3):
                runr()
            dtr = _timed(runr, 5)
            legs["host_fed_B%d_%s" % (Bh, name)] = dict(
                config="msrvtt_base_ami", dtype=main_dtype, clips_per_step=Bh, feature_transport=name,
                captions_per_s=round(Bh / dt, 1), ms_per_batch=round(dt * 1e3, 3), h2d_bytes_per_batch=nbytes,
                pcie_GBps_achieved=round(nbytes / dt / 1e9, 1), hbm_resident_captions_per_s=round(Bh / dtr, 1),
                # the last batch's decode has no copy to hide behind: 1 / nb of a decode in every per-batch time, which
                # weighs twice as much against a bf16 transfer - the copy rate itself, that tail taken out:
                pcie_GBps_copy_only=round(nbytes / max(dt - dtr / nb, 1e-9) / 1e9, 1),
                bound="PCIe H2D" if nbytes / dt / 1e9 > 35 and dt > 1.1 * dtr else "decode",
                sample="%d batches from 2 pinned host buffers, double-buffered device slots" % nb)
            del host, res

    # ---- the drop-in API itself: what translate.py calls is get_translator(opt).translate_batch
    # (models/Translator.py:35-85 via Wrapper.py:175), which returns python lists - every leg above times the engine call
    # underneath it.  Per operating point: the engine pass (device tensors out), translate_batch call by call and
    # translate_batches (batch k's lists assembled while batch k + 1 decodes) on features resident in HBM, and both again
    # fed from pinned host batches the way translate.py's loop is (FeaturePrefetcher: H2D on a side stream), with the split
    # of one batch's time.  `api_over_engine` = pipelined API on resident features / engine pass: the price of the boundary.
    from care_amd import get_translator

    def api_leg(config, B, beam, nb, iters):
        opt, eng = build(config, main_dtype, beam_size=beam, topk=1)
        model = eng_model[0]
        tr = get_translator(opt)
        feats = feats_for(opt, B)
        if beam == 1:
            eng_run = lambda: eng.translate_greedy(feats, use_graph=True, lean=True)
        else:
            eng_run = lambda: eng.translate_beam(feats, beam, beam, use_graph=True, lean=True)
        for _ in range(3):
            eng_run()
        dt_eng = _timed_median(eng_run, iters)
        batch = {"feats": feats}
        one = lambda: tr.translate_batch([model], batch)
        for _ in range(3):
            hyps, scores = one()
        dt_one = _timed_median(one, iters)
        piped = lambda: sum(len(h) for h, _ in tr.translate_batches([model], (batch for _ in range(nb))))
        piped()
        dt_pipe = _timed_median(piped, max(1, iters // nb)) / nb
        # the split of one call: enqueue (python + graph launch), device pass, copy back + list assembly
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = tr._launch([model], batch, {})
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tr._finish(pend)
        t3 = time.perf_counter()
        # host-fed: pinned batches, H2D through the prefetcher (three device slots: a batch's features stay in place
        # until its results are out), serial calls and the pipelined entry
        host = [t.cpu().pin_memory() for t in feats]
        nbytes = sum(t.numel() * t.element_size() for t in host)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dst = [torch.empty_like(t) for t in feats]
        e0.record()
        for d_, h_ in zip(dst, host):
            d_.copy_(h_, non_blocking=True)
        e1.record()
        torch.cuda.synchronize()
        h2d_ms = e0.elapsed_time(e1)
        del dst
        fed_serial = lambda: [tr.translate_batch([model], {"feats": f}) for f in FeaturePrefetcher((host for _ in range(nb)), dev, depth=3)]
        fed_piped = lambda: list(tr.translate_batches([model], ({"feats": f} for f in FeaturePrefetcher((host for _ in range(nb)), dev, depth=3))))
        for fn in (fed_serial, fed_piped):
            fn()
            fn()
        dt_fs = _timed(fed_serial, max(1, iters // nb)) / nb
        dt_fp = _timed(fed_piped, max(1, iters // nb)) / nb
        return dict(config=config, dtype=main_dtype, clips_per_batch=B, beam_size=beam, rows_per_decoder_step=B * beam,
                    entry="get_translator(opt).translate_batch([model], {'feats': ...}) -> (batch_hyps, batch_scores) python lists "
                          "(models/Translator.py:35-85); pipelined = translate_batches over %d batches; every figure the median of 3 "
                          "timed loops" % nb,
                    engine_captions_per_s=round(B / dt_eng, 1), engine_ms_per_pass=round(dt_eng * 1e3, 3),
                    api_captions_per_s=round(B / dt_one, 1), api_ms_per_call=round(dt_one * 1e3, 3),
                    api_pipelined_captions_per_s=round(B / dt_pipe, 1), api_pipelined_ms_per_batch=round(dt_pipe * 1e3, 3),
                    api_over_engine=round(dt_eng / dt_pipe, 3), api_serial_over_engine=round(dt_eng / dt_one, 3),
                    one_call_split_ms=dict(enqueue=round((t1 - t0) * 1e3, 3), device_wait=round((t2 - t1) * 1e3, 3),
                                           d2h_and_list_assembly=round((t3 - t2) * 1e3, 3)),
                    host_fed=dict(h2d_bytes_per_batch=nbytes, h2d_ms_alone=round(h2d_ms, 3),
                                  serial_captions_per_s=round(B / dt_fs, 1), serial_ms_per_batch=round(dt_fs * 1e3, 3),
                                  pipelined_captions_per_s=round(B / dt_fp, 1), pipelined_ms_per_batch=round(dt_fp * 1e3, 3),
                                  pipelined_over_engine=round(dt_eng / dt_fp, 3),
                                  bound="PCIe H2D" if h2d_ms > 1.05 * dt_eng * 1e3 else "decode"),
                    caption_length=len(hyps[0][0]), resident_launch=bool(eng.last_decode.get("resident")))

    legs["api_greedy_B128"] = api_leg("msrvtt_base_ami", 128, 1, 16, 32)
    legs["api_beam5_B128"] = api_leg("msrvtt_care_beam5", 128, 5, 16, 32)    # translate.py's defaults (translate.py:137,144)
    # (8 batches per pipelined pass: the last batch's assembly - ~28 ms of list building at 32768 clips - is not hidden behind a
    # next pass, so a stream of n batches costs about (n x pass + one assembly) / n)
    legs["api_greedy_B32768"] = api_leg("msrvtt_base_ami", 32768, 1, 8, 8)

    # ---- model ensembling through the same entry (models/Translator.py:39-52,112-133): two CARE models, beam 5, 128 clips - the
    # members step side by side with their vocabulary logits in memory (eager, off the fast forms)
    opt, eng = build("msrvtt_care_beam5", main_dtype, beam_size=5, topk=1)
    m1 = eng_model[0]
    m2 = get_framework(opt).eval()
    m2.load_state_dict(synth_state_dict(1, [(k, tuple(v.shape)) for k, v in m2.state_dict().items()]), strict=True)
    m2.set_compute_dtype(main_dtype)
    m2.to(dev)
    tr = get_translator(opt)
    batch = {"feats": feats_for(opt, 128)}
    ens = lambda: tr.translate_batch([m1, m2], batch)
    for _ in range(2):
        ens()
    dt_ens = _timed(ens, 4)
    one = lambda: tr.translate_batch([m1], batch)
    for _ in range(2):
        one()
    legs["ensemble_x2_beam5_B128"] = dict(config="msrvtt_care_beam5 x 2 (seeds 0, 1)", dtype=main_dtype, clips_per_batch=128, beam_size=5,
                                          captions_per_s=round(128 / dt_ens, 1), ms_per_call=round(dt_ens * 1e3, 3),
                                          single_model_ms_per_call=round(_timed(one, 8) * 1e3, 3))
    del m1, m2, tr, batch, ens, one

    # ---- 16-bit agreement at the size of the MSRVTT test split (2990 clips; notebooks/retrieval_robustness.ipynb:188): the
    # peaked CARE model (a softmax as peaked as a trained model's) through the Translator at translate.py's batch of 128
    # (the resident launches), greedy and beam 5: captions identical to those of the engine's fp32 mode, per 16-bit mode.
    # (fp32 mode is identical to the reference on every fixture; tests/test_gpu_scale.py holds the same 2990 clips to the CPU
    # oracle and audits every differing clip as a near-tie of the reference's own distribution.)
    # (seed 373: the weights of the peaked fixtures and of tests/test_gpu_scale.py - a model that ends its captions)
    opt, eng = build("msrvtt_care", "fp32", row_scale={"cls_head.tgt_word_prj.weight": {**{r: 12.0 for r in range(6, 46)}, 3: 20.0}}, seed=373)
    model_s = eng_model[0]
    n_test = 2990
    gen = torch.Generator().manual_seed(373)
    host_feats = [torch.randn(sh, generator=gen) for sh in feat_shapes(opt, n_test)]
    dev_batches = [{"feats": [f[lo: lo + 128].to(dev) for f in host_feats]} for lo in range(0, n_test, 128)]
    agree = {}
    for beam in (1, 5):
        tr = get_translator(dict(opt, beam_size=beam, topk=1))
        caps = {}
        for mode in ("fp32", "fp16", "bf16"):
            model_s.set_compute_dtype(mode)
            caps[mode] = [h[0] for hyps, _ in tr.translate_batches([model_s], iter(dev_batches)) for h in hyps]
        agree["greedy" if beam == 1 else "beam5"] = {m: sum(int(a == b) for a, b in zip(caps[m], caps["fp32"])) for m in ("fp16", "bf16")}
        agree["mean_caption_length_" + ("greedy" if beam == 1 else "beam5")] = round(sum(len(c) for c in caps["fp32"]) / n_test, 2)
    legs["msrvtt_test_scale_agreement"] = dict(
        config="msrvtt_care (peaked rows)", clips=n_test, batch=128, through="get_translator(opt).translate_batches",
        reference="the engine's fp32 mode (identical to the reference on every fixture; held to the CPU oracle on these clips "
                  "by tests/test_gpu_scale.py)", identical_captions=agree)
    del model_s, dev_batches, host_feats

    # the error of the throughput mode: teacher-forced hidden states, bf16 mode against fp32 mode of this
    # same engine (fp32 mode is within 1e-5 of the reference, tests/test_gpu_parity.py) on the benchmarked model
    if main_dtype == "bf16":
        from care_amd.synth import synth_feats, synth_input_ids

        opt = make_opt("msrvtt_base_ami")
        model = get_framework(opt).eval()
        model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
        model.to(dev)
        f = [x.to(dev) for x in synth_feats(3, feat_shapes(opt, 64))]
        ids = synth_input_ids(3, 64, opt["max_len"] - 1, opt["vocab_size"]).to(dev)
        hid = {}
        for dt_ in ("fp32", "bf16", "fp16"):
            model.set_compute_dtype(dt_)
            hid[dt_] = model.feedforward_step({"feats": f, "input_ids": ids})["hidden_states"].float().clone()
        for dt_ in ("bf16", "fp16"):
            diff = (hid[dt_] - hid["fp32"]).abs()
            legs[dt_ + "_hidden_state_error"] = dict(max_abs=round(float(diff.max()), 5), mean_abs=round(float(diff.mean()), 6),
                                                     against="fp32 mode of the same engine (itself within 1e-5 of the reference)",
                                                     sample="teacher-forced hidden states, 64 clips x 29 positions x 512")
    return legs


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))  # one process per GPU: a torchrun child, rank 0's line relayed
    if world != args.gpus:
        raise SystemExit("--gpus {} but WORLD_SIZE {} (torch.distributed.run --nproc-per-node must equal --gpus)".format(args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)
    if local >= torch.cuda.device_count():
        raise SystemExit("rank {}: no GPU {} on this host ({} visible)".format(rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("CARE_BENCH_FORCE_DIST") == "1"  # forced: 1-rank RCCL self-test
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    from care_amd import _lib, get_framework
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_feats, synth_state_dict

    opt = make_opt(args.config)
    B = args.batch
    model = get_framework(opt).eval()
    P = synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()])
    model.load_state_dict(P, strict=True)
    model.set_compute_dtype(args.dtype)
    model.to(dev)
    eng = model.engine()
    eng.lanes = args.lanes
    # per-rank inputs: rank r holds clips [r*B, (r+1)*B) of the global batch (weak scaling).
    # Unit-variance features generated ON the device (seeded per rank); the portable CPU generator
    # (care_amd.synth) would spend minutes producing 2.5 G values for B = 32768.
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    feats = [torch.randn(shape, generator=gen, device=dev, dtype=torch.float32) for shape in feat_shapes(opt, B)]
    torch.cuda.synchronize()

    from care_amd.sharding import all_gather_records, pack_records

    gathered = None
    if use_dist:
        gathered = [torch.empty(B, eng.T + 4, device=dev, dtype=torch.int32) for _ in range(world)]

    if args.beam > 1:
        for _ in range(3):
            eng.translate_beam(feats, args.beam, args.beam, use_graph=not args.no_graph, lean=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.translate_beam(feats, args.beam, args.beam, use_graph=not args.no_graph, lean=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        _lib.TIMING = {}   # instrumented pass (untimed): HIP events around every tagged launch
        eng.translate_beam(feats, args.beam, args.beam, use_graph=False, lean=True)
        torch.cuda.synchronize()
        timing, _lib.TIMING = _lib.TIMING, None
        kernels = {t: dict(launches=len(ev), avg_us=round(1e3 * sum(s.elapsed_time(e) for s, e in ev) / len(ev), 2),
                           total_ms=round(sum(s.elapsed_time(e) for s, e in ev), 3)) for t, ev in timing.items()}
        kernels = dict(sorted(kernels.items(), key=lambda kv: -kv[1]["total_ms"]))
        print(json.dumps(dict(metric="captions/sec (beam %d)" % args.beam, value=round(B / dt, 1), unit="captions/s",
                              n_gpus=1, steps=args.steps, ms_per_step=round(dt * 1e3, 3), dtype=args.dtype, kernels=kernels,
                              config=dict(config_name=args.config, clips_per_gpu_per_step=B, beam_size=args.beam,
                                          rows_per_decoder_step=B * args.beam))), flush=True)
        return

    def step():
        _, fed, length, score = eng.translate_greedy(feats, use_graph=not args.no_graph, lean=True)  # as the Translator does
        if use_dist:  # metrics-step exchange over RCCL: every rank gets every caption
            all_gather_records(pack_records(fed, length, score, B), gathered)
        return fed, length, score

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2)):   # >= 2: the first call allocates, the second captures the graph
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    dist_info = None
    if use_dist:
        mine = torch.tensor([elapsed, float(rank)], device=dev, dtype=torch.float64)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_s = [float(e[0]) for e in every]
        # the exchange alone: the all-gather of one batch's records, back to back
        rec = pack_records(*step(), B)
        barrier()
        t1 = time.perf_counter()
        for _ in range(20):
            all_gather_records(rec, gathered)
        barrier()
        dist_info = dict(backend="nccl (RCCL)", rccl_ranks_seen=sorted(int(e[1]) for e in every),
                         per_rank_captions_per_s=[round(B * args.steps / t, 1) for t in per_rank_s],
                         all_gather_us=round((time.perf_counter() - t1) / 20 * 1e6, 1),
                         all_gather_bytes_per_rank=int(rec.numel() * rec.element_size()))
        elapsed = max(per_rank_s)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    # what the last timed pass did (after the clock stopped): decoder steps run, caption lengths - a pass that ended early or
    # returned a previous batch's rows would be fast for the wrong reason
    _, length_chk, _ = step()
    torch.cuda.synchronize()
    work_done = dict(decoder_steps_run=int(eng.last_decode.get("steps", eng.T)), caption_length_min=int(length_chk.min()),
                     caption_length_mean=round(float(length_chk.float().mean()), 2), captions=int(length_chk.numel()))

    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- instrumented pass (untimed): HIP events around every tagged launch, same workload
    _lib.TIMING = {}
    eng.translate_greedy(feats, use_graph=False, lean=True)
    torch.cuda.synchronize()
    timing, _lib.TIMING = _lib.TIMING, None
    kernels = {}
    for tag, evs in timing.items():
        ms = [s.elapsed_time(e) for s, e in evs]
        kernels[tag] = dict(launches=len(ms), avg_us=1e3 * sum(ms) / len(ms), total_ms=sum(ms))
    tagged_ms = sum(k["total_ms"] for k in kernels.values())
    # decoder-step time: the 29 steps' tagged kernels (event-measured) per step
    step_tags = [t for t in kernels if t.startswith("step_") or t == "decode_resident"]
    step_tags = [t for t in step_tags if kernel_model(t, eng, B, args.dtype) is not None]
    dom = max(step_tags, key=lambda t: kernels[t]["total_ms"])
    km = kernel_model(dom, eng, B, args.dtype)
    # `achieved` uses the IN-SITU duration: the average over the kernel's launches inside the
    # pass, each bracketed by HIP events on the launch stream (agrees with rocprofv3's kernel
    # trace, profiles/).  The back-to-back re-launch (same arguments, 50x between two events) is
    # reported beside it; it is faster because consecutive launches re-read a warm L2/MALL.
    dom_b2b_us = _lib.relaunch_avg_us(dom, 50)

    # The same kernel INSIDE the timed configuration (hipGraph replay): its launches are bracketed by one-thread
    # timestamp kernels (device wall clock, 100 MHz) in a freshly captured copy of the pass - HIP events do not record
    # in a replayed graph.  A bracket also spans two kernel boundaries; their cost is measured the same way around an
    # empty bracket and reported beside the raw figure.
    def stamped_us(tag, cap=64):
        buf = torch.zeros(2 * cap + 64, dtype=torch.int64, device=dev)
        eng._graphs.clear()
        _lib.STAMP = dict(tag=tag, buf=buf, n=0)
        try:
            step()                      # eager: keys seen
            _lib.STAMP["n"] = 0
            step()                      # captured (+ replayed)
            n = _lib.STAMP["n"]
            step()                      # replay: the stamps of this pass are read
            torch.cuda.synchronize()
        finally:
            _lib.STAMP = None
            eng._graphs.clear()
        t = buf[: 2 * n].view(n, 2).cpu()
        raw = float((t[:, 1] - t[:, 0]).double().mean()) / 100.0   # 100 MHz ticks -> us
        # empty brackets in a small graph: stamp, stamp
        lib, base = _lib.load(), buf.data_ptr() + 16 * cap
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for i in range(16):
                lib.care_timestamp(base + 16 * i, _lib.stream_ptr())
                lib.care_timestamp(base + 16 * i + 8, _lib.stream_ptr())
        g.replay()
        torch.cuda.synchronize()
        e = buf[2 * cap: 2 * cap + 32].view(16, 2).cpu()
        return raw, float((e[:, 1] - e[:, 0]).double().mean()) / 100.0, n

    graph_us = graph_ovh = None
    if not args.no_graph and args.lanes == 1:
        graph_us, graph_ovh, _n = stamped_us(dom)
    dur_s = kernels[dom]["avg_us"] * 1e-6
    if km["bound"] == "hbm":
        achieved, peak, unit = km["bytes"] / dur_s / 1e9, HBM_PEAK_GBS, "GB/s"
    else:
        achieved, peak, unit = km["flops"] / dur_s / 1e12, MFMA_PEAK_TF[args.dtype], "TFLOP/s"
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        key = "{}|{}|B{}|{}".format(args.config, args.dtype, B, dom)
        if dom == "step_cross_attn" and not eng.latent_for(B):
            key += "|projected_kv"  # the K/V-reading kernel (CARE_LATENT=0, fp32, or < 2048 rows)
        traffic = tj.get(key)
    roofline = dict(kernel=dom, bound=km["bound"], achieved=round(achieved, 2), peak=peak, unit=unit,
                    frac=round(achieved / peak, 4), traffic=traffic,
                    avg_launch_us=round(kernels[dom]["avg_us"], 2), avg_launch_us_back_to_back=round(dom_b2b_us, 2),
                    avg_launch_us_in_graph_replay=None if graph_us is None else round(graph_us - graph_ovh, 2),
                    in_graph_bracket_raw_us=None if graph_us is None else round(graph_us, 2),
                    in_graph_bracket_overhead_us=None if graph_ovh is None else round(graph_ovh, 2),
                    frac_in_graph_replay=None if graph_us is None else round(
                        (km["bytes"] / 1e9 if km["bound"] == "hbm" else km["flops"] / 1e12) / ((graph_us - graph_ovh) * 1e-6) / peak, 4),
                    duration_sources="avg_launch_us: HIP events around every launch of an eager pass (agrees with "
                                     "rocprofv3's trace of an eager run); avg_launch_us_in_graph_replay: device "
                                     "timestamps inside the replayed hipGraph = the timed configuration",
                    launches=kernels[dom]["launches"],
                    algorithmic_bytes_per_launch=int(km["bytes"]), algorithmic_flops_per_launch=int(km["flops"]))
    per_kernel = {}
    for tag, k in sorted(kernels.items(), key=lambda kv: -kv[1]["total_ms"]):
        m = kernel_model(tag, eng, B, args.dtype)
        ent = dict(launches=k["launches"], avg_us=round(k["avg_us"], 2), total_ms=round(k["total_ms"], 3))
        if m:
            ent["GBps"] = round(m["bytes"] / (k["avg_us"] * 1e-6) / 1e9, 1)
            ent["TFLOPs"] = round(m["flops"] / (k["avg_us"] * 1e-6) / 1e12, 2)
        per_kernel[tag] = ent

    # ---- whole-pass algorithmic work: FLOPs per caption
    d, ff, V, Lk, T = eng.d, eng.ff, eng.V, eng.Lk, eng.T
    enc_fl = sum(2 * eng.rows_of[ch] * d * opt["dim_" + ch] for ch in eng.modality)
    step_fl = 2 * d * d * 6 + 4 * d * ff + 2 * d * V + 4 * Lk * d
    total_fl = enc_fl + 4 * Lk * d * d + sum(step_fl + 4 * t * d for t in range(1, T + 1))

    # ---- CPU baseline: the oracle (reference algorithm as written) on the host cores
    cpu = None
    if not args.no_cpu_baseline and world == 1:  # contract: rank 0 at N=1 only
        from oracle import care_cpu  # timed baseline only (never on the product path)

        cb = args.cpu_batch
        torch.set_num_threads(args.cpu_threads)
        cfeats = synth_feats(2000, feat_shapes(opt, cb))
        care_cpu.translate_batch(P, opt, cfeats)  # warm-up
        t1 = time.perf_counter()
        passes = 0
        while passes < 30 and (time.perf_counter() - t1 < 12.0 or passes < 2):
            care_cpu.translate_batch(P, opt, cfeats)
            passes += 1
        cpu_s = (time.perf_counter() - t1) / passes
        cpu = dict(value=round(cb / cpu_s, 2), unit="captions/s", cores=torch.get_num_threads(), kind="port",
                   sample="{} passes of greedy translate_batch, B={} clips, fp32, full-prefix recompute as in the "
                          "reference; {:.0f} us per decoder step; {} torch threads (fastest of 8..128 on this "
                          "host)".format(passes, cb, cpu_s / T * 1e6, args.cpu_threads),
                   host_cpus=os.cpu_count())

    line = dict(
        metric="captions/sec (greedy)", value=round(value, 1), unit="captions/s", n_gpus=world, steps=args.steps,
        warmup=args.warmup, ms_per_step=round(ms_per_step, 3), higher_is_better=True, scaling="weak",
        vs_baseline=None, dtype={"bf16": "bf16", "fp16": "f16", "fp32": "f32", "fp16x3": "f16x3"}[args.dtype], data="synthetic",
        config=dict(workload="MSRVTT Transformer/base task=Base feats=ViT modality=ami greedy "
                             "(BASELINE.json configs[1]): [B,28,128]+[B,28,2048]+[B,28,512] fp32 feats, d=512, "
                             "V=10547, 29 decoder steps" if args.config == "msrvtt_base_ami" else args.config,
                    config_name=args.config, clips_per_gpu_per_step=B, global_batch=B * world, lanes=args.lanes,
                    parallelism="batch-sharded dp{} (no data-path collective; all-gather of results)".format(world),
                    hip_graph=not args.no_graph, absorbed_cross_attention=bool(eng.latent_for(B)),
                    lean_encode=bool(eng.lean_ok),  # the Translator's call: bf16 memory only, no unused fp32 copy / frame means
                    work_done_per_step=work_done),
        decoder_step_us=round(ms_per_step * 1e3 * (1 - (kernels.get("enc_gemm", {"total_ms": 0})["total_ms"] +
                                                         kernels.get("cross_kv_gemm", {"total_ms": 0})["total_ms"]) /
                                                    max(tagged_ms, 1e-9)) / T, 2),
        gflop_per_caption=round(total_fl / 1e9, 4),
        pass_tflops=round(total_fl * value / 1e12, 2),
        roofline=roofline, kernels=per_kernel, cpu_baseline=cpu)
    if dist_info:
        line["distributed"] = dist_info
    if world == 1 and not args.no_legs and args.beam == 1:
        del model, eng, feats
        torch.cuda.empty_cache()
        legs = {}
        try:   # a leg that fails (e.g. no pinned host memory on the box) must not take the headline line with it
            extra_legs(dev, args.dtype, legs)
        except Exception as exc:  # noqa: BLE001 - reported in the line
            import traceback
            legs["error"] = "{}: {}".format(type(exc).__name__, exc)
            legs["error_where"] = traceback.format_exc().strip().splitlines()[-3:]
        line["legs"] = legs
        # the same workload in the default 16-bit mode (`fp16`: model.set_compute_dtype("half")) beside the bf16 headline
        # BASELINE.json names, and what the boundary costs (legs api_*)
        if "msrvtt_base_ami_fp16" in legs:
            line["config"]["fp16_mode_captions_per_s"] = legs["msrvtt_base_ami_fp16"]["captions_per_s"]
        if "api_greedy_B32768" in legs:
            line["config"]["through_translate_batches_captions_per_s"] = legs["api_greedy_B32768"]["api_pipelined_captions_per_s"]
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
