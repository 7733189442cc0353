"""CPU, 2 processes, gloo: the batch-sharded path's exchange step (care_amd/sharding.py).

The GPU run uses the same code with backend nccl (= RCCL over xGMI).  Each rank
"translates" its shard with the CPU oracle (standing in for the device decode, which needs
a GPU) and the metrics-step all-gather must reproduce the single-process result exactly.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_clips, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from care_amd.configs import feat_shapes, make_opt
        from care_amd.sharding import gather_captions, shard_feats
        from care_amd.synth import synth_feats, synth_state_dict
        from oracle import care_cpu
        from conftest import GoldenCase

        meta = GoldenCase("msvd_base_i_b10").meta
        opt = make_opt("msvd_base_i", max_len=8)
        P = synth_state_dict(5, [(k, tuple(s)) for k, s in meta["state_dict"] if "position_embeddings" not in k] +
                             [("decoder.embedding.position_embeddings.weight", (8, 512))])
        feats = synth_feats(5, feat_shapes(opt, n_clips))
        mine = shard_feats(feats, rank, world)
        T = opt["max_len"] - 1
        n = mine[0].shape[0]
        fed = torch.zeros(n, T + 1, dtype=torch.int32)
        fed[:, 0] = 2
        length = torch.zeros(n, dtype=torch.int32)
        score = torch.zeros(n)
        if n:
            hyps, scores = care_cpu.translate_batch(P, opt, mine)
            for i, (h, s) in enumerate(zip(hyps, scores)):
                fed[i, 1: len(h[0]) + 1] = torch.tensor(h[0], dtype=torch.int32)
                length[i] = len(h[0])
                score[i] = s[0]
        g_fed, g_len, g_score = gather_captions(fed, length, score, n_clips)
        if rank == 0:
            torch.save({"fed": g_fed, "len": g_len, "score": g_score, "P": None}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [5, 4])
def test_sharded_translate_equals_single_process(tmp_path, n_clips):
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_feats, synth_state_dict
    from oracle import care_cpu
    from conftest import GoldenCase

    out_path = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), n_clips, out_path), nprocs=2, join=True)
    got = torch.load(out_path)

    meta = GoldenCase("msvd_base_i_b10").meta
    opt = make_opt("msvd_base_i", max_len=8)
    P = synth_state_dict(5, [(k, tuple(s)) for k, s in meta["state_dict"] if "position_embeddings" not in k] +
                         [("decoder.embedding.position_embeddings.weight", (8, 512))])
    hyps, scores = care_cpu.translate_batch(P, opt, synth_feats(5, feat_shapes(opt, n_clips)))
    assert got["fed"].shape[0] == n_clips          # the padded tail of the ragged shard was dropped
    for i in range(n_clips):
        n = int(got["len"][i])
        assert got["fed"][i, 1: n + 1].tolist() == hyps[i][0]
        assert abs(float(got["score"][i]) - scores[i][0]) < 1e-6


def _attr_worker(rank, world, port, n_clips, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from care_amd.sharding import gather_captions, shard_feats
        from oracle import care_cpu
        from conftest import GoldenCase

        opt, P, feats, _ = GoldenCase("msrvtt_care_eos_b4").build()
        mine = [f[:n_clips] for f in feats]
        mine = shard_feats(mine, rank, world)
        n = mine[0].shape[0]
        T = opt["max_len"] - 1
        fed = torch.zeros(n, T + 1, dtype=torch.int32)
        length = torch.zeros(n, dtype=torch.int32)
        score = torch.zeros(n)
        preds = torch.zeros(n, opt["attribute_prediction_k"])
        if n:
            with torch.no_grad():
                preds = care_cpu.encoding_phase(P, opt, mine)["preds_attr"]
            hyps, scores = care_cpu.translate_batch(P, opt, mine)
            for i, (h, s) in enumerate(zip(hyps, scores)):
                fed[i, 1: len(h[0]) + 1] = torch.tensor(h[0], dtype=torch.int32)
                length[i], score[i] = len(h[0]), s[0]
        g_fed, g_len, g_score, g_preds = gather_captions(fed, length, score, n_clips, preds_attr=preds)
        torch.save({"fed": g_fed, "len": g_len, "score": g_score, "preds": g_preds, "local_preds": preds}, out_path + str(rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [4, 3])
def test_sharded_metrics_step_gathers_concept_probabilities(tmp_path, n_clips):
    """north_star's one exchange: the per-rank concept probabilities travel in the same all-gather as
    the captions, EVERY rank ends with the whole batch, and the reference's concept metrics
    (crit_attribute.py:58-89) on the gathered block equal the single-process ones (fixture values for
    the full batch of 4; ragged 2 + 1 split for 3)."""
    import numpy as np

    from care_amd.metrics import concept_metrics
    from oracle import care_cpu
    from conftest import GoldenCase

    out_path = str(tmp_path / "gathered.pt")
    mp.spawn(_attr_worker, args=(2, _free_port(), n_clips, out_path), nprocs=2, join=True)
    got = [torch.load(out_path + str(r)) for r in range(2)]
    for k in ("fed", "len", "score", "preds"):
        assert torch.equal(got[0][k], got[1][k])                      # all-gather: same on every rank
    g = GoldenCase("msrvtt_care_eos_b4")
    opt, P, feats, _ = g.build()
    # the fp32 bit patterns each rank computed survive the int32 records ...
    assert torch.equal(got[0]["preds"], torch.cat([got[r]["local_preds"] for r in range(2)]))
    with torch.no_grad():  # ... and are the single-process probabilities (up to the CPU GEMM's thread split)
        preds = care_cpu.encoding_phase(P, opt, [f[:n_clips] for f in feats])["preds_attr"]
    assert got[0]["preds"].shape == (n_clips, 500) and (got[0]["preds"] - preds).abs().max() < 1e-6
    labels = torch.from_numpy(g.z["labels_attr"])[:n_clips]
    c = concept_metrics(got[0]["preds"], labels)
    if n_clips == 4:
        np.testing.assert_allclose([c["F1-%02d" % k] for k in (5, 10, 20, 30, 40, 50)] + [c["mAP"]],
                                   g.z["metrics_attr"], rtol=1e-5, atol=1e-7)
    ref_hyps, _ = g.hyps()
    for i in range(n_clips):
        assert got[0]["fed"][i, 1: int(got[0]["len"][i]) + 1].tolist() == ref_hyps[i][0]


def _bench(*args, timeout=300):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(args), env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_launches_its_own_ranks_dry():
    """`python bench.py --gpus 2` (no torchrun around it) starts torch.distributed.run as a child, one process per
    rank; in --dry-run the ranks form a gloo group, push synthetic records through care_amd.sharding's pack /
    all-gather / unpack and rank 0's JSON line comes back through the launcher with the child's exit code."""
    import json

    out = _bench("--gpus", "2", "--dry-run", "--steps", "2")
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["dry_run"] and line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1]
    assert line["records_gathered"] == 128 and line["exchange_ok"]


def test_bench_refuses_more_gpus_than_the_host_has():
    """The plain command on a host with fewer GPUs than asked for: a clear message and a non-zero exit code, before
    anything is launched."""
    import torch

    want = torch.cuda.device_count() + 2
    out = _bench("--gpus", str(want), "--steps", "1", timeout=120)
    assert out.returncode != 0
    assert "--gpus {}".format(want) in out.stderr and "visible" in out.stderr
