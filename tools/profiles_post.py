"""gpurun_out/<round>prof/* (tools/profiles.sh) -> profiles/<round>_*, and the "Readings" section of profiles/README.md
REGENERATED from the committed files (so the text cannot drift from the data):

    python tools/profiles_post.py [round]          # default r06; copies + regenerates
    python tools/profiles_post.py [round] --readme # only regenerate the readings from profiles/<round>_*
"""
import collections
import csv
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = next((a for a in sys.argv[1:] if not a.startswith("-")), "r06")
SRC = os.path.join(ROOT, "gpurun_out", ROUND + "prof")
DST = os.path.join(ROOT, "profiles")
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()

STATS = {"trace": "bench_B32768_bf16", "trace_vatex": "vatex_care_large_B4096_bf16", "trace_vatex16k": "vatex_care_large_B16384_bf16", "greedy_B128": "small_batch_greedy_B128",
         "greedy_B1": "small_batch_greedy_B1", "beam5_B128": "small_batch_beam5_B128", "beam5_B1": "small_batch_beam5_B1",
         "train_B64": "training_step_B64", "train_B512": "training_step_B512", "train_B512_f32": "training_f32_B512", "trace_fp16": "bench_B32768_fp16", "beam5_chain_B128": "small_batch_beam5_chain_B128",
         "beam5_multilaunch_B512": "mid_batch_beam5_multilaunch_B512"}
ours = lambda k: ("anonymous namespace" in k or k.startswith("_ZN12_GLOBAL__N_1") or k.startswith("_Z17split2_act")) and "at::native" not in k


def counters(name):
    acc = collections.defaultdict(dict)
    path = os.path.join(SRC, name + "_counters.txt")
    if not os.path.exists(path):
        return acc
    for line in open(path):
        k, c, v, n = line.rstrip("\n").split("\t")
        acc[k][c] = (float(v), int(n))
    return acc


def copy_in():
    for src, dst in STATS.items():
        f = os.path.join(SRC, src + "_kernel_stats.csv")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(DST, "{}_{}_kernel_stats.csv".format(ROUND, dst)))
    for name in ("resident_phase_clocks.txt", "beam_sweep.txt", "greedy_sweep.txt"):
        f = os.path.join(SRC, name)
        if os.path.exists(f):
            txt = "".join(l for l in open(f) if "amdgpu.ids" not in l)
            open(os.path.join(DST, "{}_{}".format(ROUND, name)), "w").write("# build {}\n".format(head) + txt)
    # hardware counters PER PHASE of the beam step (the chained form: one kernel per phase of the resident launch)
    parts = [os.path.join(SRC, n + "_counters.txt") for n in ("chain_sq1", "chain_sq2")]
    if all(os.path.exists(f) for f in parts):
        with open(os.path.join(DST, ROUND + "_chain_phase_counters.txt"), "w") as out:
            out.write("# build {}\n# rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/chain_prof.py 128 1 (two passes); mean per launch, "
                      "640 rows = 128 clips x beam 5; kernels = the phases of the resident beam launch (csrc/decode_chain.hip)\n".format(head))
            for f in parts:
                out.write(open(f).read())
    fetch, write = counters("fetch"), counters("write")
    if fetch or write:
        pm = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --no-cpu-baseline --no-legs --steps 1 --warmup 2 "
                         "--no-graph; a second pass with --pmc WRITE_SIZE (tools/profiles.sh)",
              "build": head,
              "note": "average per launch, KB as rocprofv3 reports them; hbm_bytes = 2 x FETCH_SIZE (gfx950 tallies wide 16-B/lane "
                      "streaming reads at half, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, x 1024",
              "kernels": []}
        for k in sorted(set(fetch) | set(write)):
            if not ours(k):
                continue
            f = fetch.get(k, {}).get("FETCH_SIZE", (0.0, 0))
            w = write.get(k, {}).get("WRITE_SIZE", (0.0, 0))
            pm["kernels"].append(dict(kernel=k, launches=max(f[1], w[1]), FETCH_SIZE_KB=round(f[0], 1), WRITE_SIZE_KB=round(w[0], 1),
                                      hbm_bytes=int((2 * f[0] + w[0]) * 1024)))
        json.dump(pm, open(os.path.join(DST, ROUND + "_bench_B32768_bf16_pmc_fetch_write.json"), "w"), indent=1)
        # bench.py reads the dominant kernel's traffic from profiles/traffic.json
        tj_path = os.path.join(DST, "traffic.json")
        tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
        for e in pm["kernels"]:
            if "attention_latent_kernel<4, 2>" in e["kernel"] or "attention_latent_kernel<4,2>" in e["kernel"]:
                tj["msrvtt_base_ami|bf16|B32768|step_cross_attn"] = e["hbm_bytes"]
        json.dump(tj, open(tj_path, "w"), indent=1)
    for tag, fname, cmd in (("sq", "_sq_counters.json", "python3 tools/pmc_target.py 32768"),
                            ("sq_resident", "_sq_counters_resident_beam5_B128.json",
                             "python3 bench.py --batch 128 --beam 5 --config msrvtt_care_beam5 --no-graph"),
                            ("icache_resident", "_icache_resident_beam5_B128.json",
                             "python3 bench.py --batch 128 --beam 5 --config msrvtt_care_beam5 --no-graph")):
        sq = counters(tag)
        if not sq:
            continue
        out = {"command": "rocprofv3 --pmc <counters below> --kernel-trace -- " + cmd, "build": head,
               "note": "average per launch.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are summed over waves; parked = SQ_WAIT_ANY / "
                       "SQ_WAVE_CYCLES (waves at s_waitcnt / barriers / sleeping polls), issue_stall = SQ_WAIT_INST_ANY / "
                       "SQ_WAVE_CYCLES, active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES", "kernels": []}
        for k, cs in sq.items():
            if not ours(k):
                continue
            ent = dict(kernel=k, launches=next(iter(cs.values()))[1])
            for c, (v, _) in sorted(cs.items()):
                ent[c] = int(v)
            wc = max(ent.get("SQ_WAVE_CYCLES", 0), 1)
            if "SQ_WAVE_CYCLES" in ent:
                ent["parked_frac"] = round(ent.get("SQ_WAIT_ANY", 0) / wc, 3)
                ent["issue_stall_frac"] = round(ent.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
                ent["active_frac"] = round(ent.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
            out["kernels"].append(ent)
        json.dump(out, open(os.path.join(DST, ROUND + fname), "w"), indent=1)


def kernel_stats(name):
    path = os.path.join(DST, "{}_{}_kernel_stats.csv".format(ROUND, name))
    if not os.path.exists(path):
        return []
    return list(csv.DictReader(open(path)))


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", k)[:80]


def readings():
    L = ["## Round {} - readings (GENERATED by tools/profiles_post.py from the files of this directory; do not edit)".format(ROUND[1:].lstrip("0")), ""]
    rows = kernel_stats("bench_B32768_bf16")
    if rows:
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        L += ["`{}_bench_B32768_bf16_kernel_stats.csv` (`rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-legs "
              "--steps 5 --warmup 2`), our kernels by total time:".format(ROUND), "", "| kernel | calls | average us | share |", "|---|---|---|---|"]
        for r in rows[:14]:
            L.append("| `{}` | {} | {:.1f} | {:.1f} % |".format(short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                 100 * float(r["TotalDurationNs"]) / tot))
        L.append("")
        dom = next((r for r in rows if "attention_latent_kernel<4, 2>" in r["Name"] or "attention_latent_kernel<4,2>" in r["Name"]), None)
        pj = os.path.join(DST, ROUND + "_bench_B32768_bf16_pmc_fetch_write.json")
        if dom and os.path.exists(pj):
            e = next((k for k in json.load(open(pj))["kernels"] if "attention_latent_kernel<4" in k["kernel"]), None)
            alg = 32768 * (84 * 512 * 2 + 2 * 8 * 512 * 2)
            us = float(dom["AverageNs"]) / 1e3
            L.append("Dominant kernel `attention_latent_kernel<4, 2>`: {:.1f} us average over {} launches in the trace = {:.2f} TB/s of the "
                     "{:,} algorithmic bytes = **{:.3f} of the 8 TB/s peak**{}.".format(
                         us, dom["Calls"], alg / us / 1e6, alg, alg / us / 1e6 / 8.0,
                         "; PMC traffic {:,} B per launch = {:.4f} x algorithmic".format(e["hbm_bytes"], e["hbm_bytes"] / alg) if e else ""))
            L.append("")
    for name, what in (("small_batch_greedy_B1", "greedy, 1 clip"), ("small_batch_greedy_B128", "greedy, 128 clips"),
                       ("small_batch_beam5_B1", "beam 5, 1 clip"), ("small_batch_beam5_B128", "beam 5, 128 clips (640 rows)")):
        rows = kernel_stats(name)
        r = next((r for r in rows if "decode_resident" in r["Name"]), None)
        if r:
            L.append("`{}_{}_kernel_stats.csv` ({}): `{}` {:.1f} us per launch = {:.1f} us per decoder step of 29 ({} launches).".format(
                ROUND, name, what, short(r["Name"]), float(r["AverageNs"]) / 1e3, float(r["AverageNs"]) / 29e3, r["Calls"]))
    rows = [r for r in kernel_stats("vatex_care_large_B16384_bf16") if ours(r["Name"])]
    if rows:
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        L += ["", "`{}_vatex_care_large_B16384_bf16_kernel_stats.csv` (d_model 1024 at the batch that fills the chip), our kernels by share: ".format(ROUND) +
              ", ".join("`{}` {:.1f} % ({:.0f} us)".format(short(r["Name"])[:48], 100.0 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3)
                        for r in rows[:6]) + "."]
    rows = kernel_stats("training_step_B64")
    if rows:
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        calls = sum(int(r["Calls"]) for r in rows)
        L += ["", "`{}_training_step_B64_kernel_stats.csv` (`tools/train_prof.py 64 10`: 13 steps): {:.2f} ms of kernels and {:.0f} launches per "
              "training step; top: ".format(ROUND, tot / 13 / 1e6, calls / 13) +
              ", ".join("`{}` {:.2f} ms".format(short(r["Name"])[:40], float(r["TotalDurationNs"]) / 13 / 1e6) for r in rows[:5]) + "."]
    for nm, what, n in (("training_step_B512", "`tools/train_prof.py 512 5`, CARE_TRAIN_GEMM=auto: the split products", 8),
                        ("training_f32_B512", "the same with CARE_TRAIN_GEMM=f32: every product exact f32", 8)):
        rows = kernel_stats(nm)
        if rows:
            tot = sum(float(r["TotalDurationNs"]) for r in rows)
            L += ["", "`{}_{}_kernel_stats.csv` ({}; {} steps): {:.2f} ms of kernels per training step of 512 clips; top: ".format(ROUND, nm, what, n, tot / n / 1e6) +
                  ", ".join("`{}` {:.2f} ms".format(short(r["Name"])[:40], float(r["TotalDurationNs"]) / n / 1e6) for r in rows[:6]) + "."]
    for fname, title in ((ROUND + "_sq_counters.json", "SQ counters at 32768 rows (tools/pmc_target.py)"),
                         (ROUND + "_sq_counters_resident_beam5_B128.json", "SQ counters of the resident beam launch (128 clips x 5)")):
        path = os.path.join(DST, fname)
        if os.path.exists(path):
            L += ["", "`{}` - {}:".format(fname, title), "", "| kernel | parked | issue-stalled | issuing | LDS bank-conflict cycles |", "|---|---|---|---|---|"]
            for e in json.load(open(path))["kernels"]:
                if "resident" in fname and "decode_resident" not in e["kernel"]:
                    continue
                if "parked_frac" in e:
                    L.append("| `{}` | {:.0%} | {:.0%} | {:.0%} | {} |".format(short(e["kernel"]), e["parked_frac"], e["issue_stall_frac"],
                                                                        e["active_frac"], e.get("SQ_LDS_BANK_CONFLICT", "-")))
    path = os.path.join(DST, ROUND + "_icache_resident_beam5_B128.json")
    if os.path.exists(path):
        for e in json.load(open(path))["kernels"]:
            if "SQC_ICACHE_REQ" in e and "decode_resident" in e["kernel"]:
                L += ["", "`{}_icache_resident_beam5_B128.json`: `{}` {:,} instruction-cache requests per launch, {:,} misses + {:,} duplicate "
                      "misses ({:.2%}).".format(ROUND, short(e["kernel"]), e["SQC_ICACHE_REQ"], e.get("SQC_ICACHE_MISSES", 0),
                                                e.get("SQC_ICACHE_MISSES_DUPLICATE", 0),
                                                (e.get("SQC_ICACHE_MISSES", 0) + e.get("SQC_ICACHE_MISSES_DUPLICATE", 0)) / max(e["SQC_ICACHE_REQ"], 1))]
    rows = kernel_stats("bench_B32768_fp16")
    dom = next((r for r in rows if "attention_latent_kernel<4" in r["Name"]), None)
    if dom:
        tot = sum(float(r["TotalDurationNs"]) for r in rows if ours(r["Name"]))
        L += ["", "`{}_bench_B32768_fp16_kernel_stats.csv` (the headline workload in fp16 mode, libcare_hip_f16.so): dominant kernel {:.1f} us "
              "average, {:.1f} % of our kernels' time.".format(ROUND, float(dom["AverageNs"]) / 1e3, 100 * float(dom["TotalDurationNs"]) / tot)]
    rows = [r for r in kernel_stats("small_batch_beam5_chain_B128") if "chain_" in r["Name"]]
    if rows:
        def phase(name):  # chain_gemm_kernel<K, AMODE, EPI, ...> -> the phase it is (csrc/decode_resident.h enums)
            m = re.search(r"chain_gemm_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name) or re.search(r"chain_gemm_kernel<(\d+), (\d+), (\d+)", name)
            if not m:
                return short(name)[:40]
            epi = {0: "QKV", 1: "query", 2: "dense + residual (x 2)", 3: "FFN dense1", 5: "vocabulary (groups)"}.get(int(m.group(3)), "gemm")
            return epi + (" K=" + m.group(1) if m.group(1) != "512" else "")
        adv = next((r for r in rows if "chain_advance" in r["Name"]), None)
        steps = float(adv["Calls"]) if adv else 29.0  # one advance per decoder step
        per_step = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
        L += ["", "`{}_small_batch_beam5_chain_B128_kernel_stats.csv` (`tools/chain_prof.py 128 3`: the beam step as a chain of kernels, eager): "
              "{:.0f} us of kernels per decoder step; by phase (average per launch): ".format(ROUND, per_step) +
              ", ".join("{} {:.1f} us".format(phase(r["Name"]), float(r["AverageNs"]) / 1e3) for r in rows[:10]) + "."]
    path = os.path.join(DST, ROUND + "_chain_phase_counters.txt")
    if os.path.exists(path):
        L += ["", "`{}_chain_phase_counters.txt`: SQ counters per phase of the beam step at 640 rows (VALU / MFMA / LDS / VMEM instruction "
              "counts, wave cycles, waits, LDS bank conflicts) - the resident launch is ONE kernel, the chain gives its phases one each.".format(ROUND)]
    for name in ("resident_phase_clocks.txt", "beam_sweep.txt", "greedy_sweep.txt", "embedder_ablation.txt", "embedder_streams.txt",
                 "embedder_timeline.txt", "mfma_kernel_timelines.txt"):
        if os.path.exists(os.path.join(DST, "{}_{}".format(ROUND, name))):
            what = {"resident_phase_clocks.txt": "device clock at every phase boundary of decoder step 3, workgroup 0 (tools/resident_prof.py)",
                    "beam_sweep.txt": "beam 5 by batch: resident launch / chained step / multi-launch search, whole passes (tools/beam_sweep.py)",
                    "greedy_sweep.txt": "greedy by batch, d_model 512 and 1024, whole passes (tools/greedy_sweep.py)",
                    "embedder_ablation.txt": "not rocprofv3: `tools/emb_bench.py` on the shipped library and on libraries with ONE translation unit "
                                             "rebuilt with a switch (`tools/variant_lib.py gemm_ln.hip ... -DCARE_LN3_DBG=n` / `-DCARE_LN_DBG=n`) - the "
                                             "loader-wave embedder (version 3) and version 2 at 32768 clips, whole and without MFMAs / DMA streams / "
                                             "fragment reads / epilogue / stores; the concept models' split form, both versions, bit-identity checked",
                    "embedder_streams.txt": "`tools/micro/emb_stream`: the K = 2048 launch's two operand streams alone, by piece shape, depth, cache "
                                            "policy, LDS-DMA or registers, with and without a barrier per K step",
                    "embedder_timeline.txt": "`tools/emb_ts.py` (`-DCARE_LN3_DBG=64`): `s_memtime` stamps per K step of workgroup 0 - a compute wave, a "
                                             "weight-loader wave, a feature-loader wave",
                    "mfma_kernel_timelines.txt": "the same kind of stamps for the LDS-tiled GEMM (`tools/tile_ts.py`, two shapes), the vocabulary arg-max "
                                                 "(`tools/v32_ts.py`), the 256-row store GEMM (`tools/s32_ts.py`), and `tools/micro/mfma_chain` (issue "
                                                 "interval of dependent MFMAs); DESIGN.md 5.4 and 11 read these four files"}[name]
            L += ["", "`{}_{}`: {}".format(ROUND, name, what)]
    return "\n".join(L) + "\n"


if "--readme" not in sys.argv:
    copy_in()
path = os.path.join(DST, "README.md")
txt = open(path).read()
marker = "## Round {} - readings (GENERATED".format(ROUND[1:].lstrip("0"))
if marker in txt:
    txt = txt[: txt.index(marker)].rstrip() + "\n\n"
else:
    txt = txt.rstrip() + "\n\n"
open(path, "w").write(txt + readings())
print("profiles/README.md: readings of", ROUND, "regenerated")
