"""gpurun_out/r03prof/* (tools/r3_profiles.sh) -> profiles/r03_*: kernel stats, FETCH / WRITE bytes per launch, SQ counters."""
import collections
import json
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r03prof")
DST = os.path.join(ROOT, "profiles")
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()

shutil.copy(os.path.join(SRC, "trace_kernel_stats.csv"), os.path.join(DST, "r03_bench_B32768_bf16_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "trace_vatex_kernel_stats.csv"), os.path.join(DST, "r03_vatex_care_large_B4096_bf16_kernel_stats.csv"))


def counters(name):
    acc = collections.defaultdict(dict)
    for line in open(os.path.join(SRC, name + "_counters.txt")):
        k, c, v, n = line.rstrip("\n").split("\t")
        acc[k][c] = (float(v), int(n))
    return acc


ours = lambda k: "anonymous namespace" in k and "at::native" not in k
fetch, write = counters("fetch"), counters("write")
pm = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --no-cpu-baseline --no-legs --steps 1 --warmup 2 --no-graph; "
                 "a second pass with --pmc WRITE_SIZE (tools/r3_profiles.sh)",
      "build": head,
      "note": "average per launch, KB as rocprofv3 reports them; hbm_bytes = 2 x FETCH_SIZE (gfx950 tallies wide 16-B/lane streaming "
              "reads at half, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, x 1024",
      "kernels": []}
for k in sorted(set(fetch) | set(write)):
    if not ours(k):
        continue
    f = fetch.get(k, {}).get("FETCH_SIZE", (0.0, 0))
    w = write.get(k, {}).get("WRITE_SIZE", (0.0, 0))
    pm["kernels"].append(dict(kernel=k, launches=max(f[1], w[1]), FETCH_SIZE_KB=round(f[0], 1), WRITE_SIZE_KB=round(w[0], 1),
                              hbm_bytes=int((2 * f[0] + w[0]) * 1024)))
json.dump(pm, open(os.path.join(DST, "r03_bench_B32768_bf16_pmc_fetch_write.json"), "w"), indent=1)

sq = counters("sq")
out = {"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES "
                  "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -- python3 tools/pmc_target.py 32768",
       "build": head,
       "note": "average per launch.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are summed over waves, SQ_VALU_MFMA_BUSY_CYCLES over SIMDs "
               "(= 32 x the number of v_mfma_f32_32x32x16_bf16, 16 x the number of 16x16x32, 8 x 16x16x16).  parked = SQ_WAIT_ANY / "
               "SQ_WAVE_CYCLES (waves at s_waitcnt / barriers), issue_stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, active = SQ_ACTIVE_INST_ANY / "
               "SQ_WAVE_CYCLES.  MFMA pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x clock x 1024 SIMDs).",
       "kernels": []}
for k, cs in sq.items():
    if not ours(k):
        continue
    ent = dict(kernel=k, launches=cs["SQ_WAVE_CYCLES"][1])
    for c, (v, _) in sorted(cs.items()):
        ent[c] = int(v)
    wc = max(ent.get("SQ_WAVE_CYCLES", 1), 1)
    ent["parked_frac"] = round(ent.get("SQ_WAIT_ANY", 0) / wc, 3)
    ent["issue_stall_frac"] = round(ent.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
    ent["active_frac"] = round(ent.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
    out["kernels"].append(ent)
json.dump(out, open(os.path.join(DST, "r03_sq_counters.json"), "w"), indent=1)

# the dominant kernel's measured HBM bytes -> traffic.json (bench.py reports it as roofline.traffic)
tj_path = os.path.join(DST, "traffic.json")
tj = json.load(open(tj_path))
lat = [e for e in pm["kernels"] if "attention_latent_kernel<4, 2>" in e["kernel"]]
if lat:
    tj["msrvtt_base_ami|bf16|B32768|step_cross_attn"] = lat[0]["hbm_bytes"]
    tj["_note"] += "  Round 3 (build %s, profiles/r03_bench_B32768_bf16_pmc_fetch_write.json): FETCH %.0f KB x 2 + WRITE %.0f KB." % (
        head, lat[0]["FETCH_SIZE_KB"], lat[0]["WRITE_SIZE_KB"])
    json.dump(tj, open(tj_path, "w"), indent=1)
for e in pm["kernels"]:
    print("%-90s F %12.1f KB  W %12.1f KB  n=%d" % (e["kernel"][:90], e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"], e["launches"]))
print()
for e in out["kernels"]:
    print("%-80s mfma_busy %12d parked %.2f stall %.2f active %.2f conflicts %d" % (
        e["kernel"][:80], e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), e["parked_frac"], e["issue_stall_frac"], e["active_frac"], e.get("SQ_LDS_BANK_CONFLICT", 0)))
