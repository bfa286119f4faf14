"""Turn gpurun_out/<tag>/ (tools/profile_round.sh) into the committed evidence under profiles/:
<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_traffic.json (per-launch HBM bytes of the persistent kernels,
FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), <tag>_README.md."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def pmc(sub):
    f = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
stats_csv = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats_csv, os.path.join(dst, f"{tag}_kernel_stats.csv"))
fe, wr, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq")
mean = lambda xs: sum(xs) / max(1, len(xs))
names = {"imagine_fwd": "imagine_fwd_kernel", "imagine_bwd": "imagine_bwd_kernel", "observe_fwd": "observe_cfwd_kernel",
         "observe_bwd": "observe_cbwd_kernel", "mlp_fwd (all launches, mean)": "mlp_fwd_kernel",
         "mlp_bwd (all launches, mean)": "mlp_bwd_kernel", "wgrad_wide (all launches, mean)": "wgrad_wide_kernel"}
traffic = {}
for key, frag in names.items():
    kn = [k for k in fe if frag in k]
    if not kn:
        continue
    k = kn[0]
    fetch = mean(fe[k]["FETCH_SIZE"]) * 1024 * 2          # KB -> B, x2: gfx950 FETCH_SIZE reads half of a wide stream
    write = mean(wr[k]["WRITE_SIZE"]) * 1024
    hit, miss = sum(wr[k]["TCC_HIT_sum"]), sum(wr[k]["TCC_MISS_sum"])
    c = sq[k]
    wc = mean(c["SQ_WAVE_CYCLES"])
    traffic[key] = {"kernel": k.split("(")[0], "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                    "l2_hit_rate": hit / max(1.0, hit + miss), "mfma_f32_insts": mean(c["SQ_INSTS_VALU_MFMA_F32"]),
                    "mfma_busy_cycles": mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]), "gui_active_cycles_8xcd": mean(c["GRBM_GUI_ACTIVE"]),
                    "wait_any_frac": mean(c["SQ_WAIT_ANY"]) / wc, "wait_inst_frac": mean(c["SQ_WAIT_INST_ANY"]) / wc,
                    "active_frac": mean(c["SQ_ACTIVE_INST_ANY"]) / wc}
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
dom = bench["roofline"]["kernel"]
if dom in traffic:
    bench["roofline"]["traffic"] = traffic[dom]["hbm_bytes_per_launch"]
json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"))
rows = list(csv.DictReader(open(stats_csv)))
with open(os.path.join(dst, f"{tag}_README.md"), "w") as f:
    f.write(f"# {tag}: profile of `python bench.py` (BASELINE configs[1], 1x MI355X)\n\n")
    f.write("Produced by `tools/profile_round.sh` on the GPU box and `tools/summarize_profile.py` here.\n"
            "Commands: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 "
            "--no-cpu-baseline`; PMC in separate runs (`--pmc FETCH_SIZE`; `--pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum`; "
            "`--pmc SQ_*`), never combined with a trace.\n\n")
    f.write(f"Bench line (un-profiled, 50 steps): **{bench['value']:.0f} {bench['unit']}**, {bench['ms_per_step']:.3f} ms/step; "
            f"CPU oracle on {bench['cpu_baseline']['cores']} host cores: {bench['cpu_baseline']['value']:.0f} "
            f"({bench['cpu_baseline']['sample']}).\n\n")
    f.write("| kernel | calls (12 steps) | avg us | % GPU time |\n|---|---|---|---|\n")
    for r in rows[:14]:
        f.write(f"| `{r['Name'][:64]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")
    f.write("\nHIP-event averages inside bench.py (ms): " + json.dumps(bench["kernel_ms"]) + "\n\n")
    f.write("| kernel | HBM bytes/launch (FETCH x2 + WRITE) | L2 hit | MFMA f32 insts | MFMA busy / (CU-cycles used) | wait / stall / active |\n|---|---|---|---|---|---|\n")
    for key, t in traffic.items():
        f.write(f"| {key} | {t['hbm_bytes_per_launch'] / 1e6:.1f} MB ({t['fetch_bytes'] / 1e6:.1f} + {t['write_bytes'] / 1e6:.1f}) | "
                f"{100 * t['l2_hit_rate']:.1f} % | {t['mfma_f32_insts']:.3g} | {t['mfma_busy_cycles']:.3g} | "
                f"{100 * t['wait_any_frac']:.0f} / {100 * t['wait_inst_frac']:.0f} / {100 * t['active_frac']:.0f} % |\n")
print(open(os.path.join(dst, f"{tag}_README.md")).read())
