"""Turn gpurun_out/<tag>/ (tools/profile_round.sh) into the committed evidence under profiles/:
<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_traffic.json (per-launch HBM bytes of the persistent kernels,
FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), <tag>_README.md -- and, when the pixel passes were run,
<tag>_pixel_bench.json, <tag>_pixel_kernel_stats.csv, <tag>_pixel_traffic.json (configs[2]: conv / weight-gradient kernels)."""
import collections
import csv
import glob as _glob
import json
import os
import shutil
import sys


class glob:      # newest first: a tag's directory may hold the files of more than one run of a leg
    @staticmethod
    def glob(pattern):
        return sorted(_glob.glob(pattern), key=os.path.getmtime, reverse=True)

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
mean = lambda xs: sum(xs) / max(1, len(xs))


def pmc_rows(sub):
    f = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))[0]
    return list(csv.DictReader(open(f)))


def pmc(sub, select=None):
    """{kernel name: {counter: [values per dispatch]}}; `select(row)` filters dispatches."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in pmc_rows(sub):
        if select is None or select(r):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def traffic_of(fe, wr, sq, names):
    out = {}
    for key, frags in names.items():
        # a name fragment or a list of alternatives (kernels renamed between rounds: first one present wins)
        kn = [k for frag in ([frags] if isinstance(frags, str) else frags) for k in fe if frag in k]
        if not kn:
            continue
        k = kn[0]
        fetch = mean(fe[k]["FETCH_SIZE"]) * 1024 * 2          # KB -> B, x2: gfx950 FETCH_SIZE reads half of a wide stream
        write = mean(wr[k]["WRITE_SIZE"]) * 1024
        hit, miss = sum(wr[k]["TCC_HIT_sum"]), sum(wr[k]["TCC_MISS_sum"])
        c = sq[k]
        wc = mean(c["SQ_WAVE_CYCLES"])
        out[key] = {"kernel": k.split("(")[0], "launches_sampled": len(fe[k]["FETCH_SIZE"]),
                    "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                    "l2_hit_rate": hit / max(1.0, hit + miss), "mfma_f32_insts": mean(c["SQ_INSTS_VALU_MFMA_F32"]),
                    "mfma_busy_cycles": mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]), "gui_active_cycles_8xcd": mean(c["GRBM_GUI_ACTIVE"]),
                    "wait_any_frac": mean(c["SQ_WAIT_ANY"]) / wc, "wait_inst_frac": mean(c["SQ_WAIT_INST_ANY"]) / wc,
                    "active_frac": mean(c["SQ_ACTIVE_INST_ANY"]) / wc}
    return out


def table(f, traffic):
    f.write("| kernel | HBM bytes/launch (FETCH x2 + WRITE) | L2 hit | MFMA f32 insts | MFMA busy cycles | wait / stall / active |\n|---|---|---|---|---|---|\n")
    for key, t in traffic.items():
        f.write(f"| {key} | {t['hbm_bytes_per_launch'] / 1e6:.1f} MB ({t['fetch_bytes'] / 1e6:.1f} + {t['write_bytes'] / 1e6:.1f}) | "
                f"{100 * t['l2_hit_rate']:.1f} % | {t['mfma_f32_insts']:.3g} | {t['mfma_busy_cycles']:.3g} | "
                f"{100 * t['wait_any_frac']:.0f} / {100 * t['wait_inst_frac']:.0f} / {100 * t['active_frac']:.0f} % |\n")


def stats_table(f, rows, n=14):
    f.write("| kernel | calls | avg us | % GPU time |\n|---|---|---|---|\n")
    for r in rows[:n]:
        f.write(f"| `{r['Name'][:64]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")


readme = open(os.path.join(dst, f"{tag}_README.md"), "w")
readme.write(f"# {tag}: rocprofv3 evidence (1x MI355X)\n\nProduced by `tools/profile_round.sh {tag}` on the GPU box and "
             "`tools/summarize_profile.py` here.  Kernel statistics: `rocprofv3 --kernel-trace --stats --output-format csv -- "
             "python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary`; PMC in separate runs (`--pmc FETCH_SIZE`; "
             "`--pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum`; `--pmc SQ_*`), never combined with a trace.\n\n")

# ------------------------------------------------------------------------------------------------ main config
if os.path.exists(os.path.join(src, "bench.json")):
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    stats_csv = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats_csv, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    names = {"imagine_fwd": "imagine_fwd_kernel", "imagine_bwd": "imagine_bwd_kernel", "observe_fwd": ["observe_kfwd_kernel", "observe_cfwd_kernel"],
             "observe_bwd": ["observe_kbwd_kernel", "observe_cbwd_kernel"], "mlp_fwd_tall (34 300-row chains, mean)": "mlp_fwd_tall_kernel",
             "mlp_bwd_tall (34 300-row chains, mean)": "mlp_bwd_tall_kernel", "mlp_fwd (all launches, mean)": "mlp_fwd_kernel",
             "mlp_bwd (all launches, mean)": "mlp_bwd_kernel", "wgrad_wide (all launches, mean)": "wgrad_wide_kernel",
             "actor_entropy (in-kernel Philox draws)": "actor_entropy_kernel"}
    traffic = traffic_of(pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq"), names)
    json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
    dom = bench["roofline"]["kernel"]
    if dom in traffic:
        bench["roofline"]["traffic"] = traffic[dom]["hbm_bytes_per_launch"]
        bench["roofline"]["traffic_source"] = f"{tag}_traffic.json"
    json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"))
    rows = list(csv.DictReader(open(stats_csv)))
    readme.write("## BASELINE configs[1] (`python bench.py`)\n\n")
    cb = bench.get("cpu_baseline", {})
    readme.write(f"Bench line (un-profiled, {bench['steps']} steps): **{bench['value']:.0f} {bench['unit']}**, "
                 f"{bench['ms_per_step']:.3f} ms/step"
                 + (f"; through the reference surface (Dreamer.train_step, lazy logs) {bench['surface_ms_per_step']:.3f} ms/step"
                    if "surface_ms_per_step" in bench else "")
                 + (f"; CPU oracle on {cb['cores']} host cores: {cb['value']:.0f} ({cb['sample']})" if cb else "") + ".\n\n")
    stats_table(readme, rows)
    readme.write("\nHIP-event averages inside bench.py (ms): " + json.dumps(bench["kernel_ms"]) + "\n\n")
    table(readme, traffic)
    if "secondary" in bench and "error" not in bench["secondary"]:
        s2 = bench["secondary"]
        readme.write(f"\nSecondary record of the same run (configs[2], pixels): {s2['ms_per_step']:.2f} ms/step, "
                     f"{s2['value']:.0f} transitions/s; decoder weight-gradient GEMM {s2['roofline']['avg_launch_ms']:.3f} ms/launch = "
                     f"{s2['roofline']['achieved']:.1f} TFLOP/s ({s2['roofline']['frac']:.3f} of the fp32 MFMA peak).\n")

# ------------------------------------------------------------------------------------------------ pixel config
if os.path.exists(os.path.join(src, "pixel_bench.json")):
    pb = json.loads(open(os.path.join(src, "pixel_bench.json")).read().strip().splitlines()[-1])
    pstats = glob.glob(os.path.join(src, "pixel_stats", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(pstats, os.path.join(dst, f"{tag}_pixel_kernel_stats.csv"))
    # the decoder's grouped weight-gradient launch = every 4th wgrad_wide_kernel dispatch of a step (host issue order:
    # decoder [own stream], model, actor, critic); cross-checked as the longest of the four in the kernel trace
    trace = list(csv.DictReader(open(glob.glob(os.path.join(src, "pixel_stats", "*", "*_kernel_trace.csv"))[0])))
    wg = [r for r in trace if "wgrad_wide_kernel" in r["Kernel_Name"]]
    wg.sort(key=lambda r: int(r["Dispatch_Id"]))
    dur = [[], [], [], []]
    for i, r in enumerate(wg):
        dur[i % 4].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    means = [mean(x) for x in dur]
    dec = max(range(4), key=lambda i: means[i])

    def counter(sub):
        rows_ = [r for r in pmc_rows(sub) if "wgrad_wide_kernel" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows_})
        pick = {d for i, d in enumerate(ids) if i % 4 == dec}
        return pmc(sub, lambda r: "wgrad_wide_kernel" not in r["Kernel_Name"] or int(r["Dispatch_Id"]) in pick)

    names = {"wgrad_decoder": "wgrad_wide_kernel", "conv_gemm<1>": "conv_gemm_kernel<1>", "conv_gemm<2>": "conv_gemm_kernel<2>",
             "conv_gemm<4>": "conv_gemm_kernel<4>", "conv_gemm<8>": "conv_gemm_kernel<8>", "conv_patch<1>": "conv_patch_kernel<1>",
             "conv_patch<2>": "conv_patch_kernel<2>", "conv_patch<8>": "conv_patch_kernel<8>",
             "conv_thin_f<12> (Conv2d 3->32 forward)": "conv_thin_f_kernel<12>", "conv_thin_f<27> (ConvT 32->3 dgrad)": "conv_thin_f_kernel<27>",
             "gemm_nt_dma (K = 3200 dgrad)": "gemm_nt_dma_kernel",
             "imagine_fwd (A=17)": "imagine_fwd_kernel", "imagine_bwd (A=17)": "imagine_bwd_kernel"}
    ptraffic = traffic_of(counter("pixel_pmc_fetch"), counter("pixel_pmc_write"), counter("pixel_pmc_sq"), names)
    if "wgrad_decoder" in ptraffic:
        ptraffic["wgrad_decoder"]["trace_avg_us_by_position_in_step"] = [round(m, 1) for m in means]
        ptraffic["wgrad_decoder"]["trace_avg_us"] = means[dec]
    json.dump(ptraffic, open(os.path.join(dst, f"{tag}_pixel_traffic.json"), "w"), indent=1)
    json.dump(pb, open(os.path.join(dst, f"{tag}_pixel_bench.json"), "w"))
    prow = list(csv.DictReader(open(pstats)))
    readme.write("\n## BASELINE configs[2] (`python bench.py --pixel`): 64x64 pixels, A=17, conv stacks on conv.hip\n\n")
    readme.write(f"Bench line (un-profiled, {pb['steps']} steps): **{pb['value']:.0f} {pb['unit']}**, {pb['ms_per_step']:.3f} ms/step.\n\n")
    stats_table(readme, prow, 16)
    readme.write("\nHIP-event averages inside bench.py (ms): " + json.dumps(pb["kernel_ms"]) + "\n\n")
    readme.write(f"`wgrad_wide_kernel` by position in the step (kernel trace, us): {[round(m, 1) for m in means]} -> the decoder's "
                 f"grouped launch is position {dec}.\n\n")
    table(readme, ptraffic)
# ------------------------------------------------------------------------------------------------ serial schedule: kernels alone
ser = glob.glob(os.path.join(src, "serial_stats", "*", "*_kernel_stats.csv"))
if ser:
    shutil.copy(ser[0], os.path.join(dst, f"{tag}_serial_kernel_stats.csv"))
    srow = list(csv.DictReader(open(ser[0])))
    readme.write("\n## The same step on ONE stream (`BD_PIPELINE=0`): every kernel alone on the GPU\n\n"
                 "Un-contended durations of the kernels of configs[1] (`rocprofv3 --kernel-trace --stats`, serial schedule); the table at "
                 "the top has the same kernels under the three-stream pipeline.\n\n")
    stats_table(readme, srow, 12)
# ------------------------------------------------------------------------------------------------ categorical config
if os.path.exists(os.path.join(src, "cat_bench.json")):
    cb = json.loads(open(os.path.join(src, "cat_bench.json")).read().strip().splitlines()[-1])
    cstats = glob.glob(os.path.join(src, "cat_stats", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(cstats, os.path.join(dst, f"{tag}_cat_kernel_stats.csv"))
    names = {"imagine_fwd": "imagine_cat_fwd_kernel", "imagine_bwd": "imagine_cat_bwd_kernel",
             "observe_fwd": ["observe_cat_cfwd_kernel", "observe_cat_fwd_kernel"],
             "observe_bwd": ["observe_cat_cbwd_kernel", "observe_cat_bwd_kernel"],
             "mlp_fwd (all launches, mean)": "mlp_fwd_kernel", "mlp_bwd (all launches, mean)": "mlp_bwd_kernel",
             "wgrad_wide (all launches, mean)": "wgrad_wide_kernel"}
    ctraffic = traffic_of(pmc("cat_pmc_fetch"), pmc("cat_pmc_write"), pmc("cat_pmc_sq"), names)
    json.dump(ctraffic, open(os.path.join(dst, f"{tag}_cat_traffic.json"), "w"), indent=1)
    dom = cb["roofline"]["kernel"]
    if dom in ctraffic:
        cb["roofline"]["traffic"] = ctraffic[dom]["hbm_bytes_per_launch"]
        cb["roofline"]["traffic_source"] = f"{tag}_cat_traffic.json"
    json.dump(cb, open(os.path.join(dst, f"{tag}_cat_bench.json"), "w"))
    crow = list(csv.DictReader(open(cstats)))
    readme.write("\n## BASELINE configs[4] per GPU (`python bench.py --categorical pixel`): dreamerV2 32x32 Categorical latents, "
                 "64x64 pixels, A=17, batch 100 (= 800 / 8)\n\n")
    readme.write(f"Bench line (un-profiled, {cb['steps']} steps): **{cb['value']:.0f} {cb['unit']}**, {cb['ms_per_step']:.3f} ms/step")
    if os.path.exists(os.path.join(src, "cat_state_bench.json")):
        sb = json.loads(open(os.path.join(src, "cat_state_bench.json")).read().strip().splitlines()[-1])
        if os.path.isdir(os.path.join(src, "catstate_pmc_fetch")):       # counters of the state-observation run
            straffic = traffic_of(pmc("catstate_pmc_fetch"), pmc("catstate_pmc_write"), pmc("catstate_pmc_sq"), names)
            json.dump(straffic, open(os.path.join(dst, f"{tag}_catstate_traffic.json"), "w"), indent=1)
            sdom = sb["roofline"]["kernel"]
            if sdom in straffic:
                sb["roofline"]["traffic"] = straffic[sdom]["hbm_bytes_per_launch"]
                sb["roofline"]["traffic_source"] = f"{tag}_catstate_traffic.json"
        json.dump(sb, open(os.path.join(dst, f"{tag}_cat_state_bench.json"), "w"))
        readme.write(f"; the same latents on state observations (`--categorical state`): {sb['value']:.0f}, {sb['ms_per_step']:.3f} ms/step")
    readme.write(".\n\n")
    stats_table(readme, crow, 16)
    readme.write("\nHIP-event averages inside bench.py (ms): " + json.dumps(cb["kernel_ms"]) + "\n\n")
    table(readme, ctraffic)
# ------------------------------------------------------------------------------------------------ kernels alone on the GPU
alone = {k: glob.glob(os.path.join(src, f"alone_{k}", "*", "*_kernel_stats.csv")) for k in ("chain", "wgrad")}
if alone["chain"] or alone["wgrad"]:
    readme.write("\n## Dense-chain and weight-gradient kernels ALONE on the GPU (`tools/profile_alone.sh`)\n\n"
                 "`rocprofv3 --kernel-trace --stats` around `tools/tall_probe.py` (head chain 230 -> 200 x4 -> 1 over 34 300 rows = "
                 "11.43 GFLOP per pass; 16-row kernels against the tall form) and `tools/wgrad_stamps.py` (the critic's five "
                 "weight-gradient GEMMs, 11.43 GFLOP).  Inside the three-stream step the same kernels run contended (table at the top).\n\n")
    GF = 2.0 * 34300 * (230 * 200 + 3 * 200 * 200 + 200) / 1e9
    readme.write("| kernel | calls | avg us | TFLOP/s | of the fp32 MFMA peak (157.3) |\n|---|---|---|---|---|\n")
    for k in ("chain", "wgrad"):
        if not alone[k]:
            continue
        shutil.copy(alone[k][0], os.path.join(dst, f"{tag}_alone_{k}_kernel_stats.csv"))
        for r in csv.DictReader(open(alone[k][0])):
            n = r["Name"]
            if ("mlp_" in n and k == "chain") or ("wgrad_wide" in n and k == "wgrad"):
                us = float(r["AverageNs"]) / 1e3
                tf = GF / us * 1e3
                readme.write(f"| `{n[:70]}` | {r['Calls']} | {us:.1f} | {tf:.1f} | {tf / 157.3:.3f} |\n")
readme.close()
print(open(os.path.join(dst, f"{tag}_README.md")).read())
