"""Builds the committed profile artefacts of a round from gpurun_out/prof_<tag>/ (made by tools/profile_round.sh):
profiles/<tag>_*.{json,csv,txt} copies, profiles/<tag>_pmc_traffic.json (what bench.py reports as roofline.traffic)
and profiles/<tag>_summary.txt.  usage: python tools/make_profile_summary.py r01"""
import csv, json, os, re, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof_" + tag), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
for f in os.listdir(src):
    if f.startswith(tag + "_"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
L = lambda n: json.load(open(os.path.join(src, "%s_pmc_%s.json" % (tag, n))))
fetch, write, sq, sq2, l2 = L("fetch"), L("write"), L("sq"), L("sq2"), L("l2")
bench = json.loads(open(os.path.join(src, tag + "_bench_default.json")).read().strip().splitlines()[-1])
under = json.loads(open(os.path.join(src, tag + "_bench_under_rocprof.json")).read().strip().splitlines()[-1])
g = lambda d, k, c: d.get(k, {}).get(c, {}).get("avg", 0.0)
ours = sorted(k for k in sq if k.startswith("igemm") or any(s in k for s in ("render_crop", "mano_", "bn_", "raster", "huber", "col_sum", "joint2offset", "offset2joint", "x6_split", "adamw")))
traffic = {"source": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of `python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline` "
                     "(tools/profile_round.sh); FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE doubled (gfx950 reports half "
                     "the bytes of 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section); counts L2 misses to the fabric, "
                     "Infinity-Cache hits included", "kernels": {}}
for k in ours:
    fb, wb = g(fetch, k, "FETCH_SIZE") * 1024 * 2, g(write, k, "WRITE_SIZE") * 1024
    traffic["kernels"][k] = {"fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                             "launches_sampled": int(fetch.get(k, {}).get("FETCH_SIZE", {}).get("n", 0))}
json.dump(traffic, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
stats = {}
for r in csv.DictReader(open(os.path.join(src, tag + "_kernel_stats.csv"))):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    stats[(m.group(1) if m else name)[:120]] = r
out = []
out.append("%s profile summary (MI355X, B=32 ResNet_stage_18 2-stage step; see tools/profile_round.sh)" % tag)
out.append("bench default : %.2f img/s, %.3f ms/step | under rocprofv3 --kernel-trace: %.2f img/s" % (bench["value"], bench["ms_per_step"], under["value"]))
r = bench["roofline"]
out.append("dominant kernel (bench, live HIP-event replay): %s  %.1f us/launch  %.1f TFLOP/s  frac %.3f of %.1f" % (r["kernel"], r["avg_launch_us"], r["achieved"], r["frac"], r["peak"]))
st = stats.get(r["kernel"])
if st:
    out.append("same kernel in rocprofv3 --kernel-trace --stats  : %.1f us average over %s calls (all layers routed to it)" % (float(st["AverageNs"]) / 1e3, st["Calls"]))
out.append("cpu baseline  : %s" % json.dumps(bench.get("cpu_baseline")))
out.append("")
out.append("per-kernel PMC (averages per launch over the profiled run; clk = GRBM_GUI_ACTIVE/8/duration; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES /")
out.append("(GRBM_GUI_ACTIVE/8 * 1024 SIMDs); waves/simd = 4*SQ_WAVE_CYCLES / (cycles * 1024); L2hit = TCC_HIT/(TCC_HIT+TCC_MISS); MB = corrected bytes)")
out.append("%-44s %6s %8s %6s %9s %10s %8s %8s %8s %7s %9s %9s" % ("kernel", "n", "us", "clkGHz", "mfma_util", "waves/simd", "wait_any", "wait_ins", "active", "L2hit", "fetchMB", "writeMB"))
for k in ours:
    a = sq[k]; dur = a["_duration_ns_under_pmc"]["avg"]; cyc = g(sq, k, "GRBM_GUI_ACTIVE") / 8.0
    wc = max(g(sq, k, "SQ_WAVE_CYCLES"), 1.0); h, m = g(l2, k, "TCC_HIT_sum"), g(l2, k, "TCC_MISS_sum")
    out.append("%-44s %6d %8.1f %6.2f %9.3f %10.2f %8.2f %8.2f %8.2f %7.3f %9.2f %9.2f" % (
        k[:44], a["_duration_ns_under_pmc"]["n"], dur / 1e3, cyc / max(dur, 1), g(sq, k, "SQ_VALU_MFMA_BUSY_CYCLES") / max(cyc * 1024, 1),
        wc * 4 / max(cyc * 1024, 1), g(sq, k, "SQ_WAIT_ANY") / wc, g(sq, k, "SQ_WAIT_INST_ANY") / wc, g(sq, k, "SQ_ACTIVE_INST_ANY") / wc,
        h / max(h + m, 1e-9), traffic["kernels"][k]["fetch_bytes_per_launch_corrected"] / 1e6, traffic["kernels"][k]["write_bytes_per_launch"] / 1e6))
out.append("")
out.append("LDS: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS per kernel")
for k in ours:
    if k.startswith("igemm"):
        out.append("  %-44s %.3f" % (k[:44], g(sq2, k, "SQ_LDS_BANK_CONFLICT") / max(g(sq2, k, "SQ_ACTIVE_INST_LDS"), 1.0)))
out.append("")
out.append(open(os.path.join(src, tag + "_kernel_categories.txt")).read())
open(os.path.join(dst, tag + "_summary.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:60]))
