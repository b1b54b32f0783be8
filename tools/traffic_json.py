#!/usr/bin/env python3
"""profiles/traffic.json entries from the PMC passes of tools/profile_round.sh:
   tools/traffic_json.py <round tag> <dir with pmc_{fetch,write}_size_config<C>_background_prep_<P>.txt> -> updates profiles/traffic.json
HBM bytes per launch of the compose kernel = FETCH_SIZE (KB) x 2 (gfx950: a wide coalesced read is tallied at half,
MI355X_MICROARCH.md) + WRITE_SIZE (KB)."""
import glob, json, os, re, sys
tag, d = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "traffic.json")
tj = json.load(open(path)) if os.path.exists(path) else {}
WORK = {2: "mode 5, 512x384, batch 32, 16 objects", 3: "mode 9, 512x384, batch 32, 16 objects", 4: "mode 7, 1024x768, batch 8, 32 objects",
        5: "mode 7, 512x384, batch 32, 10000 x 1 MP pool"}


STEP_KERNELS = ("cs_sample_realize", "geom_kernel", "raster_kernel", "bgprep", "compose")


def all_kernels(f, counter):
    """{kernel: counter value} of every kernel of a step in a pmcstats.py summary"""
    name, out = None, {}
    for line in open(f):
        if not line.startswith(" "):
            name = line.split()[0].replace("ofdg::", "")
        elif name and any(k in name for k in STEP_KERNELS) and line.split()[0] == counter:
            out[name] = float(line.split()[1])
    return out


def dominant_kernel(stats_csv):
    """the kernel with the largest share of GPU time in a rocprofv3 --kernel-trace --stats run (kernel_stats.csv), one-off kernels excluded"""
    import csv
    rows = [r for r in csv.DictReader(open(stats_csv)) if r["Name"].startswith("ofdg::") and int(r["Calls"]) > 10]
    if not rows:
        return None
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    return {"kernel": r["Name"].split("(")[0].replace("ofdg::", ""), "share_of_the_steps_kernels": float(r["TotalDurationNs"]) / tot,
            "average_us_per_launch_in_the_pipeline": float(r["AverageNs"]) / 1e3,
            "all": {x["Name"].split("(")[0].replace("ofdg::", ""): {"share": float(x["TotalDurationNs"]) / tot, "average_us": float(x["AverageNs"]) / 1e3} for x in rows}}


def kernel_value(f, counter, which="compose"):
    name, out = None, {}
    for line in open(f):
        if not line.startswith(" "):
            name = line.split()[0]
        elif name and which in name:
            k, v = line.split()
            out[k] = float(v)
            out["kernel"] = name.replace("ofdg::", "")
    return out


for f in sorted(glob.glob(os.path.join(d, "pmc_fetch_size_config*_background_prep_*.txt"))):
    m = re.search(r"config(\d)_background_prep_(\d)", f)
    cfg, bgp = int(m.group(1)), int(m.group(2))
    fe, wr = kernel_value(f, "FETCH_SIZE"), kernel_value(f.replace("fetch", "write"), "WRITE_SIZE")
    fetch2, write = int(fe["FETCH_SIZE"] * 1024 * 2), int(wr["WRITE_SIZE"] * 1024)
    W, H, B = (1024, 768, 8) if cfg == 4 else (512, 384, 32)
    tj["config%d_background_prep_%d" % (cfg, bgp)] = {
        "kernel": fe["kernel"], "background_prep": bgp, "workload": "bench.py --config %d --background-prep %d (%s)" % (cfg, bgp, WORK[cfg]),
        "fetch_size_kb_raw": fe["FETCH_SIZE"], "write_size_kb_raw": wr["WRITE_SIZE"], "fetch_bytes_corrected_x2": fetch2, "write_bytes": write,
        "hbm_bytes_per_launch": fetch2 + write, "algorithmic_bytes_per_launch": 38 * W * H * B, "kernel_us_serialised": fe["dur_us"],
        "source": "committed PMC passes profiles/%s_pmc_fetch_write_size_config%d_background_prep_%d.txt (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                  "runs of bench.py, tools/profile_round.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 note; counter collection serialises the kernels)" % (tag, cfg, bgp)}
    # the whole step: every kernel's bytes at the L2s' memory side, summed (FETCH_SIZE x 2 + WRITE_SIZE per kernel)
    fa, wa = all_kernels(f, "FETCH_SIZE"), all_kernels(f.replace("fetch", "write"), "WRITE_SIZE")
    per_kernel = {k: int(fa.get(k, 0) * 1024 * 2) + int(wa.get(k, 0) * 1024) for k in sorted(set(fa) | set(wa))}  # (the same rounding as hbm_bytes_per_launch)
    tj["config%d_background_prep_%d" % (cfg, bgp)]["whole_step"] = {
        "hbm_bytes_per_step": sum(per_kernel.values()), "by_kernel": per_kernel,
        "over_algorithmic": sum(per_kernel.values()) / (38 * W * H * B),
        "source": "the same two passes, every kernel of the step: FETCH_SIZE x 2 + WRITE_SIZE, summed"}
    stats = os.path.join(d, "kernel_stats.csv")
    if cfg == 2 and bgp == 1 and os.path.exists(stats):
        dk = dominant_kernel(stats)
        if dk:
            dk["source"] = "profiles/%s_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats over the default bench.py command)" % tag
            tj["config%d_background_prep_%d" % (cfg, bgp)]["dominant_kernel_by_gpu_time"] = dk
    # the background preparation's kernel: its own HBM bytes from the same passes, its instruction counts from a third one
    fv = os.path.join(d, "pmc_valu_config%d_background_prep_%d.txt" % (cfg, bgp))
    if bgp and os.path.exists(fv):
        pf, pw, pv = kernel_value(f, "FETCH_SIZE", "bgprep"), kernel_value(f.replace("fetch", "write"), "WRITE_SIZE", "bgprep"), kernel_value(fv, "SQ_INSTS_VALU", "bgprep")
        if pv:
            tj["config%d_background_prep_%d" % (cfg, bgp)]["background_prep_kernel"] = {
                "kernel": pv["kernel"], "valu_instructions_per_launch": pv["SQ_INSTS_VALU"], "salu_instructions_per_launch": pv.get("SQ_INSTS_SALU"),
                "wave_cycles_per_launch": pv.get("SQ_WAVE_CYCLES"), "wave_cycles_waiting_per_launch": pv.get("SQ_WAIT_INST_ANY"),
                "kernel_us_serialised": pv["dur_us"],
                "hbm_bytes_per_launch": int(pf["FETCH_SIZE"] * 1024 * 2 + pw["WRITE_SIZE"] * 1024) if pf and pw else None,
                "source": "profiles/%s_pmc_valu_config%d_background_prep_%d.txt (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY over bench.py, "
                          "tools/profile_round.sh; wave instructions summed over the launch's waves)" % (tag, cfg, bgp)}
            with open(os.path.join(root, "profiles", "%s_pmc_valu_config%d_background_prep_%d.txt" % (tag, cfg, bgp)), "w") as o:
                o.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY over bench.py --config %d --background-prep %d --steps 60; per kernel: "
                        "average per launch (kernels serialised by the collection)\n" % (cfg, bgp))
                o.write(open(fv).read())
    with open(os.path.join(root, "profiles", "%s_pmc_fetch_write_size_config%d_background_prep_%d.txt" % (tag, cfg, bgp)), "w") as o:
        o.write("# rocprofv3 --pmc FETCH_SIZE (first block) and --pmc WRITE_SIZE (second block), separate passes over bench.py --config %d --background-prep %d "
                "--steps 60; per kernel: average of the counter (KB) and of the duration (us; kernels serialised by the collection)\n" % (cfg, bgp))
        o.write(open(f).read() + "\n" + open(f.replace("fetch", "write")).read())
json.dump(tj, open(path, "w"), indent=1)
print("\n".join("%s: %.1f MB per launch (%.3f x algorithmic)" % (k, v["hbm_bytes_per_launch"] / 1e6, v["hbm_bytes_per_launch"] / v["algorithmic_bytes_per_launch"]) for k, v in tj.items()))
