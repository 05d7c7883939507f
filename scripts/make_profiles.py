#!/usr/bin/env python3
"""Turn a gpurun_out/<tag>/ evidence directory (scripts/gpu_full.sh) into the tracked summaries under profiles/.

    python scripts/make_profiles.py <tag> [<round-name>]

Writes profiles/<round>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), profiles/<round>_bench.json,
profiles/<round>_pmc_traffic.csv (FETCH_SIZE / WRITE_SIZE per kernel and grid, mean per dispatch, as reported, KB)
and profiles/traffic.json: HBM bytes per launch of the local_laplacian kernels = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE doubled: gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md §HBM for wide coalesced reads)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else tag
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


ks = glob.glob(os.path.join(src, "kt", "*kernel_stats.csv"))
if ks:
    with open(ks[0]) as f, open(os.path.join(dst, f"{rnd}_kernel_stats.csv"), "w") as g:
        w = csv.writer(g)
        for i, row in enumerate(csv.reader(f)):
            if i:
                row[0] = short(row[0])
            w.writerow(row)
for name in ("bench.json", "bench_1stream.json", "bench_2streams.json", "bench_driver_flags.json", "bench_gpus2.log", "pytest_gpu.log",
             "rocminfo.txt", "smoke.log", "bench_apps.jsonl"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{rnd}_{name}"))

for name in ("apps_pmc.txt", "membench.log", "pmc_widths.txt", "pmc_ll_tcc.txt", "membench_sweep_1024MB.json", "ll_pmc_fma.txt", "ll_pmc_nofma.txt",
             "pytest_gpu_nofma.log"):   # scripts/gpu_r3_evidence.sh
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{rnd}_{name}"))
ka = glob.glob(os.path.join(src, "kt_apps", "*kernel_stats.csv"))
if ka:
    with open(ka[0]) as f, open(os.path.join(dst, f"{rnd}_apps_kernel_stats.csv"), "w") as g:
        w = csv.writer(g)
        for i, row in enumerate(csv.reader(f)):
            if i:
                row[0] = short(row[0])
            w.writerow(row)

def launch_bytes(prefix, csv_name):
    """per-kernel counters of gpurun_out/<tag>/<prefix>FETCH_SIZE + <prefix>WRITE_SIZE -> profiles/<csv_name>, {launch name: HBM bytes}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for p in glob.glob(os.path.join(src, f"{prefix}{ctr}", "*counter_collection.csv")):
            for r in csv.DictReader(open(p)):
                acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for (k, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
        if k.startswith("__amd"):
            continue
        f = sum(v.get("FETCH_SIZE", [0])) / max(1, len(v.get("FETCH_SIZE", [])))
        w = sum(v.get("WRITE_SIZE", [0])) / max(1, len(v.get("WRITE_SIZE", [])))
        rows.append((k, grid, len(v.get("FETCH_SIZE", [])), round(f, 1), round(w, 1), int((2 * f + w) * 1024)))
    if rows:
        with open(os.path.join(dst, csv_name), "w") as g:
            w = csv.writer(g)
            w.writerow(["kernel", "grid_size", "dispatches", "FETCH_SIZE_KB_mean", "WRITE_SIZE_KB_mean", "hbm_bytes_2xfetch_plus_write"])
            w.writerows(rows)
        # name the launches the way libhlmi's timing report does: strips and ups by descending grid size
        per = {}
        # level 1 -> 2 comes out of ll_down01f when that kernel ran (then the strips start at level 2); the level-1 collapse
        # is part of ll_up0f when only two ll_up launches ran (levels 3 and 2)
        for r in rows:   # round 5: levels 3 and 4 from level 2 in one launch, reported as ll_down_strip2:2
            if r[0].startswith("ll_down_strip2"):
                per["ll_down_strip2:2"] = r[5]
        strips = sorted([r for r in rows if r[0].startswith("ll_down_strip") and not r[0].startswith("ll_down_strip2")], key=lambda r: -r[5])
        first_strip = 2 if any(r[0].startswith("ll_down01") for r in rows) else 1
        for i, r in enumerate(strips):
            per[f"ll_down_strip:{i + first_strip}"] = r[5]
        ups = sorted([r for r in rows if r[0] == "ll_up" or r[0].startswith("ll_up<")], key=lambda r: -r[5])
        # the level-1 collapse is part of ll_up0f / ll_up0h (no ll_up:1 launch) whenever one of those ran
        first_up = 2 if any(r[0].startswith("ll_up0f") or r[0].startswith("ll_up0h") for r in rows) else 1
        for i, r in enumerate(ups):
            per[f"ll_up:{i + first_up}"] = r[5]
        for r in rows:
            for base, name in (("ll_down01f", "ll_down01"), ("ll_down01e", "ll_down01"), ("ll_down0f", "ll_down0"), ("ll_down0<", "ll_down0"), ("ll_up0", "ll_up0"),
                               ("ll_top", "ll_top"), ("ll_remap_lut", "ll_remap_lut"), ("ll_mid", "ll_mid:4")):
                if r[0].startswith(base):
                    per[name] = r[5]
            # the multi-level kernels are reported as ll_down_multi:<S> / ll_up_multi:<S>; one instantiation each per run
            for base, depth_to_s in (("ll_down_multi", lambda d: 7 - d), ("ll_up_multi", lambda d: 7 - d)):
                if r[0].startswith(base + "<"):
                    per[f"{base}:{depth_to_s(int(r[0].split('<')[1].split('>')[0]))}"] = r[5]
        return per
    return None


per = launch_bytes("pmc_", f"{rnd}_pmc_traffic.csv")
perq = launch_bytes("pmcq_", f"{rnd}_pmc_traffic_frame_queue.csv")
if per:
    import hashlib
    import subprocess
    ksrc = os.path.join(ROOT, "halide_amd", "csrc", "local_laplacian.hip")
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ""
    # bench.py prints `traffic` only while the kernel source is the one these counters were collected with
    json.dump({"source": f"profiles/{rnd}_pmc_traffic.csv", "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per dispatch",
               "kernel_source": "halide_amd/csrc/local_laplacian.hip", "kernel_source_sha256": hashlib.sha256(open(ksrc, "rb").read()).hexdigest(),
               "git_head_when_collected": head, "input": "3840x2160 frames of bench.py's headline input, uniform noise since round 6 (the re-cut dataflow moves the same bytes for every input)",
               "bytes_per_launch": per,
               "frame_queue_note": "bytes_per_launch_frame_queue: the same counters with HLMI_STREAM_SHARE=4, i.e. the launch geometry of one of four frame queues (what the headline loop runs)",
               "frame_queue_source": f"profiles/{rnd}_pmc_traffic_frame_queue.csv" if perq else None,
               "bytes_per_launch_frame_queue": perq}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
