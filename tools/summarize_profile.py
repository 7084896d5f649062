#!/usr/bin/env python3
"""Copies one tools/profile_round.sh run (gpurun_out/<tag>/<workload>/...) into profiles/ as the files the judge and bench.py
read.  Run HERE, on the committed state that was profiled (records `git rev-parse HEAD` and bench.code_hash()).

  bench                      -> <tag>_kernel_stats.csv, <tag>_pmc_means.csv (per-launch means of the timed k_render
                                instantiation), <tag>_traffic.json (HBM bytes per launch: FETCH_SIZE is in KiB and half-counts
                                on gfx950, WRITE_SIZE in KiB -- MI355X_MICROARCH.md, HBM), <tag>_meta.json
  config3/4/5, synthetic_*   -> <tag>_<workload>_kernel_stats.csv and one entry in <tag>_workloads.json: counter sums per UNIT
                                (a frame, a pass, a launch: tools/profile_workloads.py) over every kernel launched after the
                                warm-up marker, plus the per-kernel split

usage: python tools/summarize_profile.py <tag>
(also reads the old layout of tools/profile_bench.sh: gpurun_out/<tag>/{trace,pmc_*})
"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def newest(paths):
    return sorted(paths, key=os.path.getmtime)[-1:]


def short(kernel: str) -> str:
    """`void rtk::dev::k_path<2, true>(...)` -> `k_path<2, true>`"""
    k = re.sub(r"^void\s+", "", kernel)
    k = re.sub(r"\(.*$", "", k)
    return k.split("::")[-1] if "<" not in k else k[k.rfind("::", 0, k.index("<")) + 2:]


def counter_means(path, kernel_sub, stats_marker="true"):
    """mean Counter_Value per counter over the launches of the timed instantiation of the kernel: the most frequent
    non-counting (STATS = false) instantiation"""
    rows = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if kernel_sub not in k:
                continue
            args = k.split("<", 1)[-1].split(",")
            if len(args) > 1 and args[1].strip() == stats_marker:      # the one-off counting launch
                continue
            rows.append(row)
    if not rows:
        return None, {}, 0
    freq = defaultdict(int)
    for row in rows:
        freq[row["Kernel_Name"]] += 1
    name = max(freq, key=freq.get)
    sums, counts = defaultdict(float), defaultdict(int)
    for row in rows:
        if row["Kernel_Name"] != name:
            continue
        sums[row["Counter_Name"]] += float(row["Counter_Value"])
        counts[row["Counter_Name"]] += 1
    return name, {c: sums[c] / counts[c] for c in sums}, (max(counts.values()) if counts else 0)


def unit_sums(path, units):
    """Counter sums over every launch after the LAST k_to_rgb8 launch (the warm-up marker), per unit and per kernel."""
    rows = list(csv.DictReader(open(path, newline="")))
    if not rows:
        return {}, {}, 0
    mark = max((int(r["Dispatch_Id"]) for r in rows if "k_to_rgb8" in r["Kernel_Name"]), default=-1)
    total, per_kernel, launches = defaultdict(float), defaultdict(lambda: defaultdict(float)), set()
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d <= mark or r["Kernel_Name"].startswith("__amd_rocclr"):
            continue
        v = float(r["Counter_Value"])
        total[r["Counter_Name"]] += v
        per_kernel[short(r["Kernel_Name"])][r["Counter_Name"]] += v
        launches.add(d)
    return ({c: v / units for c, v in total.items()}, {k: {c: v / units for c, v in d.items()} for k, d in per_kernel.items()},
            len(launches) / units)


def bench_part(tag, src, dst, meta):
    stats = newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    merged, launches, kname = {}, 0, None
    for pass_dir in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(pass_dir):
            continue
        for path in newest(glob.glob(os.path.join(pass_dir, "*", "*_counter_collection.csv"))):
            name, means, n = counter_means(path, "k_render")
            if name:
                kname, launches = name, max(launches, n)
                merged.update(means)
    if not merged:
        return
    with open(os.path.join(dst, f"{tag}_pmc_means.csv"), "w") as f:
        f.write("kernel,counter,mean_per_launch,launches\n")
        for c in sorted(merged):
            f.write(f"\"{kname}\",{c},{merged[c]:.3f},{launches}\n")
    json.dump(meta, open(os.path.join(dst, f"{tag}_meta.json"), "w"), indent=1)
    if "FETCH_SIZE" in merged and "WRITE_SIZE" in merged:
        hit, miss = merged.get("TCC_HIT_sum", 0.0), merged.get("TCC_MISS_sum", 0.0)
        out = {
            "workload": "config2", "trace_mode": "auto", "kernel": kname,
            "FETCH_SIZE_KiB_per_launch": merged["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": merged["WRITE_SIZE"],
            "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
            "hbm_bytes_per_launch": merged["FETCH_SIZE"] * 1024 * 2 + merged["WRITE_SIZE"] * 1024,
            "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate runs of "
                      "`python bench.py --no-cpu-baseline --no-extras --no-verify --steps 5 --warmup 20`, tools/profile_round.sh), mean over "
                      f"{launches} launches; FETCH_SIZE*1024*2 (gfx950 half-count correction, MI355X_MICROARCH.md, HBM) "
                      f"+ WRITE_SIZE*1024; profiles/{tag}_pmc_means.csv",
            **meta,
        }
        json.dump(out, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
        print(json.dumps(out, indent=1))
    for c in sorted(merged):
        print(f"{c:28s} {merged[c]:16.1f}")


def main():
    import bench

    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "simd-raytracer_amd"], capture_output=True, text=True).stdout.strip()
    meta = {"git_head": head + ("+dirty" if dirty else ""), "code_hash": bench.code_hash()}
    if os.path.isdir(os.path.join(src, "bench")):
        bench_part(tag, os.path.join(src, "bench"), dst, meta)
    elif os.path.isdir(os.path.join(src, "trace")):
        bench_part(tag, src, dst, meta)                      # old layout
    workloads = {}
    for wdir in sorted(glob.glob(os.path.join(src, "*"))):
        w = os.path.basename(wdir)
        if not os.path.isdir(wdir) or w in ("bench", "trace", "trace_extras") or w.startswith("pmc_"):
            continue
        log = os.path.join(wdir, "trace.log")
        units = 4
        if os.path.exists(log):
            for line in open(log):
                if line.startswith("{") and "units" in line:
                    units = json.loads(line)["units"]
        stats = newest(glob.glob(os.path.join(wdir, "trace", "*", "*_kernel_stats.csv")))
        if stats:
            shutil.copy(stats[0], os.path.join(dst, f"{tag}_{w}_kernel_stats.csv"))
        per_unit, per_kernel, launches = {}, defaultdict(dict), 0
        for pass_dir in sorted(glob.glob(os.path.join(wdir, "pmc_*"))):
            if not os.path.isdir(pass_dir):
                continue
            for path in newest(glob.glob(os.path.join(pass_dir, "*", "*_counter_collection.csv"))):
                tot, pk, n = unit_sums(path, units)
                per_unit.update(tot)
                for k, d in pk.items():
                    per_kernel[k].update(d)
                launches = max(launches, n)
        if per_unit:
            workloads[w] = {"unit": {"config3": "one frame", "config4": "one pass of 16 samples", "config5": "one pass of 8 samples"}.get(w, "one launch of 2^24 rays"),
                            "units_profiled": units, "launches_per_unit": launches, "per_unit": per_unit, "per_kernel": per_kernel}
            c = per_unit
            print(f"{w:32s} launches/unit {launches:8.1f}  VALU {c.get('SQ_INSTS_VALU', 0) / 1e6:9.1f} M  wait share "
                  f"{c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f}  HBM "
                  f"{(c.get('FETCH_SIZE', 0) * 2048 + c.get('WRITE_SIZE', 0) * 1024) / 1e6:9.1f} MB")
    if workloads:
        json.dump({**meta, "how": "tools/profile_round.sh: rocprofv3 --pmc passes (separate runs) of tools/profile_workloads.py; counter sums over "
                                  "every kernel launch after the warm-up marker divided by the units profiled; FETCH_SIZE / WRITE_SIZE in KiB "
                                  "(HBM bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024 on gfx950)",
                   "workloads": workloads}, open(os.path.join(dst, f"{tag}_workloads.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
