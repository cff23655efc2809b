#!/usr/bin/env python3
"""Condense rocprofv3 output dirs (gpurun_out/...) into a small tracked summary under profiles/.

usage: summarize_rocprof.py TAG STATS_DIR [PMC_FETCH_DIR] [PMC_WRITE_DIR] [BENCH_JSON] [SRC_HASH_FILE CONFIG [PMC_SQ_DIR]]
Writes profiles/TAG_kernel_stats.csv (verbatim rocprofv3 --stats table) and profiles/TAG_summary.md; with SRC_HASH_FILE
(bench.py --print-src-hash of the build that was profiled) also profiles/<round>_traffic.json, which bench.py reads for
`roofline.traffic` -- and ignores when the kernel sources have changed since.
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reads 1/2 of the bytes of a wide coalesced stream, so the read side is reported both
raw and doubled (k_raster's reads are scalar record loads and 4-byte texel gathers -- uncalibrated widths --
so the truth lies between the two); WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def pmc_means(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    fetch_dir = sys.argv[3] if len(sys.argv) > 3 else None
    write_dir = sys.argv[4] if len(sys.argv) > 4 else None
    bench = sys.argv[5] if len(sys.argv) > 5 else None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    ks = glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(ks, os.path.join(prof, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(ks)))
    fetch = pmc_means(fetch_dir, "FETCH_SIZE") if fetch_dir else {}
    write = pmc_means(write_dir, "WRITE_SIZE") if write_dir else {}
    lines = [f"# rocprofv3 summary `{tag}`", "",
             "`rocprofv3 --kernel-trace --stats -- python3 bench.py ...` (kernel time) and separate `--pmc FETCH_SIZE` / "
             "`--pmc WRITE_SIZE` passes (HBM-side traffic per launch, KiB -> MB).", "",
             "| kernel | calls | avg us | % | FETCH MB raw (x2) | WRITE MB |", "|---|---|---|---|---|---|"]
    for r in rows:
        name = r["Name"].split("(")[0]
        f = fetch.get(name); w = write.get(name)
        lines.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} | "
                     f"{'' if f is None else f'{f * 1024 / 1e6:.1f} ({2 * f * 1024 / 1e6:.1f})'} | "
                     f"{'' if w is None else f'{w * 1024 / 1e6:.1f}'} |")
    # Frames in flight: a raster launch either shares the chip with the next frame's front end (the pipelined bursts of the command:
    # warm-up, timed region) or has it to itself (bench.py's one-stream passes behind the timed region, and the synced setup frames).
    # The aggregate above mixes the two; split by whether any front-end kernel's interval intersects the launch's.
    kt = glob.glob(os.path.join(stats_dir, "**", "*_kernel_trace.csv"), recursive=True)
    if kt:
        tr = [(r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kt[0]))]
        front = sorted((s0, e0) for n, s0, e0 in tr if "swr::" in n and "k_raster_c" not in n and "k_reduce" not in n and "k_clear" not in n
                       and "k_flatten" not in n)
        import bisect
        starts = [f[0] for f in front]
        beside, alone = [], []
        for n, s0, e0 in tr:
            if "k_raster_c" not in n:
                continue
            i = bisect.bisect_left(starts, e0)
            hit = any(front[k][1] > s0 and front[k][0] < e0 for k in range(max(0, i - 12), i))
            (beside if hit else alone).append((e0 - s0) / 1e3)
        if beside or alone:
            med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
            lines += ["", "`k_raster_c` launches of this command, split by the kernel trace (does a front-end kernel of the NEXT frame run during "
                      "the launch?): "
                      f"**beside a front end** {len(beside)} launches, average {sum(beside) / max(len(beside), 1):.1f} us, median {med(beside):.1f} "
                      "(= `roofline.timed_region`, frames in flight); "
                      f"**alone** {len(alone)} launches, average {sum(alone) / max(len(alone), 1):.1f} us, median {med(alone):.1f} "
                      "(= `roofline.kernel_ms`, the kernel with the chip to itself: bench.py's one-stream pass behind the timed region)."]
    if bench and os.path.exists(bench):
        txt = open(bench).read().strip().splitlines()
        js = [l for l in txt if l.startswith("{")]
        if js:
            line = json.loads(js[-1])
            rk = [n for n in fetch if "k_raster_c" in n]
            if rk and write and "roofline" in line and line["roofline"].get("traffic") is None:
                # the line was printed before this script wrote the traffic file of the build: fill in what the PMC passes of the
                # SAME call measured, so that the tracked summary never shows a stale or empty traffic field
                f, w = fetch[rk[0]], write[rk[0]]
                line["roofline"]["traffic"] = round((2.0 * f + w) * 1024 / 1e9, 4)
                line["roofline"]["traffic_detail"] = {"file": "this profile round's PMC passes", "fetch_raw_gb": round(f * 1024 / 1e9, 4),
                                                      "write_gb": round(w * 1024 / 1e9, 4), "raw_sum_gb": round((f + w) * 1024 / 1e9, 4)}
            lines += ["", "bench.py line of the same command:", "", "```json", json.dumps(line, indent=1), "```"]
    open(os.path.join(prof, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    if len(sys.argv) > 7 and fetch and write:
        src_hash = open(sys.argv[6]).read().strip()
        config = sys.argv[7]
        k = [n for n in fetch if "k_raster_c" in n][0]
        tpath = os.path.join(prof, f"{tag.split('_')[0]}_traffic.json")
        j = json.load(open(tpath)) if os.path.exists(tpath) else {}
        if j.get("csrc_sha256") != src_hash:
            j = {}
        j["_provenance"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- python3 bench.py --steps 3 "
                            "--warmup 1 --prime 4 --no-cpu-baseline on MI355X; per-launch means of k_raster_c, KiB as reported. bench.py "
                            "applies the guide's gfx950 correction (FETCH_SIZE x 2) and reports the raw figures beside it.")
        j["csrc_sha256"] = src_hash
        j[config] = {"kernel": k, "fetch_kib_raw": round(fetch[k], 1), "write_kib": round(write[k], 1)}
        if len(sys.argv) > 8 and sys.argv[8]:
            # instruction mix of the same kernel (SQ pass): what bench.py's roofline.valu_floor is computed from
            for cn, key in (("SQ_INSTS_VALU", "valu_wave_insts"), ("SQ_INSTS_SALU", "salu_wave_insts"), ("SQ_INSTS_LDS", "lds_wave_insts"),
                            ("SQ_INSTS_VMEM_RD", "vmem_rd_wave_insts"), ("SQ_WAVE_CYCLES", "wave_quad_cycles"),
                            ("SQ_WAIT_INST_ANY", "wait_any_quad_cycles"), ("SQ_ACTIVE_INST_VALU", "valu_active_quad_cycles")):
                m = pmc_means(sys.argv[8], cn)
                if k in m:
                    j[config][key] = round(m[k], 1)
        json.dump(j, open(tpath, "w"), indent=1)
    print("\n".join(lines[:20]))


if __name__ == "__main__":
    main()
