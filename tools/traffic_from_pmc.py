#!/usr/bin/env python3
"""profiles/rNN_rocprofv3_pmc_*.txt (written by tools/profile.sh: one `--pmc FETCH_SIZE` and one `--pmc WRITE_SIZE` pass of the same
bench.py command, summarised by tools/rocpd_summary.py)  ->  profiles/rNN_traffic.json, the file bench.py's `roofline.traffic`
is looked up in.  Nothing is typed by hand: every number of the JSON follows from the PMC files.

    bytes per dispatch = FETCH_SIZE [KiB] * 1024 * 2  +  WRITE_SIZE [KiB] * 1024
        (FETCH_SIZE doubled: gfx950's rocprofv3 tallies the 128-byte requests of wide coalesced reads at 64 bytes,
         /opt/skills/guides/MI355X_MICROARCH.md, HBM section; WRITE_SIZE as reported)
    bytes per step     = sum over the library's kernels of bytes per dispatch * dispatches per step

usage: python tools/traffic_from_pmc.py r03
"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_sections(path):
    """[(bench args, counter, {kernel: (calls, avg_us)}, {kernel: pmc avg})] for every `# rocprofv3 ...` section of a summary file."""
    out, cur = [], None
    for line in open(path):
        line = line.rstrip("\n")
        m = re.match(r"# rocprofv3 (.*) -- python bench.py (.*)", line)
        if m:
            ctr = re.search(r"--pmc (\w+)", m.group(1))
            cur = {"args": m.group(2).split(), "counter": ctr.group(1) if ctr else None, "kern": {}, "pmc": {}}
            out.append(cur)
            continue
        if cur is None or not line.strip() or line.startswith("kernel ") or line.startswith("PMC"):
            continue
        m = re.match(r"(.+?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) avg=([0-9.e+]+)", line)
        if m:
            cur["pmc"][m.group(1).strip()] = float(m.group(4))
            continue
        m = re.match(r"(.+?)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+[0-9.]+\s+[0-9.]+\s+[0-9.]+\s+\d+", line)
        if m:
            cur["kern"][m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    return out


def bench_key(args):
    def val(flag, dflt):
        return args[args.index(flag) + 1] if flag in args else dflt
    wl = val("--workload", "groupby")
    n, g, s = float(val("--rows", "1e9")), float(val("--groups", "1e8")), float(val("--selectivity", "0.5"))
    key = f"{wl}_N{n:.0e}_G{g:.0e}_s{s}"
    if "--shape" in args:
        key += "_" + val("--shape", "hot")
    if wl == "topk" and "--limit" in args:
        key += "_limit" + val("--limit", "10")
    return key + ("_hint" if "--hint" in args else ""), int(val("--steps", "10")) + int(val("--warmup", "3"))


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    result = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{rnd}_rocprofv3_pmc_*.txt"))):
        secs = parse_sections(path)
        by_key = {}
        for sec in secs:
            if sec["counter"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            key, calls_total = bench_key(sec["args"])
            by_key.setdefault(key, {"calls_total": calls_total})[sec["counter"]] = sec
        for key, d in by_key.items():
            if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
                continue
            detail, total, kernel_us = {}, 0.0, {}
            for kern, fetch_kib in d["FETCH_SIZE"]["pmc"].items():
                if "vnm::" not in kern or kern not in d["WRITE_SIZE"]["pmc"]:
                    continue
                calls, avg_us = d["FETCH_SIZE"]["kern"].get(kern, (d["calls_total"], 0.0))
                per_step = calls / d["calls_total"]
                fb, wb = fetch_kib * 1024 * 2, d["WRITE_SIZE"]["pmc"][kern] * 1024
                if (fb + wb) * per_step < 1e6:
                    continue      # sampling / bookkeeping kernels: below a megabyte per step
                name = kern.replace("void vnm::", "").replace("vnm::", "")
                detail[name] = {"fetch": round(fb), "write": round(wb), "dispatches_per_step": per_step, "avg_us": avg_us}
                total += (fb + wb) * per_step
            result[key] = {"bytes_per_step": round(total), "detail": detail, "source": os.path.relpath(path, ROOT),
                           "how": "tools/traffic_from_pmc.py: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KiB per dispatch), "
                                  "FETCH_SIZE doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md HBM "
                                  "section), WRITE_SIZE as reported; summed over the library's kernels x dispatches per step"}
    out = os.path.join(ROOT, "profiles", f"{rnd}_traffic.json")
    with open(out, "w") as f:
        json.dump(result, f, indent=1, sort_keys=True)
    for k, v in result.items():
        print(f"{k}: {v['bytes_per_step'] / 1e9:.2f} GB per step  ({v['source']})")


if __name__ == "__main__":
    main()
