"""Summaries of rocprofv3 output for profiles/ (run on the GPU box, after the rocprofv3 commands in DESIGN.md section 6).

  python tests/prof_summary.py stats <results.db> [...]          per-kernel time table (rocprofv3 --kernel-trace --stats, rocpd db)
  python tests/prof_summary.py pmc <windows_per_launch> <out.json> <counter_collection.csv> [...]
        average FETCH_SIZE / WRITE_SIZE per launch of every kernel -> HBM traffic in bytes per launch
        (MI355X_MICROARCH.md, "HBM": the counters are in KiB; FETCH_SIZE is doubled on gfx950), merged into out.json as
        {kernel: {windows_per_launch: {"fetch_bytes":..,"write_bytes":..,"traffic_bytes":..,"launches":..}}}
"""
import csv
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha256   # the kernel sources these counters belong to: bench.py prints no traffic figure for other sources


def short(name):
    m = re.search(r"(k_[a-z_A-Z0-9]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:48]


def stats(paths):
    for name in paths:
        con = sqlite3.connect(name)
        cur = con.cursor()
        rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                           "from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        print(f"# {os.path.basename(name)}: total kernel time {tot / 1e3:.2f} ms")
        print(f"{'kernel':58s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'min us':>9s} {'max us':>9s} {'%':>6s}")
        for r in rows:
            print(f"{short(r[0]):58s} {r[1]:6d} {r[2] / 1e3:10.3f} {r[3]:10.2f} {r[4]:9.2f} {r[5]:9.2f} {100 * r[2] / tot:6.2f}")


def pmc(wpl, out, paths):
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = json.load(open(out)) if os.path.exists(out) else {}
    print(f"{'kernel':58s} {'launches':>8s} {'fetch MB':>10s} {'write MB':>10s}   (per launch; FETCH_SIZE x2 x1024, WRITE_SIZE x1024)")
    for k, c in sorted(acc.items()):
        f = c.get("FETCH_SIZE", [])
        w = c.get("WRITE_SIZE", [])
        fb = 2.0 * 1024.0 * sum(f) / len(f) if f else None
        wb = 1024.0 * sum(w) / len(w) if w else None
        e = res.setdefault(k.split("<")[0] if not k.startswith("k_vis_eval") else k, {}).setdefault(str(wpl), {})
        if fb is not None: e["fetch_bytes"] = fb; e["launches"] = len(f)
        if wb is not None: e["write_bytes"] = wb; e["launches"] = len(w)
        if "fetch_bytes" in e and "write_bytes" in e: e["traffic_bytes"] = e["fetch_bytes"] + e["write_bytes"]
        print(f"{k:58s} {max(len(f), len(w)):8d} {(fb or 0) / 1e6:10.3f} {(wb or 0) / 1e6:10.3f}")
    res["_csrc_sha256"] = csrc_sha256()
    res["_source"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) of tools/profile_round6.sh, {wpl} windows per launch, single stream; "
                      "FETCH_SIZE doubled, KiB -> bytes (MI355X_MICROARCH.md); kernel sources: _csrc_sha256")
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


def counters(out, paths):
    """Average per launch of every counter found, per kernel -> table + JSON (MFMA busy cycles, ops, wave cycles ...)."""
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    names = sorted({c for k in acc.values() for c in k})
    res = {}
    print(f"{'kernel':52s} " + " ".join(f"{n[-26:]:>26s}" for n in names))
    for k, c in sorted(acc.items()):
        e = {n: (sum(v) / len(v)) for n, v in c.items()}
        e["launches"] = max(len(v) for v in c.values())
        base = k.split("<")[0]          # keyed without template arguments (bench.py looks kernels up by their plain name)
        if base not in res or res[base]["launches"] < e["launches"]:
            res[base] = e
        print(f"{k[:52]:52s} " + " ".join(f"{e.get(n, float('nan')):26.4g}" for n in names))
    res["_csrc_sha256"] = csrc_sha256()
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        counters(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2:])
    elif sys.argv[1] == "pmc":
        pmc(int(sys.argv[2]), sys.argv[3], sys.argv[4:])
    else:
        raise SystemExit(__doc__)
