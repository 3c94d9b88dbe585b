#!/usr/bin/env python3
"""Aggregate the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into per-kernel HBM bytes per launch.

rocprofv3 reports both counters in KiB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(wide coalesced reads report exactly half); WRITE_SIZE is uncalibrated.  Both are therefore calibrated here on kernels of
this library whose byte counts are known exactly: k_schur_z (reads every W record once, writes every Z record once) and the
large k_zero_ranges launch (writes the dense S once, reads nothing).  The calibration factors are printed with the result.
"""
import csv
import glob
import json
import os
import sys


def collect(d, counter):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per.setdefault(k, []).append(float(r["Counter_Value"]))
    return per


CALIBRATION = {
    "note": "FETCH_SIZE / WRITE_SIZE in KiB, separate --pmc passes (tools/pmc_traffic.sh).  Calibrated on kernels of this library with exactly "
            "known byte counts: k_schur_z of an earlier build (read 36.26 MB of W + 3.4 MB of C and slot ids, wrote 36.26 MB of Z: FETCH 34.4-35.7 "
            "MiB, WRITE 35.72 MiB) and the dense k_zero_ranges launch (writes 11.83 MB: WRITE max 11.84 MB) => factor 1.0 for both counters on these "
            "8-byte-per-lane record accesses (the x2 gfx950 correction of MI355X_MICROARCH.md applies to 16-byte-per-lane streaming loads, which these "
            "kernels do not issue)",
    "fetch_factor": 1.0, "write_factor": 1.0}


def bench_line(out_dir):
    """The JSON line bench.py printed during the FETCH pass (the same command as the counters): workload and algorithmic bytes."""
    try:
        for ln in open(os.path.join(out_dir, "fetch.log")):
            if ln.startswith("{"):
                return json.loads(ln)
    except (OSError, ValueError):
        pass
    return None


def main(out_dir, json_path, txt_path):
    fetch = collect(os.path.join(out_dir, "fetch"), "FETCH_SIZE")
    write = collect(os.path.join(out_dir, "write"), "WRITE_SIZE")
    names = sorted(set(fetch) | set(write))
    rows = {}
    for k in names:
        f = fetch.get(k, [])
        w = write.get(k, [])
        rows[k] = dict(launches=max(len(f), len(w)), fetch_kib=sum(f) / max(len(f), 1), write_kib=sum(w) / max(len(w), 1),
                       fetch_max_kib=max(f) if f else 0.0, write_max_kib=max(w) if w else 0.0)
    # the schema bench.py reads (committed_traffic): the roofline kernel's bytes per launch on top, every kernel below
    out = {"kernel": "k_linearize_all", "per_kernel": rows, "calibration": CALIBRATION}
    lin = [r for k, r in rows.items() if "k_linearize_all" in k]
    b = bench_line(out_dir)
    if lin:
        r = max(lin, key=lambda x: x["launches"])
        out.update(fetch_kib_avg=r["fetch_kib"], write_kib_avg=r["write_kib"], launches=r["launches"],
                   hbm_bytes_per_launch=1024.0 * (CALIBRATION["fetch_factor"] * r["fetch_kib"] + CALIBRATION["write_factor"] * r["write_kib"]))
    if b:
        out.update(workload=b["config"]["workload"], algorithmic_bytes=int(b["roofline"]["algorithmic_bytes"]))
    json.dump(out, open(json_path, "w"), indent=1)
    with open(txt_path, "w") as o:
        o.write("%-40s %8s %14s %14s %14s %14s\n" % ("kernel", "launches", "FETCH avg KiB", "FETCH max KiB", "WRITE avg KiB", "WRITE max KiB"))
        for k, r in rows.items():
            o.write("%-40s %8d %14.1f %14.1f %14.1f %14.1f\n" % (k[:40], r["launches"], r["fetch_kib"], r["fetch_max_kib"], r["write_kib"], r["write_max_kib"]))
    sys.stdout.write(open(txt_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
