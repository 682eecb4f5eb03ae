#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as
MI355X_MICROARCH.md section HBM prescribes) into profiles/traffic.json, which bench.py
reads for roofline.traffic.

  tools/pmc_summary.py <workload> <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <kernel substring> [round tag]

Units and corrections (MI355X_MICROARCH.md, HBM):
  * both counters are in KiB;
  * gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming
    read -> doubled here.  Checked in the same run on kernels of known traffic: the
    calibration kernel stream_rows_kernel (same 16-byte-per-lane loads as the sweep) when it
    is in the trace, else check_rows_kernel (reads start and end once = 16 B/row);
  * WRITE_SIZE is exact for streaming stores (checked on synth_rows_kernel = 24 B/row).
"""
import csv
import glob
import json
import os
import sys


def counter_rows(d, counter):
    f = glob.glob(os.path.join(d, f"pmc_{counter}", "*counter_collection.csv"))[0]
    return [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]


def mean(rows, key):
    """the MEDIAN over the kernel's launches (the first launches of a run may be another workload of the same kernel: the
    dense-row sweep reads all the dense rows until the k-class view is built)"""
    v = sorted(float(r["Counter_Value"]) for r in rows if key in r["Kernel_Name"])
    return (v[len(v) // 2], len(v)) if v else (None, 0)


def main():
    wl, d, kern = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else "r01"
    fr, wr = counter_rows(d, "FETCH_SIZE"), counter_rows(d, "WRITE_SIZE")
    f_kib, nf = mean(fr, kern)
    w_kib, nw = mean(wr, kern)
    fetch = 2.0 * f_kib * 1024          # gfx950: FETCH_SIZE = half the streamed bytes
    write = w_kib * 1024
    out = {"kernel": kern, "launches_fetch_pass": nf, "launches_write_pass": nw,
           "FETCH_SIZE_KiB_raw": f_kib, "WRITE_SIZE_KiB_raw": w_kib,
           "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B)",
           "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write,
           "hbm_bytes_per_launch": fetch + write}
    cal = {}
    for name, bytes_per_row in (("stream_rows_kernel", 24), ("check_rows_kernel", 16), ("pack_rows_kernel", 24)):
        v, n = mean(fr, name)
        if v:
            cal[name] = {"FETCH_SIZE_KiB_raw": v, "bytes_per_row_known": bytes_per_row}
    v, n = mean(wr, "synth_rows_kernel")
    if v:
        cal["synth_rows_kernel"] = {"WRITE_SIZE_KiB_raw": v, "bytes_per_row_known": 24}
    out["calibration_kernels_same_run"] = cal
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    allw = json.load(open(path)) if os.path.exists(path) else {}
    out["round"] = tag
    out["result_bytes_per_position"] = int(os.environ.get("RESULT_BYTES", "1"))   # bench default: uint8 results
    # what bench.py prices this launch at: WITHOUT it an entry would match any run of the same workload name (round 4: the
    # k = 31 packed view's counters ended up under an all-rows k = 101 line) -- no entry is written without it
    if not os.environ.get("ALG_BYTES"):
        sys.exit("pmc_summary.py: set ALG_BYTES to roofline.algorithmic_bytes of the bench line printed inside the counter pass")
    out["algorithmic_bytes"] = float(os.environ["ALG_BYTES"])
    if out["hbm_bytes_per_launch"] < 0.98 * out["algorithmic_bytes"]:
        sys.exit("pmc_summary.py: %.4g B measured under %.4g B algorithmic: these counters are not of that launch"
                 % (out["hbm_bytes_per_launch"], out["algorithmic_bytes"]))
    allw[wl] = out
    json.dump(allw, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
