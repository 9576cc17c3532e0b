"""HBM traffic per launch from rocprofv3 --pmc passes, with FETCH_SIZE calibrated on a known-bytes stream.
usage: pmc_hbm.py <calib FETCH dir> <calib WRITE dir> <bench FETCH dir> <bench WRITE dir>  ->  JSON on stdout
(each dir = output of one `rocprofv3 --kernel-trace --pmc X` run; tools/fetch_calib for the first two,
bench.py for the others).  FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived metrics)."""
import collections
import csv
import glob
import json
import re
import sys

CALIB_BYTES = 96 << 20


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        meta = {}
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            meta[r["Dispatch_Id"]] = (re.sub(r"^void |\(.*$", "", r["Kernel_Name"]), int(r["Grid_Size"]))
        for (disp, cname), v in per.items():
            agg[meta[disp]][cname].append(v)
    return agg


def mean(v):
    return sum(v) / len(v)


cf, cw, bf, bw = (load(d) for d in sys.argv[1:5])
calib = {}
for (name, grid), cs in cf.items():
    if "FETCH_SIZE" in cs and ("k_read4" in name or "k_read16" in name):
        calib[name.split("::")[-1]] = {"known_bytes": CALIB_BYTES, "FETCH_SIZE_KiB": mean(cs["FETCH_SIZE"]),
                                       "factor": CALIB_BYTES / (mean(cs["FETCH_SIZE"]) * 1024.0)}
for (name, grid), cs in cw.items():
    if "WRITE_SIZE" in cs and "k_write4" in name:
        calib["k_write4"] = {"known_bytes": CALIB_BYTES, "WRITE_SIZE_KiB": mean(cs["WRITE_SIZE"]),
                             "factor": CALIB_BYTES / (mean(cs["WRITE_SIZE"]) * 1024.0)}
f4 = calib.get("k_read4", {}).get("factor", 2.0)
f16 = calib.get("k_read16", {}).get("factor", 2.0)
w4 = calib.get("k_write4", {}).get("factor", 1.0)
kern = {}
for key in sorted(set(bf) | set(bw), key=str):
    name, grid = key
    e = {"launches": 0}
    if "FETCH_SIZE" in bf.get(key, {}):
        v = bf[key]["FETCH_SIZE"]
        e["launches"] = len(v)
        e["FETCH_SIZE_KiB_per_launch"] = mean(v)
        e["read_bytes_per_launch"] = mean(v) * 1024.0 * f4
        e["read_bytes_per_launch_if_16B_factor"] = mean(v) * 1024.0 * f16
    if "WRITE_SIZE" in bw.get(key, {}):
        v = bw[key]["WRITE_SIZE"]
        e["launches"] = max(e["launches"], len(v))
        e["WRITE_SIZE_KiB_per_launch"] = mean(v)
        e["write_bytes_per_launch"] = mean(v) * 1024.0 * w4
    kern[f"{name} grid_threads={grid}"] = e
out = {"calibration": calib,
       "calibration_note": "factor = known bytes / counter bytes for a streaming pass over 96 MiB; the RAM rows are read 4 B per lane "
                           "(k_read4's pattern), the prepared keys 16 B per lane (L2 resident, a small share of the fabric reads)",
       "kernels": kern}


def dominant(kern):
    """The dominant kernel = k_read_chain (a row's products of coordinate 0 + its alone packer levels as one launch, one workgroup
    per row): the launch with the largest share of GPU time.  Its steps hand over through LDS and registers, so a launch reads its
    int32 input once, its prepared operands (one GGSW digit per product, one key per trace step) and writes its int32 output once."""
    for pat in ("k_read_chain<4, 4>", "k_read_chain<5, 4>", "k_write_chain<4, 4>"):
        cand = [(k, e) for k, e in kern.items() if pat in k and "read_bytes_per_launch" in e and "write_bytes_per_launch" in e]
        if cand:
            k, e = max(cand, key=lambda ke: ke[1]["launches"])
            blocks = int(k.split("grid_threads=")[1]) // 512
            return {"kernel": k, "launches": e["launches"], "ciphertexts_per_launch": blocks,
                    "read_bytes_per_launch": e["read_bytes_per_launch"], "write_bytes_per_launch": e["write_bytes_per_launch"],
                    "hbm_bytes_per_launch": e["read_bytes_per_launch"] + e["write_bytes_per_launch"],
                    "note": "compulsory per launch in the device layout: 98 304 B of int32 input per ciphertext (+ the same written where the "
                            "products are in place), 98 304 B of output, 1.5 MiB per GGSW digit and 0.75 MiB per trace key ONCE; FETCH_SIZE "
                            "also counts what the eight XCD L2s fetch from the Infinity Cache (each takes its own copy of the operands)"}
    return None


dk = dominant(kern)
if dk:
    out["dominant_kernel"] = dk
print(json.dumps(out, indent=1))
