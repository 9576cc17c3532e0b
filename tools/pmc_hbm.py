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
    """The dominant kernel shape = one fused trace step over 256 ciphertexts.  It runs inside k_keyswitch_chain (6 or 12 steps
    per launch at 2^18) or as k_keyswitch<1,3,4,3,2,0>.  Limb-form chain: every step writes blocks * 98 304 B; Y-form chain
    (round 3: intermediates as one double per coefficient and column): blocks * 65 536 B per inner step, 98 304 B for the last —
    which gives the steps of a launch from its WRITE_SIZE."""
    for pat in ("k_keyswitch_chain<3, 4, 3", "k_keyswitch<1, 3, 4, 3, 2, 0>"):
        cand = [(k, e) for k, e in kern.items() if pat in k and "read_bytes_per_launch" in e and "write_bytes_per_launch" in e
                and int(k.split("grid_threads=")[1]) >= 65536]
        if cand:
            k, e = max(cand, key=lambda ke: ke[1]["launches"])
            blocks = int(k.split("grid_threads=")[1]) // 512
            yform = ("true" in k.split("grid_threads=")[0]) or ("(bool)1" in k)
            per_ct = e["write_bytes_per_launch"] / blocks
            steps = ((per_ct - 98304.0) / 65536.0 + 1.0) if yform else per_ct / 98304.0
            comp = (2 * 65536 if yform else 2 * 98304)
            return {"kernel": k, "launches": e["launches"], "ciphertexts_per_step": blocks, "steps_per_launch": steps,
                    "intermediate_form": "Y = ceil(A/2), 8 B per coefficient and column" if yform else "int32 limbs",
                    "read_bytes_per_step": e["read_bytes_per_launch"] / steps, "write_bytes_per_step": e["write_bytes_per_launch"] / steps,
                    "hbm_bytes_per_launch": (e["read_bytes_per_launch"] + e["write_bytes_per_launch"]) / steps,
                    "note": "hbm_bytes_per_launch is per STEP (one trace step over all ciphertexts = what bench.py's roofline calls a launch); "
                            f"compulsory for the device layout of an inner step: {comp} B per ciphertext + the 786 432 B key"}
    return None


dk = dominant(kern)
if dk:
    out["dominant_kernel"] = dk
print(json.dumps(out, indent=1))
