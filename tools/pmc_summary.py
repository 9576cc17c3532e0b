"""Summarise rocprofv3 --pmc counter_collection CSVs per (kernel, grid size): mean counter value per launch.
usage: pmc_summary.py <dir> [<dir> ...]  ->  JSON on stdout"""
import collections
import csv
import glob
import json
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(float)
        meta = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])      # summed over XCDs / instances
            meta[r["Dispatch_Id"]] = (re.sub(r"^void |\(.*$", "", r["Kernel_Name"]), r["Grid_Size"])
        for (disp, cname), v in per_dispatch.items():
            agg[meta[disp]][cname].append(v)
out = {}
for k, cs in sorted(agg.items(), key=lambda kv: str(kv[0])):
    out[str(k)] = {"launches": max(len(v) for v in cs.values()), **{c + "_per_launch": sum(v) / len(v) for c, v in cs.items()}}
print(json.dumps(out, indent=1))
