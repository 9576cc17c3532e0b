"""Per (kernel, grid) means of SQ counters from rocprofv3 --pmc passes, as percentages of SQ_WAVE_CYCLES.
usage: pmc_sq_summary.py <label>=<dir> [...]"""
import collections, csv, glob, re, sys
for arg in sys.argv[1:]:
    label, d = arg.split("=", 1)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per, meta = collections.defaultdict(float), {}
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            meta[r["Dispatch_Id"]] = (re.sub(r"^void |\(.*$", "", r["Kernel_Name"]), r["Grid_Size"])
        for (disp, cname), v in per.items():
            agg[meta[disp]][cname].append(v)
    print(f"== {label}: mean per launch, percentages are of SQ_WAVE_CYCLES")
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
        wc = sum(cs["SQ_WAVE_CYCLES"]) / len(cs["SQ_WAVE_CYCLES"]) if "SQ_WAVE_CYCLES" in cs else 0
        n = max(len(v) for v in cs.values())
        if wc <= 0 or n < 2:
            continue
        parts = " ".join(f"{c.replace('SQ_', '')}={sum(v) / len(v):.3g}({100 * sum(v) / len(v) / wc:.1f}%)" for c, v in sorted(cs.items()) if c != "SQ_WAVE_CYCLES")
        print(f"{k} n={n}: {parts}")
