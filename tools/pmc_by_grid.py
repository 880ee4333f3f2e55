"""Mean per-launch value of every counter in a rocprofv3 --pmc CSV, grouped by (kernel name, grid size): one row per GEMM
shape when the same kernel instance serves several.  usage: pmc_by_grid.py <pmc_dir> [name-filter ...]"""
import collections, csv, glob, json, re, sys
d = sys.argv[1]
filt = sys.argv[2:] or ["gemm"]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"\(.*", "", k)[:70]
    if not any(s in k for s in filt):
        continue
    key = f"{k} grid={r.get('Grid_Size', '?')}"
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "End_Timestamp" in r and "Start_Timestamp" in r:
        dur[key][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {}
for k, cs in sorted(acc.items()):
    out[k] = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
    out[k]["launches"] = len(next(iter(cs.values())))
    if dur[k]:
        out[k]["avg_us_under_pmc"] = round(sum(dur[k].values()) / len(dur[k]) / 1e3, 2)
print(json.dumps(out, indent=1))
