"""Mean per-launch value of every counter in a rocprofv3 --pmc CSV, grouped by (shortened) kernel name."""
import csv, glob, collections, re, sys, json
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"\(.*", "", k)[:60]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    if not any(s in k for s in ("gemm", "attn", "ln_kernel")):
        continue
    out[k] = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
    out[k]["launches"] = len(next(iter(cs.values())))
print(json.dumps(out, indent=1))
