"""FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, as MI355X_MICROARCH.md prescribes) -> per-kernel HBM
bytes per launch.  Counter unit = KB; gfx950 correction: FETCH_SIZE counts 64-byte requests as 32 -> x2.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> <note>"""
import collections, csv, glob, json, re, sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*", "", k)[:80]
        acc[k].append(float(r["Counter_Value"]))
    return acc


fd, wd, out, note = sys.argv[1:5]
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
res = {"note": note, "kernels": {}}
for k in sorted(set(F) | set(W)):
    f = sum(F.get(k, [0])) / max(len(F.get(k, [0])), 1)
    w = sum(W.get(k, [0])) / max(len(W.get(k, [0])), 1)
    res["kernels"][k] = {"launches": len(F.get(k, W.get(k, []))), "FETCH_SIZE_KB_per_launch": round(f, 1),
                         "WRITE_SIZE_KB_per_launch": round(w, 1), "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
json.dump(res, open(out, "w"), indent=1)
for k, v in res["kernels"].items():
    if any(s in k for s in ("gemm", "attn", "ln_kernel", "sae", "adam", "topk", "wenc", "csr", "inv_norm")):
        print(k[:60], v)
