"""MFMA utilisation and effective shader clock per kernel instance from ONE rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE).
Units (MI355X_MICROARCH.md, checked on round-1 passes): SQ_VALU_MFMA_BUSY_CYCLES = cycles summed over the 1024 SIMDs
(= 32 x #v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE = cycles summed over the 8 XCDs; SQ_BUSY_CYCLES = summed over 32 SEs;
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY = quad-cycles summed over waves.
    mfma_util   = MFMA_BUSY / (1024 * GRBM_GUI_ACTIVE / 8)
    clock_GHz   = GRBM_GUI_ACTIVE / 8 / kernel duration (the PMC pass's own timestamps: profiled clocks run 2-5 % low)
usage: pmc_mfma.py <pmc_dir> <out.json> <note> [name-filter ...]"""
import collections, csv, glob, json, re, sys

d, out, note = sys.argv[1:4]
filt = sys.argv[4:] or ["gemm", "attn", "sae", "adam", "ln_kernel"]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"\(.*", "", k)[:90]
    if not any(s in k for s in filt):
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
res = {"note": note, "kernels": {}}
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    us = sum(dur[k].values()) / len(dur[k]) / 1e3
    e = {"launches": len(dur[k]), "avg_us_under_pmc": round(us, 2)}
    e.update({c: round(v, 1) for c, v in m.items()})
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        e["clock_GHz"] = round(cyc / (us * 1e3), 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            e["mfma_util"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
    if "SQ_WAVE_CYCLES" in m:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in m:
                e[c + "_frac"] = round(m[c] / m["SQ_WAVE_CYCLES"], 4)
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
for k, e in res["kernels"].items():
    print(k[:70], {x: e[x] for x in ("launches", "avg_us_under_pmc", "clock_GHz", "mfma_util") if x in e})
