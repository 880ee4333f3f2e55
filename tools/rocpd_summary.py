"""Per-kernel summary (count / total / avg / min / max) of a rocprofv3 rocpd SQLite database ->
markdown table.   python tools/rocpd_summary.py gpurun_out/prof_vit/vit_results.db > profiles/x.md"""
import sqlite3
import sys


def main(path, title=""):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, "
         f"max(d.end-d.start)/1e3, max(s.arch_vgpr_count), max(d.group_segment_size) from {kd} d join {ks} s "
         f"on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    print(f"# {title or path}\n\nrocprofv3 --kernel-trace; total kernel time {tot / 1e3:.3f} ms\n")
    print("| % | calls | total ms | avg us | min us | max us | VGPR | LDS B | kernel |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r[2] / tot * 100:.1f} | {r[1]} | {r[2] / 1e3:.3f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} | {r[6]} | {r[7]} | `{r[0][:120]}` |")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
