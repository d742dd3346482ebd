"""Summarise rocprofv3 --pmc output (rocpd .db): per kernel name, average of each counter per dispatch."""
import glob, os, sqlite3, sys
from collections import defaultdict

def main(src, out=None, flt=None):
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*results.db"), recursive=True)
    agg = defaultdict(lambda: defaultdict(list))
    for f in files:
        c = sqlite3.connect(f)
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        q = "select kernel_name, counter_name, value, dispatch_id from counters_collection" if "kernel_name" in cols else None
        if q is None:
            print("columns:", cols); return
        per = defaultdict(lambda: defaultdict(float))
        for kn, cn, val, did in c.execute(q):
            per[(kn, did)][cn] += val          # sum over dimensions (XCD/SE instances)
        for (kn, did), d in per.items():
            for cn, v in d.items():
                agg[kn][cn].append(v)
    lines = []
    for kn, d in sorted(agg.items()):
        if flt and flt not in kn:
            continue
        lines.append(kn[:110])
        for cn, v in sorted(d.items()):
            lines.append(f"    {cn:28s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")

if __name__ == "__main__":
    main(*sys.argv[1:])
