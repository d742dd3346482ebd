"""Condense a rocprofv3 --kernel-trace run (rocpd sqlite .db or *_kernel_trace.csv) into a per-kernel table."""
import csv, glob, os, sqlite3, sys
from collections import defaultdict

def rows_from(path):
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        for name, start, end in c.execute("select name, start, end from kernels"):
            yield name, (end - start)
    else:
        with open(path) as f:
            for r in csv.DictReader(f):
                yield r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])

def main(src, out=None):
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*results.db"), recursive=True) + \
        glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    agg = defaultdict(list)
    for f in files:
        for n, d in rows_from(f):
            agg[n].append(d)
    tot = sum(sum(v) for v in agg.values())
    lines = [f"{'kernel':80s} {'calls':>6s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'%':>6s}"]
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{n[:80]:80s} {len(v):6d} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:9.2f} {max(v)/1e3:9.2f} "
                     f"{sum(v)/1e6:9.3f} {100*sum(v)/tot:6.2f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")

if __name__ == "__main__":
    main(*sys.argv[1:])
