"""Per-kernel duration table from a rocprofv3 --kernel-trace output directory (rocpd *results.db). usage: kstats.py <dir>
Prints calls / avg / median / total per kernel (us) and the total per step (one k_render_bwd launch = one step)."""
import glob, os, sqlite3, sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    depth, out = 0, []
    for ch in n:
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).strip()


def main(src):
    dbs = glob.glob(os.path.join(src, "**", "*results.db"), recursive=True)
    if not dbs:
        print("no results.db under", src)
        return
    c = sqlite3.connect(dbs[0])
    agg = defaultdict(list)
    for n, s, e in c.execute("select name, start, end from kernels"):
        agg[short(n)].append((e - s) / 1e3)
    steps = max(1, sum(len(v) for k, v in agg.items() if "k_render_bwd" in k))
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':64s} {'calls':>6s} {'/step':>6s} {'avg_us':>9s} {'med_us':>9s} {'us/step':>9s} {'%':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[:64]:64s} {len(v):6d} {len(v)/steps:6.1f} {sum(v)/len(v):9.2f} {sorted(v)[len(v)//2]:9.2f} "
              f"{sum(v)/steps:9.2f} {100*sum(v)/tot:6.2f}")
    print(f"# steps (k_render_bwd launches): {steps}; GPU time per step: {tot/steps:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
