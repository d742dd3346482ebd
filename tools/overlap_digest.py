"""Do kernels of different HIP streams overlap? From a rocprofv3 --kernel-trace output directory (rocpd *results.db): per queue /
stream the busy time, the union of all kernel intervals against their sum (sum > union = concurrency), and an excerpt of
consecutive launches with their queue, start and end relative to the excerpt's first kernel.
usage: python tools/overlap_digest.py <dir> [skip_fraction] [excerpt_rows]"""
import glob
import os
import sqlite3
import sys


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


def main(src, skip=0.5, rows=64):
    dbs = glob.glob(os.path.join(src, "**", "*results.db"), recursive=True)
    if not dbs:
        print("no results.db under", src)
        return
    c = sqlite3.connect(dbs[0])
    cols = [r[1] for r in c.execute("PRAGMA table_info(kernels)")]
    qcol = next((x for x in ("stream_id", "queue_id", "stream", "queue") if x in cols), None)
    print("# columns of `kernels`:", cols, "-> stream column:", qcol)
    q = f"select name, start, end, {qcol} from kernels order by start" if qcol else "select name, start, end, 0 from kernels order by start"
    ks = [(short(n), s, e, qq) for n, s, e, qq in c.execute(q) if "k_" in n]
    ks = ks[int(len(ks) * float(skip)):]          # steady state: drop the warm-up part of the trace
    if not ks:
        print("no kernels")
        return
    t0, t1 = ks[0][1], max(k[2] for k in ks)
    tot = sum(e - s for _, s, e, _ in ks)
    # union of the intervals
    union, cur_s, cur_e = 0, None, None
    for _, s, e, _ in sorted(ks, key=lambda k: k[1]):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    per_q = {}
    for _, s, e, qq in ks:
        per_q[qq] = per_q.get(qq, 0) + (e - s)
    print(f"# kernels {len(ks)}; span {(t1 - t0) / 1e3:.1f} us; sum of kernel durations {tot / 1e3:.1f} us; union (GPU busy with >= 1 "
          f"kernel) {union / 1e3:.1f} us; sum / union = {tot / union:.3f} (1.0 = nothing ever overlaps); busy share of the span "
          f"{union / (t1 - t0):.3f}")
    print("# busy time per stream / queue (us):", {str(k): round(v / 1e3, 1) for k, v in sorted(per_q.items(), key=lambda kv: -kv[1])})
    print(f"# excerpt: {rows} consecutive launches (by start time); times in us relative to the first")
    b = ks[len(ks) // 3:len(ks) // 3 + int(rows)]
    z = b[0][1]
    print(f"{'stream':>8s} {'start':>9s} {'end':>9s} {'dur':>8s}  kernel")
    for n, s, e, qq in b:
        print(f"{str(qq):>8s} {(s - z) / 1e3:9.1f} {(e - z) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n[:60]}")


if __name__ == "__main__":
    main(*sys.argv[1:])
