"""Digest gpurun_out/<tag>/ (tools/profile_round.sh) into the committed summaries:
   profiles/<tag>_kernel_stats.txt  per-kernel durations (rocprofv3 --kernel-trace)
   profiles/<tag>_pmc.txt           per-kernel FETCH_SIZE / WRITE_SIZE / SQ counters (separate --pmc passes)
   profiles/traffic.json            HBM bytes per LAUNCH of every kernel and per STEP of every stage, corrected as
                                    MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950 for wide coalesced reads;
                                    both counters are KiB).
Every pass (trace, pmc_fetch, pmc_write, pmc_sq) is a separate run of bench.py, and rocprofv3 serialises kernels in the
PMC passes, so the runs do not dispatch the same number of kernels: every pass is normalised by ITS OWN dispatch
counts (per launch = median over that pass's dispatches of the kernel; per step = the pass's sum over the stage's
kernels / the pass's number of k_render_bwd launches, one per step). Profile batched configurations with
`bench.py --no-dropin` so that the per-kernel statistics are those of the batched launches."""
import glob, json, os, sqlite3, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE_OF = {"k_preprocess_bwd": "preprocess_bwd", "k_preprocess": "preprocess", "k_render_fwd": "render_fwd",
            "k_render_bwd": "render_bwd", "k_radix": "sort", "k_os_": "sort", "k_gsr_zero_words": "sort",
            "k_row_hist": "sort", "k_row_scatter": "sort",
            "k_depth": "sort", "k_emit_pairs": "duplicate",
            "k_emit_cols": "duplicate", "k_tile_ranges": "ranges", "k_sorted_block_sums": "scan", "k_scan_blocks": "scan",
            "k_col_hist": "scan", "k_col_plan": "scan", "k_work_order_fwd": "render_fwd", "k_work_order_bwd": "render_bwd"}
# kernels whose per-launch traffic bench.py's `roofline.traffic` looks up (stage -> the stage's dominant kernel)
MAIN_KERNEL = {"preprocess": "k_preprocess", "preprocess_bwd": "k_preprocess_bwd", "render_fwd": "k_render_fwd",
               "render_bwd": "k_render_bwd", "duplicate": "k_emit", "sort": None, "scan": None, "ranges": None}


def stage(name):
    for k, v in STAGE_OF.items():
        if k in name:
            return v
    return None


def short(name):
    """`void (anonymous namespace)::k_preprocess_views<16>(GsrView, ...)` -> `k_preprocess_views<16>`"""
    n = name.split("(anonymous namespace)::")[-1] if "k_" in name.split("(anonymous namespace)::")[-1] else name
    n = n[n.index("k_"):] if "k_" in n else n
    depth, out = 0, []
    for ch in n:
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).strip()


def kernel_rows(db):
    c = sqlite3.connect(db)
    return list(c.execute("select name, start, end from kernels"))


def pmc_rows(db):
    c = sqlite3.connect(db)
    per = defaultdict(lambda: defaultdict(float))
    for kn, cn, val, did in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        per[(kn, did)][cn] += val
    return per


def main(tag, scene_key=None):
    src = os.path.join(ROOT, "gpurun_out", tag)
    # GSR_PROFILE_OUT: digest on the GPU box itself (the raw rocprofv3 databases exceed what gpurun copies back) into a
    # small directory under gpurun_out/; its files are then copied into profiles/ here
    out_dir = os.environ.get("GSR_PROFILE_OUT") or os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    bench = {}
    try:
        bench = json.loads(open(os.path.join(src, "bench_line.json")).read())
    except Exception:
        pass
    # ---- kernel stats (the trace pass)
    db = glob.glob(os.path.join(src, "trace", "**", "*results.db"), recursive=True)[0]
    agg = defaultdict(list)
    for n, s, e in kernel_rows(db):
        agg[n].append(e - s)
    tot = sum(sum(v) for v in agg.values())
    lines = [f"# rocprofv3 --kernel-trace --stats of: bench.py (see bench line below); durations in us",
             f"# bench: {json.dumps(bench)[:1500]}",
             f"{'kernel':86s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'%':>6s}  median_us"]
    per_stage = defaultdict(float)
    vps = (bench.get("config", {}) or {}).get("views_per_step_per_gpu", 1) if bench else 1
    # steps of THIS pass, counted from the trace itself: one k_render_bwd launch per step in the batched path, one per
    # view otherwise (the bench line says which)
    batched = bool((bench.get("config", {}) or {}).get("batched_call", vps > 1))
    n_bwd_launches = sum(len(v) for n, v in agg.items() if "k_render_bwd" in n)
    n_views_trace = max(1, n_bwd_launches * (vps if batched else 1))
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{n[:86]:86s} {len(v):6d} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:9.2f} {max(v)/1e3:9.2f} "
                     f"{sum(v)/1e6:9.3f} {100*sum(v)/tot:6.2f}  med {sorted(v)[len(v)//2]/1e3:8.2f}")
        st = stage(n)
        if st:
            per_stage[st] += sum(v) / n_views_trace / 1e3
    lines.append("")
    lines.append("# per-stage GPU time per view (us), summed over the kernels of the stage: " +
                 json.dumps({k: round(v, 2) for k, v in per_stage.items()}))
    open(os.path.join(out_dir, f"{tag}_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))
    # ---- PMC (each pass normalised by its own dispatch counts)
    plines = ["# rocprofv3 --pmc passes (one counter set per run) of the same bench command; values are per-dispatch "
              "averages (mean over the dispatches OF THAT PASS) summed over XCD/SE instances", ""]
    launch = defaultdict(lambda: defaultdict(float))     # kernel short name -> counter -> mean per launch
    step = defaultdict(lambda: defaultdict(float))       # stage -> counter -> per step
    sq = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        dbs = glob.glob(os.path.join(src, sub, "**", "*results.db"), recursive=True)
        if not dbs:
            continue
        per = pmc_rows(dbs[0])
        byk = defaultdict(lambda: defaultdict(list))
        for (kn, did), d in per.items():
            for cn, v in d.items():
                byk[kn][cn].append(v)
        steps_pass = max(1, sum(len(next(iter(d.values()))) for kn, d in byk.items() if "k_render_bwd" in kn))
        plines.append(f"## {sub}   (steps in this pass: {steps_pass})")
        for kn, d in sorted(byk.items()):
            if not stage(kn):
                continue
            plines.append(kn[:120])
            for cn, v in sorted(d.items()):
                plines.append(f"    {cn:24s} dispatches={len(v):5d} avg={sum(v)/len(v):16.1f} median={sorted(v)[len(v)//2]:16.1f}")
                # per launch: the MEDIAN over the pass's dispatches (the warm-up renders a few views one at a time before the
                # batched launches start; the median is the batched launch's value, the mean would be diluted)
                med = sorted(v)[len(v) // 2]
                if cn in ("FETCH_SIZE", "WRITE_SIZE"):
                    launch[short(kn)][cn] = med
                    step[stage(kn)][cn] += sum(v) / steps_pass
                elif sub == "pmc_sq":
                    sq.setdefault(short(kn), {})[cn] = med
        plines.append("")
    open(os.path.join(out_dir, f"{tag}_pmc.txt"), "w").write("\n".join(plines) + "\n")
    tj_path = os.path.join(out_dir, "traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    cfg = bench.get("config", {})
    wl = str(cfg.get("workload", ""))
    scene = "indoor" if "indoor" in wl else ("object-init" if "init" in wl else "object")
    key = scene_key or (f"{scene}_{cfg.get('gaussians', 500000)}_{(cfg.get('resolution') or [1024, 1024])[1]}" +
                        ("" if batched else "_dropin"))
    tobytes = lambda d: int((2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024)
    tj = {k: v for k, v in tj.items() if k != key and not k.startswith(key + "_raw")}
    tj[key] = {
        "tag": tag, "views_per_step": vps, "batched_call": batched,
        "per_launch_bytes": {k: tobytes(d) for k, d in sorted(launch.items())},
        "per_step_bytes": {st: tobytes(d) for st, d in sorted(step.items())},
        "per_launch_raw_KiB": {k: {c: round(x, 1) for c, x in d.items()} for k, d in sorted(launch.items())},
        "sq_per_launch": {k: {c: round(x, 1) for c, x in d.items()} for k, d in sorted(sq.items())
                          if any(s in k for s in ("k_render", "k_preprocess"))},
    }
    tj["_note"] = ("bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE/WRITE_SIZE are KiB and on gfx950 FETCH_SIZE "
                   "reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM); gather / atomic patterns are "
                   "uncalibrated, so treat the render stages' figures as indicative. per_launch = mean over the dispatches "
                   "of the kernel in its own PMC pass; per_step = that pass's sum over the stage's kernels / that pass's "
                   "number of steps")
    json.dump(tj, open(tj_path, "w"), indent=1)
    print(json.dumps(tj[key]["per_launch_bytes"]))
    print(json.dumps(tj[key]["per_step_bytes"]))


if __name__ == "__main__":
    main(*sys.argv[1:])
