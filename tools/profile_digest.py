"""Digest gpurun_out/<tag>/ (tools/profile_round.sh) into the committed summaries:
   profiles/<tag>_kernel_stats.txt  per-kernel durations (rocprofv3 --kernel-trace)
   profiles/<tag>_pmc.txt           per-kernel FETCH_SIZE / WRITE_SIZE / SQ counters (separate --pmc passes)
   profiles/traffic.json            per-stage HBM bytes per view, corrected as MI355X_MICROARCH.md prescribes
                                    (FETCH_SIZE x2 on gfx950 for wide coalesced reads; both in KiB)."""
import glob, json, os, sqlite3, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE_OF = {"k_preprocess_bwd": "preprocess_bwd", "k_preprocess": "preprocess", "k_render_fwd": "render_fwd",
            "k_render_bwd": "render_bwd", "k_radix": "sort", "k_row_hist": "sort", "k_row_scatter": "sort", "k_emit_pairs": "duplicate",
            "k_emit_cols": "duplicate", "k_tile_ranges": "ranges", "k_sorted_block_sums": "scan", "k_scan_blocks": "scan",
            "k_col_hist": "scan", "k_col_plan": "scan", "k_work_order_fwd": "render_fwd", "k_work_order_bwd": "render_bwd"}


def stage(name):
    for k, v in STAGE_OF.items():
        if k in name:
            return v
    return None


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
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    bench = {}
    try:
        bench = json.loads(open(os.path.join(src, "bench_line.json")).read())
    except Exception:
        pass
    n_views = (bench.get("steps", 20) + bench.get("warmup", 5)) if bench else 25
    # ---- kernel stats
    db = glob.glob(os.path.join(src, "trace", "**", "*results.db"), recursive=True)[0]
    agg = defaultdict(list)
    for n, s, e in kernel_rows(db):
        agg[n].append(e - s)
    tot = sum(sum(v) for v in agg.values())
    lines = [f"# rocprofv3 --kernel-trace --stats of: bench.py (see bench line below); durations in us",
             f"# bench: {json.dumps(bench)[:1500]}",
             f"{'kernel':86s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'%':>6s}"]
    per_stage = defaultdict(float)
    # per VIEW: the bench line says how many views the run rendered (a batched launch covers several views)
    vps = (bench.get("config", {}) or {}).get("views_per_step_per_gpu", 1) if bench else 1
    # (counted from the trace itself: one k_render_bwd launch per step, whatever warm-up / stage passes bench.py ran)
    n_bwd_launches = sum(len(v) for n, v in agg.items() if "k_render_bwd" in n)
    n_fwd = max(1, n_bwd_launches * vps)
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{n[:86]:86s} {len(v):6d} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:9.2f} {max(v)/1e3:9.2f} "
                     f"{sum(v)/1e6:9.3f} {100*sum(v)/tot:6.2f}")
        st = stage(n)
        if st:
            per_stage[st] += sum(v) / n_fwd / 1e3
    lines.append("")
    lines.append("# per-stage GPU time per view (us), summed over the kernels of the stage: " +
                 json.dumps({k: round(v, 2) for k, v in per_stage.items()}))
    open(os.path.join(out_dir, f"{tag}_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))
    # ---- PMC
    plines = ["# rocprofv3 --pmc passes (one counter set per run) of the same bench command; values are per-dispatch "
              "averages summed over XCD/SE instances", ""]
    traffic = defaultdict(lambda: defaultdict(float))
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        dbs = glob.glob(os.path.join(src, sub, "**", "*results.db"), recursive=True)
        if not dbs:
            continue
        per = pmc_rows(dbs[0])
        byk = defaultdict(lambda: defaultdict(list))
        for (kn, did), d in per.items():
            for cn, v in d.items():
                byk[kn][cn].append(v)
        plines.append(f"## {sub}")
        for kn, d in sorted(byk.items()):
            if not stage(kn):
                continue
            plines.append(kn[:120])
            n_f = n_fwd
            for cn, v in sorted(d.items()):
                plines.append(f"    {cn:24s} dispatches={len(v):5d} avg={sum(v)/len(v):16.1f}")
                if cn in ("FETCH_SIZE", "WRITE_SIZE"):
                    traffic[stage(kn)][cn] += sum(v) / n_f        # KiB per view for this stage
        plines.append("")
    open(os.path.join(out_dir, f"{tag}_pmc.txt"), "w").write("\n".join(plines) + "\n")
    tj_path = os.path.join(out_dir, "traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    cfg = bench.get("config", {})
    key = scene_key or f"object_{cfg.get('gaussians', 500000)}_{(cfg.get('resolution') or [1024, 1024])[1]}"
    tj[key] = {st: int((2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024) for st, d in traffic.items()}
    tj[key + "_raw_KiB"] = {st: {k: round(v, 1) for k, v in d.items()} for st, d in traffic.items()}
    tj["_note"] = ("bytes per view per stage = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE/WRITE_SIZE are KiB and on "
                   "gfx950 FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM); gather / "
                   "atomic patterns are uncalibrated, so treat the render stages' figures as indicative")
    json.dump(tj, open(tj_path, "w"), indent=1)
    print(json.dumps(tj[key]))


if __name__ == "__main__":
    main(*sys.argv[1:])
