#!/usr/bin/env python3
"""Read the scaling curve out of bench.py's lines: one JSON line per GPU count (files, or stdin) -> a table with, per N, the whole-job rate, the
speed-up over the N = 1 line, where a step's time goes on rank 0 (render / exchange + assembly / host gaps) and how evenly the ranks finished.

    for n in 1 2 4 8; do python bench.py --gpus $n --steps 20 --warmup 5 > scale_n$n.json; done
    python tools/scale_report.py scale_n*.json

Everything it prints is in the lines themselves (`per_rank`, `gather_assemble_ms_rank0`, `step_ms_breakdown_rank0`, `roofline.kernel_ms_per_rank`);
the efficiency is value(N) / (N x value(1)) — total work is fixed ("strong" scaling)."""
import json
import sys


def lines(paths):
    srcs = [open(p) for p in paths] if paths else [sys.stdin]
    for f in srcs:
        for ln in f:
            ln = ln.strip()
            if ln.startswith("{"):
                yield json.loads(ln)


def main():
    rows = sorted(lines(sys.argv[1:]), key=lambda d: d["n_gpus"])
    if not rows:
        raise SystemExit("no bench.py line found")
    base = next((d for d in rows if d["n_gpus"] == 1), None)
    print("| N | ranks seen (backend) | Msamples/s | ms / step | speed-up | efficiency | render ms min..max (imbalance) | dominant kernel ms per rank | exchange + assembly ms (rank 0) | host gaps ms | assembled frame verified |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for d in rows:
        n = d["n_gpus"]
        sp = d["value"] / base["value"] if base else float("nan")
        if n == 1:
            r = d.get("roofline", {})
            print(f"| 1 | - | {d['value']:.1f} | {d['ms_per_step']:.3f} | 1.00 | 1.00 | {d.get('kernel_ms_per_step_rank0', float('nan')):.3f} | {r.get('kernel_ms', float('nan'))} | - | - | - |")
            continue
        pr, bd = d.get("per_rank", {}), d.get("step_ms_breakdown_rank0", {})
        print(f"| {n} | {d.get('ranks_seen')} ({d.get('backend')}) | {d['value']:.1f} | {d['ms_per_step']:.3f} | {sp:.2f} | {sp / n:.2f} | "
              f"{pr.get('render_ms_min')}..{pr.get('render_ms_max')} ({pr.get('render_imbalance')}) | {d.get('roofline', {}).get('kernel_ms_per_rank')} | "
              f"{d.get('gather_assemble_ms_rank0')} | {bd.get('host_and_launch_gaps')} | {d.get('assembly_verified')} |")


if __name__ == "__main__":
    main()
