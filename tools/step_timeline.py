#!/usr/bin/env python
"""Timeline of ONE steady-state step from a rocprofv3 --kernel-trace CSV: for every kernel its start offset within the
step, its duration and the queue it ran on; plus the busy time per queue and the gaps on the main queue.

    python tools/step_timeline.py gpurun_out/prof_x/.../NNN_kernel_trace.csv [--step -2]
A step is delimited by consecutive launches of the kernel named by --marker (default: preprocess_fwd_kernel)."""
import argparse
import csv
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", default="preprocess_fwd_kernel")
    ap.add_argument("--step", type=int, default=-2, help="which step (index into the marker launches; -2 = last complete one)")
    ap.add_argument("--min-us", type=float, default=8.0)
    ap.add_argument("--all-queues", action="store_true")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
    if len(marks) < 3:
        sys.exit("not enough marker launches")
    i0, i1 = marks[a.step], marks[a.step + 1] if a.step + 1 != 0 else len(rows)
    t0 = int(rows[i0]["Start_Timestamp"])
    t1 = int(rows[i1]["Start_Timestamp"]) if i1 < len(rows) else int(rows[-1]["End_Timestamp"])
    print("step length %.3f ms (%d kernels)" % ((t1 - t0) / 1e6, i1 - i0))
    queues = {}
    for r in rows[i0:i1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        q = r.get("Queue_Id", "?")
        queues.setdefault(q, []).append((s, e, r["Kernel_Name"]))
    main_q = max(queues, key=lambda q: sum(e - s for s, e, _ in queues[q] if "specular" not in _ and "cubemap" not in _))
    for q, ks in queues.items():
        busy = sum(e - s for s, e, _ in ks)
        print("queue %s: %d kernels, busy %.3f ms%s" % (q, len(ks), busy / 1e6, "  <- main" if q == main_q else ""))
    prev_end = 0
    for s, e, name in sorted(queues[main_q]):
        gap = s - prev_end
        if (e - s) / 1e3 >= a.min_us or gap / 1e3 >= a.min_us:
            print("  +%8.1f us  dur %8.1f us  gap-before %7.1f us  %s" % (s / 1e3, (e - s) / 1e3, gap / 1e3, name[:70]))
        prev_end = max(prev_end, e)
    for q, ks in queues.items():
        if q == main_q:
            continue
        print("queue %s spans +%.1f .. +%.1f us" % (q, min(s for s, _, _ in ks) / 1e3, max(e for _, e, _ in ks) / 1e3))
        if a.all_queues:
            prev_end = 0
            for s, e, name in sorted(ks):
                if (e - s) / 1e3 >= a.min_us or (s - prev_end) / 1e3 >= a.min_us:
                    print("  +%8.1f us  dur %8.1f us  gap-before %7.1f us  %s" % (s / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[:70]))
                prev_end = max(prev_end, e)


if __name__ == "__main__":
    main()
