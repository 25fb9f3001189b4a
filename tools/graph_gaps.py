"""Reads rocprofv3 kernel traces (CSV) of tools/graph_gap_probe.py and prints, for the last 5 forwards of each, the sum of kernel durations and the
idle time between consecutive kernels.   usage: python tools/graph_gaps.py eager_trace.csv graph_trace.csv"""
import csv, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda v: v[0])
    n = len(ks)
    tail = ks[n // 2:]  # the timed half (warm-up launches come first)
    busy = sum(e - s for s, e, _ in tail)
    gaps = [max(0, tail[i + 1][0] - tail[i][1]) for i in range(len(tail) - 1)]
    gaps_sorted = sorted(gaps)
    span = tail[-1][1] - tail[0][0]
    print(f"{path}: {len(tail)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle between kernels {sum(gaps) / 1e6:.3f} ms "
          f"(median gap {gaps_sorted[len(gaps) // 2] / 1e3:.1f} us, p90 {gaps_sorted[int(0.9 * len(gaps))] / 1e3:.1f} us, max {gaps_sorted[-1] / 1e3:.1f} us)")
