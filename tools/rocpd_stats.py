"""Per-kernel summary (calls, total/avg/min/max duration, share) from a rocprofv3 rocpd SQLite database.
usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
    "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size) "
    "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPRs", "AGPRs", "LDS_bytes")]
for name, calls, tot, avg, mn, mx, vg, ag, lds in rows:
    out.append((name, calls, int(tot), round(avg, 1), int(mn), int(mx), round(100.0 * tot / total, 2), vg, ag, lds))
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerows(out)
