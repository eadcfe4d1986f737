"""rocprofv3 results .db -> per-kernel stats CSV (what `--stats` prints).  usage: db2stats.py results.db out.csv"""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                  "group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / tot, 2)])
print("kernels", len(rows), "total ms", tot / 1e6)
