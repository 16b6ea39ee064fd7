#!/usr/bin/env python
"""Dump the per-kernel summary (name, calls, total us, avg us, %) of a rocprofv3 rocpd sqlite database as CSV."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
for r in rows:
    w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.3f}"])
