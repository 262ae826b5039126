#!/usr/bin/env python
"""print the top kernels of a rocprofv3 kernel_stats.csv: name, calls, mean us, percentage"""
import csv
import sys
for r in list(csv.DictReader(open(sys.argv[1])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), r["Percentage"])
