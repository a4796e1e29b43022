#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats CSV compactly: name, calls, average us, share."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > (float(sys.argv[2]) if len(sys.argv) > 2 else 0.3):
        print("%-72s %6s  avg_us %9.1f  %6.2f%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
