"""Summarise rocprofv3 `--pmc ... --output-format csv` output (*counter_collection.csv files under a directory) into
kernel,counter,dispatches,avg_value — one line per (kernel, counter).  Usage: python tools/pmc_summary.py DIR [DIR ...]"""
import csv
import glob
import os
import re
import sys

agg = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            cols = {c.lower(): c for c in rd.fieldnames or []}
            kcol = next((cols[c] for c in cols if "kernel" in c and "name" in c), None)
            ccol = next((cols[c] for c in cols if "counter" in c and "name" in c), None)
            vcol = next((cols[c] for c in cols if "counter" in c and "value" in c), None)
            if not (kcol and ccol and vcol):
                print(f"# {f}: unexpected columns {rd.fieldnames}", file=sys.stderr)
                continue
            for row in rd:
                k = (re.sub(r"\s+", " ", row[kcol])[:160], row[ccol])
                a = agg.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += float(row[vcol])
print("kernel,counter,dispatches,avg_value")
for (k, c), (n, t) in sorted(agg.items(), key=lambda kv: (-kv[1][1], kv[0])):
    print(f"\"{k}\",{c},{n},{t / n:.1f}")
