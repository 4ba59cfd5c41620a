#!/usr/bin/env python3
"""Per-kernel mean of every counter in rocprofv3 --pmc output (…_counter_collection.csv files under a directory)."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path


def main(root, pattern="ba_"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in Path(root).rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pattern in row["Kernel_Name"]:
                    acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} | {"launches": max(len(v) for v in cs.values())} for k, cs in acc.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
