#!/usr/bin/env python3
"""Per-launch averages of rocprofv3 --pmc counter CSVs.

    python tools/pmc_summary.py [--library-version TEXT] gpurun_out/pmcA gpurun_out/pmcB ... > profiles/rNN/pmc_per_launch.json

Each directory is the -d output of one `rocprofv3 --pmc ...` pass; the script finds the
*counter_collection.csv files below it, averages every counter per kernel name over the
launches seen, and prints one JSON object {kernel: {counter: mean}}.  --library-version stores the text of
stereo_hip_version() of the library the passes ran on under "__library_version__": bench.py quotes counters only
from a summary whose entry equals the version of the library it is running.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main() -> None:
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    args = sys.argv[1:]
    version = None
    if args and args[0] == "--library-version":
        version, args = args[1], args[2:]
    for root in args:
        for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    slot = acc[row["Kernel_Name"]][row["Counter_Name"]]
                    slot[0] += float(row["Counter_Value"])
                    slot[1] += 1
    out = {k: {c: v[0] / v[1] for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
    if version is not None:
        out["__library_version__"] = version
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
