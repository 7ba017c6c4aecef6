#!/usr/bin/env python3
"""Copies the summaries of a tools/profile_round.sh run from gpurun_out/<round>/ into profiles/<round>/."""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main() -> None:
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src, dst = os.path.join(ROOT, "gpurun_out", rnd), os.path.join(ROOT, "profiles", rnd)
    os.makedirs(os.path.join(dst, "config5"), exist_ok=True)
    for name in ("bench_line.json", "pair_bench_under_rocprof.json", "bench_single_rank_rccl_overlap.json",
                 "two_call_bench_line.json", "config4_4k_bench_line.json"):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    for f in ("guided", "median", "gaussian", "bilateral"):
        shutil.copy(os.path.join(src, f"config5_{f}_8k_bench_line.json"), os.path.join(dst, "config5", f"{f}_8k_bench_line.json"))
    shutil.copy(os.path.join(src, "pmc_pair", "pmc_per_launch.json"), os.path.join(dst, "pair_pmc_per_launch.json"))
    for trace, out in (("trace", "pair_kernel_stats.csv"), ("trace2", "two_call_kernel_stats.csv"), ("trace4k", "config4_4k_kernel_stats.csv")):
        hits = glob.glob(os.path.join(src, trace, "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[0], os.path.join(dst, out))
    for name in ("param_table.txt", "eps_0.005.json", "eps_0.0001.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    for f in ("bilateral", "median"):   # counter passes of the 8K filters, when tools/pmc_passes.sh was run for them
        hit = os.path.join(src, f"pmc_{f}", "pmc_per_launch.json")
        if os.path.exists(hit):
            shutil.copy(hit, os.path.join(dst, "config5", f"{f}_pmc_per_launch.json"))
    print("collected into", dst)


if __name__ == "__main__":
    main()
