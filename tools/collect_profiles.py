#!/usr/bin/env python3
"""Copies the summaries of a tools/profile_round.sh run from gpurun_out/<round>/ into profiles/<round>/."""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main() -> None:
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src, dst = os.path.join(ROOT, "gpurun_out", rnd), os.path.join(ROOT, "profiles", rnd)
    os.makedirs(os.path.join(dst, "config5"), exist_ok=True)

    def copy(a: str, b: str) -> None:
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, b))
        else:
            print("missing:", a)

    for name in ("bench_line.json", "bench_single_rank_rccl_overlap.json", "two_call_bench_line.json",
                 "config4_4k_bench_line.json", "param_table.txt", "eps_0.005.json", "eps_0.0001.json",
                 "precompute_loop.txt", "shard_compute.txt", "shard_compute_4k.txt"):
        copy(name, name)
    copy("pair.json", "pair_bench_under_rocprof.json")
    for f in ("guided", "median", "gaussian", "bilateral"):
        copy(f"config5_{f}_8k_bench_line.json", os.path.join("config5", f"{f}_8k_bench_line.json"))
    # kernel-trace summaries (and, for the N > 1 rehearsal, the trace itself: which stream ran when)
    for trace, out in (("pair", "pair_kernel_stats.csv"), ("two_call", "two_call_kernel_stats.csv"),
                       ("config4_4k", "config4_4k_kernel_stats.csv"), ("dist_single", "dist_single_kernel_stats.csv"),
                       ("config5_guided", "config5/guided_kernel_stats.csv"), ("config5_median", "config5/median_kernel_stats.csv"),
                       ("config5_gaussian", "config5/gaussian_kernel_stats.csv"),
                       ("config5_bilateral", "config5/bilateral_kernel_stats.csv")):
        hits = glob.glob(os.path.join(src, trace, "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[0], os.path.join(dst, out))
        else:
            print("missing trace:", trace)
    hits = glob.glob(os.path.join(src, "dist_single", "**", "*kernel_trace.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(dst, "dist_single_kernel_trace.csv"))
    # counter passes
    copy(os.path.join("pmc_pair", "pmc_per_launch.json"), "pair_pmc_per_launch.json")
    copy(os.path.join("pmc_two_call", "pmc_per_launch.json"), "two_call_pmc_per_launch.json")
    for f in ("bilateral", "median", "gaussian"):
        copy(os.path.join(f"pmc_{f}", "pmc_per_launch.json"), os.path.join("config5", f"{f}_pmc_per_launch.json"))
    print("collected into", dst)


if __name__ == "__main__":
    main()
