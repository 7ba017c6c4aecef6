#!/usr/bin/env python3
"""Weighted VALU-pipe cost of the innermost loops of a kernel in a gfx950 assembly listing.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o k.s kernel.hip
    python tools/isa_cost.py k.s <mangled-kernel-name-substring> [waves_per_simd]

Weights are cycles of one SIMD per wave-instruction with every SIMD busy, measured on MI355X by
tools/valu_rate.hip: plain two-operand f32 / int32 adds, muls, logic and moves 2.05; v_fma/fmac_f32
2.6; DPP forms, three-operand integer ops, dot, conversions, selects, compares, min/max 4.1;
float64 and packed-f32 arithmetic 3.1 with four waves per SIMD, 4.25 with two.
"""
import re
import sys
from collections import Counter


def weight(op: str, wps: int) -> float:
    wide = 3.12 if wps >= 4 else 4.25
    if op.endswith("_dpp"):
        return 4.13
    if re.match(r"v_(fma|add|mul|fmac|min|max)_f64", op) or op.startswith("v_pk_") or op in ("v_lshl_add_u64", "v_mov_b64_e32"):
        return wide
    if op.startswith("v_cvt_") or op.startswith("v_cmp") or op.startswith("v_cndmask") or op.startswith("v_dot"):
        return 4.1
    if op.startswith(("v_add3", "v_mad_", "v_mul_i32", "v_mul_u32", "v_mul_lo", "v_mul_hi", "v_and_or", "v_bfi", "v_bfe",
                      "v_min_", "v_max_", "v_med3", "v_rcp", "v_lshl_add", "v_lshl_or", "v_xad", "v_perm", "v_alignbit")):
        return 4.1
    if op.startswith(("v_fma_f32", "v_fmac_f32")):
        return 2.6
    if op.startswith("v_bitop3"):
        return 2.45
    if op.startswith("v_"):
        return 2.05
    return 0.0


def main() -> None:
    path, name = sys.argv[1], sys.argv[2]
    wps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    for s, e in loops:
        ops = Counter()
        for l in body[s:e]:
            l = l.strip()
            if not l or l.startswith((";", ".")):
                continue
            ops[l.split()[0]] += 1
        nv = sum(n for o, n in ops.items() if o.startswith("v_"))
        if nv < 200:
            continue
        cost = sum(weight(o, wps) * n for o, n in ops.items())
        classes = Counter()
        for o, n in ops.items():
            w = weight(o, wps)
            key = "dpp" if o.endswith("_dpp") else "wide(f64/pk)" if w in (3.12, 4.25) else "cvt" if o.startswith("v_cvt") else \
                "other4" if w == 4.1 else "cheap" if w > 0 else "non-valu"
            classes[key] += w * n
        print(f"loop lines {s}-{e}: {nv} VALU, {cost:.0f} pipe cycles;", ", ".join(f"{k} {v:.0f}" for k, v in classes.most_common() if v))
        print("   ", " ".join(f"{o}:{n}" for o, n in ops.most_common(40)))


if __name__ == "__main__":
    main()
