#!/usr/bin/env python3
"""Timing-only variants of the two-column fused kernel (wrong results on purpose): each removes one
group of instructions from the row loop, so that the difference of the kernel times prices that group.
Builds tools/ab/libstereo_ab_<name>.so; run `python bench.py --no-cpu-baseline` with STEREO_HIP_LIB set.

The edits are text replacements anchored on the kernel source as it was at commit 345c115 (the build whose
prices the design notes quote): `git show 345c115:stereo-matching_amd/csrc/guided_pair.hip > /tmp/gp.hip` and point SRC
at that copy to reproduce them.  Against a later source some anchors no longer exist and the script stops with
the name of the first one; tools/ab/build_variant.sh builds variants of the current source from -D macros."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "stereo-matching_amd")
SRC = os.environ.get("AB_SRC", os.path.join(PKG, "csrc", "guided_pair.hip"))
OUT = os.path.join(ROOT, "tools", "ab")

KEEP = 'asm volatile("" :: "v"({0}));'


def variant(name, text):
    def rep(a, b):
        nonlocal text
        assert text.count(a) == 1, (name, a, text.count(a))
        text = text.replace(a, b)
    if name == "base":
        pass
    elif name == "nocorr":      # no A * (G(x) - c) correction and no lane shifts of A
        rep("                        P[u].x = dpp_mov_f<DPP_WAVE_SHL1>(P[u - 1].x);\n", "")
        rep("                        P[u].y = dpp_mov_f<DPP_WAVE_SHL1>(P[u - 1].y);\n", "")
        rep("    for (int u = 1; u < RG; ++u) dsum = fma2(gp[0] - gp[u].xx, P[u].yy, dsum);", "    for (int u = 1; u < 1; ++u) {}")
        rep("                    dsum = fma2(gp[0] - gp[RG].xx, P[RG], dsum);", "                    dsum += P[0];")
    elif name == "nof2":        # no float32 g_lo * p chain
        rep("sf[0] = ef_ + dpp_f<DPP_WAVE_SHR1>(ofr);", "sf[0] = F2.x;")
        rep("sf[1] = of_ + efr;", "sf[1] = F2.y;")
        rep("                    ef_ = F2.x + dpp_f<DPP_WAVE_SHR1>(ef_);\n", "")
        rep("                    of_ = F2.y + dpp_f<DPP_WAVE_SHR1>(of_);\n", "")
    elif name == "noechain":    # no lane chain of E
        rep("                        ee = E.x + dpp_f<DPP_WAVE_SHL1>(ee);\n", "")
        rep("                        eo = E.y + dpp_f<DPP_WAVE_SHL1>(eo);\n", "")
        rep("se.y = eo + dpp_f<DPP_WAVE_SHL1>(eer);", "se.y = eo + eer;")
    elif name == "nodpp1":      # no stage-1 lane chains at all
        a0 = text.index("#pragma unroll\n                for (int u = 0; u < RG; ++u) {\n                    e1r = e1;")
        a1 = text.index("                // (the integer partial sums are pinned")
        text = text[:a0] + text[a1:]
        rep("const uint32_t tp = (uint32_t)__builtin_amdgcn_mov_dpp((int)cpr, DPP_WAVE_SHR1, 0xf, 0xf, true);", "const uint32_t tp = cpr;")
        rep("si[0] = ei + dpp_i<DPP_WAVE_SHR1>(oir);", "si[0] = ei + oir;")
        rep("sf[0] = ef_ + dpp_f<DPP_WAVE_SHR1>(ofr);", "sf[0] = ef_ + ofr;")
    elif name == "f32ab":       # a, b and their sums in float32
        a0 = text.index("                // 4. a, b; 5. their vertical running sums")
        a1 = text.index("                if (STEADY || i >= 2 * HALO) {")
        blk = text[a0:a1].replace("double", "float").replace("fma(", "fmaf(")
        text = text[:a0] + blk + text[a1:]
        rep("    double A[2] = {0.0, 0.0}, B[2] = {0.0, 0.0};", "    float A[2] = {0.f, 0.f}, B[2] = {0.f, 0.f};")
        rep("    double ar[2][WIN], br[2][WIN];", "    float ar[2][WIN], br[2][WIN];")
        rep("const double cref = (double)gp[0].x;", "const float cref = gp[0].x;")
        rep("E.x = (float)fma(A[0], cref, B[0]);", "E.x = fmaf(A[0], cref, B[0]);")
        rep("E.y = (float)fma(A[1], cref, B[1]);", "E.y = fmaf(A[1], cref, B[1]);")
        rep("const double mG[2] = {cur.mG[0], cur.mG[1]};", "const float mG[2] = {(float)cur.mG[0], (float)cur.mG[1]};")
    elif name == "nokey":       # no key update
        a0 = text.index("                        const unsigned long long key = ((unsigned long long)ordered_bits(tf) << 32)")
        a1 = text.index("                    }\n                }\n            }\n            cur = nxt;")
        text = text[:a0] + "                        " + KEEP.format("tf") + "\n" + text[a1:]
    elif name == "noab":        # a, b replaced by trivial values (no float64 arithmetic, no conversions)
        a0 = text.index("                    const double s1d = (double)s1[j];")
        a1 = text.index("                    A[j] += av;")
        text = text[:a0] + "                    const double av = __hiloint2double((int)s1[j], si[j]), bv = __hiloint2double(__float_as_int(sf[j]), (int)s1[j]);\n" + text[a1:]
    elif name == "novert2":     # no vertical running sums of a, b
        rep("                        A[j] -= ar[j][k2];\n", "")
        rep("                        B[j] -= br[j][k2];\n", "")
    elif name == "nocost":      # no census cost (p = lane constant)
        rep("const uint32_t fs = field_sum((WordT)(cur.L[j] ^ Rcur[j]));", "const uint32_t fs = (uint32_t)cur.L[j] & 127u;")
    else:
        raise SystemExit(f"unknown variant {name}")
    return text


def main():
    names = sys.argv[1:] or ["base", "nocorr", "nof2", "noechain", "nodpp1", "f32ab", "nokey", "nocost", "noab", "novert2"]
    src = open(SRC).read()
    objs = [os.path.join(PKG, "csrc", "obj", f) for f in os.listdir(os.path.join(PKG, "csrc", "obj"))
            if f.endswith(".o") and f != "guided_pair.o"]
    procs = []
    for n in names:
        path = os.path.join(OUT, f"guided_pair_{n}.hip")
        open(path, "w").write(variant(n, src))
        obj = os.path.join(OUT, f"guided_pair_{n}.o")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-w",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-c", path, "-o", obj]
        procs.append((n, obj, subprocess.Popen(cmd)))
        if len(procs) % 4 == 0:
            for _, _, p in procs[-4:]:
                p.wait()
    for n, obj, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"compile of {n} failed")
        so = os.path.join(OUT, f"libstereo_ab_{n}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + objs)
        print("built", so)


if __name__ == "__main__":
    main()
