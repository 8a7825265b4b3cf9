#!/usr/bin/env python3
"""Static instruction mix of a K1 hot loop (development aid).

    python tools/valu_mix.py [kernel-substring]        default: k1_pairsILi2ELi5E  (the c4 kernel)

Compiles icikt_kernels.hip to gfx950 assembly, finds the innermost loop that holds the packed in-step chain
(`row_shr:15`) and classifies its vector instructions by the issue classes measured in tools/ubench/valu_rate.hip:
  half-rate: DPP forms, VOPC / carry forms (they write or read VCC / an SGPR pair), v_bcnt, v_perm, every three-operand
             integer form (v_and_or, v_lshl_or, v_add3, v_lshl_add, v_bfe, v_alignbit, v_mad_*, v_cndmask with an SGPR
             mask), 16-bit packed forms, permlane swaps (8 cycles: counted twice)
  full-rate: everything else (v_add_u32, v_sub_u32, v_and, v_or, v_xor, shifts, v_mov, v_mul_u32_u24 is half)
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HALF3 = ("v_and_or", "v_lshl_or", "v_add3", "v_lshl_add", "v_add_lshl", "v_bfe", "v_bfi", "v_alignbit", "v_alignbyte", "v_mad_", "v_perm_b32",
         "v_or3", "v_xad", "v_med3", "v_min3", "v_max3", "v_mul_u32_u24", "v_mul_i32_i24", "v_mul_lo", "v_mul_hi", "v_pk_", "v_bcnt", "v_mbcnt",
         "v_cmp", "v_addc", "v_subb", "v_add_co", "v_sub_co", "v_subrev_co", "v_readlane", "v_readfirstlane", "v_writelane")


def classify(l):
    op = l.split()[0]
    if not op.startswith("v_"):
        return op.split("_")[0]            # s / ds / global / buffer ...
    if "permlane" in op:
        return "v8"
    if "_dpp" in op or "row_" in l or "quad_perm" in l or "wave_sh" in l or "_sdwa" in op:
        return "vH"
    if op.startswith(HALF3):
        return "vH"
    if op.startswith("v_cndmask") and ("_e64" in op or re.search(r"s\[\d+:\d+\]", l)):
        return "vH"
    return "vF"


def hot_loops(want, verbose=False):
    """[(first line, last line, {class: count})] of the loops of kernel `want` that hold the in-step compare chain,
    innermost first."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "icikendalltau_amd", "csrc"), "--cuda-device-only", "-S", "-o", out,
                        os.path.join(ROOT, "icikendalltau_amd", "csrc", "icikt_kernels.hip")], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if want in l and re.match(r"^_Z\w+:", l))
    # (the function's end label, not its first s_endpgm: a kernel with early returns -- a wave whose segment is empty -- has several)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end + 1]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"\bs_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    res = []
    for a, b in sorted(loops, key=lambda ab: ab[1] - ab[0]):
        seg = [l.strip() for l in body[a:b + 1] if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
        if not any("row_shr:15" in l for l in seg):
            continue
        cnt = {}
        for l in seg:
            c = classify(l)
            cnt[c] = cnt.get(c, 0) + 1
        cnt["_inner_loops"] = sum(1 for x in loops if a < x[0] and x[1] < b)
        res.append((a, b, cnt, seg if verbose else None))
    return res


def hot_loop_mix(want):
    """{"full": n, "half": n, "permlane": n, "ds": n, "salu": n, ...} of the innermost hot loop of kernel `want`."""
    # the innermost loop that holds the compare chain AND gathers rows (a step of the walk)
    # (the one-pair / two-pair long-column kernels have two such loops: the one with the PACKED chain -- v_perm_b32 -- is the
    #  one that runs on columns without odd-positioned tie groups, i.e. on every benchmark workload)
    loops = [x for x in hot_loops(want, verbose=True) if x[2].get("global", 0) > 0]
    packed = [x for x in loops if any("v_perm_b32" in l for l in x[3])]
    # two long-column pairs per wave: the singleton region runs in the half layout (the loop with v_permlane32_swap)
    swapped = [x for x in packed if any("permlane32_swap" in l for l in x[3])] if "Li2ELi0E" in want else []
    a, b, cnt, _ = (swapped or packed or loops)[0]
    return {"kernel": want, "full": cnt.get("vF", 0), "half": cnt.get("vH", 0), "permlane": cnt.get("v8", 0),
            "ds": cnt.get("ds", 0), "global": cnt.get("global", 0), "salu_and_waits": cnt.get("s", 0)}


if __name__ == "__main__":
    want = next((x for x in sys.argv[1:] if not x.startswith("-")), "k1_pairsILi2ELi5E")
    for a, b, cnt, seg in hot_loops(want, verbose="-v" in sys.argv):
        nF, nH, n8 = cnt.get("vF", 0), cnt.get("vH", 0), cnt.get("v8", 0)
        print(f"loop at lines {a}..{b} ({cnt['_inner_loops']} inner loops): VALU full-rate {nF}, half-rate {nH}, "
              f"permlane {n8}; other: " + ", ".join(f"{k} {v}" for k, v in sorted(cnt.items()) if not k.startswith(("v", "_"))))
        print(f"  issue floor if every VALU took a 4-cycle slot: {4 * (nF + nH + 2 * n8)} cycles; "
              f"if full-rate forms of different waves overlapped completely: {4 * (nH + 2 * n8) + 2 * nF}")
        if seg:
            for l in seg:
                print("   ", classify(l), l)
