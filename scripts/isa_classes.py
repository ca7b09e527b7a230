"""VALU instruction mix of the traversal loop, by issue class (static count over the loop body of the first-round kernel).

    hipcc ... -S --cuda-device-only kkd_kernels.hip -o /tmp/kk.s ;  python scripts/isa_classes.py /tmp/kk.s > profiles/r3_isa_classes.json

Every VALU mnemonic of the loop (from its 'Inner Loop Header' label to the last branch back to it) is mapped to the instruction
kind of tests/tools/valu_peak.hip whose measured issue rate applies to it (profiles/valu_peak_gfx950.txt); bench.py weighs the
dynamic instruction count of the kernel (PMC SQ_INSTS_VALU) with this mix.  The mix is static: blocks the walk skips at run time
(stack spills, pops) count as much as the slab tests, so it is an approximation, good to a few per cent for a loop whose body is
almost straight-line code."""
import json
import re
import sys

# mnemonic prefix -> valu_peak kind (first match wins)
CLASS = [
    ("v_fma_mix", "v_fma_f32"), ("v_pk_fma_f32", "v_pk_fma_f32"), ("v_fma_f32", "v_fma_f32"), ("v_fmac_f32", "v_fma_f32"), ("v_mul_f32", "v_mul_f32"),
    ("v_div_scale", "v_fma_f32"), ("v_div_fmas", "v_fma_f32"), ("v_div_fixup", "v_fma_f32"), ("v_rcp", "v_rcp_f32"), ("v_rsq", "v_rcp_f32"), ("v_sqrt", "v_rcp_f32"),
    ("v_cvt_f32_ubyte", "v_cvt_f32_ubyte1"), ("v_cvt", "v_cvt_f32_ubyte1"),
    ("v_max3", "v_max3_f32"), ("v_min3", "v_max3_f32"), ("v_med3", "v_med3_f32"), ("v_max_f32", "v_min_f32"), ("v_min_f32", "v_min_f32"),
    ("v_min_u32", "v_min_u32"), ("v_max_u32", "v_min_u32"), ("v_min_i32", "v_min_u32"), ("v_max_i32", "v_min_u32"),
    ("v_cmp", "v_cmp_lt_f32_e64"), ("v_cndmask", "v_cndmask_e64 sgpr"), ("v_perm", "v_perm_b32"),
    ("v_lshl_or", "v_lshl_or_b32"), ("v_lshl_add", "v_lshl_or_b32"), ("v_and_or", "v_lshl_or_b32"), ("v_bfe", "v_lshl_or_b32"), ("v_bfi", "v_lshl_or_b32"),
    ("v_add3", "v_lshl_or_b32"), ("v_mad_u", "v_lshl_or_b32"), ("v_mul_lo", "v_rcp_f32"), ("v_mul_hi", "v_rcp_f32"), ("v_bitop3", "v_lshl_or_b32"),
    ("v_sub_f32", "v_sub_f32"), ("v_add_f32", "v_sub_f32"), ("v_subrev_f32", "v_sub_f32"),
    ("v_and", "v_and_b32"), ("v_or", "v_and_b32"), ("v_xor", "v_and_b32"), ("v_not", "v_and_b32"), ("v_lshlrev", "v_and_b32"), ("v_lshrrev", "v_and_b32"), ("v_ashrrev", "v_and_b32"),
    ("v_add", "v_add_u32"), ("v_sub", "v_add_u32"), ("v_mov", "v_mov_b32"), ("v_ffb", "v_and_b32"), ("v_bcnt", "v_and_b32"), ("v_readlane", "v_mov_b32"), ("v_readfirstlane", "v_mov_b32"),
    ("v_mbcnt", "v_and_b32"), ("v_accvgpr", "v_mov_b32"),
]


def loop_mix(txt, kernel_re):
    m = re.search(r"\n(%s[A-Za-z0-9_]*):" % kernel_re, txt)
    if not m:
        raise SystemExit("kernel %s not found" % kernel_re)
    body = txt[m.end():txt.find("s_endpgm", m.end())]
    hd = re.search(r"\n(\.LBB\d+_\d+):\s*;[^\n]*Inner Loop Header", body)
    label = hd.group(1)
    tag = "Header=" + label[2:]                      # blocks of the loop are annotated "in Loop: Header=BBn_m"
    lines, inloop = [], False
    for line in body.split("\n"):
        st = line.strip()
        if re.match(r"\.LBB\d+_\d+:", st) or st.startswith("; %bb."):
            inloop = st.startswith(label + ":") or tag in st
            continue
        if inloop:
            lines.append(line)
    loop = "\n".join(lines)
    mix, kinds, unknown = {}, {}, {}
    for line in loop.split("\n"):
        t = line.split()
        if not t or not t[0].startswith("v_"):
            continue
        mn = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])
        mix[mn] = mix.get(mn, 0) + 1
        for pre, kind in CLASS:
            if mn.startswith(pre):
                kinds[kind] = kinds.get(kind, 0) + 1
                break
        else:
            unknown[mn] = unknown.get(mn, 0) + 1
            kinds["v_lshl_or_b32"] = kinds.get("v_lshl_or_b32", 0) + 1
    n = sum(kinds.values())
    salu = sum(1 for line in loop.split("\n") if line.split() and line.split()[0].startswith("s_"))
    return {"kernel": m.group(1), "loop": label, "valu_instructions": n, "salu_instructions": salu,
            "share_by_kind": {k: round(v / n, 4) for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])},
            "mnemonics": dict(sorted(mix.items(), key=lambda kv: -kv[1])), "unclassified": unknown}


if __name__ == "__main__":
    txt = open(sys.argv[1]).read()
    out = {"source": "hipcc -O3 -ffp-contract=on -fno-slp-vectorize --offload-arch=gfx950 -S --cuda-device-only kkd_kernels.hip",
           "bsdf": loop_mix(txt, "_ZN3kkd11kTraceRoundILb0ELb0ELi0ELb1ELb1"),      # closest hit, projection test, first round
           "shadow": loop_mix(txt, "_ZN3kkd11kTraceRoundILb1ELb0ELi2ELb1ELb1")}    # any hit
    print(json.dumps(out, indent=1))
