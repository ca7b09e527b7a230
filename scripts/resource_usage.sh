#!/bin/bash
# Register / scratch / LDS use of every kernel of kkd_kernels.hip (cross-compiles, no GPU needed): name, VGPRs, SGPR spills, scratch bytes, occupancy
cd "$(dirname "$0")/../ray-tracer_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize --offload-arch=gfx950 $EXTRA -Rpass-analysis=kernel-resource-usage -c ${1:-kkd_kernels.hip} -o /tmp/kkd_ru.o 2>&1 |
python3 -c '
import re, sys, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark: +([A-Za-z ]+(?:\[[^\]]*\])?): +(\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        if cur: rows.append(cur)
        cur = {"name": v}
    else: cur[k] = v
if cur: rows.append(cur)
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void kkd::", "")
    print("%-70s vgpr %3s  sgpr-spill %3s  vgpr-spill %3s  scratch %4s  occ %s  lds %s" % (n[:70], r.get("VGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
