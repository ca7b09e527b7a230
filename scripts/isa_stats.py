"""Static instruction statistics per kernel of a gfx950 assembly dump (hipcc -S --cuda-device-only)."""
import re, sys
for f in sys.argv[1:]:
    txt = open(f).read()
    for m in re.finditer(r'\n(_Z[A-Za-z0-9_]+):', txt):
        name = m.group(1)
        i = m.end(); j = txt.find('s_endpgm', i)
        lines = [l.split()[0] for l in txt[i:j].split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith((';', '.'))]
        tail = txt[j:j + 6000]
        vg = re.search(r'; NumVgprs: (\d+)', tail); sc = re.search(r'; ScratchSize: (\d+)', tail); oc = re.search(r'; Occupancy: (\d+)', tail)
        short = re.sub(r'^_ZN3kkd', '', name)[:40]
        print('%-42s insts %6d valu %6d salu %6d div %4d sqrt %4d vgprs %s scratch %s occ %s' % (short, len(lines),
              sum(l.startswith('v_') for l in lines), sum(l.startswith('s_') for l in lines), lines.count('v_div_fixup_f32'),
              sum(l.startswith('v_sqrt') for l in lines), vg and vg.group(1), sc and sc.group(1), oc and oc.group(1)))
