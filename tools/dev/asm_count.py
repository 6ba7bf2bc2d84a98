"""dev: instruction-class census of the kernels in a hipcc -S dump.  usage: asm_count.py file.s [name-substring]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\S+): +; @', s, flags=re.M):
    name = m.group(1)
    if pat not in name: continue
    j = s.index('.Lfunc_end', m.end())
    c = Counter()
    n = 0
    for l in s[m.end():j].split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;': continue
        k = t.split()[0]; n += 1
        if k.startswith('v_mfma'): c['mfma'] += 1
        elif k.startswith('v_accvgpr'): c['accvgpr'] += 1
        elif k.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')): c['trans'] += 1
        elif k.startswith(('v_mul_lo', 'v_mul_hi', 'v_mad_u64')): c['imul'] += 1
        elif k.startswith('v_cndmask'): c['cndmask'] += 1
        elif k.startswith('v_cmp'): c['cmp'] += 1
        elif k.startswith('v_'): c['valu'] += 1
        elif k.startswith('ds_'): c['ds'] += 1
        elif k.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif k.startswith('s_barrier'): c['barrier'] += 1
        elif k.startswith('s_'): c['salu'] += 1
        elif k.startswith(('global', 'buffer', 'flat', 'scratch')): c['vmem'] += 1
        else: c[k] += 1
    print(name[-50:], n, dict(c))
