"""static instruction mix of a kernel's main loop (loop header .. first s_barrier, and .. end of loop) from hipcc -S output: python scripts/isa_count.py file.s <mangled-name-prefix>"""
import collections, sys
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
a = [i for i, l in enumerate(lines) if l.startswith(pref)][0]
b = [i for i, l in enumerate(lines) if '.amdhsa_kernel ' + pref in l][0]


def cls(l):
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        return None
    op = l.split()[0]
    for p, c in (('v_mfma', 'mfma'), ('v_pk', 'valu_pk'), ('v_', 'valu'), ('ds_', 'lds'), ('s_waitcnt', 'waitcnt'), ('s_barrier', 'barrier'), ('s_nop', 'nop'), ('s_', 'salu'),
                 ('global', 'vmem'), ('buffer', 'vmem')):
        if op.startswith(p):
            return c
    return 'other'


hs = [i for i in range(a, b) if 'Loop Header' in lines[i]]
for h in hs:
    seq = [cls(l) for l in lines[h:b]]
    c = collections.Counter(); nb = 0
    for s in seq:
        if not s:
            continue
        if s == 'barrier':
            nb += 1
            print(f"loop at line {h}: up to barrier {nb}: {dict(c)}")
            if nb == 4:
                break
        c[s] += 1
