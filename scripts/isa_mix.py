"""Developer tool: instruction mix of one kernel in a hipcc -S dump.  usage: isa_mix.py file.s mangled_prefix"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith(pref) and ':' in l)
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
lines = [l.strip() for l in s[start + 1:end + 1]]
# find the main loop: last backward branch target with the largest span
labels = {l[:-1]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:$', l)}
best = None
for i, l in enumerate(lines):
    m = re.match(r'^s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'^s_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = i - labels[m.group(1)]
        if best is None or span > best[0]:
            best = (span, labels[m.group(1)], i)
def mix(ls, title):
    c = Counter()
    for l in ls:
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
            continue
        c[l.split()[0]] += 1
    g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print(title, 'total', sum(c.values()), '| valu', g('v_'), '(pk', g('v_pk_'), ') | ds_read', g('ds_read'), 'ds_write', g('ds_write'),
          '| gload', g('global_load'), 'gstore', g('global_store'), '| scratch', g('scratch_'), '| barrier', c['s_barrier'], 'waitcnt', c['s_waitcnt'], 'salu', g('s_'))
    return c
mix(lines, 'whole kernel:')
if best:
    c = mix(lines[best[1]:best[2] + 1], 'main loop   :')
    if len(sys.argv) > 3:
        for k, v in c.most_common(45):
            print('   ', k, v)
