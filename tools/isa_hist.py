"""Diagnostic: instruction histogram of one kernel of a device assembly file (hipcc -S --cuda-device-only), per loop.
usage: python tools/isa_hist.py file.s <substring of the kernel symbol> [first_line last_line]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and key in l)
end = next(i for i in range(start, len(lines)) if '.end_amdhsa_kernel' in lines[i])
if len(sys.argv) > 4:
    start, end = int(sys.argv[3]) - 1, int(sys.argv[4])
labels = {}
for i in range(start, end):
    m = re.match(r'^(\.LBB\w+):', lines[i])
    if m: labels[m.group(1)] = i
# back edges: branch to a label at an earlier line
loops = []
for i in range(start, end):
    m = re.match(r'^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\w+)', lines[i])
    if m and m.group(2) in labels and labels[m.group(2)] <= i:
        loops.append((labels[m.group(2)], i))
def hist(a, b):
    h = collections.Counter()
    for i in range(a, b + 1):
        m = re.match(r'^\s+([a-z_0-9]+)\s', lines[i] + ' ')
        if m and not lines[i].lstrip().startswith(('.', ';')): h[m.group(1)] += 1
    return h
def classes(h):
    c = collections.Counter()
    for k, v in h.items():
        if k.startswith(('v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_rcp_f64', 'v_rsq_f64', 'v_ldexp_f64', 'v_rndne_f64', 'v_max_f64', 'v_min_f64', 'v_cmp_', 'v_cmpx')) and 'f64' in k: c['f64'] += v
        elif k.startswith('v_cvt'): c['cvt'] += v
        elif k.startswith('ds_'): c['lds:' + k] += v
        elif k.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): c['mem:' + k] += v
        elif k.startswith('v_readlane') or k.startswith('v_writelane') or k.startswith('v_readfirstlane'): c[k] += v
        elif k.startswith('v_'): c['valu_other'] += v
        elif k.startswith('s_'): c['salu'] += v
        else: c[k] += v
    return c
print('kernel lines', start + 1, end + 1, 'total', sum(hist(start, end).values()))
for a, b in sorted(set(loops), key=lambda t: t[0] - t[1])[:int(sys.argv[5]) if len(sys.argv) > 5 else 12]:
    h = hist(a, b)
    print('loop', a + 1, b + 1, 'instr', sum(h.values()), dict(classes(h).most_common(14)))
if len(sys.argv) > 4:
    h = hist(start, end)
    print(h.most_common(60))
