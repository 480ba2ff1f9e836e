"""Dev tool: per-basic-block instruction class counts of one kernel in a hipcc -S listing."""
import re, sys, collections
path, start = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
i0 = next(i for i, l in enumerate(lines) if l.startswith(start) and l.rstrip().endswith(start.split(':')[0]) or l.startswith(start + ':'))
blocks = []; blk = ['entry', collections.Counter(), i0]; blocks.append(blk)
for i in range(i0 + 1, len(lines)):
    l = lines[i].strip()
    if l.startswith('s_endpgm'): break
    if re.match(r'^\.LBB\d+_\d+:', l):
        blk = [l, collections.Counter(), i]; blocks.append(blk); continue
    if not l or l.startswith(';') or l.startswith('.'): continue
    op = l.split()[0]
    cls = ('mfma' if 'mfma' in op else 'exp' if op.startswith('v_exp') else 'cvt' if op.startswith('v_cvt') else 'ds' if op.startswith('ds_') else
           'vmem' if op.startswith(('global_', 'buffer_', 'scratch_')) else 'wait' if op.startswith('s_waitcnt') else 'nop' if op.startswith('s_nop') else
           'branch' if op.startswith(('s_cbranch', 's_branch')) else 'salu' if op.startswith('s_') else 'valu')
    blk[1][cls] += 1; blk[1]['_n'] += 1
for b in blocks:
    if b[1]['_n'] > int(sys.argv[3]) if len(sys.argv) > 3 else 20: print(b[2], b[0], dict(b[1]))
