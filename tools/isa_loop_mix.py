"""Instruction mix of the hottest loop (the backward branch spanning the most v_mfma) of kernels in a gfx950 .s
dump:  hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o conv.s rrnet_amd/csrc/conv.hip
       python tools/isa_loop_mix.py conv.s <substring of the mangled kernel name> ..."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\S+:', l)]
for idx, (i0, name) in enumerate(starts):
    if not any(p in name for p in sys.argv[2:]):
        continue
    i1 = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
    body = lines[i0:i1]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'(\.LBB\d+_\d+):', l)] if m}
    best = None
    for i, l in enumerate(body):
        m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = body[labels[m.group(1)]:i]
            c = sum('v_mfma' in x for x in seg)
            # the innermost hot loop: most MFMAs per instruction
            dens = c / max(len(seg), 1)
            if c > 0 and (best is None or dens > best[3]):
                best = (c, labels[m.group(1)], i, dens)
    if not best:
        continue
    seg = body[best[1]:best[2]]
    cnt = collections.Counter()
    for x in seg:
        x = x.strip()
        if not x or x[0] in ';.':
            continue
        op = x.split()[0]
        key = ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'waitcnt' if op.startswith('s_waitcnt')
               else 'salu' if op.startswith('s_') else 'ds' if op.startswith('ds_') else 'buffer' if op.startswith('buffer_')
               else 'global' if op.startswith('global_') else op)
        cnt[key] += 1
    print(name[:100])
    print('  loop:', dict(cnt), ' waterfall:', sum('s_and_saveexec' in x for x in seg))
    m = re.search(r'; NumVgprs: (\d+)', '\n'.join(lines[i1:i1 + 400]) if False else '\n'.join(body))
