"""Where a kernel of a hipcc -S listing spills, waits and multiplies: per basic block the counts of MFMA / scratch / s_waitcnt vmcnt /
LDS-DMA / barrier instructions (build container, no GPU).   python tools/asm_report.py file.s <mangled-name substring>"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith(key.split(':')[0]) or (l.startswith('_Z') and key in l.split(':')[0]))
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
body = s[start:end + 1]
blocks, cur = [], ['entry', 0, {}]
for i, l in enumerate(body):
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t):
        blocks.append(cur); cur = [t.split(':')[0], i, {}]
        continue
    for tag, pat in (('mfma', 'v_mfma'), ('scratch', 'scratch_'), ('vmcnt', 'vmcnt('), ('dma', ' lds'), ('barrier', 's_barrier'), ('ds_read', 'ds_read'), ('ds_write', 'ds_write'),
                     ('gstore', 'global_store'), ('gload', 'global_load'), ('branch', 's_cbranch'), ('readlane', 'v_readlane'), ('writelane', 'v_writelane')):
        if pat in t and not t.startswith(';'):
            cur[2][tag] = cur[2].get(tag, 0) + 1
    if 'vmcnt(' in t:
        cur[2].setdefault('vm', []).append(re.search(r'vmcnt\((\d+)\)', t).group(1))
    if 's_cbranch' in t or t.startswith('s_branch'):
        cur[2].setdefault('to', []).append(t.split()[-1])
blocks.append(cur)
print(len(body), 'lines')
for name, i, c in blocks:
    if c:
        print(f"{name:12s} @{i:6d} " + ' '.join(f"{k}={v}" for k, v in c.items()))
