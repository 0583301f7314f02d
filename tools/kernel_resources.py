"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output for this library's kernels."""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else 'gi-gs_amd/build/resource_usage.txt'
txt = open(path).read()
blocks = re.split(r'(?=remark: [^\n]*Function Name:)', txt)
keys = [('sgpr', r'TotalSGPRs'), ('vgpr', r' VGPRs'), ('agpr', r'AGPRs'), ('scratch', r'ScratchSize \[bytes/lane\]'),
        ('occ', r'Occupancy \[waves/SIMD\]'), ('lds', r'LDS Size \[bytes/block\]')]
for b in blocks:
    m = re.search(r'Function Name: (\S+)', b)
    if not m or 'gigs' not in m.group(1) or 'rocprim' in m.group(1):
        continue
    vals = []
    for name, k in keys:
        mm = re.search(k + r': (\d+)', b)
        vals.append('%s=%s' % (name, mm.group(1) if mm else '?'))
    print('%-72s %s' % (m.group(1)[:72], ' '.join(vals)))
