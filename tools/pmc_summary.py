"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc/*/**/_counter_collection.csv) per kernel."""
import csv, glob, json, os, re, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        m = re.search(r'gigs::(\w+)', k)
        name = m.group(1) if m else ('rocprim_sort' if 'radix' in k or 'merge' in k else None)
        if name is None:
            continue
        t = re.search(r'gigs::\w+(<[^>]*>)', k)
        if t: name += t.group(1).replace(' ', '')
        if name.startswith('specular_apply'): name += '@' + r.get('Grid_Size', r.get('Grid_Size_X', '?'))
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in sorted(agg.items()):
    out[k] = {c: {'mean_per_launch': sum(v) / len(v), 'launches': len(v)} for c, v in cs.items()}
json.dump(out, open(os.path.join(root, 'summary.json'), 'w'), indent=1)
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print('   %-28s %14.4g  (%d launches)' % (c, v['mean_per_launch'], v['launches']))
