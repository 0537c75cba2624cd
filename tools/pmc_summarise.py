#!/usr/bin/env python3
"""Summarise the rocprofv3 output directories written by tools/pmc_collect.sh: per kernel family (search / emit /
value) the mean counter value and duration per launch, plus the --stats table of the first pass."""
import csv, glob, json, os, sys
from collections import defaultdict

def family(name):
    for key in ('search_fast', 'emit_fast', 'value_compact', 'value_kernel', 'keys_to_partials', 'refine_targets',
                'rollout_all', 'forecast', 'search_kernel', 'emit_kernel'):
        if key in name:
            return key
    return None

def main(root):
    out = {'counters_mean_per_launch': {}, 'launches': {}, 'stats': []}
    for d in sorted(glob.glob(os.path.join(root, 'pmc*'))):
        acc, dur = defaultdict(list), defaultdict(dict)
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            per = defaultdict(float)
            for r in csv.DictReader(open(f)):
                fam = family(r['Kernel_Name'])
                if fam:
                    per[(fam, r['Counter_Name'], r['Dispatch_Id'])] += float(r['Counter_Value'])
            for (fam, cn, _), v in per.items():
                acc[(fam, cn)].append(v)
        for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                fam = family(r['Kernel_Name'])
                if fam:
                    dur[fam].setdefault('ns', []).append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        tag = os.path.basename(d)
        for (fam, cn), v in acc.items():
            v = v[len(v) // 4:]                      # skip warm-up launches
            out['counters_mean_per_launch'][f'{fam}.{cn}.{tag}'] = sum(v) / len(v)
            out['launches'][f'{fam}.{tag}'] = len(v)
        for fam, dd in dur.items():
            v = dd['ns'][len(dd['ns']) // 4:]
            out['counters_mean_per_launch'][f'{fam}.dur_ns.{tag}'] = sum(v) / len(v)
    for f in glob.glob(os.path.join(root, 'stats', '**', '*kernel_stats.csv'), recursive=True):
        out['stats'] = [r for r in csv.DictReader(open(f))][:12]
    json.dump(out, sys.stdout, indent=1)

if __name__ == '__main__':
    main(sys.argv[1])
