#!/usr/bin/env python3
"""Copies what tools/collect.sh <tag> left under gpurun_out/ into profiles/ under the names DESIGN.md cites.
usage: publish.py <tag> [round=r04]"""
import glob, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else 'r04'
G, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
def cp(src, dst):
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(P, dst)); print('published', dst)
    else:
        print('MISSING', src)
for k in ('bench_b4096', 'bench_b4096_driver_args', 'bench_b65536', 'bench_gt_sc1_b65536', 'bench_gt_sc3_b65536'):
    cp(f'{tag}_{k}.json', f'{rnd}_{k}.json')
for k in ('inflight_sweep', 'f64_probe', 'family_probe', 'latency', 'f32_margin', 'closed_loop', 'closed_loop_n40', 'closed_loop_breakdown',
          'closed_loop_scale', 'envelope_sweep', 'valu_microbench', 'mfma64_microbench'):
    cp(f'{tag}_{k}.txt', f'{rnd}_{k}.txt')
for k in ('f64_b4096', 'f64_track_b4096', 'f64_b65536', 'f32_b4096', 'f32_gt1_b65536', 'f64_gt1_b65536'):
    cp(f'{tag}_{k}/summary.json', f'{rnd}_pmc_{k}.json')
    st = glob.glob(os.path.join(G, f'{tag}_{k}', 'stats', '**', '*kernel_stats.csv'), recursive=True)
    if st:
        newest = max(st, key=os.path.getmtime)      # gpurun_out/ accumulates the runs of a session: rocprofv3 names its files by pid
        shutil.copy(newest, os.path.join(P, f'{rnd}_kernel_stats_{k}.csv')); print('published', f'{rnd}_kernel_stats_{k}.csv')
# the literal-mapping lines of the probe, on their own
fp = os.path.join(G, f'{tag}_f64_probe.txt')
if os.path.exists(fp):
    lines = open(fp).read().splitlines()
    keep = [l for l in lines if 'B=  4096' in l and ('dev=    0' in l or 'dev= 2048' in l) and l.startswith('f64')] + [l for l in lines if 'literal' in l]
    if keep:
        open(os.path.join(P, f'{rnd}_literal_mapping_b4096.txt'), 'w').write(
            'f64 search pass at B = 4096: production (dev=0) vs the literal north_star mapping, one wave per (scenario, candidate)\n'
            'trajectory with the horizon staged in LDS and a stage-parallel cost (dev=2048, search_literal_f64_kernel)\n' + '\n'.join(keep) + '\n')
        print(f'published {rnd}_literal_mapping_b4096.txt')
