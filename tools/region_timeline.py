#!/usr/bin/env python3
"""Timeline of the timed region of a `bench.py --steps K --warmup W` run from a rocprofv3 --kernel-trace directory:
which kernel of which lane (HIP stream = queue) ran when, relative to the first kernel of the region, and how long the
region took next to the K steps before it (the settle phase: a full pipeline).

  usage: region_timeline.py <rocprof dir> K W [--all]
The region is found from the sentinel fills bench.py enqueues right before the warm-up steps (the last elementwise
fill kernels of the trace that precede a search kernel), not from timestamps."""
import csv, glob, os, sys

def short(n):
    for k in ('search_f64', 'emit_seg_f64', 'emit_f64', 'build_queues', 'accel_rows', 'search_fast', 'emit_fast'):
        if k in n:
            return k
    return None

def main(root, K, W, show_all=False):
    rows = []
    for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
    rows.sort()
    # searches of the first measurement only (the headline: f64 lattice); the fills are torch's elementwise kernels
    is_search = [short(n) == 'search_f64' for _, _, n, _, _ in rows]
    fills = [i for i, (_, _, n, _, _) in enumerate(rows) if 'FillFunctor' in n]
    # the first group of fills that is followed by >= W + K searches
    start = None
    for i in fills:
        if sum(is_search[i:i + 40 * (W + K)]) >= W + K and (i + 1 >= len(rows) or 'FillFunctor' not in rows[i + 1][2]):
            start = i + 1
            break
    if start is None:
        print('no region found'); return
    s_idx = [i for i in range(start, len(rows)) if is_search[i]]
    first_timed = s_idx[W]
    # the timed region begins with the accel_rows kernel ahead of the first timed search
    j = first_timed
    while j > 0 and short(rows[j - 1][2]) in ('build_queues', 'accel_rows') and rows[j - 1][0] > rows[s_idx[W - 1]][0]:
        j -= 1
    last_timed_search = s_idx[W + K - 1]
    # ... and ends with the last emit after the K-th search
    e = last_timed_search
    for i in range(last_timed_search, min(len(rows), last_timed_search + 12)):
        if short(rows[i][2]) in ('emit_f64', 'emit_seg_f64'):
            e = i
    t0 = rows[j][0]
    region = [r for r in rows[j:e + 1] if short(r[2])]
    t_end = max(r[1] for r in region)
    print(f'timed region: {K} steps in {(t_end - t0) * 1e-6:.3f} ms (kernel clock), {len(region)} kernels')
    # the K steps before the warm-up: steady state
    pre = [i for i in range(0, start) if is_search[i]]
    if len(pre) >= K + 1:
        a, b = pre[-K - 1], pre[-1]
        print(f'{K} steps of the settle phase before it: {(rows[b][0] - rows[a][0]) * 1e-6:.3f} ms (search start to search start)')
    lanes = {}
    for r in region:
        lanes.setdefault(r[3], len(lanes))
    # busy profile: fraction of the region with 0, 1, 2, ... search kernels running
    ev = []
    for r in region:
        if short(r[2]) == 'search_f64':
            ev += [(r[0], 1), (r[1], -1)]
    ev.sort()
    cur, last, hist = 0, t0, {}
    for t, d in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        cur += d; last = t
    print('time with n searches running:', ', '.join(f'{n}: {v * 1e-6:.3f} ms' for n, v in sorted(hist.items())))
    for r in region:
        k = short(r[2])
        if show_all or k in ('search_f64', 'emit_f64', 'emit_seg_f64'):
            print(f'  lane {lanes[r[3]]}  {k:13s} {(r[0] - t0) * 1e-3:8.1f} -> {(r[1] - t0) * 1e-3:8.1f} us   ({(r[1] - r[0]) * 1e-3:6.1f})')

if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), '--all' in sys.argv)
