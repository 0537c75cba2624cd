import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%s: %.4f ms/step  %.2f M solves/s   search %.4f emit %.4f   one at a time %.2f M   tracking %.2f M' % (
    sys.argv[1], d['ms_per_step'], d['value'] / 1e6, d['kernels_ms']['search'], d['kernels_ms']['emit'],
    d.get('one_solve_in_flight', {}).get('value', 0) / 1e6, d.get('tracking_family', {}).get('value', 0) / 1e6))
