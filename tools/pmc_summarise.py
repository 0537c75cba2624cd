#!/usr/bin/env python3
"""Summarise the rocprofv3 output directories written by tools/pmc_collect.sh: per kernel family (search / emit /
value) the mean counter value and duration per launch, plus the --stats table of the first pass."""
import csv, glob, json, os, sys
from collections import defaultdict

def family(name):
    if 'emit_seg_f64' in name:          # the float64 emit in pieces (round 4) reports as the float64 emit
        return 'emit_f64'
    for key in ('search_f64', 'emit_f64', 'search_fast', 'emit_fast', 'emit_seg', 'build_queues', 'accel_rows', 'value_mfma', 'value_bound', 'value_select', 'value_compact', 'value_kernel',
                'keys_to_partials', 'refine_targets', 'rollout_all', 'forecast', 'search_kernel', 'emit_kernel'):
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
        out['stats'] = [{k: (v[:90] if k == 'Name' else v) for k, v in r.items()} for r in csv.DictReader(open(f))][:6]
    c = out['counters_mean_per_launch']
    g = lambda k: c.get(k)
    B = int(os.environ.get('IGT_PMC_BATCH', '4096'))
    dtype = os.environ.get('IGT_PMC_DTYPE', 'f64')
    SF, EF = ('search_f64', 'emit_f64') if dtype == 'f64' else ('search_fast', 'emit_fast')
    es = 8 if dtype == 'f64' else 4
    units = (4 if dtype == 'f64' else 2) * B          # C = 256: 64-candidate units (f64) or 128-candidate units (f32)
    out['batch'] = B
    out['dtype'] = dtype
    out['search_kernel'] = SF
    # HBM bytes per launch, MI355X_MICROARCH.md HBM section: (FETCH_SIZE + WRITE_SIZE) * 1024, separate passes.  The
    # guide's x2 FETCH_SIZE correction is for 16 B/lane coalesced vector streams; the search kernel reads its inputs
    # with scalar (SMEM) loads, so the raw counter is reported.
    for fam, key in ((SF, 'search'), (EF, 'emit')):
        f, w = g(f'{fam}.FETCH_SIZE.pmc1'), g(f'{fam}.WRITE_SIZE.pmc2')
        if f is not None and w is not None:
            out[f'{key}_kernel_hbm_bytes_per_launch'] = (f + w) * 1024.0
    rd = (7 + 2 + 3 + 42) * es + 4                      # x0, u_prev, kparams, obs (x,y)[21], flags
    out['algorithmic_bytes_per_launch'] = {'search': (rd + 12) * B, 'emit': (rd + (147 + 40) * es + es + 8) * B}
    d = {}
    if g(f'{SF}.GRBM_GUI_ACTIVE.pmc4') and g(f'{SF}.dur_ns.pmc4'):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        d['shader_clock_GHz_during_search'] = g(f'{SF}.GRBM_GUI_ACTIVE.pmc4') / 8.0 / g(f'{SF}.dur_ns.pmc4')
    if g(f'{SF}.SQ_WAVES.pmc3'):
        d['waves_per_launch'] = g(f'{SF}.SQ_WAVES.pmc3')
        d['valu_instructions_per_launch'] = g(f'{SF}.SQ_INSTS_VALU.pmc3')
        d['valu_instructions_per_unit'] = g(f'{SF}.SQ_INSTS_VALU.pmc3') / units
        d['salu_instructions_per_launch'] = g(f'{SF}.SQ_INSTS_SALU.pmc3')
        clk = d.get('shader_clock_GHz_during_search', 2.23)
        simd_cycles = 1024.0 * g(f'{SF}.dur_ns.pmc3') * clk
        d['simd_valu_busy_fraction'] = 4.0 * g(f'{SF}.SQ_ACTIVE_INST_VALU.pmc3') / simd_cycles     # SQ_* are quad-cycles
        # cycles the vector ALU is busy per VALU wave-instruction, from the counters alone (a v_fma_f64 holds it longer
        # than the 4 cycles of a 32-bit operation; tools/valu_microbench.hip has the per-opcode rates)
        d['busy_cycles_per_valu_instruction'] = 4.0 * g(f'{SF}.SQ_ACTIVE_INST_VALU.pmc3') / g(f'{SF}.SQ_INSTS_VALU.pmc3')
        d['search_kernel_ms'] = g(f'{SF}.dur_ns.pmc3') * 1e-6
        # instruction-based issue fraction: VALU wave-instructions x 4 issue cycles / (SIMDs x kernel cycles)
        d['valu_issue_fraction_4_cycles_per_instruction'] = 4.0 * g(f'{SF}.SQ_INSTS_VALU.pmc3') / simd_cycles
        d['mean_waves_resident_per_simd'] = 4.0 * g(f'{SF}.SQ_WAVE_CYCLES.pmc3') / simd_cycles
        wc = g(f'{SF}.SQ_WAVE_CYCLES.pmc3')
        d['wave_cycles_split'] = {'active': g(f'{SF}.SQ_ACTIVE_INST_ANY.pmc3') / wc,
                                  'issue_stall': g(f'{SF}.SQ_WAIT_INST_ANY.pmc3') / wc}
    mb = os.environ.get('IGT_PMC_MICROBENCH')
    if mb and os.path.exists(mb):      # issue rates measured on the same box, 2 waves per SIMD (the search kernels' residency)
        import re
        rates = {}
        for l in open(mb):
            m = re.match(r'^(.+?)\s+waves/SIMD=2\s+[\d.]+ ms\s+[\d.]+ wave-instr/cycle/SIMD @2.4GHz\s+\(([\d.]+) cycles per wave-instr\)', l)
            if m:
                rates[m.group(1).strip()] = float(m.group(2))
        if rates:
            d['microbenchmark_cycles_per_instruction'] = dict(rates, note='tools/valu_microbench.hip at 2 waves per SIMD, cycles '
                                                              'counted at an assumed 2.4 GHz')
    out['derived'] = d
    # the kernel sources this profile belongs to (bench.py quotes PMC figures only at a matching hash)
    import hashlib
    hh = hashlib.sha256()
    cs = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'igt-mpc-int_amd', 'csrc')
    for fn in sorted(os.listdir(cs)):
        if fn.endswith(('.hip', '.h', '.inc')):
            with open(os.path.join(cs, fn), 'rb') as f:
                hh.update(fn.encode() + b'\0' + f.read())
    out['source_hash'] = hh.hexdigest()[:16]
    json.dump(out, sys.stdout, indent=1)

if __name__ == '__main__':
    main(sys.argv[1])
