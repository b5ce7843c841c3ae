#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc passes (one counter set per pass, csv output) into a per-kernel summary.

  python tools/pmc_summary.py gpurun_out/pmc profiles/r01_pmc_dccrn --steps 3

Reads <dir>/<PASS>/p_counter_collection.csv for PASS in FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES and writes
<out>.json / <out>.md.  Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE reports half of the bytes of coalesced reads - doubled
here (calibrated on se::add_kernel in this repo: 2 x 519 MB read -> FETCH_SIZE 506 873, 519 MB written -> WRITE_SIZE
506 864).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs).
"""
import collections
import csv
import json
import os
import sys


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        d[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return d


def main():
    src, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 1
    commit = sys.argv[sys.argv.index('--commit') + 1] if '--commit' in sys.argv else None      # HEAD of the profiled tree
    fe = load(os.path.join(src, 'FETCH_SIZE', 'p_counter_collection.csv'))
    wr = load(os.path.join(src, 'WRITE_SIZE', 'p_counter_collection.csv'))
    sq = load(os.path.join(src, 'SQ_VALU_MFMA_BUSY_CYCLES', 'p_counter_collection.csv'))
    rows = []
    for k in sorted(set(fe) | set(wr) | set(sq)):
        if not (k.startswith('void se::') or k.startswith('se::')):
            continue
        f = fe.get(k, {}).get('FETCH_SIZE', [])
        w = wr.get(k, {}).get('WRITE_SIZE', [])
        s = sq.get(k, {})
        n = max(len(f), len(w), len(s.get('GRBM_GUI_ACTIVE', [])))
        rd = 2.0 * 1024 * sum(f)
        wb = 1024.0 * sum(w)
        busy, act = sum(s.get('SQ_VALU_MFMA_BUSY_CYCLES', [])), sum(s.get('GRBM_GUI_ACTIVE', []))
        mops = sum(s.get('SQ_INSTS_VALU_MFMA_MOPS_F32', []))
        rows.append({'kernel': k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0], 'launches': n,
                     'read_GB_per_launch': round(rd / max(len(f), 1) / 1e9, 4),
                     'write_GB_per_launch': round(wb / max(len(w), 1) / 1e9, 4),
                     'read_GB_per_step': round(rd / steps / 1e9, 3), 'write_GB_per_step': round(wb / steps / 1e9, 3),
                     'mfma_util': round(busy / (act / 8 * 256 * 4), 4) if act else None,
                     'mfma_tflop_per_step': round(mops * 512 / steps / 1e12, 3)})
    rows.sort(key=lambda r: -(r['read_GB_per_step'] + r['write_GB_per_step']))
    # everything gc_launch dispatches + the sum / combine passes of the three-product complex layers (the profiler in the
    # engine - and bench.py's roofline.launches_per_step - counts those as launches of the same family)
    gc = [r for r in rows if 'gc_kernel' in r['kernel'] or 'gc_small_' in r['kernel'] or 'gauss_' in r['kernel']]
    mf = [r for r in rows if r['mfma_tflop_per_step']]
    fam = {'launches_per_step': sum(r['launches'] for r in gc) / steps,
           'read_GB_per_step': round(sum(r['read_GB_per_step'] for r in gc), 3),
           'write_GB_per_step': round(sum(r['write_GB_per_step'] for r in gc), 3)}
    fam['traffic_GB_per_launch'] = round((fam['read_GB_per_step'] + fam['write_GB_per_step']) / fam['launches_per_step'], 4)
    fam['executed_mfma_tflop_per_step'] = round(sum(r['mfma_tflop_per_step'] for r in mf), 3)   # SQ_INSTS_VALU_MFMA_MOPS_F32 x 512
    fam['commit'] = commit
    # hash of the kernel sources the DCCRN decode runs through, as bench.py computes it: the bench quotes these counters only
    # while the tree's sources still hash to the same value (else `traffic` is null and `traffic_stale` true)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        fam['csrc_sha16'] = bench.dccrn_csrc_sha16()
    except Exception as ex:       # (summaries of other models do not need it)
        fam['csrc_sha16'] = None
        print('csrc hash unavailable:', ex, file=sys.stderr)
    res = {'steps': steps, 'gc_family': fam, 'kernels': rows}
    json.dump(res, open(out + '.json', 'w'), indent=1)
    with open(out + '.md', 'w') as f:
        f.write('# rocprofv3 PMC summary (separate --pmc passes; FETCH_SIZE doubled per the gfx950 correction)\n\n')
        f.write('| kernel | launches | read GB/launch | write GB/launch | read GB/step | write GB/step | MFMA util | MFMA TFLOP/step |\n|---|---|---|---|---|---|---|---|\n')
        for r in rows:
            f.write('| {kernel} | {launches} | {read_GB_per_launch} | {write_GB_per_launch} | {read_GB_per_step} | {write_GB_per_step} | {mfma_util} | {mfma_tflop_per_step} |\n'.format(**r))
        f.write('\ngc_kernel family: %s\n' % json.dumps(fam))
    print(json.dumps(fam))


if __name__ == '__main__':
    main()
