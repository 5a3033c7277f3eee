#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof/...) into the small summaries
committed under profiles/.

    python tools/summarize_profile.py <tag> <kernel_stats.csv> [<fetch_counter.csv> <write_counter.csv>] [<bench.log>]

Writes profiles/<tag>_kernel_stats.csv (nmsa kernels + memsets only),
profiles/<tag>_traffic.json (per-launch HBM bytes per kernel; FETCH_SIZE is
doubled for the 16-B-per-lane streaming kernels as MI355X_MICROARCH.md §HBM
prescribes for gfx950) and copies the bench JSON line.
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')

# kernels whose global loads are wide (8 / 16 B per lane) coalesced streams: FETCH_SIZE x2
# (MI355X_MICROARCH.md, HBM section).  Exact kernel names or name + '<' (templates): a bare
# prefix would also catch e.g. k_confmat_reduce (4-B loads) behind k_confmat.
WIDE_STREAM = ('k_panoptic_fused', 'k_paint', 'k_paint2', 'k_semantic_argmax', 'k_semantic_softmax',
               'k_semantic_softmax_reg', 'k_group_offsets', 'k_confmat', 'k_pq_count',
               'k_nms_rows3', 'k_ce_fwd', 'k_ce_bwd', 'k_elem_fwd', 'k_elem_bwd', 'k_vm_fwd',
               'k_vm_bwd', 'k_cos_emb_lds', 'k_ce_fused', 'k_elem_fused', 'k_vm_fused', 'k_count_u8',
               # LDS-DMA loads (`global_load_lds_dwordx4`) count like 16-B register loads
               'k_resized_tile', 'k_ce_split', 'k_multi_loss', 'k_multi_count', 'k_cos_split', 'k_cos_parts')


def is_wide(kernel):
    base = kernel.split('<')[0]
    return base in WIDE_STREAM


def short(name):
    """nmsa::k_x<1, 8, true>(args...) -> k_x<1,8,true>: template arguments stay, they tell the
    forward from the backward instantiation of one kernel template"""
    name = name.replace('void ', '')
    head = name.split('(')[0]
    if 'k_' in head or 'rocclr' in head:
        return head.split('::')[-1].replace(' ', '')
    for tok in name.replace('(', ' ').replace('<', ' ').split():
        if 'k_' in tok or 'rocclr' in tok:
            return tok.split('::')[-1]
    return name[:40]


def main():
    tag = sys.argv[1]
    stats = sys.argv[2]
    os.makedirs(PROF, exist_ok=True)
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if 'nmsa' in r['Name'] or 'rocclr' in r['Name']]
    with open(os.path.join(PROF, f'{tag}_kernel_stats.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Kernel', 'Calls', 'AverageNs', 'MinNs', 'MaxNs', 'StdDev', 'TotalDurationNs',
                    'FullName'])
        for r in keep:
            w.writerow([short(r['Name']), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'],
                        r['StdDev'], r['TotalDurationNs'], r['Name'][:160]])
    print(f'wrote profiles/{tag}_kernel_stats.csv ({len(keep)} kernels)')

    args = sys.argv[3:]
    counter_files = [a for a in args if a.endswith('.csv')]
    logs = [a for a in args if not a.endswith('.csv')]
    if counter_files:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for cf in counter_files:
            for r in csv.DictReader(open(cf)):
                if 'nmsa' in r['Kernel_Name']:
                    agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
        out = {}
        for k, d in agg.items():
            fetch_kb = sum(d['FETCH_SIZE']) / max(len(d['FETCH_SIZE']), 1) if 'FETCH_SIZE' in d else None
            write_kb = sum(d['WRITE_SIZE']) / max(len(d['WRITE_SIZE']), 1) if 'WRITE_SIZE' in d else None
            wide = is_wide(k)
            e = {'FETCH_SIZE_KB_raw': fetch_kb, 'WRITE_SIZE_KB_raw': write_kb,
                 'fetch_correction': 2.0 if wide else 1.0}
            if fetch_kb is not None and write_kb is not None:
                e['hbm_bytes_per_launch'] = (fetch_kb * e['fetch_correction'] + write_kb) * 1024
            out[k] = e
        with open(os.path.join(PROF, f'{tag}_traffic.json'), 'w') as f:
            json.dump(out, f, indent=1)
        print(f'wrote profiles/{tag}_traffic.json')
    for lg in logs:
        for line in open(lg):
            if line.startswith('{"metric"'):
                with open(os.path.join(PROF, f'{tag}_bench.json'), 'w') as f:
                    f.write(line)
                print(f'wrote profiles/{tag}_bench.json')


if __name__ == '__main__':
    main()
