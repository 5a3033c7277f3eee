#!/usr/bin/env python3
"""Timeline of the headline step out of a rocprofv3 kernel trace:
   (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o kt -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline)
   python tools/timeline_bench.py OUT/**/kt_kernel_trace.csv
prints, for the last 20 launches of the fused kernel: when each kernel of the step ran relative to
the fused kernel's start, which kernels ran concurrently, and how much of the span no streaming
kernel covered."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    name = r['Kernel_Name']
    m = __import__('re').search(r'(k_[a-z0-9_]+)', name)
    short = m.group(1) if m else name.split('(')[0][-40:]
    ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short, str(r.get('Queue_Id', '?'))))
ks.sort()
# the timed region = the last 20 fused launches that start before the last metric kernel ends
# (the isolated pass behind it runs the pipeline without the metric updates)
last_metric = max(k[1] for k in ks if k[2] == 'k_pq_accumulate')
idx = [i for i, k in enumerate(ks) if k[2] == 'k_panoptic_fused' and k[0] < last_metric]
timed = idx[-20:]
t_first, t_last = ks[timed[0]][0], ks[timed[-1]][1]
span = [k for k in ks if k[0] >= t_first - 30000 and k[0] <= last_metric]
t0 = span[0][0]
span_us = (ks[timed[-1]][1] - t_first) / 1e3
print(f'{len(timed)} timed fused launches, first start .. last end {span_us:.1f} us = {span_us / len(timed):.1f} us per step; '
      f'at 198 algorithmic B/px x 32 x 640 x 480 px per step that is {198 * 32 * 640 * 480 / (span_us / len(timed)) / 1e6:.2f} TB/s; '
      f'first pipeline kernel .. last metric kernel {(last_metric - min(k[0] for k in span)) / 1e3:.1f} us')
by = {}
for s, e, n, q in span:
    by.setdefault(n, []).append((e - s) / 1e3)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f'  {n:28s} x{len(v):3d}  mean {sum(v) / len(v):7.1f} us  total {sum(v):8.1f}')
# fused-to-fused gaps
fs = [ks[i] for i in timed]
gaps = [(fs[i + 1][0] - fs[i][1]) / 1e3 for i in range(len(fs) - 1)]
print('  gap between consecutive fused kernels (us):', ' '.join(f'{g:.0f}' for g in gaps))
print('  fused durations (us):', ' '.join(f'{(e - s) / 1e3:.0f}' for s, e, _, _ in fs))
if '-v' in sys.argv:
    for s, e, n, q in span:
        print(f'  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  q{q:>3s}  {n}')
