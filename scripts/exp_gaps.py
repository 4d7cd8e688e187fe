"""per-step timeline from a rocprofv3 kernel trace: span, busy time, gaps (usage: exp_gaps.py <dir>)"""
import csv, glob, sys
fs = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))
rows = []
for f in fs:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_gray' in r['Kernel_Name']]
for a, b in zip(idx[-4:-1], idx[-3:]):
    seg = rows[a:b]
    t0 = int(seg[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in seg)
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    nxt = int(rows[b]['Start_Timestamp'])
    print('ops', len(seg), 'span %.1f us busy %.1f us gaps %.1f us; idle until next step %.1f us' % ((t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3, (nxt - t1) / 1e3))
prev = None
for r in rows[idx[-2]:idx[-1]]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print('   ', r['Kernel_Name'][:34].ljust(34), 'dur %.1f gap %.1f' % ((e - s) / 1e3, (s - prev) / 1e3 if prev else 0)); prev = e
