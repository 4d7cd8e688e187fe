"""Gaps of a kernel trace (rocprofv3 --kernel-trace --output-format csv): where the GPU idles between launches of a steady-state loop.
usage: python3 scripts/trace_gaps.py <dir with *_kernel_trace.csv> [skip_fraction]"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("nvca::", "")[:28]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
gaps = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps[(last_name if 'last_name' in dir() else '?', n)][0] += 1
        gaps[(last_name if 'last_name' in dir() else '?', n)][1] += s - cur_e
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = n
busy += cur_e - cur_s
print("window %.2f ms, %d launches, busy %.2f ms (%.0f %%)" % ((t1 - t0) / 1e6, len(rows), busy / 1e6, 100.0 * busy / (t1 - t0)))
print("gaps (kernel that ended last -> kernel that starts), count, total ms, mean us:")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-28s -> %-28s %5d  %7.2f  %6.1f" % (a, b, c, t / 1e6, t / c / 1e3))
dur = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    dur[n][0] += 1; dur[n][1] += e - s
print("kernels: name, calls, total ms, mean us")
for n, (c, t) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-28s %5d  %7.2f  %6.1f" % (n, c, t / 1e6, t / c / 1e3))
