import numpy as np
d = np.loadtxt('/tmp/band_clocks.txt', dtype=np.float64)
t0, t1, nt = d[:, 0], d[:, 1], d[:, 2]
T0, T1 = t0.min(), t1.max()
span = T1 - T0
busy = (t1 - t0).sum()
print('blocks', len(d), 'kernel span (clk ticks)', span, 'sum of block times / (512 slots * span) = %.3f' % (busy / (512 * span)))
print('per-tile time (ticks): mean %.0f  p50 %.0f p90 %.0f max %.0f' % tuple([((t1 - t0) / nt).mean()] + list(np.percentile((t1 - t0) / nt, [50, 90, 100]))))
# concurrency over time
ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
ts = ev[:, 0]
for frac in (0.1, 0.25, 0.5, 0.75, 0.9, 0.95, 0.99):
    i = np.searchsorted(ts, T0 + frac * span)
    print('at %.0f%% of the span: %d blocks running' % (frac * 100, conc[min(i, len(conc) - 1)]))

ph = d[:, 3:7].sum(0)
print('phase shares: fill %.3f  stage0 %.3f  visited+carry %.3f  stages 1.. %.3f' % tuple(ph / ph.sum()))
