"""Reads a rocprofv3 kernel trace of bench.py (…_kernel_trace.csv) and prints, for the LAST profiled step, the span and the
busy time of its phases: forward, decoder backward (up to the last gate-derivative launch), the rest of backward (condition
embed / encoder: small latent-rate launches), optimizer.  usage: python tools/phase_trace.py <kernel_trace.csv>"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if 'adam_kernel' in k[2]]
assert len(adam) >= 2
a0, a1 = adam[-2], adam[-1]
step = ks[a0 + 1:a1 + 1]
def idx(pred, last=False):
    c = [i for i, k in enumerate(step) if pred(k[2])]
    return (c[-1] if last else c[0]) if c else None
i_xent = idx(lambda n: 'xent_bwd' in n)
i_gbwd_last = idx(lambda n: 'conv_gemm_x3_kernel<2,' in n, last=True)
def span(lo, hi, name):
    seg = step[lo:hi]
    if not seg:
        return
    busy = sum(e - s for s, e, _ in seg)
    wall = seg[-1][1] - seg[0][0]
    print('%-46s launches %4d  span %7.3f ms  kernel time %7.3f ms  idle %6.3f ms' % (name, len(seg), wall / 1e6, busy / 1e6, (wall - busy) / 1e6))
print('step: %d launches, %.3f ms from the first launch to the end of adam' % (len(step), (step[-1][1] - step[0][0]) / 1e6))
span(0, i_xent, 'forward (to the loss backward)')
span(i_xent, i_gbwd_last + 1, 'decoder backward (to the last gate-derivative)')
span(i_gbwd_last + 1, len(step) - 1, 'rest of backward (condition embed, encoder, VQ)')
span(len(step) - 1, len(step), 'adam')
import collections
c = collections.Counter(); n = collections.Counter()
for s, e, name in step[i_gbwd_last + 1:len(step) - 1]:
    key = name.split('(')[0][:70]
    c[key] += e - s; n[key] += 1
print('rest of backward, by kernel:')
for k, v in c.most_common(14):
    print('  %-72s x%3d  %7.3f ms' % (k, n[k], v / 1e6))
print('gaps > 4 us in the step:')
for (s0, e0, n0), (s1, e1, n1) in zip(step[:-1], step[1:]):
    if s1 - e0 > 4000:
        print('  %6.1f us  after %-50s before %s' % ((s1 - e0) / 1e3, n0.split('(')[0][-50:], n1.split('(')[0][-50:]))
