"""Debugging aid: the device-built read words of one soak seed built alone against the same seed built after others in one process."""
import sys
import numpy as np
for s in sys.argv[1:]:
    a = np.load('gpurun_out/dbg/alone%s.npz' % s); b = np.load('gpurun_out/dbg/loop%s.npz' % s)
    wa, wb = a['words'], b['words']
    print(s, 'words equal', np.array_equal(wa, wb), 'ustart equal', np.array_equal(a['ustart'], b['ustart']), 'loci equal', a['loci'].tobytes() == b['loci'].tobytes())
    d = np.flatnonzero(wa != wb)
    print(' diff words', len(d))
    L = a['loci']; starts = 4 * L['read_off4'].astype(np.int64)
    for i in d[:16]:
        l = np.searchsorted(starts, i, side='right') - 1
        print('  slot', i, 'locus', l, 'rank', i - starts[l], 'of', L['n_reads'][l], 'alone %08x loop %08x' % (wa[i], wb[i]))
    du = np.flatnonzero(a['ustart'] != b['ustart'])
    print(' diff ustart', len(du), du[:10])
    dl = [k for k in range(len(L)) if a['loci'][k].tobytes() != b['loci'][k].tobytes()]
    print(' diff loci', len(dl), dl[:10], [(a['loci'][k], b['loci'][k]) for k in dl[:3]])
