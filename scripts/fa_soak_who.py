"""Debugging aid: which alignments are the reads whose words differ between a seed built alone and built after others?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import synth
from smcounter_amd.params import VcParams
seed = int(sys.argv[1])
a = np.load('gpurun_out/dbg/alone%d.npz' % seed); b = np.load('gpurun_out/dbg/loop%d.npz' % seed)
A = np.load('gpurun_out/dbg/aln%d.npz' % seed)
aln, loc, cig = A['aln'], A['loc'], A['cig']
start0 = int(A['start0'])
wa, wb = a['words'], b['words']
L = a['loci']; starts = 4 * L['read_off4'].astype(np.int64)
for i in np.flatnonzero(wa != wb)[:10]:
    l = int(np.searchsorted(starts, i, side='right') - 1); rank = int(i - starts[l])
    p = start0 + l
    w0, w1 = int(loc['w0'][l]), int(loc['w1'][l])
    idx = np.arange(w0, w1)
    cov = idx[(aln['pos'][w0:w1] <= p) & (aln['end'][w0:w1] > p)]
    order = cov[np.lexsort((cov, aln['pair_gid'][cov], aln['bc_gid'][cov]))]
    ai = int(order[rank]); r = aln[ai]
    t = l // 64
    tl0, tl1 = t * 64, min(len(loc), t * 64 + 64) - 1
    tw0, tw1 = int(loc['w0'][tl0]), int(loc['w1'][tl1])
    tidx = np.arange(tw0, tw1)
    tord = tidx[np.lexsort((tidx, aln['pair_gid'][tw0:tw1], aln['bc_gid'][tw0:tw1]))]
    j = int(np.flatnonzero(tord == ai)[0])
    ops = [(int(w) & 15, int(w) >> 4) for w in cig[int(r['cig_off']):int(r['cig_off']) + int(r['n_cig'])]]
    print('slot %d locus %d (tile %d, locus %d of it) rank %d of %d: alignment %d pos %d end %d cigar %s oflag %d mapq %d l_seq %d seq_off %d; tile list %d entries, this one at %d (batch %d lane %d); qpos %d' % (
        i, l, t, l % 64, rank, len(order), ai, int(r['pos']), int(r['end']), ops, int(r['oflag']), int(r['mapq']), int(r['l_seq']), int(r['seq_off']), len(tord), j, j // 64, j % 64, p - int(r['pos'])))
