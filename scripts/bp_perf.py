"""Time of the device plane builder alone (smc_build_planes: sort + count + scan + walk) on the e2e fixture's alignments,
HIP events around repeated calls (dev tool, GPU box).  usage: bp_perf.py [n_loci depth rpu] [all_planes]"""
import ctypes, os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import bamio, bedops, fasta, abi, _lib
from smcounter_amd.params import VcParams
from smcounter_amd.engine import Engine, DevBuf
from smcounter_amd.features import LOCUS_DTYPE

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rpu = int(sys.argv[3]) if len(sys.argv) > 3 else 20
all_planes = len(sys.argv) > 4 and sys.argv[4] == "4"
sys.argv = [sys.argv[0], str(n_loci), str(depth), str(rpu)]
import importlib.util
# the fixture writer of e2e_perf.py (its first part), without running its passes
src = open(os.path.join(ROOT, "scripts", "e2e_perf.py")).read().split("P = VcParams(")[0]
g = {"__name__": "fixture", "__file__": os.path.join(ROOT, "scripts", "e2e_perf.py")}
exec(compile(src, "e2e_fixture", "exec"), g)
bam_path, fa, bed = g["bam"], g["fa"], g["bed"]
P = VcParams(mtDepth=max(1, depth // rpu), rpb=float(rpu), hpLen=8)
ref_f = fasta.FastaFile(fa)
loci = bedops.expand_loci(bed)
eng = Engine(0)
L = eng.L
bam = bamio.NativeBam(bam_path)
chrom = loci[0][0]
lo, hi = int(loci[0][1]) - 1, int(loci[-1][1])
A = bam.alignments_run(chrom, lo, hi, 1 << 40, P, len(os.sched_getaffinity(0)), host_array=eng.pinned)
nl, ns = A["nl"], A["n_slots"]
reads = int(A["loc"]["n"].sum())
print("run: %d loci, %d alignments, %d pileup reads, %d barcodes, %d fragments" % (nl, len(A["aln"]), reads, A["n_bc"], A["n_pair"]))
up = lambda a: DevBuf(eng, a.nbytes + 64).upload(a.view(np.uint8).reshape(-1))
d_aln, d_cig, d_bq, d_loc = up(A["aln"]), up(A["cig"]), up(A["bq"]), up(A["loc"])
run_ref = ref_f.fetch(chrom, lo, hi).upper()
d_ref = up(np.frombuffer(run_ref[:nl].encode().ljust(nl, b"\0"), np.uint8).copy())
planes = [DevBuf(eng, 4 * ns)] + [DevBuf(eng, 4 * ns) if all_planes else None for k in range(4)]   # words, meta, umi, frag, dist
uaux = [DevBuf(eng, 4 * (ns + nl + 8)) for _ in range(3)]
d_loci = DevBuf(eng, nl * LOCUS_DTYPE.itemsize)
xcap = 4 * nl + 4096
d_x = DevBuf(eng, 20 * xcap); d_cnt = DevBuf(eng, 8)
loc_host = np.ascontiguousarray(A["loc"])
bi = abi.SmcBuildIn(d_aln.data_ptr(), d_cig.data_ptr(), d_bq.data_ptr(), d_loc.data_ptr(), d_ref.data_ptr(),
                    lo, nl, A["n_bc"], A["n_pair"], int(A["loc"]["n"].max()), len(A["aln"]), loc_host.ctypes.data)
cp = abi.c_params(P)
def call():
    pp = [t.data_ptr() if t is not None else None for t in planes]
    _lib.check(L.smc_build_planes(eng.ctx, ctypes.byref(cp), ctypes.byref(bi), 0, 0, pp[0], pp[1], pp[2], pp[3], pp[4], uaux[0].data_ptr(),
                                  uaux[1].data_ptr(), uaux[2].data_ptr(), d_loci.data_ptr(), d_x.data_ptr(), xcap,
                                  d_cnt.data_ptr(), ctypes.c_void_p(0)), "smc_build_planes")
for _ in range(3):
    call()
L.smc_device_sync(eng.ctx)
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
L.smc_event_create(ctypes.byref(e0)); L.smc_event_create(ctypes.byref(e1))
reps = int(os.environ.get("BP_REPS", "20"))
L.smc_event_record(e0, None)
t0 = time.perf_counter()
for _ in range(reps):
    call()
t_host = (time.perf_counter() - t0) / reps
L.smc_event_record(e1, None)
ms = ctypes.c_float()
L.smc_event_elapsed_ms(e0, e1, ctypes.byref(ms))
ms = ms.value / reps
print("smc_build_planes (%d planes): %.3f ms per run (host issue %.3f ms) -> %.1f G pileup reads/s; status %s" % (
    5 if all_planes else 1, ms, t_host * 1e3, reads / ms / 1e6, d_cnt.download(np.uint32, 2).tolist()))
