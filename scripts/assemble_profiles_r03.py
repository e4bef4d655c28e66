"""profiles/r03_* from gpurun_out/r03final (scripts/collect_round3.sh) and gpurun_out/call_pmc_C3 (scripts/r03_call_pmc.sh): copies of
the summaries, the counter analysis of k_call_v2, and profiles/traffic.json (dev tool, build container)."""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
O, P = "gpurun_out/r03final", "profiles"
for src, dst in (("bench.json", "r03_bench_C3_200k.json"), ("bench_under_trace.json", "r03_bench_under_trace.json"),
                 ("kernel_stats.csv", "r03_bench_kernel_stats.csv"), ("kernel_trace_by_grid.txt", "r03_bench_kernel_trace_by_grid.txt"),
                 ("pmc_summary.txt", "r03_bench_pmc.txt"), ("shapes.txt", "r03_other_shapes.txt")):
    shutil.copy(os.path.join(O, src), os.path.join(P, dst))
open(P + "/r03_from_alignments_pmc.txt", "w").write(
    "# smcounter_amd.fa_leg --config C3 (200,000 loci, 4.8 M alignments, 599 M pileup reads) under rocprofv3: kernel trace by grid, then\n"
    "# PMC counters per dispatch (one --pmc pass per counter group; scripts/collect_round3.sh)\n## kernels\n" +
    open(O + "/fa_kernels.txt").read() + "## counters\n" + open(O + "/fa_pmc_summary.txt").read())
txt = ("# scripts/e2e_perf.py / scripts/bp_perf.py on the GPU box (round 3): the command-line path on synthetic BAMs, stage by stage, and\n"
       "# the device plane builder alone (HIP events around smc_build_planes, alignments resident)\n")
for n, label in (("2000", "2000 loci x 3000x, 60 reads per UMI"), ("20000", "20000 loci x 1000x, 20 reads per UMI"),
                 ("500", "500 loci x 58000x, 9 reads per UMI (the depth of the reference's example run)")):
    txt += "## " + label + "\n" + open(O + "/e2e_%s.txt" % n).read() + "# plane builder alone:\n" + open(O + "/bp_%s.txt" % n).read()
txt += "## kernels of the 20000-locus run (rocprofv3 --kernel-trace, by grid)\n" + "".join(open(O + "/e2e_kernels.txt").readlines()[:30])
txt += "## kernels of the 500 x 58000x run\n" + "".join(open(O + "/e2e_deep_kernels.txt").readlines()[:30])
open(P + "/r03_e2e_cli.txt", "w").write(txt)
c = subprocess.run([sys.executable, "scripts/pmc_summary.py"] + sorted(glob.glob("gpurun_out/call_pmc_C3/p*")), capture_output=True, text=True).stdout
c = c[c.index("void k_call_v2"):c.index("k_filter_loci")]
open(P + "/r03_call_v2_counters.txt", "w").write('''# What binds k_call_v2<64> on C3 (200,000 loci x 3000 reads, one wavefront per locus): rocprofv3 --pmc passes over
# `bench.py --steps 5 --blocks 1 --config C3` (scripts/r03_call_pmc.sh), mean per dispatch.  Units: SQ_WAVE_CYCLES / SQ_WAIT_* /
# SQ_ACTIVE_INST_* count quad-cycles summed over the chip (x 4 = cycles); SQ_BUSY_CYCLES sums 32 shader engines.
#
# Per wavefront (= per locus; / 200,000): lifetime 4.213e9 x 4 / 2e5 = 84.3 k cycles, of which
#     issuing an instruction    (ACTIVE_INST_ANY) 28.6 k  34 %
#     waiting for an issue slot (WAIT_INST_ANY)   22.9 k  27 %
#     parked in s_waitcnt       (WAIT_ANY)        32.7 k  39 %      (the three are disjoint and add up to the lifetime)
#   instructions: 4,042 VALU, 2,061 SALU, 588 branches, 214 LDS, 39 vector loads, 15 SMEM = 7,280.
# Resident: SQ_WAVE_CYCLES x 4 / (kernel 2.6e6 cycles x 256 CUs) = 25 wavefronts per CU = the 6 per SIMD the 80 VGPRs allow.
# Pipes over a wavefront's lifetime (6 wavefronts per SIMD, 24 per CU):
#     vector ALU   6 x 4,042 x 2 cycles (a wave64 instruction occupies the SIMD-32 for 2) / 84.3 k = 58 %
#                  (at 4 cycles it would be 115 %: round 2's "the vector units bind" model was wrong - MI355X_MICROARCH.md,
#                   row v_fma_f32; SQ_ACTIVE_INST_VALU counts one quad-cycle per instruction: its granularity, not the occupancy)
#     scalar ALU   24 x 2,061 / 84.3 k = 59 % of the CU's one scalar pipe; branch unit 24 x 588 / 84.3 k = 17 %
#     LDS          SQ_LDS_IDX_ACTIVE 1.04e8 / (2.6e6 x 256) = 16 %, bank conflicts 5 % of that
# Memory: 40.5 M read requests per launch, 39.2 M of them 128 B (TCC_EA0_RDREQ_128B, profiles/r03_bench_pmc.txt), 1.3 M 64 B
#     = 5.10 GB read + 0.09 GB written = 4.2 TB/s = 52 % of the 8 TB/s peak, 66 % of the 6.3 TB/s a copy reaches; L2 hit rate 10 % (a
#     streaming kernel); mean L2 read latency TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ = 747 cycles; TCC_EA0_RDREQ_DRAM_CREDIT_STALL
#     0.2 M cycles, TCC_TAG_STALL 0.25 M: the memory system does not stall the requests.
# Reading: no unit is saturated.  The kernel is a chain per locus - init, 12 scan steps with the next step's loads in flight, the
# barcode pass, the general calProb walk (dependent meta loads), rank, row - and a wavefront spends 39 % of its life waiting for the
# loads of the step it is on; six wavefronts per SIMD cover that only partly (27 % of a wavefront's cycles go to waiting for an issue
# slot: co-resident loci are in the same phase at the same time more often than not).  What would move it: more loci in flight per
# SIMD (77-79 VGPRs now; 7 waves per SIMD measured 2-5 % slower with its spills) or a deeper prefetch in the scan (two steps ahead
# measured slower in round 2).  Instruction trimming alone - round 2's lever - can buy at most the 34 %.
''' + c)
t = json.load(open(P + "/traffic.json"))
if "_round2" not in t:
    t["_round2"] = {k: t[k] for k in ("C3:200000", "C5:100000", "C2:10000")}
src = "profiles/r03_bench_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_* in separate passes, bench.py --steps 5 --blocks 1; scripts/collect_round3.sh)"
rd = 3.915e7 * 128 + 1.315e6 * 64
t["C3:200000"] = {"hbm_bytes_per_launch": rd + 8.75e4 * 1024, "fetch_size_kb": 2.528e6, "write_size_kb": 8.75e4,
                  "read_requests": {"TCC_EA0_RDREQ_sum": 4.047e7, "TCC_EA0_RDREQ_128B_sum": 3.915e7, "TCC_EA0_RDREQ_64B_sum": 1.315e6, "TCC_EA0_RDREQ_32B_sum": 0},
                  "correction": "read bytes = 128 B x RDREQ_128B + 64 B x RDREQ_64B (measured request sizes); FETCH_SIZE = RDREQ_sum x 64 B tallies the 128-B requests of 16-B-per-lane streaming loads at 64 B (MI355X_MICROARCH.md, HBM): 2 x FETCH_SIZE over-states the reads by 1.6 %; WRITE_SIZE as is",
                  "source": src, "kernel": "k_call_v2<64>", "needed_bytes_per_launch": 4934400000.0}
t["C5:100000"] = {"hbm_bytes_per_launch": 2 * 3.316e6 * 1024 + 4.378e4 * 1024, "fetch_size_kb": 3.316e6, "write_size_kb": 4.378e4,
                  "correction": "FETCH_SIZE x 2 (the request-size split was measured on C3 only: 96.7 % of its requests are 128 B), WRITE_SIZE as is",
                  "source": src, "kernel": "k_call_v2<128>", "needed_bytes_per_launch": 6484400000.0}
t["C2:10000"] = {"hbm_bytes_per_launch": 2 * 1.73e4 * 1024 + 4375 * 1024, "fetch_size_kb": 1.73e4, "write_size_kb": 4375.0,
                 "correction": "FETCH_SIZE x 2, WRITE_SIZE as is", "source": src, "kernel": "k_call_v2<64>", "needed_bytes_per_launch": 29920000.0}
t["fa:C3:200000"] = {"hbm_bytes_per_launch": 4.114e7 * 128 + 1.3e5 * 64 + 1.006e7 * 1024, "fetch_size_kb": 2.58e6, "write_size_kb": 1.006e7,
                     "read_requests": {"TCC_EA0_RDREQ_sum": 4.127e7, "TCC_EA0_RDREQ_128B_sum": 4.114e7},
                     "write_requests": {"TCC_EA0_WRREQ_sum": 2.484e8, "TCC_EA0_WRREQ_64B_sum": 7.316e7},
                     "correction": "reads by request size as above; WRITE_SIZE as is - 70 % of the write requests are 32-byte ones: a flush of the staging buffer writes 8 plane words per locus and the L2 evicts most lines before the next flush completes them (2.1 x the 4.8 GB of planes)",
                     "source": "profiles/r03_from_alignments_pmc.txt (smcounter_amd.fa_leg under rocprofv3 --pmc, separate passes)",
                     "kernel": "k_bp_emit", "needed_bytes_per_launch": 6192108532.0}
json.dump(t, open(P + "/traffic.json", "w"), indent=1)
print("ok")
