"""profiles/r03_* from gpurun_out/<tag> (scripts/collect_round3.sh <tag>) and gpurun_out/call_pmc_C3 (scripts/r03_call_pmc.sh): copies
of the summaries, the counter analysis of k_call_v2 (numbers computed from the passes, not transcribed), and profiles/traffic.json
(dev tool, build container).  usage: assemble_profiles_r03.py [tag]"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
O, P = "gpurun_out/" + (sys.argv[1] if len(sys.argv) > 1 else "r03final"), "profiles"


def parse(path):
    """pmc_summary.py text -> {section: {kernel: {counter: mean}}} ('' = no '== section' lines)"""
    out, sec, ker = {"": {}}, "", None
    for line in open(path):
        if line.startswith("== "):
            sec = line[3:].strip(); out.setdefault(sec, {}); continue
        m = re.match(r"\s+(\S+)\s+mean (\S+) over", line)
        if m and ker is not None:
            out[sec][ker][m.group(1)] = float(m.group(2))
        elif line.strip() and not line.startswith(" "):
            ker = line.strip(); out[sec].setdefault(ker, {})
    return out


for src, dst in (("bench.json", "r03_bench_C3_200k.json"), ("bench_under_trace.json", "r03_bench_under_trace.json"),
                 ("kernel_stats.csv", "r03_bench_kernel_stats.csv"), ("kernel_trace_by_grid.txt", "r03_bench_kernel_trace_by_grid.txt"),
                 ("pmc_summary.txt", "r03_bench_pmc.txt"), ("shapes.txt", "r03_other_shapes.txt")):
    shutil.copy(os.path.join(O, src), os.path.join(P, dst))
bench = json.load(open(O + "/bench.json"))
fa = bench["from_alignments"]
open(P + "/r03_from_alignments_pmc.txt", "w").write(
    "# smcounter_amd.fa_leg --config C3 (%s) under rocprofv3: kernel trace by grid, then\n"
    "# PMC counters per dispatch (one --pmc pass per counter group; scripts/collect_round3.sh)\n## kernels\n" % fa["workload"].split(";")[0] +
    open(O + "/fa_kernels.txt").read() + "## counters\n" + open(O + "/fa_pmc_summary.txt").read())
txt = ("# scripts/e2e_perf.py / scripts/bp_perf.py on the GPU box (round 3): the command-line path on synthetic BAMs, stage by stage, and\n"
       "# the device plane builder alone (HIP events around smc_build_planes, alignments resident)\n")
for n, label in (("2000", "2000 loci x 3000x, 60 reads per UMI"), ("20000", "20000 loci x 1000x, 20 reads per UMI"),
                 ("500", "500 loci x 58000x, 9 reads per UMI (the depth of the reference's example run)")):
    txt += "## " + label + "\n" + open(O + "/e2e_%s.txt" % n).read() + "# plane builder alone:\n" + open(O + "/bp_%s.txt" % n).read()
txt += "## kernels of the 20000-locus run (rocprofv3 --kernel-trace, by grid)\n" + "".join(open(O + "/e2e_kernels.txt").readlines()[:30])
txt += "## kernels of the 500 x 58000x run\n" + "".join(open(O + "/e2e_deep_kernels.txt").readlines()[:30])
open(P + "/r03_e2e_cli.txt", "w").write(txt)

# ---- k_call_v2: what the counters say
cp_txt = open("gpurun_out/call_pmc_C3/pmc_summary.txt").read()
c = parse("gpurun_out/call_pmc_C3/pmc_summary.txt")[""]
K = c[[k for k in c if k.startswith("void k_call_v2")][0]]
NL = K["SQ_WAVES"]                                   # one wavefront per locus
life = 4 * K["SQ_WAVE_CYCLES"] / NL
act, wis, wany = (4 * K[x] / NL for x in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"))
valu, salu, br, lds, vrd, smem = (K[x] / NL for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"))
kcyc = K["SQ_BUSY_CYCLES"] / 32                      # (summed over 32 shader engines)
wpc = 4 * K["SQ_WAVE_CYCLES"] / (kcyc * 256)
b = parse(O + "/pmc_summary.txt")
R = b["C3 request sizes"][[k for k in b["C3 request sizes"] if k.startswith("void k_call_v2")][0]]
rd = 128 * R["TCC_EA0_RDREQ_128B_sum"] + 64 * R["TCC_EA0_RDREQ_64B_sum"] + 32 * R.get("TCC_EA0_RDREQ_32B_sum", 0)
wr = K["WRITE_SIZE"] * 1024
kms = bench["roofline"]["kernel_ms"]
body = cp_txt[cp_txt.index("void k_call_v2"):cp_txt.index("k_filter_loci")]
open(P + "/r03_call_v2_counters.txt", "w").write('''# What binds k_call_v2<64> on C3 (200,000 loci x 3000 reads, one wavefront per locus, one uint32 per read): rocprofv3 --pmc passes
# over `bench.py --steps 5 --blocks 1 --config C3` (scripts/r03_call_pmc.sh), mean per dispatch; the numbers below are computed from
# the passes by scripts/assemble_profiles_r03.py.  Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
# the chip (x 4 = cycles); SQ_BUSY_CYCLES sums 32 shader engines.
#
# Per wavefront (= per locus): lifetime %.1f k cycles, of which
#     issuing an instruction    (ACTIVE_INST_ANY) %.1f k  %.0f %%
#     waiting for an issue slot (WAIT_INST_ANY)   %.1f k  %.0f %%
#     parked in s_waitcnt       (WAIT_ANY)        %.1f k  %.0f %%      (the three are disjoint and add up to the lifetime)
#   instructions: %.0f VALU, %.0f SALU, %.0f branches, %.0f LDS, %.0f vector loads, %.0f SMEM = %.0f (7,280 with two plane words per read).
# Resident: SQ_WAVE_CYCLES x 4 / (kernel %.2e cycles x 256 CUs) = %.1f wavefronts per CU.
# Pipes over a wavefront's lifetime (%.0f wavefronts per CU):
#     vector ALU   %.1f x %.0f x 2 cycles (a wave64 instruction occupies the SIMD-32 for 2) / 4 SIMDs / %.1f k = %.0f %%
#     scalar ALU   %.1f x %.0f / %.1f k = %.0f %% of the CU's one scalar pipe; branch unit %.0f %%
#     LDS          SQ_LDS_IDX_ACTIVE %.3g / (kernel cycles x 256) = %.0f %%
# Memory: %.2f M read requests per launch, %.2f M of them 128 B (TCC_EA0_RDREQ_128B, profiles/r03_bench_pmc.txt)
#     = %.2f GB read + %.2f GB written in %.3f ms = %.1f TB/s = %.0f %% of the 8 TB/s peak (the words: 2.40 GB); L2 hit rate %.0f %% (a streaming
#     kernel); mean L2 read latency TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ = %.0f cycles; TCC_EA0_RDREQ_DRAM_CREDIT_STALL %.2g,
#     TCC_TAG_STALL %.2g cycles: the memory system does not stall the requests.
# Reading: no unit is saturated, and halving the bytes per read (8 -> 4) bought 11 %% of the time, not 50.  By phase (ablation builds,
# scripts/ab_variants.py, same process): the scan of the reads 0.53 ms - VALU-bound now: ~ 190 instructions per 252 reads, 4.5 TB/s,
# asking for the reads two steps ahead instead of one changes nothing -, the per-barcode passes 0.16 ms, the general calProb walk of
# the ~ 15 %% of barcodes with a second allele 0.30 ms, ranking + row + filter 0.16 ms.  The kernel is a chain per locus and six
# wavefronts per SIMD hide only part of each other's waits.
''' % (life / 1e3, act / 1e3, 100 * act / life, wis / 1e3, 100 * wis / life, wany / 1e3, 100 * wany / life,
       valu, salu, br, lds, vrd, smem, valu + salu + br + lds + vrd + smem, kcyc, wpc, wpc,
       wpc, valu, life / 1e3, 100 * wpc * valu * 2 / 4 / life, wpc, salu, life / 1e3, 100 * wpc * salu / life, 100 * wpc * br / life,
       K["SQ_LDS_IDX_ACTIVE"], 100 * K["SQ_LDS_IDX_ACTIVE"] / (kcyc * 256),
       R["TCC_EA0_RDREQ_sum"] / 1e6, R["TCC_EA0_RDREQ_128B_sum"] / 1e6, rd / 1e9, wr / 1e9, kms, (rd + wr) / kms / 1e9, 100 * (rd + wr) / kms / 1e9 / 8,
       100 * K["TCC_HIT_sum"] / K["TCC_REQ_sum"], K["TCP_TCC_READ_REQ_LATENCY_sum"] / K["TCP_TCC_READ_REQ_sum"],
       K["TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"], K["TCC_TAG_STALL_sum"]) + body)

# ---- traffic.json
t = json.load(open(P + "/traffic.json"))
if "_round3a" not in t:          # the kernels of the first half of round 3 (two plane words per read)
    t["_round3a"] = {k: t[k] for k in ("C3:200000", "C5:100000", "C2:10000", "fa:C3:200000") if k in t}
src = "profiles/r03_bench_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_* in separate passes, bench.py --steps 5 --blocks 1; scripts/collect_round3.sh)"
F = lambda sec: b[sec][[k for k in b[sec] if k.startswith("void k_call_v2")][0]]
c3 = F("C3")
per_fetch_kb = rd / c3["FETCH_SIZE"]               # bytes a FETCH_SIZE unit really stands for on this kernel's request mix
need = {"C3": bench["roofline"]["needed_bytes_per_launch"]}
for k in ("C5", "C2"):
    need[k] = bench["other_configs"][k]["roofline"]["needed_bytes_per_launch"]
t["C3:200000"] = {"hbm_bytes_per_launch": rd + c3["WRITE_SIZE"] * 1024, "fetch_size_kb": c3["FETCH_SIZE"], "write_size_kb": c3["WRITE_SIZE"],
                  "read_requests": {k: R[k] for k in R if k.startswith("TCC_EA0_RDREQ")},
                  "correction": "read bytes = 128 B x RDREQ_128B + 64 B x RDREQ_64B (measured request sizes); FETCH_SIZE = RDREQ_sum x 64 B tallies the 128-B requests of 16-B-per-lane streaming loads at 64 B (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is",
                  "source": src, "kernel": "k_call_v2<64>", "needed_bytes_per_launch": need["C3"]}
for key, sec, kern in (("C5:100000", "C5", "k_call_v2<128>"), ("C2:10000", "C2", "k_call_v2<64>")):
    f = F(sec)
    t[key] = {"hbm_bytes_per_launch": per_fetch_kb * f["FETCH_SIZE"] + f["WRITE_SIZE"] * 1024, "fetch_size_kb": f["FETCH_SIZE"],
              "write_size_kb": f["WRITE_SIZE"],
              "correction": "FETCH_SIZE x %.0f B per unit (the request-size split measured on C3), WRITE_SIZE as is" % per_fetch_kb,
              "source": src, "kernel": kern, "needed_bytes_per_launch": need[sec]}
f = parse(O + "/fa_pmc_summary.txt")[""]
E = f[[k for k in f if "k_bp_emit" in k][0]]
frd = 128 * E["TCC_EA0_RDREQ_128B_sum"] + 64 * (E["TCC_EA0_RDREQ_sum"] - E["TCC_EA0_RDREQ_128B_sum"])
t["fa:C3:200000"] = {"hbm_bytes_per_launch": frd + E["WRITE_SIZE"] * 1024, "fetch_size_kb": E["FETCH_SIZE"], "write_size_kb": E["WRITE_SIZE"],
                     "read_requests": {k: E[k] for k in E if k.startswith("TCC_EA0_RDREQ")},
                     "write_requests": {k: E[k] for k in E if k.startswith("TCC_EA0_WRREQ")},
                     "correction": "reads by request size as above (a window of ~ 43 bases and qualities costs two to four 128-byte lines: the reads are 3-4 x the 1.4 GB needed); WRITE_SIZE as is: %.0f %% of the write requests are 32-byte ones (a flush writes up to 16 words per locus; the L2 evicts most lines before the next flush completes them)" % (100 * (1 - E["TCC_EA0_WRREQ_64B_sum"] / E["TCC_EA0_WRREQ_sum"])),
                     "source": "profiles/r03_from_alignments_pmc.txt (smcounter_amd.fa_leg under rocprofv3 --pmc, separate passes)",
                     "kernel": "k_bp_emit", "needed_bytes_per_launch": fa["roofline"]["needed_bytes_per_launch"]}
json.dump(t, open(P + "/traffic.json", "w"), indent=1)
print("C3 k_call_v2: %.3f GB read + %.3f GB written (needed %.3f); emit: %.2f GB read + %.2f GB written (needed %.2f)" % (
    rd / 1e9, wr / 1e9, need["C3"] / 1e9, frd / 1e9, E["WRITE_SIZE"] * 1024 / 1e9, fa["roofline"]["needed_bytes_per_launch"] / 1e9))
