"""profiles/r05_* from gpurun_out/<tag> (scripts/collect_round5.sh <tag>): copies of the summaries, the traffic records of the walk
and of the locus kernel in profiles/traffic.json (computed from the passes, each tied to the hash of the library it was measured
on) and a short reading of the counters (dev tool, build container).  usage: assemble_profiles_r05.py [tag]"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
O, P = "gpurun_out/" + (sys.argv[1] if len(sys.argv) > 1 else "r05final"), "profiles"


def parse(path):
    """pmc_summary.py text -> {kernel: {counter: mean}}"""
    out, ker = {}, None
    for line in open(path):
        m = re.match(r"\s+(\S+)\s+mean (\S+) over", line)
        if m and ker is not None:
            out[ker][m.group(1)] = float(m.group(2))
        elif line.strip() and not line.startswith(" "):
            ker = line.strip(); out.setdefault(ker, {})
    return out


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


for src, dst in (("bench.json", "r05_bench_C3_200k.json"), ("bench_under_trace.json", "r05_bench_under_trace.json"),
                 ("kernel_stats.csv", "r05_bench_kernel_stats.csv"), ("kernel_trace_by_grid.txt", "r05_bench_kernel_trace_by_grid.txt"),
                 ("shapes.txt", "r05_other_shapes.txt"), ("bench_2ranks_functional.json", "r05_bench_2ranks_one_gpu_functional.json"),
                 ("bench_C4_strong_n1.json", "r05_bench_C4_strong_n1.json")):
    if os.path.exists(os.path.join(O, src)) and os.path.getsize(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
bench = last_json(O + "/bench.json")
fa_tr = last_json(O + "/fa_under_trace.json")
lib = open(O + "/lib_sha16.txt").read().strip()
f = parse(O + "/fa_pmc_summary.txt")


def reading(C, need, kms, what):
    rd = 128 * C["TCC_EA0_RDREQ_128B_sum"] + 64 * C["TCC_EA0_RDREQ_64B_sum"] + 32 * C.get("TCC_EA0_RDREQ_32B_sum", 0)
    wr = C["WRITE_SIZE"] * 1024
    nw = C["SQ_WAVES"]
    life = 4 * C["SQ_WAVE_CYCLES"] / nw
    act, wis, wany = (4 * C[x] / nw for x in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"))
    txt = ("""# %s, per launch:
#   reads   %.2f M requests of 128 B + %.2f M of 64 B = %.2f GB   (FETCH_SIZE %.3g KB: counts every request at 64 B)
#   writes  WRITE_SIZE %.2f GB; %.2f M write requests, %.0f %% of them 64-byte ones
#   total   %.2f GB against %.2f GB needed = %.2f x; in %.3f ms (HIP events of the bench run) = %.2f TB/s
#   L2      %.0f %% of %.1f M requests hit
#   a wavefront (%d of them): %.1f k cycles resident, %.0f %% issuing, %.0f %% waiting for an issue slot, %.0f %% parked in s_waitcnt;
#   %.0f VALU + %.0f SALU + %.0f LDS + %.0f vector loads + %.0f vector stores per wavefront
""" % (what, C["TCC_EA0_RDREQ_128B_sum"] / 1e6, C["TCC_EA0_RDREQ_64B_sum"] / 1e6, rd / 1e9, C["FETCH_SIZE"],
       wr / 1e9, C["TCC_EA0_WRREQ_sum"] / 1e6, 100 * C["TCC_EA0_WRREQ_64B_sum"] / C["TCC_EA0_WRREQ_sum"],
       (rd + wr) / 1e9, need / 1e9, (rd + wr) / need, kms, (rd + wr) / kms / 1e9,
       100 * C["TCC_HIT_sum"] / (C["TCC_HIT_sum"] + C["TCC_MISS_sum"]), (C["TCC_HIT_sum"] + C["TCC_MISS_sum"]) / 1e6,
       int(nw), life / 1e3, 100 * act / life, 100 * wis / life, 100 * wany / life,
       C["SQ_INSTS_VALU"] / nw, C["SQ_INSTS_SALU"] / nw, C["SQ_INSTS_LDS"] / nw, C["SQ_INSTS_VMEM_RD"] / nw, C["SQ_INSTS_VMEM_WR"] / nw))
    return txt, rd, wr


E = f[[k for k in f if "k_bp_emit2" in k][0]]
need = bench["roofline"]["needed_bytes_per_launch"]
kms = bench["roofline"]["kernel_ms"]
head = ("# The step bench.py times (smc_build_planes -> smc_plan_create_dev -> smc_plan_run_words) on the C3-shaped run, `python3 -m bench_fa\n"
        "# --config C3 --slots 1` under rocprofv3 (scripts/collect_round5.sh): kernels by grid, the timeline of one step, PMC counters per\n"
        "# dispatch (one --pmc pass per counter group).  Library %s.  The reading below is computed from the passes by\n"
        "# scripts/assemble_profiles_r05.py.\n#\n" % lib)
t_e, rd, wr = reading(E, need, kms, "k_bp_emit2 (the walk that writes the read words; one batch of 64 alignments per wavefront since round 5)")
head += t_e
t = json.load(open(P + "/traffic.json"))
if "_round4" not in t and "fa:C3:200000" in t:
    t["_round4"] = {"fa:C3:200000": t["fa:C3:200000"], "C3:200000": t.get("C3:200000")}
t["fa:C3:200000"] = {"hbm_bytes_per_launch": rd + wr, "fetch_size_kb": E["FETCH_SIZE"], "write_size_kb": E["WRITE_SIZE"],
                     "read_requests": {k: E[k] for k in E if k.startswith("TCC_EA0_RDREQ")},
                     "write_requests": {k: E[k] for k in E if k.startswith("TCC_EA0_WRREQ")},
                     "correction": "read bytes = 128 B x RDREQ_128B + 64 B x RDREQ_64B (+ 32 B x RDREQ_32B): the measured request sizes (FETCH_SIZE tallies "
                                   "every request at 64 B: MI355X_MICROARCH.md, HBM); WRITE_SIZE as is",
                     "source": "profiles/r05_from_alignments_pmc.txt (python3 -m bench_fa under rocprofv3 --pmc, one pass per counter group; scripts/collect_round5.sh)",
                     "kernel": "k_bp_emit2", "needed_bytes_per_launch": need, "lib_sha16": lib}
Ck = [k for k in f if k.startswith("void k_call_v2<64")]
if Ck:
    C = f[Ck[0]]
    cms = bench["step_breakdown"]["k_call_v2_ms"]
    bits = bench["config"].get("read_word_bits", 32)
    reads = bench["roofline"]["reads_per_s_kernel"] * bench["roofline"]["kernel_ms"] * 1e-3
    # (consumer_only's figure is for 32-bit words: the step's own locus kernel reads 16-bit ones when the run is built that way)
    cneed = bench["consumer_only"]["roofline"]["needed_bytes_per_launch"] - (4 - bits // 8) * reads
    t_c, crd, cwr = reading(C, cneed, cms, "%s (one wavefront per locus, 200,000 of them; %d-bit read words), same passes" % (Ck[0].split("(")[0].replace("void ", ""), bits))
    head += "#\n" + t_c
    t["fa:C3:200000:k_call_v2"] = {"hbm_bytes_per_launch": crd + cwr, "fetch_size_kb": C["FETCH_SIZE"], "write_size_kb": C["WRITE_SIZE"],
                                   "read_requests": {k: C[k] for k in C if k.startswith("TCC_EA0_RDREQ")},
                                   "correction": "as for k_bp_emit2", "source": "profiles/r05_from_alignments_pmc.txt (the locus kernel of the same step)",
                                   "kernel": Ck[0].split("(")[0].replace("void ", ""), "needed_bytes_per_launch": cneed, "lib_sha16": lib}
for k in ("C5:100000", "C2:10000"):          # (measured on round 3's library: bench.py no longer prints them - the hash differs)
    if k in t and "lib_sha16" not in t[k]:
        t[k]["lib_sha16"] = "(round 3's library)"
json.dump(t, open(P + "/traffic.json", "w"), indent=1)
body = head + "## kernels (traced run: %.3f ms per step, k_bp_emit2 %.3f ms by its HIP events)\n" % (fa_tr["ms_per_step"], fa_tr["roofline"]["kernel_ms"])
body += open(O + "/fa_kernels.txt").read() + "## one step (C3)\n" + open(O + "/fa_timeline.txt").read()
for c in ("C5", "X3", "EX", "C2"):
    pth = O + "/fa_timeline_%s.txt" % c
    if os.path.exists(pth):
        body += "## one step (%s from alignments)\n" % c + open(pth).read()
body += "## counters\n" + open(O + "/fa_pmc_summary.txt").read()
open(P + "/r05_from_alignments_pmc.txt", "w").write(body)
# the library's placement against the bench-side trials
rows = []
for kind in ("place0", "place30"):
    for i in (1, 2, 3):
        pth = O + "/%s_%d.json" % (kind, i)
        if os.path.exists(pth) and os.path.getsize(pth):
            d = last_json(pth)
            sb = d["step_breakdown"]
            rows.append("%-8s process %d: %6.2f M loci/s, %.3f ms per step, k_bp_emit2 %.3f ms; library's probe: %s%s" % (
                "--place 0" if kind == "place0" else "--place 30", i, d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"],
                (lambda B: ", ".join("%.3f of %.3f ms (%d tried)" % (b["probe_ms_kept"], b["probe_ms_slowest"], b["candidates"]) for b in B[:2]) +
                           ("" if len(B) <= 2 else "; %d more blocks for the trials, kept %.3f-%.3f ms" % (
                               len(B) - 2, min(b["probe_ms_kept"] for b in B[2:]), max(b["probe_ms_kept"] for b in B[2:]))))(sb["allocation"]["blocks"]),
                ("; bench-side walk times %s" % sb["placement"]["walk_ms_by_allocation"]) if sb.get("placement") and "walk_ms_by_allocation" in sb["placement"] else ""))
open(P + "/r05_place0_vs_place30.txt", "w").write(
    "# bench.py --no-cpu-baseline --no-other-configs --no-parity in fresh processes on one box (scripts/collect_round5.sh): the read words in the\n"
    "# block the LIBRARY chose (smc_mem_alloc_best through engine.DevBuf(walk_output=True); the default) and with round 4's bench-side\n"
    "# trials on top (30 more allocations, each timed with the real walk, the two fastest kept)\n" + "\n".join(rows) + "\n")
txt = ("# scripts/e2e_perf.py on the GPU box (round 5): the command-line path on synthetic BAMs, stage by stage\n")
for n, label in (("2000_3000", "2000 loci x 3000x, 60 reads per UMI"), ("20000_1000", "20000 loci x 1000x, 20 reads per UMI"),
                 ("500_58000", "500 loci x 58000x, 9 reads per UMI (the depth of the reference's example run)"),
                 ("2000_58000", "2000 loci x 58000x, 9 reads per UMI (the same depth, four times the loci: more than one run)")):
    if os.path.exists(O + "/e2e_%s.txt" % n):
        txt += "## " + label + "\n" + open(O + "/e2e_%s.txt" % n).read()
open(P + "/r05_e2e_cli.txt", "w").write(txt)
print("k_bp_emit2: %.2f GB read + %.2f GB written = %.2f x the %.2f GB needed; %.3f ms" % (rd / 1e9, wr / 1e9, (rd + wr) / need, need / 1e9, kms))
print("\n".join(rows))
print("bench: %.1f M loci/s, %.3f ms per step; under trace: %.3f ms, emit %.3f ms by events" % (
    bench["value"] / 1e6, bench["ms_per_step"], last_json(O + "/bench_under_trace.json")["ms_per_step"],
    last_json(O + "/bench_under_trace.json")["roofline"]["kernel_ms"]))
