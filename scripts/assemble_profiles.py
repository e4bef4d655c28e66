"""profiles/<prefix>_* from gpurun_out/<tag> (scripts/collect_round.sh <tag>): copies of the summaries, the traffic records of the walk
(every shape) and of the locus kernel (C3) in profiles/traffic.json - computed from the PMC passes, each tied to the hash of the library
it was measured on - and a short reading of the counters (dev tool, build container).  usage: assemble_profiles.py [tag] [prefix]"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
O, P = "gpurun_out/" + (sys.argv[1] if len(sys.argv) > 1 else "r06final"), "profiles"
PRE = sys.argv[2] if len(sys.argv) > 2 else "r06"


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


for src, dst in (("bench.json", "bench_line.json"), ("bench_detail.json", "bench_detail.json"), ("bench_under_trace.json", "bench_under_trace_line.json"),
                 ("kernel_stats.csv", "bench_kernel_stats.csv"), ("kernel_trace_by_grid.txt", "bench_kernel_trace_by_grid.txt"),
                 ("bench_2ranks_functional.json", "bench_2ranks_one_gpu_functional.json"),
                 ("bench_C4_strong_n1.json", "bench_C4_strong_n1_line.json"), ("bench_C4_strong_n1_detail.json", "bench_C4_strong_n1_detail.json")):
    if os.path.exists(os.path.join(O, src)) and os.path.getsize(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, "%s_%s" % (PRE, dst)))
detail = json.load(open(O + "/bench_detail.json"))
lib = open(O + "/lib_sha16.txt").read().strip()
t = json.load(open(P + "/traffic.json"))
prev = {k: t[k] for k in list(t) if k.startswith("fa:")}
if prev and "_round5" not in t:
    t["_round5"] = prev
body = ("# The step bench.py times (smc_build_planes_w16 -> smc_plan_create_dev_spec -> smc_plan_run_words16) shape by shape, `python3 bench_fa.py\n"
        "# --config <shape> --slots 1` under rocprofv3 (scripts/collect_round.sh): kernels by grid, the timeline of one step, PMC counters per\n"
        "# dispatch (one --pmc pass per counter group).  Library %s.  Readings computed by scripts/assemble_profiles.py.\n#\n" % lib)
CORR = ("read bytes = 128 B x RDREQ_128B + 64 B x RDREQ_64B (+ 32 B x RDREQ_32B): the measured request sizes (FETCH_SIZE tallies every "
        "request at 64 B: MI355X_MICROARCH.md, HBM); WRITE_SIZE as is")
legs = {"C3": detail}
legs.update(detail.get("from_alignments", {}))
for c in ("C3", "C5", "X3", "EX", "C2"):
    pth = O + "/pmc_summary_%s.txt" % c
    if not os.path.exists(pth) or c not in legs:
        continue
    f = parse(pth)
    ek = [k for k in f if "k_bp_emit2" in k]
    if not ek:
        continue
    E = f[ek[0]]
    rf = legs[c]["roofline"]
    need, kms = rf["needed_bytes_per_launch"], rf["kernel_ms"]
    rd = 128 * E.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * E.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * E.get("TCC_EA0_RDREQ_32B_sum", 0)
    wr = E["WRITE_SIZE"] * 1024
    nl = int(re.search(r"(\d+) loci", legs[c]["config"]["workload"] if c == "C3" else legs[c]["workload"]).group(1))
    t["fa:%s:%d" % (c, nl)] = {"hbm_bytes_per_launch": rd + wr, "fetch_size_kb": E["FETCH_SIZE"], "write_size_kb": E["WRITE_SIZE"],
                               "read_requests": {k: E[k] for k in E if k.startswith("TCC_EA0_RDREQ")},
                               "write_requests": {k: E[k] for k in E if k.startswith("TCC_EA0_WRREQ")}, "correction": CORR,
                               "source": "profiles/%s_from_alignments_pmc.txt (python3 bench_fa.py under rocprofv3 --pmc, one pass per counter group; "
                                         "scripts/collect_round.sh)" % PRE,
                               "kernel": "k_bp_emit2", "needed_bytes_per_launch": need, "lib_sha16": lib}
    body += ("# %s: k_bp_emit2 per launch: %.2f M read requests of 128 B + %.2f M of 64 B = %.3f GB; WRITE_SIZE %.3f GB (%.2f M write requests, %.0f %% of\n"
             "#   them 64-byte ones); %.3f GB against %.3f GB needed = %.2f x; in %.3f ms (HIP events of the bench run) = %.2f TB/s of traffic; L2 %.0f %% hits\n"
             % (c, E.get("TCC_EA0_RDREQ_128B_sum", 0) / 1e6, E.get("TCC_EA0_RDREQ_64B_sum", 0) / 1e6, rd / 1e9, wr / 1e9, E.get("TCC_EA0_WRREQ_sum", 0) / 1e6,
                100 * E.get("TCC_EA0_WRREQ_64B_sum", 0) / max(1.0, E.get("TCC_EA0_WRREQ_sum", 1)), (rd + wr) / 1e9, need / 1e9, (rd + wr) / need, kms,
                (rd + wr) / kms / 1e9, 100 * E.get("TCC_HIT_sum", 0) / max(1.0, E.get("TCC_HIT_sum", 0) + E.get("TCC_MISS_sum", 0))))
    if c == "C3":
        ck = [k for k in f if k.startswith("void k_call_v2<64")]
        if ck:
            C = f[ck[0]]
            crd = 128 * C.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * C.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * C.get("TCC_EA0_RDREQ_32B_sum", 0)
            cwr = C["WRITE_SIZE"] * 1024
            cms = detail["step_breakdown"]["k_call_v2_ms"]
            reads = rf["reads_per_s_kernel"] * rf["kernel_ms"] * 1e-3
            cneed = detail["consumer_only"]["roofline"]["needed_bytes_per_launch"] - 2 * reads if "consumer_only" in detail else None
            t["fa:C3:%d:k_call_v2" % nl] = {"hbm_bytes_per_launch": crd + cwr, "fetch_size_kb": C["FETCH_SIZE"], "write_size_kb": C["WRITE_SIZE"],
                                            "read_requests": {k: C[k] for k in C if k.startswith("TCC_EA0_RDREQ")}, "correction": CORR,
                                            "source": "profiles/%s_from_alignments_pmc.txt (the locus kernel of the same step)" % PRE,
                                            "kernel": "k_call_v2<64, true>", "needed_bytes_per_launch": cneed, "lib_sha16": lib}
            if cneed:
                body += ("# C3: k_call_v2<64, true> per launch: %.3f GB read + %.3f GB written = %.2f x the %.3f GB needed; %.3f ms = %.3f of the 8 TB/s on needed bytes\n"
                         % (crd / 1e9, cwr / 1e9, (crd + cwr) / cneed, cneed / 1e9, cms, cneed / cms / 1e6 / 8000.0))
json.dump(t, open(P + "/traffic.json", "w"), indent=1)
sq = O + "/sq_summary_C3.txt"
if os.path.exists(sq):
    f = parse(sq)
    for key, name in (("k_bp_emit2", "k_bp_emit2"), ("void k_call_v2<64", "k_call_v2<64, true>")):
        ks = [k for k in f if key in k]
        if ks and "SQ_WAVES" in f[ks[0]]:
            C = f[ks[0]]
            nw = C["SQ_WAVES"]
            life = 4 * C["SQ_WAVE_CYCLES"] / nw
            body += ("# C3: %s: %d wavefronts, %.1f k cycles each: %.0f %% issuing / %.0f %% waiting for an issue slot / %.0f %% in s_waitcnt; %.0f VALU + %.0f SALU + %.0f LDS"
                     " + %.0f vector loads + %.0f vector stores per wavefront\n"
                     % (name, int(nw), life / 1e3, 100 * 4 * C["SQ_ACTIVE_INST_ANY"] / nw / life, 100 * 4 * C["SQ_WAIT_INST_ANY"] / nw / life,
                        100 * 4 * C["SQ_WAIT_ANY"] / nw / life, C["SQ_INSTS_VALU"] / nw, C["SQ_INSTS_SALU"] / nw, C["SQ_INSTS_LDS"] / nw,
                        C.get("SQ_INSTS_VMEM_RD", 0) / nw, C.get("SQ_INSTS_VMEM_WR", 0) / nw))
for c in ("C3", "C5", "X3", "EX", "C2"):
    for what, name in (("fa_kernels_%s.txt", "kernels by grid"), ("fa_timeline_%s.txt", "one step")):
        pth = O + "/" + what % c
        if os.path.exists(pth):
            body += "## %s (%s from alignments, one step at a time)\n" % (name, c) + open(pth).read()
for c in ("C3", "C5", "X3", "EX", "C2"):
    pth = O + "/pmc_summary_%s.txt" % c
    if os.path.exists(pth):
        body += "## counters (%s)\n" % c + open(pth).read()
if os.path.exists(sq):
    body += "## instruction / wait counters (C3)\n" + open(sq).read()
open(P + "/%s_from_alignments_pmc.txt" % PRE, "w").write(body)
rows = []
for i in (1, 2):
    pth = O + "/fresh_%d.json" % i
    if os.path.exists(pth) and os.path.getsize(pth):
        d = last_json(pth)
        rows.append("fresh process %d: %.2f M loci/s, %.3f ms per step, k_bp_emit2 %.3f ms, one at a time %.3f ms" % (
            i, d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["step_breakdown"]["ms_per_step_one_at_a_time"]))
b = last_json(O + "/bench.json")
rows.insert(0, "the driver's command:  %.2f M loci/s, %.3f ms per step, k_bp_emit2 %.3f ms, one at a time %.3f ms; slot choice %s" % (
    b["value"] / 1e6, b["ms_per_step"], b["roofline"]["kernel_ms"], b["step_breakdown"]["ms_per_step_one_at_a_time"], detail["step_breakdown"].get("slot_choice")))
open(P + "/%s_bench_fresh_processes.txt" % PRE, "w").write("# bench.py in fresh processes on one box (scripts/collect_round.sh)\n" + "\n".join(rows) + "\n")
txt = "# scripts/e2e_perf.py on the GPU box: the command-line path on synthetic BAMs, stage by stage\n"
for n, label in (("2000_3000", "2000 loci x 3000x, 60 reads per UMI"), ("20000_1000", "20000 loci x 1000x, 20 reads per UMI"),
                 ("500_58000", "500 loci x 58000x, 9 reads per UMI (the depth of the reference's example run)"),
                 ("2000_58000", "2000 loci x 58000x, 9 reads per UMI (the same depth, four times the loci: more than one run)")):
    if os.path.exists(O + "/e2e_%s.txt" % n):
        txt += "## " + label + "\n" + open(O + "/e2e_%s.txt" % n).read()
open(P + "/%s_e2e_cli.txt" % PRE, "w").write(txt)
print("\n".join(rows))
print(body[:3000])
