"""profiles/r04_* from gpurun_out/<tag> (scripts/collect_round4.sh <tag>): copies of the summaries, the walk's traffic record in
profiles/traffic.json (computed from the passes, not transcribed) and a short reading of its counters (dev tool, build container).
usage: assemble_profiles_r04.py [tag]"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
O, P = "gpurun_out/" + (sys.argv[1] if len(sys.argv) > 1 else "r04final"), "profiles"


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


for src, dst in (("bench.json", "r04_bench_C3_200k.json"), ("bench_under_trace.json", "r04_bench_under_trace.json"),
                 ("kernel_stats.csv", "r04_bench_kernel_stats.csv"), ("kernel_trace_by_grid.txt", "r04_bench_kernel_trace_by_grid.txt"),
                 ("shapes.txt", "r04_other_shapes.txt"), ("bench_2ranks_functional.json", "r04_bench_2ranks_one_gpu_functional.json")):
    if not os.path.exists(os.path.join(O, src)):
        continue
    shutil.copy(os.path.join(O, src), os.path.join(P, dst))
bench = last_json(O + "/bench.json")
fa_tr = last_json(O + "/fa_under_trace.json")
f = parse(O + "/fa_pmc_summary.txt")
E = f[[k for k in f if "k_bp_emit2" in k][0]]
rd = 128 * E["TCC_EA0_RDREQ_128B_sum"] + 64 * E["TCC_EA0_RDREQ_64B_sum"] + 32 * E.get("TCC_EA0_RDREQ_32B_sum", 0)
wr = E["WRITE_SIZE"] * 1024
need = bench["roofline"]["needed_bytes_per_launch"]
kms = bench["roofline"]["kernel_ms"]
NW = E["SQ_WAVES"]
life = 4 * E["SQ_WAVE_CYCLES"] / NW
act, wis, wany = (4 * E[x] / NW for x in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"))
head = ('''# The step bench.py times (smc_build_planes -> smc_plan_create_dev -> smc_plan_run_words) on the C3-shaped run, `python3 -m bench_fa
# --config C3` under rocprofv3 (scripts/collect_round4.sh): kernels by grid, the timeline of one step, PMC counters per dispatch (one
# --pmc pass per counter group).  The reading below is computed from the passes by scripts/assemble_profiles_r04.py.
#
# k_bp_emit2 (the walk that writes the read words), per launch:
#   reads   %.2f M requests of 128 B + %.2f M of 64 B = %.2f GB   (FETCH_SIZE %.3g KB: counts every request at 64 B)
#   writes  WRITE_SIZE %.2f GB; %.2f M write requests, %.0f %% of them 64-byte ones
#   total   %.2f GB against %.2f GB needed = %.2f x; in %.3f ms (HIP events of the bench run) = %.2f TB/s
#   L2      %.0f %% of %.1f M requests hit
#   a wavefront (%d of them, one per (tile, part)): %.1f k cycles resident, %.0f %% issuing, %.0f %% waiting for an issue slot, %.0f %% parked in s_waitcnt;
#   %.0f VALU + %.0f SALU + %.0f LDS + %.0f vector loads + %.0f vector stores per wavefront
''' % (E["TCC_EA0_RDREQ_128B_sum"] / 1e6, E["TCC_EA0_RDREQ_64B_sum"] / 1e6, rd / 1e9, E["FETCH_SIZE"],
       wr / 1e9, E["TCC_EA0_WRREQ_sum"] / 1e6, 100 * E["TCC_EA0_WRREQ_64B_sum"] / E["TCC_EA0_WRREQ_sum"],
       (rd + wr) / 1e9, need / 1e9, (rd + wr) / need, kms, (rd + wr) / kms / 1e9,
       100 * E["TCC_HIT_sum"] / (E["TCC_HIT_sum"] + E["TCC_MISS_sum"]), (E["TCC_HIT_sum"] + E["TCC_MISS_sum"]) / 1e6,
       int(NW), life / 1e3, 100 * act / life, 100 * wis / life, 100 * wany / life,
       E["SQ_INSTS_VALU"] / NW, E["SQ_INSTS_SALU"] / NW, E["SQ_INSTS_LDS"] / NW, E["SQ_INSTS_VMEM_RD"] / NW, E["SQ_INSTS_VMEM_WR"] / NW))
# the locus kernel in the same passes (the step's second kernel; round 3's reading of it: profiles/r03_call_v2_counters.txt)
Ck = [k for k in f if k.startswith("void k_call_v2<64>")]
if Ck:
    C = f[Ck[0]]
    cw = C["SQ_WAVES"]
    clife = 4 * C["SQ_WAVE_CYCLES"] / cw
    cact, cwis, cwany = (4 * C[x] / cw for x in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"))
    crd = 128 * C["TCC_EA0_RDREQ_128B_sum"] + 64 * C["TCC_EA0_RDREQ_64B_sum"] + 32 * C.get("TCC_EA0_RDREQ_32B_sum", 0)
    cms = bench["step_breakdown"]["k_call_v2_ms"]
    cneed = bench["consumer_only"]["roofline"]["needed_bytes_per_launch"]
    head += ("""#
# k_call_v2<64> (one wavefront per locus, 200,000 of them), per launch, same passes:
#   a wavefront lives %.1f k cycles (round 3: 69.1 k): %.0f %% issuing, %.0f %% waiting for an issue slot, %.0f %% parked in s_waitcnt;
#   %.0f VALU + %.0f SALU + %.0f LDS + %.0f vector loads per wavefront (round 3: 3411 + 1882 + 162 + 26)
#   reads %.2f GB (%.2f M requests of 128 B) + writes %.2f GB against %.2f GB needed = %.2f x; in %.3f ms (HIP events of the bench run) = %.2f TB/s
#   L2 %.0f %% of %.1f M requests hit
""" % (clife / 1e3, 100 * cact / clife, 100 * cwis / clife, 100 * cwany / clife,
       C["SQ_INSTS_VALU"] / cw, C["SQ_INSTS_SALU"] / cw, C["SQ_INSTS_LDS"] / cw, C["SQ_INSTS_VMEM_RD"] / cw,
       crd / 1e9, C["TCC_EA0_RDREQ_128B_sum"] / 1e6, C["WRITE_SIZE"] * 1024 / 1e9, cneed / 1e9, (crd + C["WRITE_SIZE"] * 1024) / cneed, cms,
       (crd + C["WRITE_SIZE"] * 1024) / cms / 1e9, 100 * C["TCC_HIT_sum"] / (C["TCC_HIT_sum"] + C["TCC_MISS_sum"]), (C["TCC_HIT_sum"] + C["TCC_MISS_sum"]) / 1e6))
open(P + "/r04_from_alignments_pmc.txt", "w").write(
    head + "## kernels (traced run: %.3f ms per step, k_bp_emit2 %.3f ms by its HIP events)\n" % (fa_tr["ms_per_step"], fa_tr["roofline"]["kernel_ms"]) +
    open(O + "/fa_kernels.txt").read() + "## one step\n" + open(O + "/fa_timeline.txt").read() + "## counters\n" + open(O + "/fa_pmc_summary.txt").read())
txt = ("# scripts/e2e_perf.py / scripts/bp_perf.py on the GPU box (round 4): the command-line path on synthetic BAMs, stage by stage, and\n"
       "# the device plane builder alone (HIP events around smc_build_planes, alignments resident)\n")
for n, label in (("2000", "2000 loci x 3000x, 60 reads per UMI"), ("20000", "20000 loci x 1000x, 20 reads per UMI"),
                 ("500", "500 loci x 58000x, 9 reads per UMI (the depth of the reference's example run)")):
    txt += "## " + label + "\n" + open(O + "/e2e_%s.txt" % n).read() + "# plane builder alone:\n" + open(O + "/bp_%s.txt" % n).read()
txt += "## kernels of the 20000-locus run (rocprofv3 --kernel-trace, by grid)\n" + "".join(open(O + "/e2e_kernels.txt").readlines()[:30])
txt += "## kernels of the 500 x 58000x run\n" + "".join(open(O + "/e2e_deep_kernels.txt").readlines()[:30])
open(P + "/r04_e2e_cli.txt", "w").write(txt)

t = json.load(open(P + "/traffic.json"))
if "_round3" not in t and "fa:C3:200000" in t:
    t["_round3"] = {"fa:C3:200000": t["fa:C3:200000"]}
t["fa:C3:200000"] = {"hbm_bytes_per_launch": rd + wr, "fetch_size_kb": E["FETCH_SIZE"], "write_size_kb": E["WRITE_SIZE"],
                     "read_requests": {k: E[k] for k in E if k.startswith("TCC_EA0_RDREQ")},
                     "write_requests": {k: E[k] for k in E if k.startswith("TCC_EA0_WRREQ")},
                     "correction": "read bytes = 128 B x RDREQ_128B + 64 B x RDREQ_64B (+ 32 B x RDREQ_32B): the measured request sizes (FETCH_SIZE tallies "
                                   "every request at 64 B: MI355X_MICROARCH.md, HBM); WRITE_SIZE as is",
                     "source": "profiles/r04_from_alignments_pmc.txt (python3 -m bench_fa under rocprofv3 --pmc, one pass per counter group; scripts/collect_round4.sh)",
                     "kernel": "k_bp_emit2", "needed_bytes_per_launch": need}
json.dump(t, open(P + "/traffic.json", "w"), indent=1)
print("k_bp_emit2: %.2f GB read + %.2f GB written = %.2f x the %.2f GB needed; %.3f ms" % (rd / 1e9, wr / 1e9, (rd + wr) / need, need / 1e9, kms))
kt = open(O + "/kernel_trace_by_grid.txt").read().splitlines()
print("\n".join(l for l in kt if "k_bp_emit2" in l or "k_call_v2<64> grid 12800000" in l))
print("bench: %.1f M loci/s, %.3f ms per step; under trace: %.3f ms, emit %.3f ms by events" % (
    bench["value"] / 1e6, bench["ms_per_step"], last_json(O + "/bench_under_trace.json")["ms_per_step"],
    last_json(O + "/bench_under_trace.json")["roofline"]["kernel_ms"]))
