#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
extern "C" {
int smc_bam_open(const char* path, void** out);
void smc_bam_close(void* h);
typedef void (*smc_aln_alloc)(void* ctx, int64_t n_aln, int64_t n_cig, int64_t n_seq, int64_t n_loci, void** out);
int64_t smc_bam_alignments(void* h, const char* chrom, int64_t start0, int64_t end0, int64_t max_reads, double mismatch_thr,
                           int nthreads, smc_aln_alloc alloc, void* alloc_ctx, int64_t* n_loci_done, int64_t* n_slots,
                           int32_t* n_bc, int32_t* n_pair, int32_t* status);
}
static std::vector<char> A, C, B, Lc;
static void alloc(void*, int64_t na, int64_t nc, int64_t ns, int64_t nl, void** out) {
    A.resize(36 * (size_t)na + 64); C.resize(4 * (size_t)nc + 64); B.resize(2 * (size_t)ns + 256); Lc.resize(16 * (size_t)nl + 64);
    out[0] = A.data(); out[1] = C.data(); out[2] = B.data(); out[3] = Lc.data();
}
int main() {
    void* h = nullptr;
    if (smc_bam_open("/tmp/tsan/t.bam", &h)) return 1;
    for (int rep = 0; rep < 3; ++rep) {
        int64_t nl = 0, ns = 0; int32_t nb = 0, np = 0, st = 0;
        const int64_t n = smc_bam_alignments(h, "chrW", 500, 29000, 1ll << 40, 4.0, 8, alloc, nullptr, &nl, &ns, &nb, &np, &st);
        printf("reads %lld loci %lld bc %d pair %d status %d\n", (long long)n, (long long)nl, nb, np, st);
    }
    smc_bam_close(h);
    return 0;
}
