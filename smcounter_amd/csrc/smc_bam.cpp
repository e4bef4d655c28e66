// smc_bam.cpp - native BGZF/BAM reader and single-position pileup.
//
// Same semantics as smcounter_amd/bamio.py (which stays as the readable reference and the fallback):
// what the reference takes from `samfile.pileup(region, truncate=True, max_depth=1000000,
// stepper='nofilter')` at smCounter.py:316-448 - every mapped alignment covering the position in file
// order, no filtering; query_position incl. soft clips; is_del inside D/N; indel = the I (+) / D (-) that
// starts right after the position (samtools 0.1.19 resolve_cigar).  Barcode / read ids are made dense per
// locus in order of first appearance (smCounter.py:463-464), alleles are ids into a per-locus key table.
// Keys of deletion-start alleles need the reference sequence; they are emitted as "D<len>|<site>" and
// completed by the Python caller from the FASTA (bamio.NativeBam).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "smcounter_hip.h"   // smc_locus: the descriptor smc_bam_planes fills

#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

const char SEQ_CODE[] = "=ACMGRSVTWYHKDBN";

// Worker threads shared by every stage of a decode (inflate, parse, interning, packing): started once per process and
// reused, because a run's stages are each a millisecond or two of work and starting a hundred threads per stage costs
// more than that.  run(n, f) executes f(0) .. f(n - 1), the caller taking part; tasks are claimed one by one.
class Pool {
    std::vector<std::thread>* th = new std::vector<std::thread>();   // (a pointer: dropped, not destroyed, in a forked child)
    std::mutex m, run_m;
    std::condition_variable cv, cv_done;
    const std::function<void(int)>* job = nullptr;
    int n_tasks = 0, busy = 0, invited = 0;
    std::atomic<int> next{0};
    std::atomic<uint64_t> gen_hint{0};
    static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    uint64_t gen = 0;
    bool stop = false;
    pid_t owner = 0;
    void worker(int id) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* f;
            // a decode is a chain of short stages: a worker that just finished one spins briefly for the next before it sleeps
            for (int spin = 0; spin < 2000 && gen_hint.load(std::memory_order_relaxed) == seen; ++spin) cpu_relax();
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || (gen != seen && id < invited); });
                if (stop) return;
                seen = gen; f = job;
                ++busy;
            }
            for (int i = next.fetch_add(1); i < n_tasks; i = next.fetch_add(1)) (*f)(i);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--busy == 0) cv_done.notify_all();
            }
        }
    }
public:
    static Pool& get() { static Pool* p = new Pool; return *p; }   // (never destroyed: no join at exit)
    // at most `par` threads (incl. the caller) work on the n tasks
    void run(int n, int par, const std::function<void(int)>& f) {
        if (n <= 0) return;
        par = std::min(par, n);
        if (par <= 1) { for (int i = 0; i < n; ++i) f(i); return; }
        std::lock_guard<std::mutex> one_at_a_time(run_m);         // (two decoders on two threads take turns; tasks never call run)
        std::unique_lock<std::mutex> lk(m);
        if (owner != getpid()) { if (owner) th = new std::vector<std::thread>(); owner = getpid(); busy = 0; }
        while ((int)th->size() < par - 1 && th->size() < 255) { const int id = (int)th->size(); th->emplace_back([this, id] { worker(id); }); }
        job = &f; n_tasks = n; next.store(0); invited = par - 1; ++gen;
        gen_hint.store(gen, std::memory_order_relaxed);
        lk.unlock();
        cv.notify_all();
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i);
        lk.lock();
        // every invited worker has either finished or not yet looked: un-invite the latecomers, wait for the busy ones
        invited = 0;
        cv_done.wait(lk, [&] { return busy == 0; });
        job = nullptr;
    }
};

// Grow-only byte buffer without value initialisation (a std::vector would zero - and page in - tens of megabytes per run
// on one thread; here the inflating threads touch the pages first, and the memory is reused from run to run).
struct ByteBuf {
    uint8_t* p = nullptr;
    size_t n = 0, cap = 0;
    ByteBuf() = default;
    ByteBuf(const ByteBuf&) = delete;
    ByteBuf& operator=(const ByteBuf&) = delete;
    ~ByteBuf() { free(p); }
    uint8_t* data() { return p; }
    const uint8_t* data() const { return p; }
    size_t size() const { return n; }
    bool resize(size_t want) {                       // contents up to min(old, new) size are kept
        if (want > cap) {
            size_t c = cap ? cap : (1u << 20);
            while (c < want) c += c / 2;
            uint8_t* q = (uint8_t*)realloc(p, c);
            if (!q) return false;
            p = q; cap = c;
        }
        n = want;
        return true;
    }
    void swap(ByteBuf& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
};
// The same for arrays of trivially copyable records (the run's alignments: ~ 120 bytes each, all fields written by the parser)
template <class T> struct RawVec {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    RawVec() = default;
    RawVec(const RawVec&) = delete;
    RawVec& operator=(const RawVec&) = delete;
    ~RawVec() { free(p); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    T* begin() { return p; }
    T* end() { return p + n; }
    const T* begin() const { return p; }
    const T* end() const { return p + n; }
    void clear() { n = 0; }
    void resize_uninit(size_t want) {                // (old contents kept; aborts like operator new on exhaustion)
        if (want > cap) {
            size_t c = cap ? cap : 4096;
            while (c < want) c += c / 2;
            T* q = (T*)realloc((void*)p, c * sizeof(T));
            if (!q) abort();
            p = q; cap = c;
        }
        n = want;
    }
};

// Views into the inflated record bytes (kept in Bam::rec_data until the next run): an alignment owns no memory, so
// parsing copies nothing and dropping a run's alignments frees nothing.
struct CigView {                   // CIGAR words, len << 4 | op (not necessarily 4-byte aligned in the record)
    const uint8_t* p = nullptr;
    uint32_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    uint32_t operator[](size_t k) const { uint32_t c; memcpy(&c, p + 4 * k, 4); return c; }
    struct It {
        const uint8_t* p;
        uint32_t operator*() const { uint32_t c; memcpy(&c, p, 4); return c; }
        It& operator++() { p += 4; return *this; }
        bool operator!=(const It& o) const { return p != o.p; }
    };
    It begin() const { return It{p}; }
    It end() const { return It{p + 4 * (size_t)n}; }
};
struct SeqView {                   // 4-bit packed bases, decoded on access
    const uint8_t* p = nullptr;
    uint32_t n = 0;
    char operator[](size_t i) const { return SEQ_CODE[(p[i >> 1] >> ((i & 1) ? 0 : 4)) & 15]; }
    std::string substr(size_t o, size_t len) const {
        std::string r;
        for (size_t i = o; i < n && i < o + len; ++i) r.push_back((*this)[i]);
        return r;
    }
};
struct QualView {
    const uint8_t* p = nullptr;
    uint8_t operator[](size_t i) const { return p[i]; }
};

struct Aln {
    int32_t pos, end;
    uint16_t flag;
    uint8_t mapq, has_nm;
    uint32_t nm, l_seq;
    std::string_view qname;
    SeqView seq;
    QualView qual;
    CigView cigar;
    int32_t bc_gid, pair_gid;      // run-wide ids of the barcode and of (barcode, read id)
    int32_t c1, c2;                // positions of the last two ':' of qname (-1: fewer than 3 fields)
    uint64_t h_bc, h_pair;         // hashes of the barcode and of the whole <readid>:<barcode> prefix
    uint32_t n_ind, qalen, left_sp;
    uint8_t oflag;
};

struct Bam {
    FILE* fh = nullptr;
    const uint8_t* map = nullptr;      // the file, mapped (the block stream inflates straight from the page cache; nullptr: pread into `comp`)
    size_t map_len = 0;
    std::string err;
    int io_threads = 1;                // threads inflating BGZF blocks in collect_reads
    ByteBuf rec_data;                  // inflated records of the last collect_reads (its alignments point into it)
    ByteBuf comp;                      // compressed bytes of the blocks being inflated (reused)
    // streaming cursor: where the previous collect_reads found its first overlapping record - a later run on the
    // same reference that starts at or after the previous one never needs anything before it
    int cur_tid = -1;
    int64_t cur_start = -1, cur_end = -1;      // checkpoint positions: start and end of the previous run
    uint64_t cur_voff = 0, cur_voff_end = 0;   // first record with end > cur_start / with end > cur_end
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_lens;
    std::vector<std::vector<uint64_t>> lin;   // BAI linear index per reference
    uint64_t first_record = 0;
    // BGZF state
    std::vector<uint8_t> buf;
    size_t off = 0;
    uint64_t block_start = 0, next_block = 0;
    // last pileup result
    std::vector<uint32_t> umi, frag, nm, n_indel, left_sp, qlen, qalen;
    std::vector<int32_t> qpos, indel;
    std::vector<uint8_t> flag, mq, is_del, allele, bq;
    std::vector<int64_t> read_off;     // per locus + 1
    std::string keys;                  // allele keys beyond the six fixed ones, '\n' separated per locus, loci '\0'
    std::vector<int32_t> n_keys;       // per locus
    // last smc_bam_planes result (keys / n_keys shared with the pileup result)
    std::vector<uint32_t> p_umi_start;
    std::string ds_info;               // loci over the barcode cap: "<locus>\t<u>:<barcode>\t...\n", barcodes by first included read
    // last smc_bam_alignments result: the run's alignments (views into rec_data) and its barcode texts by run-wide id
    RawVec<Aln> d_reads;
    RawVec<Aln> parsed;                // (scratch of collect_reads, reused)
    RawVec<uint64_t> it_key;           // (interning tables of collect_reads: hash, first record, the records' slots - reused, their pages stay)
    RawVec<uint32_t> it_first, it_slot;
    std::vector<std::string> d_bc_names;

    bool load_block(uint64_t coff) {
        if (fseeko(fh, (off_t)coff, SEEK_SET) != 0) return false;
        uint8_t hdr[18];
        if (fread(hdr, 1, 18, fh) != 18) { buf.clear(); off = 0; return false; }
        if (hdr[0] != 31 || hdr[1] != 139 || hdr[12] != 'B' || hdr[13] != 'C') { err = "not a BGZF block"; return false; }
        const unsigned xlen = hdr[10] | (hdr[11] << 8), bsize = (hdr[16] | (hdr[17] << 8)) + 1;
        std::vector<uint8_t> rest(bsize - 18);
        if (fread(rest.data(), 1, rest.size(), fh) != rest.size()) { err = "truncated BGZF block"; return false; }
        const size_t c0 = xlen - 6, clen = rest.size() - c0 - 8;
        const uint32_t isize = rest[rest.size() - 4] | (rest[rest.size() - 3] << 8) | (rest[rest.size() - 2] << 16) |
                               ((uint32_t)rest[rest.size() - 1] << 24);
        buf.resize(isize);
        if (isize) {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { err = "inflateInit2"; return false; }
            zs.next_in = rest.data() + c0; zs.avail_in = (uInt)clen;
            zs.next_out = buf.data(); zs.avail_out = isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END) { err = "inflate failed"; return false; }
        }
        off = 0; block_start = coff; next_block = coff + bsize;
        return true;
    }
    void seek(uint64_t v) { load_block(v >> 16); off = v & 0xFFFF; }
    uint64_t tell() const { return (block_start << 16) | off; }
    size_t read(void* dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            if (off >= buf.size()) {
                if (!load_block(next_block)) break;
                if (buf.empty() && feof(fh)) break;
                continue;
            }
            const size_t take = std::min(n - got, buf.size() - off);
            memcpy((uint8_t*)dst + got, buf.data() + off, take);
            off += take; got += take;
        }
        return got;
    }
};

// one alignment record (without its block_size word) -> Aln
void parse_body(const uint8_t* p, size_t r_size, Aln& a, int32_t& tid) {
    int32_t pos, l_seq;
    memcpy(&tid, p, 4); memcpy(&pos, p + 4, 4);
    const unsigned l_name = p[8];
    a.mapq = p[9];
    const unsigned n_cig = p[12] | (p[13] << 8);
    a.flag = (uint16_t)(p[14] | (p[15] << 8));
    memcpy(&l_seq, p + 16, 4);
    a.pos = pos; a.l_seq = (uint32_t)l_seq;
    size_t o = 32;
    a.qname = std::string_view((const char*)p + o, l_name ? l_name - 1 : 0);
    o += l_name;
    a.cigar.p = p + o; a.cigar.n = n_cig;
    o += 4 * n_cig;
    a.seq.p = p + o; a.seq.n = (uint32_t)l_seq;
    o += (l_seq + 1) / 2;
    a.qual.p = p + o;
    o += l_seq;
    a.nm = 0; a.has_nm = 0;
    while (o + 3 <= r_size) {   // aux: find NM (smCounter.py:329-334)
        const char t0 = p[o], t1 = p[o + 1], ty = p[o + 2];
        o += 3;
        size_t sz = 0;
        switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': { while (o < r_size && p[o]) ++o; ++o; continue; }
            case 'B': {
                const char sub = p[o];
                uint32_t cnt; memcpy(&cnt, p + o + 1, 4);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                o += 5 + (size_t)cnt * es;
                continue;
            }
            default: o = r_size; continue;
        }
        if (t0 == 'N' && t1 == 'M' && ty != 'A' && ty != 'f') {
            int64_t v = 0;
            if (ty == 'c') v = (int8_t)p[o]; else if (ty == 'C') v = p[o];
            else if (ty == 's') { int16_t x; memcpy(&x, p + o, 2); v = x; }
            else if (ty == 'S') { uint16_t x; memcpy(&x, p + o, 2); v = x; }
            else if (ty == 'i') { int32_t x; memcpy(&x, p + o, 4); v = x; }
            else { uint32_t x; memcpy(&x, p + o, 4); v = x; }
            a.nm = (uint32_t)v; a.has_nm = 1;
            break;
        }
        o += sz;
    }
    int32_t e = pos;
    for (uint32_t c : a.cigar) { const unsigned op = c & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) e += (int32_t)(c >> 4); }
    a.end = e;
}

// libdeflate, when the image has it (the shared object without its header: the three entry points are declared here),
// inflates a BGZF block 2-3 x faster than zlib; zlib stays the fallback.
#include <dlfcn.h>
// (environment switches of the decoder are experiment knobs: read only under SMC_EXPERIMENTAL, as in the HIP library)
static const char* exp_env(const char* name) {
    const char* const xv = getenv("SMC_EXPERIMENTAL");
    const bool on = xv && *xv && strcmp(xv, "0") != 0;                             // (looked at every time: tests switch it on and off)
    const char* v = getenv(name);
    return on ? v : nullptr;
}
struct Libdeflate {
    void* (*alloc)(void) = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    Libdeflate() {
        if (exp_env("SMC_BAM_ZLIB")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void* (*)(void))dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        if (!alloc || !decompress || !release) alloc = nullptr;
    }
    bool ok() const { return alloc != nullptr; }
};
static const Libdeflate g_ld;

// Record stream over consecutive BGZF blocks starting at a virtual offset: a stretch of the file is read in one piece,
// its blocks (independent deflate streams) are inflated by the pool into one growing buffer, records are then parsed in
// place.  Consumed bytes are kept (record offsets stay valid for the whole run).
struct BlockStream {
    Bam& b;
    int nthreads;
    uint64_t next_coff;
    ByteBuf data;                  // inflated bytes, consumed up to `pos`
    size_t pos = 0;
    bool eof = false;
    std::vector<std::pair<size_t, uint64_t>> blocks;   // (offset in data, file offset) of every block
    uint64_t soft_stop = ~0ull;    // hint: file offset beyond which the caller expects to need (almost) nothing
    BlockStream(Bam& bam, uint64_t voff, int nt, uint64_t stop_hint = ~0ull)
        : b(bam), nthreads(nt < 1 ? 1 : nt), next_coff(voff >> 16), soft_stop(stop_hint) {
        data.swap(b.rec_data);     // (the previous run's buffer: its pages are already there)
        data.n = 0;
        refill();
        pos = (size_t)(voff & 0xFFFF);
        if (pos > data.size()) pos = data.size();
    }
    bool refill() {
        if (eof) return false;
        // the stretch to read: up to two blocks behind the hint, or 8 MB when there is none (the caller refills again if it must)
        size_t want = 8u << 20;
        if (soft_stop != ~0ull) want = soft_stop > next_coff ? (size_t)(soft_stop - next_coff) + (3u << 16) : (3u << 16);
        // (no further per refill than the threads can share: a consumer that finds its last record early stops refilling)
        {
            // (128 KB of compressed bytes per thread and refill: every refill is two rounds of the pool - inflate, record walk -, and
            // at 64 KB the 58,000x run's seven refills cost 2 ms more than its four do)
            static const int sh = exp_env("SMC_BAM_REFILL_SHIFT") ? atoi(exp_env("SMC_BAM_REFILL_SHIFT")) : 17;
            want = std::min<size_t>(want, std::max<size_t>(1u << 20, (size_t)nthreads << sh));
        }
        want = std::min<size_t>(std::max<size_t>(want, 1u << 18), 256u << 20);
        const auto tr0 = std::chrono::steady_clock::now();
        size_t got = 0;
        const uint8_t* cb = nullptr;
        if (b.map && next_coff <= b.map_len) {                         // (a copy of 20 MB per 58,000x run saved: 2.5 ms)
            cb = b.map + next_coff;
            got = std::min<size_t>(want, b.map_len - (size_t)next_coff);
        } else {
            if (!b.comp.resize(want)) { b.err = "out of memory"; eof = true; return false; }
            while (got < want) {
                const ssize_t r = pread(fileno(b.fh), b.comp.data() + got, want - got, (off_t)(next_coff + got));
                if (r <= 0) break;
                got += (size_t)r;
            }
            cb = b.comp.data();
        }
        struct Blk { size_t c0, clen; uint32_t isize; size_t out_off; };
        std::vector<Blk> blks;
        size_t o = 0, total = 0;
        int past = 0;
        while (o + 18 <= got) {
            const uint8_t* hdr = cb + o;
            if (hdr[0] != 31 || hdr[1] != 139 || hdr[12] != 'B' || hdr[13] != 'C') { b.err = "not a BGZF block"; eof = true; break; }
            const size_t xlen = hdr[10] | (hdr[11] << 8), bsize = (size_t)(hdr[16] | (hdr[17] << 8)) + 1;
            if (bsize < 12 + xlen + 8) { b.err = "not a BGZF block"; eof = true; break; }
            if (o + bsize > got) break;                                 // cut by the end of the stretch: next refill
            Blk k;
            k.c0 = o + 12 + xlen; k.clen = bsize - 12 - xlen - 8;
            const uint8_t* tl = cb + o + bsize - 4;
            k.isize = tl[0] | (tl[1] << 8) | (tl[2] << 16) | ((uint32_t)tl[3] << 24);
            k.out_off = total; total += k.isize;
            blocks.emplace_back(data.size() + k.out_off, next_coff + o);
            blks.push_back(k);
            o += bsize;
            if (next_coff + o > soft_stop && ++past >= 2) break;
        }
        if (blks.empty()) {
            if (b.err.empty() && got >= 18 && o + 18 <= got && got == want && want < (256u << 20)) { b.err = "BGZF block larger than the read window"; }
            else if (b.err.empty() && got > o && got < want) b.err = "truncated BGZF block";
            eof = true;
            return false;
        }
        next_coff += o;
        const auto tr1 = std::chrono::steady_clock::now();
        t_read += std::chrono::duration<double, std::milli>(tr1 - tr0).count();
        const size_t base = data.size();
        if (!data.resize(base + total)) { b.err = "out of memory"; eof = true; return false; }
        std::atomic<int> bad(0);
        uint8_t* const out = data.data() + base;
        // tasks of ~4 blocks (a block inflates in 20-200 us)
        const int per = 4, nt = ((int)blks.size() + per - 1) / per;
        Pool::get().run(nt, nthreads, [&](int t) {
            void* ld = g_ld.ok() ? g_ld.alloc() : nullptr;
            for (int i = t * per; i < std::min<int>((t + 1) * per, (int)blks.size()); ++i) {
                const Blk& k = blks[(size_t)i];
                if (!k.isize) continue;
                if (ld) {
                    size_t n = 0;
                    if (g_ld.decompress(ld, cb + k.c0, k.clen, out + k.out_off, k.isize, &n) != 0 || n != k.isize) bad = 1;
                    continue;
                }
                z_stream zs;
                memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; continue; }
                zs.next_in = const_cast<uint8_t*>(cb + k.c0); zs.avail_in = (uInt)k.clen;
                zs.next_out = out + k.out_off; zs.avail_out = k.isize;
                if (inflate(&zs, Z_FINISH) != Z_STREAM_END) bad = 1;
                inflateEnd(&zs);
            }
            if (ld) g_ld.release(ld);
        });
        t_inflate += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr1).count();
        if (bad.load()) { b.err = "inflate failed"; eof = true; return false; }
        return true;
    }
    double t_read = 0, t_inflate = 0;   // ms, for SMC_BAM_TIMING
    // BAM virtual offset of the byte at data offset `o` (keep mode)
    uint64_t voffset_of(size_t o) const {
        size_t lo = 0, hi = blocks.size();
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (blocks[mid].first <= o) lo = mid; else hi = mid; }
        return (blocks[lo].second << 16) | (uint64_t)(o - blocks[lo].first);
    }
    // next record body (without block_size), contiguous in memory; nullptr at the end of the file
    const uint8_t* next_record(size_t& n) {
        while (data.size() - pos < 4) if (!refill()) return nullptr;
        int32_t bs;
        memcpy(&bs, data.data() + pos, 4);
        while (data.size() - pos < 4 + (size_t)bs) if (!refill()) return nullptr;
        const uint8_t* p = data.data() + pos + 4;
        pos += 4 + (size_t)bs;
        n = (size_t)bs;
        return p;
    }
};

// Mapped alignments overlapping [start0, end0) on `chrom`, in file order, with run-wide barcode / fragment ids
// and the per-read CIGAR summaries.  Returns 0, or the negative error code of smc_bam_pileup.
int collect_reads(Bam& b, const char* chrom, int64_t start0, int64_t end0, RawVec<Aln>& reads, int& n_bc, int& n_pair,
                  std::vector<std::string>* bc_names = nullptr) {
    int tid = -1;
    for (size_t i = 0; i < b.ref_names.size(); ++i) if (b.ref_names[i] == chrom) tid = (int)i;
    if (tid >= 0) {
        uint64_t voff = b.first_record;
        if ((size_t)tid < b.lin.size() && !b.lin[(size_t)tid].empty()) {
            const auto& iv = b.lin[(size_t)tid];
            long w = (long)std::min<int64_t>(start0 >> 14, (int64_t)iv.size() - 1);
            while (w >= 0 && iv[(size_t)w] == 0) --w;
            if (w >= 0) voff = iv[(size_t)w];
        }
        if (tid == b.cur_tid && !exp_env("SMC_BAM_NO_CURSOR")) {
            if (b.cur_end >= 0 && start0 >= b.cur_end && b.cur_voff_end > voff) voff = b.cur_voff_end;
            else if (start0 >= b.cur_start && b.cur_voff > voff) voff = b.cur_voff;
        }
        b.err.clear();
        // where the linear index says the alignments of the 16 kb window after end0's begin: nothing this run needs
        // lies (much) beyond it, so the stream does not inflate further ahead than that
        uint64_t stop_hint = ~0ull;
        if ((size_t)tid < b.lin.size()) {
            const auto& iv = b.lin[(size_t)tid];
            for (size_t w = (size_t)(end0 >> 14) + 1; w < iv.size(); ++w)
                if (iv[w]) { stop_hint = iv[w] >> 16; break; }
        }
        // 1. inflate (threads) and find the record boundaries up to the first alignment starting at or after end0
        const auto tc0 = std::chrono::steady_clock::now();
        BlockStream bs(b, voff, b.io_threads, stop_hint);
        std::vector<std::pair<size_t, size_t>> recs;    // (offset of the body in bs.data, size)
        recs.reserve(bs.data.size() / 160 + 16);
        int n_pieces = 0, n_rewalked = 0;               // (for SMC_BAM_TIMING)
        // The record walk is a chain of dependent loads, one cache line per record (4 ms for the 310,000 records of a 58,000x run on
        // one thread).  So: what a refill added is cut into pieces, every piece but the first GUESSES where a record starts (a
        // position whose fields make sense as a record header, three records in a row) and walks from there on its own thread; then
        // the pieces are joined in order - a piece counts only if the walk before it ended exactly where it began, otherwise its
        // stretch is walked again from where the truth stands.  A wrong guess costs time, never a record.
        {
            const int n_ref = (int)b.ref_names.size();
            struct Piece { size_t start = 0, stop = 0; bool ok = false, terminal = false; std::vector<std::pair<size_t, size_t>> r; };
            // walk records from o while they start before `limit`; -> where it stopped (a record not consumed: incomplete, at / beyond
            // the limit, or the terminal one)
            auto walk = [&](size_t o, size_t limit, size_t avail, std::vector<std::pair<size_t, size_t>>& out, bool& terminal) -> size_t {
                const uint8_t* d = bs.data.data();
                terminal = false;
                while (o < limit && o + 4 <= avail) {
                    int32_t sz; memcpy(&sz, d + o, 4);
                    if (sz < 0 || o + 4 + (size_t)sz > avail) break;                 // (cut by the end of what is inflated)
                    if (sz < 8) { terminal = true; break; }                             // (malformed: the parser reports it)
                    int32_t rt, rpos; memcpy(&rt, d + o + 4, 4); memcpy(&rpos, d + o + 8, 4);
                    if (rt < 0 || rt > tid || (rt == tid && rpos >= end0)) { terminal = true; break; }
                    if (rt == tid) out.emplace_back(o + 4, (size_t)sz);
                    o += 4 + (size_t)sz;
                }
                return o;
            };
            auto plausible = [&](size_t o, size_t avail, size_t& next) -> bool {
                const uint8_t* d = bs.data.data();
                if (o + 36 > avail) return false;
                int32_t sz, rt, rpos, lseq; memcpy(&sz, d + o, 4); memcpy(&rt, d + o + 4, 4); memcpy(&rpos, d + o + 8, 4); memcpy(&lseq, d + o + 20, 4);
                const uint32_t lname = d[o + 12], ncig = (uint32_t)d[o + 16] | (uint32_t)d[o + 17] << 8;
                if (sz < 32 || sz > (1 << 24) || o + 4 + (size_t)sz > avail) return false;
                if (rt < -1 || rt >= n_ref || rpos < -1 || lseq < 0 || lname < 1) return false;
                const uint64_t need = 32ull + lname + 4ull * ncig + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq;
                if (need > (uint64_t)sz || d[o + 4 + 32 + lname - 1] != 0) return false;
                next = o + 4 + (size_t)sz;
                return true;
            };
            const int walk_test = exp_env("SMC_BAM_WALK_TEST") ? atoi(exp_env("SMC_BAM_WALK_TEST")) : 0;
            bool done = false;
            while (!done) {
                const size_t avail = bs.data.size();
                {   // a whole record at bs.pos?  (else: more data, or the end of the file)
                    int32_t sz = -1;
                    if (avail - bs.pos >= 4) memcpy(&sz, bs.data.data() + bs.pos, 4);
                    if (!(sz >= 0 && bs.pos + 4 + (size_t)sz <= avail)) { if (!bs.refill()) break; continue; }
                }
                const size_t span = avail - bs.pos;
                const int K = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, b.io_threads), span >> 18));
                std::vector<Piece> pc((size_t)K);
                Pool::get().run(K, b.io_threads, [&](int k) {
                    Piece& P = pc[(size_t)k];
                    const size_t nominal = bs.pos + span * (size_t)k / (size_t)K, limit = bs.pos + span * (size_t)(k + 1) / (size_t)K;
                    size_t st = nominal;
                    if (k > 0) {
                        bool found = false;
                        int skip = (walk_test == 2 && k % 3 == 1) ? 1 : 0;          // (tests: a guess that lies behind the first record of the piece)
                        if (walk_test == 1 && (k & 1)) return;                        // (tests: no guess)
                        for (; st < limit; ++st) {
                            size_t n1, n2, n3;
                            if (plausible(st, avail, n1) && plausible(n1, avail, n2) && plausible(n2, avail, n3)) { if (skip-- > 0) continue; found = true; break; }
                        }
                        if (!found) return;
                    }
                    P.start = st; P.ok = true;
                    P.r.reserve((limit - st) / 200 + 16);
                    P.stop = walk(st, limit, avail, P.r, P.terminal);
                });
                size_t cur = bs.pos;
                bool stuck = false;                                                       // an incomplete record at `cur`: more data first
                for (int k = 0; k < K && !done && !stuck; ++k) {
                    Piece& P = pc[(size_t)k];
                    const size_t limit = bs.pos + span * (size_t)(k + 1) / (size_t)K;
                    ++n_pieces;
                    if (!(P.ok && P.start == cur)) {
                        ++n_rewalked;
                        // the truth is not at the piece's start: walk up to it - or, without a usable guess, through the piece's stretch
                        bool term = false;
                        const size_t to = (P.ok && P.start > cur) ? P.start : limit;
                        size_t got = walk(cur, to, avail, recs, term);
                        if (term) { cur = got; done = true; break; }
                        if (got < to) { cur = got; stuck = true; break; }
                        cur = got;
                        if (!(P.ok && P.start == cur)) {
                            if (cur < limit) {
                                got = walk(cur, limit, avail, recs, term);
                                if (term) { cur = got; done = true; break; }
                                if (got < limit) { cur = got; stuck = true; break; }
                                cur = got;
                            }
                            continue;
                        }
                    }
                    recs.insert(recs.end(), P.r.begin(), P.r.end());
                    cur = P.stop;
                    if (P.terminal) done = true;
                    else if (P.stop < limit) stuck = true;
                }
                bs.pos = cur;
            }
        }
        // (the alignments are views into the inflated bytes: those move into the handle and live until the next run)
        b.rec_data.swap(bs.data);
        const uint8_t* const rec_base = b.rec_data.data();
        // 2. parse them (threads), 3. filter and intern barcode / read ids in file order
        const auto tc1 = std::chrono::steady_clock::now();
        RawVec<Aln>& parsed = b.parsed;
        parsed.resize_uninit(recs.size());
        {
            const int T = (int)(recs.size() / 1024 + 1);            // tasks of ~1024 records
            auto work = [&](int t) {
                const size_t lo = recs.size() * (size_t)t / (size_t)T, hi = recs.size() * (size_t)(t + 1) / (size_t)T;
                int32_t rt;
                for (size_t i = lo; i < hi; ++i) {
                    Aln& a = parsed[i];
                    parse_body(rec_base + recs[i].first, recs[i].second, a, rt);
                    a.n_ind = 0; a.qalen = 0;
                    for (uint32_t c : a.cigar) {
                        const unsigned op = c & 15;
                        if (op == 1 || op == 2) a.n_ind += c >> 4;
                        if (op == 0 || op == 1 || op == 7 || op == 8) a.qalen += c >> 4;
                    }
                    a.left_sp = (!a.cigar.empty() && (a.cigar[0] & 15) == 4) ? (a.cigar[0] >> 4) : 0u;
                    {
                        const std::string_view qn = a.qname;
                        const size_t c1 = qn.rfind(':');
                        const size_t c2 = (c1 == std::string::npos || c1 == 0) ? std::string::npos : qn.rfind(':', c1 - 1);
                        a.c1 = c1 == std::string::npos ? -1 : (int32_t)c1;
                        a.c2 = c2 == std::string::npos ? -1 : (int32_t)c2;
                        if (a.c2 >= 0) {                     // FNV-1a; equal strings are confirmed by memcmp at interning
                            uint64_t h = 1469598103934665603ull;
                            for (size_t k = c2 + 1; k < c1; ++k) { h ^= (uint8_t)qn[k]; h *= 1099511628211ull; }
                            a.h_bc = h;
                            for (size_t k = 0; k < c2; ++k) { h ^= (uint8_t)qn[k]; h *= 1099511628211ull; }
                            a.h_pair = h;
                        }
                    }
                    a.oflag = (uint8_t)(((a.flag & 0x40) ? 1 : 0) | ((a.flag & 0x80) ? 2 : 0) | ((a.flag & 0x10) ? 4 : 0) | (a.has_nm ? 8 : 0));
                }
            };
            Pool::get().run(T, b.io_threads, work);
        }
        const auto tc2 = std::chrono::steady_clock::now();
        bool have_first = false, have_end = false;
        b.cur_end = -1; b.cur_voff_end = 0;
        // 3a. which records the run keeps, the cursor checkpoints and the first malformed record: stretches of the file in parallel,
        // the stretches' first kept / first beyond-the-end / first malformed records combined in file order
        std::vector<uint8_t> keep(parsed.size(), 0);
        size_t n_keep = 0;
        {
            const size_t NPk = parsed.size(), NONE = ~(size_t)0;
            const int NK = (int)std::min<size_t>(256, NPk / 4096 + 1);
            struct Part { size_t n, first, beyond, bad; };
            std::vector<Part> part((size_t)NK, Part{0, NONE, NONE, NONE});
            Pool::get().run(NK, b.io_threads, [&](int c) {
                const size_t lo = NPk * (size_t)c / (size_t)NK, hi = NPk * (size_t)(c + 1) / (size_t)NK;
                Part P{0, NONE, NONE, NONE};
                for (size_t pi = lo; pi < hi; ++pi) {
                    const Aln& a = parsed[pi];
                    if ((a.flag & 4) || a.cigar.empty()) continue;
                    if (a.end <= start0) continue;
                    if (P.first == NONE) P.first = pi;
                    if (P.beyond == NONE && a.end > end0) P.beyond = pi;
                    if (a.c2 < 0 || a.l_seq == 0) { if (P.bad == NONE) P.bad = pi; continue; }
                    keep[pi] = 1; ++P.n;
                }
                part[(size_t)c] = P;
            });
            size_t first = NONE, beyond = NONE, bad = NONE;
            for (const Part& P : part) {
                n_keep += P.n;
                if (first == NONE) first = P.first;
                if (beyond == NONE) beyond = P.beyond;
                if (bad == NONE) bad = P.bad;
            }
            if (first != NONE) {                       // (its block_size word sits 4 bytes before the body)
                have_first = true;
                b.cur_tid = tid; b.cur_start = start0; b.cur_voff = bs.voffset_of(recs[first].first - 4);
            }
            if (beyond != NONE) {                      // first record a run starting at or after end0 can need
                have_end = true; (void)have_end;
                b.cur_end = end0; b.cur_voff_end = bs.voffset_of(recs[beyond].first - 4);
            }
            if (bad != NONE) {
                // qname -> barcode / read id (smCounter.py:320-325): <readid...>:<UMI>:<x>
                const Aln& a = parsed[bad];
                if (a.c2 < 0) { b.err = "read name '" + std::string(a.qname) + "' has fewer than 3 ':' fields"; return -3; }
                b.err = "alignment " + std::string(a.qname) + " has no sequence"; return -4;
            }
        }
        const auto tc2b = std::chrono::steady_clock::now();
        auto tc2c = tc2b;
        int SH = 1;
        auto bc_equal = [&](const Aln& o, const Aln& a) {
            return (o.c1 - o.c2) == (a.c1 - a.c2) && memcmp(o.qname.data() + o.c2 + 1, a.qname.data() + a.c2 + 1, (size_t)(a.c1 - a.c2 - 1)) == 0;
        };
        auto pair_equal = [&](const Aln& o, const Aln& a) {
            return o.c2 == a.c2 && bc_equal(o, a) && memcmp(o.qname.data(), a.qname.data(), (size_t)a.c2) == 0;
        };
        // 3b. run-wide barcode / read-name ids, numbered by first appearance in the file.  Every thread takes a stretch of the records
        // and enters their 64-bit hashes into one shared open-addressing table (a compare-and-swap claims a slot, an atomic minimum
        // keeps the slot's first record); a second pass confirms every record against its slot's first record (memcmp) and counts the
        // first records per stretch; a prefix over the stretches gives the ids.  Two different strings with one hash - never seen -
        // send the run to the sharded string-map path below.  Ids in file order are also what the device builder's sort likes: the
        // ids under a tile's window then lie in a narrow range (k_bp_sort_seg narrows its keys to it).
        bool collided = exp_env("SMC_BAM_SHARDS") != nullptr;                    // (tests: the sharded path)
        const size_t NP = parsed.size();
        const int NCH = (int)std::min<size_t>(256, NP / 2048 + 1);
        std::vector<uint32_t> nf_bc((size_t)NCH + 1, 0), nf_pair((size_t)NCH + 1, 0), n_kept((size_t)NCH + 1, 0);
        if (!collided) {
            size_t cap = 1024;
            while (cap < 2 * n_keep + 16) cap <<= 1;
            const size_t mask = cap - 1;
            b.it_key.resize_uninit(2 * cap); b.it_first.resize_uninit(2 * cap); b.it_slot.resize_uninit(2 * NP + 2);
            uint64_t* const key = b.it_key.p;
            uint32_t* const first = b.it_first.p;
            uint32_t* const slot = b.it_slot.p;
            {
                const int NZ = (int)std::min<size_t>(256, 2 * cap / 65536 + 1);
                Pool::get().run(NZ, b.io_threads, [&](int c) {
                    const size_t lo = 2 * cap * (size_t)c / (size_t)NZ, hi = 2 * cap * (size_t)(c + 1) / (size_t)NZ;
                    memset(key + lo, 0, 8 * (hi - lo)); memset(first + lo, 0xFF, 4 * (hi - lo));
                });
            }
            auto enter = [&](uint64_t h, size_t base, uint32_t pi) -> uint32_t {
                const uint64_t k = h ? h : 1ull;
                size_t i = (size_t)(k * 0x9E3779B97F4A7C15ull >> 20) & mask;
                for (;;) {
                    uint64_t cur = __atomic_load_n(&key[base + i], __ATOMIC_RELAXED);
                    if (cur == 0ull && __atomic_compare_exchange_n(&key[base + i], &cur, k, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
                    if (cur == k) break;
                    i = (i + 1) & mask;
                }
                uint32_t f = __atomic_load_n(&first[base + i], __ATOMIC_RELAXED);
                while (pi < f && !__atomic_compare_exchange_n(&first[base + i], &f, pi, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
                return (uint32_t)i;
            };
            Pool::get().run(NCH, b.io_threads, [&](int c) {
                const size_t lo = NP * (size_t)c / (size_t)NCH, hi = NP * (size_t)(c + 1) / (size_t)NCH;
                for (size_t pi = lo; pi < hi; ++pi) {
                    if (!keep[pi]) continue;
                    slot[pi] = enter(parsed[pi].h_bc, 0, (uint32_t)pi);
                    slot[NP + pi] = enter(parsed[pi].h_pair, cap, (uint32_t)pi);
                }
            });
            std::atomic<int> clash(0);
            Pool::get().run(NCH, b.io_threads, [&](int c) {
                const size_t lo = NP * (size_t)c / (size_t)NCH, hi = NP * (size_t)(c + 1) / (size_t)NCH;
                uint32_t fb = 0, fp = 0, nk = 0;
                bool bad = false;
                for (size_t pi = lo; pi < hi; ++pi) {
                    if (!keep[pi]) continue;
                    ++nk;
                    const Aln& a = parsed[pi];
                    const uint32_t ob = first[slot[pi]], op = first[cap + slot[NP + pi]];
                    if (ob == pi) ++fb; else if (!bc_equal(parsed[ob], a)) bad = true;
                    if (op == pi) ++fp; else if (!pair_equal(parsed[op], a)) bad = true;
                }
                nf_bc[(size_t)c + 1] = fb; nf_pair[(size_t)c + 1] = fp; n_kept[(size_t)c + 1] = nk;
                if (bad) clash.store(1);
            });
            collided = clash.load() != 0;
            if (!collided) {
                for (int c = 0; c < NCH; ++c) { nf_bc[(size_t)c + 1] += nf_bc[(size_t)c]; nf_pair[(size_t)c + 1] += nf_pair[(size_t)c]; n_kept[(size_t)c + 1] += n_kept[(size_t)c]; }
                n_bc = (int)nf_bc[(size_t)NCH]; n_pair = (int)nf_pair[(size_t)NCH];
                if (bc_names) bc_names->assign((size_t)n_bc, std::string());
                // the first records take their ids (file order) ...
                Pool::get().run(NCH, b.io_threads, [&](int c) {
                    const size_t lo = NP * (size_t)c / (size_t)NCH, hi = NP * (size_t)(c + 1) / (size_t)NCH;
                    int ib = (int)nf_bc[(size_t)c], ip = (int)nf_pair[(size_t)c];
                    for (size_t pi = lo; pi < hi; ++pi) {
                        if (!keep[pi]) continue;
                        Aln& a = parsed[pi];
                        if (first[slot[pi]] == pi) {
                            if (bc_names) (*bc_names)[(size_t)ib] = std::string(a.qname.substr((size_t)a.c2 + 1, (size_t)(a.c1 - a.c2 - 1)));
                            a.bc_gid = ib++;
                        }
                        if (first[cap + slot[NP + pi]] == pi) a.pair_gid = ip++;
                    }
                });
                tc2c = std::chrono::steady_clock::now();
                // ... the others copy theirs from their slot's first record; the kept records, in file order, are the run
                const size_t base = reads.size();
                reads.resize_uninit(base + n_keep);
                Pool::get().run(NCH, b.io_threads, [&](int c) {
                    const size_t lo = NP * (size_t)c / (size_t)NCH, hi = NP * (size_t)(c + 1) / (size_t)NCH;
                    size_t o = base + n_kept[(size_t)c];
                    for (size_t pi = lo; pi < hi; ++pi) {
                        if (!keep[pi]) continue;
                        Aln& a = parsed[pi];
                        const uint32_t ob = first[slot[pi]], op = first[cap + slot[NP + pi]];
                        if (ob != pi) a.bc_gid = parsed[ob].bc_gid;
                        if (op != pi) a.pair_gid = parsed[op].pair_gid;
                        reads[o++] = a;
                    }
                });
            }
        }
        if (collided) {
            // The earlier path: ids by 64-bit hash, a hit confirmed against the first record that produced the id (memcmp) and a
            // colliding hash resolved through a string map.  Sharded by hash over the threads: a shard owns its hashes, so ids are
            // dense and exact whatever the thread count (they number DISTINCT strings, shard by shard).
        while (SH < 64 && SH * 2 <= b.io_threads && (size_t)SH * 2048 < n_keep) SH *= 2;
        if (const char* e = exp_env("SMC_BAM_SHARDS")) { SH = 1; while (SH < 64 && SH * 2 <= atoi(e)) SH *= 2; }   // (tests)
        // the kept records are first binned by shard (stretches of the file in parallel, a bin per stretch and shard; a
        // shard then walks its bins stretch by stretch, i.e. in file order) so that a shard touches only its own records
        const int NB = SH == 1 ? 1 : (int)std::min<size_t>(256, parsed.size() / 2048 + 1);
        std::vector<std::vector<uint32_t>> bins_bc((size_t)NB * (size_t)SH), bins_pair((size_t)NB * (size_t)SH);
        if (SH > 1) {
            const uint64_t msk = (uint64_t)SH - 1;
            Pool::get().run(NB, b.io_threads, [&](int c) {
                const size_t lo = parsed.size() * (size_t)c / (size_t)NB, hi = parsed.size() * (size_t)(c + 1) / (size_t)NB;
                for (size_t pi = lo; pi < hi; ++pi) {
                    if (!keep[pi]) continue;
                    bins_bc[(size_t)c * SH + (size_t)((parsed[pi].h_bc >> 20) & msk)].push_back((uint32_t)pi);
                    bins_pair[(size_t)c * SH + (size_t)((parsed[pi].h_pair >> 20) & msk)].push_back((uint32_t)pi);
                }
            });
        }
        // open-addressing table hash -> id (linear probing, grows at half load): one allocation, nothing to free node by node
        struct FlatMap {
            std::vector<uint64_t> key;
            std::vector<int> val;                                       // -1: empty
            size_t mask = 0, used = 0;
            void init(size_t expect) {
                size_t c = 64;
                while (c < 2 * expect) c <<= 1;
                key.assign(c, 0); val.assign(c, -1); mask = c - 1; used = 0;
            }
            int* find(uint64_t k) {
                for (size_t i = (size_t)(k * 0x9E3779B97F4A7C15ull >> 20) & mask;; i = (i + 1) & mask) {
                    if (val[i] < 0) return nullptr;
                    if (key[i] == k) return &val[i];
                }
            }
            void emplace(uint64_t k, int v) {
                if (2 * (used + 1) > mask + 1) {
                    std::vector<uint64_t> ok; std::vector<int> ov;
                    ok.swap(key); ov.swap(val);
                    init(ok.size());
                    for (size_t i = 0; i < ok.size(); ++i) if (ov[i] >= 0) emplace(ok[i], ov[i]);
                }
                size_t i = (size_t)(k * 0x9E3779B97F4A7C15ull >> 20) & mask;
                while (val[i] >= 0) i = (i + 1) & mask;
                key[i] = k; val[i] = v; ++used;
            }
        };
        struct Shard {
            FlatMap hb, hp;
            std::vector<size_t> bc_rep, pair_rep;                       // index into `parsed` of the id's first record
            std::unordered_map<std::string, int> bc_str, pair_str;     // only for colliding hashes
        };
        std::vector<Shard> shards((size_t)SH);
        auto intern = [&](int t) {
            Shard& S = shards[(size_t)t];
            {
                // (barcodes: a few reads each at least; read names: two alignments each, typically)
                size_t mine = 0;
                if (SH == 1) mine = n_keep;
                else for (int c = 0; c < NB; ++c) mine += bins_pair[(size_t)c * SH + (size_t)t].size();
                S.hb.init(mine / 8 + 64); S.hp.init(mine / 2 + 64);
            }
            auto do_bc = [&](size_t pi) {
                Aln& a = parsed[pi];
                const std::string_view qn = a.qname;
                const size_t c1 = (size_t)a.c1, c2 = (size_t)a.c2;
                {
                    const int* it = S.hb.find(a.h_bc);
                    if (it && bc_equal(parsed[S.bc_rep[(size_t)*it]], a)) a.bc_gid = *it;
                    else if (!it) { a.bc_gid = (int)S.bc_rep.size(); S.hb.emplace(a.h_bc, a.bc_gid); S.bc_rep.push_back(pi); }
                    else {                                   // hash collision: the string decides
                        auto r = S.bc_str.emplace(std::string(qn.substr(c2 + 1, c1 - c2 - 1)), (int)S.bc_rep.size());
                        if (r.second) S.bc_rep.push_back(pi);
                        a.bc_gid = r.first->second;
                    }
                }
            };
            auto do_pair = [&](size_t pi) {
                Aln& a = parsed[pi];
                const std::string_view qn = a.qname;
                const size_t c1 = (size_t)a.c1, c2 = (size_t)a.c2;
                {
                    const int* it = S.hp.find(a.h_pair);
                    const bool hit = it && [&] {
                        const Aln& o = parsed[S.pair_rep[(size_t)*it]];
                        return o.c2 == a.c2 && bc_equal(o, a) && memcmp(o.qname.data(), qn.data(), c2) == 0;
                    }();
                    if (hit) a.pair_gid = *it;
                    else if (!it) { a.pair_gid = (int)S.pair_rep.size(); S.hp.emplace(a.h_pair, a.pair_gid); S.pair_rep.push_back(pi); }
                    else {
                        auto r = S.pair_str.emplace(std::string(qn.substr(c2 + 1, c1 - c2 - 1)) + "\x01" + std::string(qn.substr(0, c2)), (int)S.pair_rep.size());
                        if (r.second) S.pair_rep.push_back(pi);
                        a.pair_gid = r.first->second;
                    }
                }
            };
            if (SH == 1) {
                for (size_t pi = 0; pi < parsed.size(); ++pi) if (keep[pi]) { do_bc(pi); do_pair(pi); }
                return;
            }
            for (int c = 0; c < NB; ++c) for (uint32_t pi : bins_bc[(size_t)c * SH + (size_t)t]) do_bc(pi);
            for (int c = 0; c < NB; ++c) for (uint32_t pi : bins_pair[(size_t)c * SH + (size_t)t]) do_pair(pi);
        };
        Pool::get().run(SH, b.io_threads, intern);
        std::vector<int> bc_off((size_t)SH + 1, 0), pair_off((size_t)SH + 1, 0);
        for (int t = 0; t < SH; ++t) {
            bc_off[(size_t)t + 1] = bc_off[(size_t)t] + (int)shards[(size_t)t].bc_rep.size();
            pair_off[(size_t)t + 1] = pair_off[(size_t)t] + (int)shards[(size_t)t].pair_rep.size();
        }
        if (bc_names) {
            bc_names->assign((size_t)bc_off[(size_t)SH], std::string());
            for (int t = 0; t < SH; ++t)
                for (size_t k = 0; k < shards[(size_t)t].bc_rep.size(); ++k) {
                    const Aln& o = parsed[shards[(size_t)t].bc_rep[k]];
                    (*bc_names)[(size_t)bc_off[(size_t)t] + k] = std::string(o.qname.substr((size_t)o.c2 + 1, (size_t)(o.c1 - o.c2 - 1)));
                }
        }
        tc2c = std::chrono::steady_clock::now();
        {
            // the kept records, in file order, with the shard-local ids made run-wide (stretches in parallel)
            const uint64_t msk = (uint64_t)SH - 1;
            const int NC = (int)std::min<size_t>(256, parsed.size() / 4096 + 1);
            std::vector<size_t> first((size_t)NC + 1, 0);
            for (int c = 0; c < NC; ++c) {
                const size_t lo = parsed.size() * (size_t)c / (size_t)NC, hi = parsed.size() * (size_t)(c + 1) / (size_t)NC;
                size_t k = 0;
                for (size_t pi = lo; pi < hi; ++pi) k += keep[pi];
                first[(size_t)c + 1] = first[(size_t)c] + k;
            }
            const size_t base = reads.size();
            reads.resize_uninit(base + n_keep);
            Pool::get().run(NC, b.io_threads, [&](int c) {
                const size_t lo = parsed.size() * (size_t)c / (size_t)NC, hi = parsed.size() * (size_t)(c + 1) / (size_t)NC;
                size_t o = base + first[(size_t)c];
                for (size_t pi = lo; pi < hi; ++pi) {
                    if (!keep[pi]) continue;
                    Aln& a = parsed[pi];
                    a.bc_gid += bc_off[(size_t)((a.h_bc >> 20) & msk)];
                    a.pair_gid += pair_off[(size_t)((a.h_pair >> 20) & msk)];
                    reads[o++] = a;
                }
            });
        }
        n_bc = bc_off[(size_t)SH]; n_pair = pair_off[(size_t)SH];
        }
        if (exp_env("SMC_BAM_TIMING")) {
            const auto tc3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "collect_reads: %zu records, %zu kept: inflate + boundaries %.1f ms (read %.1f, inflate %.1f), parse %.1f ms, keep %.1f ms, intern %.1f ms, gather %.1f ms (%zu KB inflated, %d shards; record walk: %d pieces, %d not joined at once)\n",
                    recs.size(), reads.size(), ms(tc0, tc1), bs.t_read, bs.t_inflate, ms(tc1, tc2), ms(tc2, tc2b), ms(tc2b, tc2c), ms(tc2c, tc3), b.rec_data.size() >> 10, SH, n_pieces, n_rewalked);
        }
        if (!have_first) { b.cur_tid = tid; b.cur_start = start0; b.cur_voff = recs.empty() ? voff : bs.voffset_of(recs.back().first - 4); }
        if (!b.err.empty()) return -2;                  // corrupt / truncated BGZF
    }
    return 0;
}

// (qpos, is_del, indel) of `a` at reference position p0 (samtools 0.1.19 resolve_cigar); false = not covered
inline bool resolve_column(const Aln& a, int64_t p0, int& qpos, bool& isdel, int& indel) {
    int64_t x = a.pos, y = 0;
    qpos = -1; indel = 0; isdel = false;
    for (size_t k = 0; k < a.cigar.size(); ++k) {
        const unsigned op = a.cigar[k] & 15; const int64_t l = a.cigar[k] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            if (x <= p0 && p0 < x + l) {
                qpos = (int)(y + (p0 - x));
                if (p0 == x + l - 1 && k + 1 < a.cigar.size()) {
                    const unsigned nop = a.cigar[k + 1] & 15; const int nl = (int)(a.cigar[k + 1] >> 4);
                    if (nop == 1) indel = nl; else if (nop == 2) indel = -nl;
                }
                return true;
            }
            x += l; y += l;
        } else if (op == 1 || op == 4) {
            y += l;
        } else if (op == 2 || op == 3) {
            if (x <= p0 && p0 < x + l) {
                qpos = (int)y; isdel = true;
                // the peek at the next operation applies to every current operation (resolve_cigar2): the last column
                // of a D/N block followed by I or D carries that indel, and the reference tests `indel` before
                // `is_del` (smCounter.py:371,392,416)
                if (p0 == x + l - 1 && k + 1 < a.cigar.size()) {
                    const unsigned nop = a.cigar[k + 1] & 15; const int nl = (int)(a.cigar[k + 1] >> 4);
                    if (nop == 1) indel = nl; else if (nop == 2) indel = -nl;
                }
                if (indel != 0 && qpos >= (int)a.l_seq) indel = 0;   // no base to name the allele with: plain 'DEL'
                return true;
            }
            x += l;
        }
    }
    return false;
}

// allele id of what `a` shows at the column: 0-5 fixed, 6+ index into `extra` (appended on first sight)
inline int allele_of(const Aln& a, int qpos, bool isdel, int indel, std::vector<std::string>& extra) {
    if (isdel && indel == 0) return 5;
    const char site = a.seq[(size_t)qpos];
    if (indel == 0) {
        switch (site) { case 'A': return 0; case 'T': return 1; case 'G': return 2; case 'C': return 3; case 'N': return 4; default: break; }
    }
    std::string key;
    if (indel > 0) key = std::string("INS|") + site + "|" + site + a.seq.substr((size_t)qpos + 1, (size_t)indel);
    else if (indel < 0) key = "D" + std::to_string(-indel) + "|" + site;
    else key = std::string(1, site);
    size_t k = 0;
    while (k < extra.size() && extra[k] != key) ++k;
    if (k == extra.size()) extra.push_back(key);
    return 6 + (int)k;
}

// The large buffers of a closed handle (inflated records, compressed stretch, alignment arrays: tens to hundreds of
// megabytes) are parked here for the next handle of the process instead of being unmapped: unmapping takes milliseconds
// - on the closing thread, or, from a helper thread, through the address-space lock on every thread that allocates meanwhile.
struct SpareBuffers {
    std::mutex m;
    ByteBuf rec, comp;
    RawVec<Aln> reads, parsed;
};
static SpareBuffers& spare_buffers() { static SpareBuffers* s = new SpareBuffers; return *s; }
template <class B> static void keep_larger(B& spare, B& mine) {
    if (mine.cap > spare.cap) { std::swap(spare.p, mine.p); std::swap(spare.cap, mine.cap); spare.n = 0; mine.n = 0; }
}

static void take_spare(Bam& b) {
    SpareBuffers& S = spare_buffers();
    std::lock_guard<std::mutex> lk(S.m);
    keep_larger(b.rec_data, S.rec); keep_larger(b.comp, S.comp);
    keep_larger(b.d_reads, S.reads); keep_larger(b.parsed, S.parsed);
}

}  // namespace

extern "C" {

int smc_bam_open(const char* path, void** out) {
    Bam* b = new Bam();
    b->fh = fopen(path, "rb");
    if (!b->fh) { delete b; return -1; }
    if (!exp_env("SMC_BAM_NO_MMAP")) {
        struct stat sb;
        if (fstat(fileno(b->fh), &sb) == 0 && sb.st_size > 0) {
            void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fileno(b->fh), 0);
            if (m != MAP_FAILED) { b->map = (const uint8_t*)m; b->map_len = (size_t)sb.st_size; }
        }
    }
    b->load_block(0);
    char magic[4];
    int32_t l_text, n_ref;
    if (b->read(magic, 4) != 4 || memcmp(magic, "BAM\1", 4) != 0) { if (b->map) munmap((void*)b->map, b->map_len); fclose(b->fh); delete b; return -2; }
    b->read(&l_text, 4);
    std::vector<char> text((size_t)l_text);
    b->read(text.data(), text.size());
    b->read(&n_ref, 4);
    for (int i = 0; i < n_ref; ++i) {
        int32_t l_name, l_ref;
        b->read(&l_name, 4);
        std::string nm((size_t)l_name, '\0');
        b->read(&nm[0], (size_t)l_name);
        nm.resize(l_name ? l_name - 1 : 0);
        b->read(&l_ref, 4);
        b->ref_names.push_back(nm);
        b->ref_lens.push_back(l_ref);
    }
    b->first_record = b->tell();
    // BAI linear index (path + ".bai")
    std::string bai = std::string(path) + ".bai";
    FILE* f = fopen(bai.c_str(), "rb");
    if (f) {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> d((size_t)sz);
        if (fread(d.data(), 1, d.size(), f) == d.size() && sz >= 8 && memcmp(d.data(), "BAI\1", 4) == 0) {
            size_t o = 8;
            int32_t nr; memcpy(&nr, d.data() + 4, 4);
            for (int r = 0; r < nr && o + 4 <= d.size(); ++r) {
                int32_t n_bin; memcpy(&n_bin, d.data() + o, 4); o += 4;
                for (int k = 0; k < n_bin; ++k) { int32_t nc; memcpy(&nc, d.data() + o + 4, 4); o += 8 + 16 * (size_t)nc; }
                int32_t n_intv; memcpy(&n_intv, d.data() + o, 4); o += 4;
                std::vector<uint64_t> iv((size_t)n_intv);
                memcpy(iv.data(), d.data() + o, 8 * (size_t)n_intv); o += 8 * (size_t)n_intv;
                b->lin.push_back(iv);
            }
        }
        fclose(f);
    }
    take_spare(*b);
    *out = b;
    return 0;
}

void smc_bam_close(void* h) {
    Bam* b = (Bam*)h;
    if (!b) return;
    if (b->map) munmap((void*)b->map, b->map_len);
    b->map = nullptr;
    if (b->fh) fclose(b->fh);
    b->fh = nullptr;
    {
        SpareBuffers& S = spare_buffers();
        std::lock_guard<std::mutex> lk(S.m);
        keep_larger(S.rec, b->rec_data); keep_larger(S.comp, b->comp);
        keep_larger(S.reads, b->d_reads); keep_larger(S.parsed, b->parsed);
    }
    delete b;
}

// compressed bytes of the file between the first alignments of the 16 kb windows holding start0 and the one behind end0,
// from the linear index; -1 = not known (no index, reference not found).  Coarse: what a caller sizes a first run with.
int64_t smc_bam_span_bytes(void* h, const char* chrom, int64_t start0, int64_t end0) {
    Bam& b = *(Bam*)h;
    int tid = -1;
    for (size_t i = 0; i < b.ref_names.size(); ++i) if (b.ref_names[i] == chrom) tid = (int)i;
    if (tid < 0 || (size_t)tid >= b.lin.size() || b.lin[(size_t)tid].empty() || end0 <= start0) return -1;
    const auto& iv = b.lin[(size_t)tid];
    long w0 = (long)std::min<int64_t>(start0 >> 14, (int64_t)iv.size() - 1);
    while (w0 >= 0 && iv[(size_t)w0] == 0) --w0;
    if (w0 < 0) return -1;
    uint64_t v1 = 0;
    for (size_t w = (size_t)(end0 >> 14) + 1; w < iv.size(); ++w) if (iv[w]) { v1 = iv[w]; break; }
    if (!v1) {                                         // the run reaches the last indexed window: up to the end of the file
        struct stat stt;
        if (fstat(fileno(b.fh), &stt) != 0) return -1;
        v1 = (uint64_t)stt.st_size << 16;
    }
    const uint64_t c0 = iv[(size_t)w0] >> 16, c1 = v1 >> 16;
    return c1 > c0 ? (int64_t)(c1 - c0) : 0;
}

int smc_bam_n_refs(void* h) { return (int)((Bam*)h)->ref_names.size(); }
const char* smc_bam_ref_name(void* h, int i) { return ((Bam*)h)->ref_names[(size_t)i].c_str(); }
int64_t smc_bam_ref_len(void* h, int i) { return ((Bam*)h)->ref_lens[(size_t)i]; }
const char* smc_bam_error(void* h) { return ((Bam*)h)->err.c_str(); }

// Pileup of positions start0, start0+1, ... < end0 on `chrom`; stops after the locus at which the batch
// reaches max_reads pileup reads.  *n_loci_done = loci emitted.  Returns the number of pileup reads, or < 0:
// -3 a read name with fewer than 3 ':' fields, -4 no sequence, -5 > 255 alleles at a locus.
// (An unknown chromosome is not an error: every locus is empty, like pysam's pileup of nothing.)
int64_t smc_bam_pileup(void* h, const char* chrom, int64_t start0, int64_t end0, int64_t max_reads, int64_t* n_loci_done) {
    Bam& b = *(Bam*)h;
    b.umi.clear(); b.frag.clear(); b.nm.clear(); b.n_indel.clear(); b.left_sp.clear(); b.qlen.clear(); b.qalen.clear();
    b.qpos.clear(); b.indel.clear(); b.flag.clear(); b.mq.clear(); b.is_del.clear(); b.allele.clear(); b.bq.clear();
    b.read_off.assign(1, 0); b.keys.clear(); b.n_keys.clear();
    *n_loci_done = 0;
    RawVec<Aln> reads;
    int n_bc = 0, n_pair = 0;
    { const int rc = collect_reads(b, chrom, start0, end0, reads, n_bc, n_pair); if (rc) return rc; }
    // per-locus dense ids through epoch-stamped tables over the run-wide ids
    std::vector<int64_t> bc_stamp((size_t)n_bc, -1), pair_stamp((size_t)n_pair, -1);
    std::vector<int32_t> bc_local((size_t)n_bc), pair_local((size_t)n_pair), n_frag_of;
    size_t w0 = 0;
    std::vector<std::string> extra;
    for (int64_t p0 = start0; p0 < end0; ++p0) {
        while (w0 < reads.size() && reads[w0].end <= p0) ++w0;
        extra.clear();
        n_frag_of.clear();
        for (size_t ri = w0; ri < reads.size(); ++ri) {
            const Aln& a = reads[ri];
            if (a.pos > p0) break;
            if (p0 >= a.end) continue;
            int qpos, indel; bool isdel;
            if (!resolve_column(a, p0, qpos, isdel, indel)) continue;
            int u, f;
            if (bc_stamp[(size_t)a.bc_gid] != p0) {
                bc_stamp[(size_t)a.bc_gid] = p0; u = bc_local[(size_t)a.bc_gid] = (int)n_frag_of.size(); n_frag_of.push_back(0);
            } else u = bc_local[(size_t)a.bc_gid];
            if (pair_stamp[(size_t)a.pair_gid] != p0) {
                pair_stamp[(size_t)a.pair_gid] = p0; f = pair_local[(size_t)a.pair_gid] = n_frag_of[(size_t)u]++;
            } else f = pair_local[(size_t)a.pair_gid];
            const int ai = allele_of(a, qpos, isdel, indel, extra);
            const int bqv = (isdel && indel == 0) ? 0 : a.qual[(size_t)qpos];
            if (ai > 255) { b.err = "more than 255 alleles at one locus"; return -5; }
            b.umi.push_back((uint32_t)u); b.frag.push_back((uint32_t)f);
            b.flag.push_back(a.oflag);
            b.mq.push_back(a.mapq); b.nm.push_back(a.nm); b.n_indel.push_back(a.n_ind);
            b.left_sp.push_back(a.left_sp);
            b.qlen.push_back(a.l_seq); b.qalen.push_back(a.qalen);
            b.qpos.push_back(qpos); b.indel.push_back(indel); b.is_del.push_back(isdel ? 1 : 0);
            b.allele.push_back((uint8_t)ai); b.bq.push_back((uint8_t)bqv);
        }
        b.read_off.push_back((int64_t)b.umi.size());
        b.n_keys.push_back((int32_t)extra.size());
        for (const std::string& k : extra) { b.keys += k; b.keys += '\n'; }
        ++*n_loci_done;
        if ((int64_t)b.umi.size() >= max_reads) break;
    }
    return (int64_t)b.umi.size();
}

int64_t smc_bam_keys_len(void* h) { return (int64_t)((Bam*)h)->keys.size(); }
const char* smc_bam_ds_info(void* h) { return ((Bam*)h)->ds_info.c_str(); }

// copy the last pileup into caller-owned arrays (sizes from smc_bam_pileup / n loci = end0 - start0)
void smc_bam_copy(void* h, uint32_t* umi, uint32_t* frag, uint8_t* flag, uint8_t* mq, uint32_t* nm, uint32_t* n_indel,
                  uint32_t* left_sp, uint32_t* qlen, uint32_t* qalen, int32_t* qpos, int32_t* indel, uint8_t* is_del,
                  uint8_t* allele, uint8_t* bq, int64_t* read_off, int32_t* n_keys, char* keys) {
    Bam& b = *(Bam*)h;
    const size_t n = b.umi.size();
    memcpy(umi, b.umi.data(), 4 * n); memcpy(frag, b.frag.data(), 4 * n); memcpy(flag, b.flag.data(), n);
    memcpy(mq, b.mq.data(), n); memcpy(nm, b.nm.data(), 4 * n); memcpy(n_indel, b.n_indel.data(), 4 * n);
    memcpy(left_sp, b.left_sp.data(), 4 * n); memcpy(qlen, b.qlen.data(), 4 * n); memcpy(qalen, b.qalen.data(), 4 * n);
    memcpy(qpos, b.qpos.data(), 4 * n); memcpy(indel, b.indel.data(), 4 * n); memcpy(is_del, b.is_del.data(), n);
    memcpy(allele, b.allele.data(), n); memcpy(bq, b.bq.data(), n);
    memcpy(read_off, b.read_off.data(), 8 * b.read_off.size());
    memcpy(n_keys, b.n_keys.data(), 4 * b.n_keys.size());
    memcpy(keys, b.keys.data(), b.keys.size());
}

// ---------------------------------------------------------------------------------------------------
// Fused decode -> device planes: what bamio.iter_pileup_batches + features.extract_features produce,
// without the intermediate columns.  Per read (smCounter.py:327-366, :432-452):
//   meta = allele | bq << 8 | flags << 16 | mapq << 24, flags = R2 | reverse << 1 | mmOK << 2 | kind << 3
//   dist = distToBcEnd | distToPrimerEnd << 16 (regular bases only, saturated to 16 bits)
// Reads of a locus are emitted barcode-major (barcode, fragment slot, pileup order), each locus padded to a
// 4-read boundary; umi_start holds the first read of every barcode (+ closing entry) per locus.
// refseq: the upper-cased reference letters of [start0, end0).  Loci are processed by `nthreads` threads.
// Returns pileup reads (unpadded), or < 0: -3/-4/-5 as smc_bam_pileup, -6 first read of a locus has neither
// READ1 nor READ2, -7 base quality > 126, -8 more than 64 alleles at a locus.
// ds > 0: for every locus with more barcodes than ds, the barcodes that own an included read (bq >= min_bq or inside a
// deletion, mapq >= min_mq, mismatches within mismatch_thr: incCond, smCounter.py:378) are listed with their text in
// order of that read (smc_bam_ds_info) - the host needs them for the reference's down-sampling (:496-498).
// The planes and the descriptors are written straight into caller memory: once the sizes are known `alloc(ctx, n_slots,
// n_loci, out)` is called and must fill out[0..3] with four uint32[n_slots] buffers (meta, umi, frag, dist) and out[4]
// with an smc_locus[n_loci] buffer (uninitialised memory is fine: every entry, padding included, is written).
typedef void (*smc_planes_alloc)(void* ctx, int64_t n_slots, int64_t n_loci, void** out);
int64_t smc_bam_planes(void* h, const char* chrom, int64_t start0, int64_t end0, int64_t max_reads, double mismatch_thr,
                       const char* refseq, int nthreads, int ds, int min_bq, int min_mq, int primer_dist,
                       smc_planes_alloc alloc, void* alloc_ctx, int64_t* n_loci_done, int64_t* n_slots,
                       int64_t* n_umi_start) {
    Bam& b = *(Bam*)h;
    b.keys.clear(); b.n_keys.clear(); b.ds_info.clear();
    b.io_threads = nthreads;
    *n_loci_done = *n_slots = *n_umi_start = 0;
    RawVec<Aln> reads;
    int n_bc = 0, n_pair = 0;
    const auto t_0 = std::chrono::steady_clock::now();
    std::vector<std::string> bc_names;
    { const int rc = collect_reads(b, chrom, start0, end0, reads, n_bc, n_pair, ds > 0 ? &bc_names : nullptr); if (rc) return rc; }
    const auto t_1 = std::chrono::steady_clock::now();
    // coverage per position: every read covers exactly [pos, end) (M/=/X/D/N are contiguous on the reference)
    const int64_t span = end0 - start0;
    std::vector<int64_t> cov((size_t)span + 1, 0);
    for (const Aln& a : reads) {
        const int64_t lo = std::max<int64_t>(a.pos, start0), hi = std::min<int64_t>(a.end, end0);
        if (lo < hi) { ++cov[(size_t)(lo - start0)]; --cov[(size_t)(hi - start0)]; }
    }
    int64_t nl = 0, total = 0, run = 0;
    std::vector<int64_t> off;   // padded slot offset per locus (+ end)
    off.push_back(0);
    std::vector<int64_t> cum;   // unpadded cumulative reads
    for (int64_t k = 0; k < span; ++k) {
        run += cov[(size_t)k];
        total += run;
        off.push_back(off.back() + (run + 3) / 4 * 4);
        cum.push_back(total);
        ++nl;
        if (total >= max_reads) break;
    }
    const int64_t slots = off.back();
    void* bufs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    alloc(alloc_ctx, slots, nl, bufs);
    uint32_t* const pm = (uint32_t*)bufs[0]; uint32_t* const pu = (uint32_t*)bufs[1];
    uint32_t* const pf = (uint32_t*)bufs[2]; uint32_t* const pd = (uint32_t*)bufs[3];
    smc_locus* const ploci = (smc_locus*)bufs[4];
    if ((slots && (!pm || !pu || !pf || !pd)) || (nl && !ploci)) { b.err = "smc_bam_planes: allocation callback returned no memory"; return -9; }
    b.n_keys.assign((size_t)nl, 0);
    const auto t_a = std::chrono::steady_clock::now();
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, nl));
    std::vector<int64_t> cut((size_t)T + 1, nl);   // loci ranges of roughly equal read counts
    cut[0] = 0;
    for (int t = 1; t < T; ++t) {
        const int64_t want = total * t / T;
        cut[(size_t)t] = std::lower_bound(cum.begin(), cum.end(), want) - cum.begin();
        cut[(size_t)t] = std::max(cut[(size_t)t], cut[(size_t)t - 1]);
    }
    std::vector<std::vector<uint32_t>> t_ustart((size_t)T);
    std::vector<std::string> t_keys((size_t)T), t_ds((size_t)T);
    std::atomic<int> err(0);
    std::vector<std::string> t_err((size_t)T);
    auto work = [&](int t) {
        std::vector<int64_t> bc_stamp((size_t)n_bc, -1), pair_stamp((size_t)n_pair, -1);
        std::vector<int32_t> bc_local((size_t)n_bc), pair_local((size_t)n_pair), n_frag_of, n_reads_of, slot_base, slot_cnt;
        std::vector<uint32_t> c_meta, c_umi, c_frag, c_dist;
        std::vector<uint8_t> c_class;
        std::vector<int32_t> gid_of_u, inc_order;
        std::vector<uint8_t> inc_seen;
        std::vector<std::string> extra;
        size_t w0 = 0;
        for (int64_t l = cut[(size_t)t]; l < cut[(size_t)t + 1] && !err.load(std::memory_order_relaxed); ++l) {
            const int64_t p0 = start0 + l;
            while (w0 < reads.size() && reads[w0].end <= p0) ++w0;
            extra.clear(); n_frag_of.clear(); n_reads_of.clear(); gid_of_u.clear(); inc_order.clear(); inc_seen.clear();
            c_meta.clear(); c_umi.clear(); c_frag.clear(); c_dist.clear(); c_class.clear();
            bool r2 = false, first = true;
            for (size_t ri = w0; ri < reads.size(); ++ri) {
                const Aln& a = reads[ri];
                if (a.pos > p0) break;
                if (p0 >= a.end) continue;
                int qpos, indel; bool isdel;
                if (!resolve_column(a, p0, qpos, isdel, indel)) continue;
                int u, f;
                if (bc_stamp[(size_t)a.bc_gid] != p0) {
                    bc_stamp[(size_t)a.bc_gid] = p0; u = bc_local[(size_t)a.bc_gid] = (int)n_frag_of.size();
                    n_frag_of.push_back(0); n_reads_of.push_back(0); gid_of_u.push_back(a.bc_gid); inc_seen.push_back(0);
                } else u = bc_local[(size_t)a.bc_gid];
                if (pair_stamp[(size_t)a.pair_gid] != p0) {
                    pair_stamp[(size_t)a.pair_gid] = p0; f = pair_local[(size_t)a.pair_gid] = n_frag_of[(size_t)u]++;
                } else f = pair_local[(size_t)a.pair_gid];
                ++n_reads_of[(size_t)u];
                const int ai = allele_of(a, qpos, isdel, indel, extra);
                if (ai >= SMC_MAX_ALLELES) { t_err[(size_t)t] = "more than 64 distinct alleles at " + std::string(chrom) + ":" + std::to_string(p0 + 1); err = -8; return; }
                const unsigned bq = (isdel && indel == 0) ? 0u : a.qual[(size_t)qpos];
                if (bq > 126) { t_err[(size_t)t] = "base quality " + std::to_string(bq) + " > 126 in " + std::string(a.qname); err = -7; return; }
                // pairOrder: R2 wins over R1; neither -> the previous read's value (smCounter.py:359-362)
                if (a.oflag & 3) r2 = (a.oflag & 2) != 0;
                else if (first) { t_err[(size_t)t] = "first pileup read at " + std::string(chrom) + ":" + std::to_string(p0 + 1) + " has neither read1 nor read2 set"; err = -6; return; }
                first = false;
                const bool rev = (a.oflag & 4) != 0;
                const int64_t mism = std::max<int64_t>(0, (int64_t)a.nm - (int64_t)a.n_ind);
                const double mm100 = a.l_seq > 0 ? 100.0 * (double)mism / (double)a.l_seq : 0.0;
                const unsigned kind = indel > 0 ? 2u : indel < 0 ? 3u : isdel ? 1u : 0u;
                const unsigned flags = (r2 ? 1u : 0u) | (rev ? 2u : 0u) | (mm100 <= mismatch_thr ? 4u : 0u) | kind << 3;
                uint32_t dist = 0;
                if (kind == 0) {
                    const int64_t rel = (int64_t)qpos - (int64_t)a.left_sp, far = (int64_t)a.qalen - rel;
                    const int64_t dbc = r2 ? (rev ? rel : far) : (rev ? far : rel), dpr = r2 ? (rev ? far : rel) : 0;
                    dist = (uint32_t)std::min<int64_t>(65535, std::max<int64_t>(0, dbc)) | (uint32_t)std::min<int64_t>(65535, std::max<int64_t>(0, dpr)) << 16;
                }
                const bool bq_ok = (int)bq >= min_bq;
                const bool inc = (bq_ok || kind == 1) && (int)a.mapq >= min_mq && (flags & 4u);     // incCond, :378
                if (ds > 0 && !inc_seen[(size_t)u] && inc) { inc_seen[(size_t)u] = 1; inc_order.push_back(u); }
                c_class.push_back((uint8_t)smc_read_class((int)kind, rev, r2, inc, bq_ok, (dist & 0xffffu) <= 20u,
                                                         (int)(dist >> 16) <= primer_dist));
                c_meta.push_back((uint32_t)ai | (kind == 1 ? (unsigned)min_bq : bq) << 8 | flags << 16 | (uint32_t)a.mapq << 24);   // in-deletion: minBQ, :418
                c_umi.push_back((uint32_t)u); c_frag.push_back((uint32_t)f); c_dist.push_back(dist);
            }
            // fragment slots UMI-major, then a stable counting sort of the reads by slot
            const size_t nu = n_frag_of.size(), n = c_meta.size();
            slot_base.assign(nu + 1, 0);
            for (size_t u = 0; u < nu; ++u) slot_base[u + 1] = slot_base[u] + n_frag_of[u];
            const size_t nf = (size_t)slot_base[nu];
            slot_cnt.assign(nf + 1, 0);
            for (size_t i = 0; i < n; ++i) { c_frag[i] += (uint32_t)slot_base[c_umi[i]]; ++slot_cnt[c_frag[i] + 1]; }
            for (size_t k = 0; k < nf; ++k) slot_cnt[k + 1] += slot_cnt[k];
            const size_t o = (size_t)off[(size_t)l];
            for (size_t i = 0; i < n; ++i) {
                const size_t d = o + (size_t)slot_cnt[c_frag[i]]++;
                pm[d] = c_meta[i]; pu[d] = c_umi[i]; pf[d] = c_frag[i] | (uint32_t)c_class[i] << SMC_FRAG_CLASS_SHIFT; pd[d] = c_dist[i];
            }
            for (size_t d = o + n; d < (size_t)off[(size_t)l + 1]; ++d) pm[d] = pu[d] = pf[d] = pd[d] = 0u;   // padding
            smc_locus& L = ploci[(size_t)l];
            L.read_off4 = (uint32_t)(o / 4);
            L.umi_off = (uint32_t)t_ustart[(size_t)t].size();   // thread-relative; rebased after the join
            L.n_reads = (int32_t)n; L.n_umi = (int32_t)nu; L.n_frag = (int32_t)nf;
            uint32_t acc = 0;
            t_ustart[(size_t)t].push_back(0);
            for (size_t u = 0; u < nu; ++u) { acc += (uint32_t)n_reads_of[u]; t_ustart[(size_t)t].push_back(acc); }
            // allele-table facts
            const char rc = refseq[l];
            int ra = 255;
            switch (rc) { case 'A': ra = 0; break; case 'T': ra = 1; break; case 'G': ra = 2; break; case 'C': ra = 3; break; case 'N': ra = 4; break; default: break; }
            uint64_t mask = 0x1f;
            for (size_t k = 0; k < extra.size(); ++k) {
                if (extra[k].size() == 1) { mask |= 1ull << (6 + k); if (ra == 255 && extra[k][0] == rc) ra = 6 + (int)k; }
                t_keys[(size_t)t] += extra[k]; t_keys[(size_t)t] += '\n';
            }
            L.ref_allele = (uint8_t)ra; L.n_alleles = (uint8_t)(6 + extra.size());
            L.flags = (uint16_t)(smc_param_fingerprint(min_bq, min_mq, mismatch_thr, primer_dist) << SMC_LF_FP_SHIFT); L.snp_mask = mask;
            b.n_keys[(size_t)l] = (int32_t)extra.size();
            if (ds > 0 && (int)nu > ds && (int)inc_order.size() > ds) {
                std::string& o = t_ds[(size_t)t];
                o += std::to_string(l);
                for (int32_t u : inc_order) { o += '\t'; o += std::to_string(u); o += ':'; o += bc_names[(size_t)gid_of_u[(size_t)u]]; }
                o += '\n';
            }
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    const auto t_b = std::chrono::steady_clock::now();
    if (err.load()) { for (const auto& e : t_err) if (!e.empty()) { b.err = e; break; } return err.load(); }
    b.p_umi_start.clear();
    for (int t = 0; t < T; ++t) {
        const uint32_t base = (uint32_t)b.p_umi_start.size();
        for (int64_t l = cut[(size_t)t]; l < cut[(size_t)t + 1]; ++l) ploci[(size_t)l].umi_off += base;
        b.p_umi_start.insert(b.p_umi_start.end(), t_ustart[(size_t)t].begin(), t_ustart[(size_t)t].end());
        b.keys += t_keys[(size_t)t];
        b.ds_info += t_ds[(size_t)t];
    }
    *n_loci_done = nl; *n_slots = slots; *n_umi_start = (int64_t)b.p_umi_start.size();
    if (exp_env("SMC_BAM_TIMING")) {
        const auto t_2 = std::chrono::steady_clock::now();
        fprintf(stderr, "smc_bam_planes: %zu alignments, %lld loci, %lld reads: fetch %.1f ms, pileup %.1f ms (%d threads) [prep %.1f, threads %.1f]\n",
                reads.size(), (long long)nl, (long long)total, std::chrono::duration<double, std::milli>(t_1 - t_0).count(),
                std::chrono::duration<double, std::milli>(t_2 - t_1).count(), T,
                std::chrono::duration<double, std::milli>(t_a - t_1).count(), std::chrono::duration<double, std::milli>(t_b - t_a).count());
    }
    return total;
}

// the small per-run outputs of the last smc_bam_planes: umi_start, extra allele keys
void smc_bam_planes_copy(void* h, uint32_t* umi_start, int32_t* n_keys, char* keys) {
    Bam& b = *(Bam*)h;
    memcpy(umi_start, b.p_umi_start.data(), 4 * b.p_umi_start.size());
    memcpy(n_keys, b.n_keys.data(), 4 * b.n_keys.size());
    memcpy(keys, b.keys.data(), b.keys.size());
}

// ---------------------------------------------------------------------------------------------------------------
// Device plane builder, host half: what the GPU needs of a run's alignments, as a structure of arrays (decode only - one
// entry per ALIGNMENT; everything per pileup read - which reads cover which locus, query position, allele, quality,
// flags, read class, barcode / fragment ids by first appearance, the barcode-major order - is done by k_build_planes).
//   aln[n_aln]   smc_dev_aln (36 B): position, end, CIGAR / sequence offsets, flags, MAPQ, run-wide barcode / read-name ids
//   cig[n_cig]   the CIGAR words of those alignments; bq[2 * n_seq]: their bases as (ASCII letter, quality) byte pairs
//                (include/smcounter_hip.h: smc_build_in.bq - one stream, so that the device reads both with one fetch)
//   loc[n_loci]  per locus: window [w0, w1) of alignments that can cover it, first padded slot, depth
// status bits: 1 = an alignment has neither READ1 nor READ2 (pairOrder is carried over between pileup reads,
// smCounter.py:359-362: that needs the sequential path), 2 = a field does not fit the packed record.
// `alloc(ctx, n_aln, n_cig, n_seq, n_loci, out)` must fill out[0..3] with buffers for aln, cig, bq (2 * n_seq bytes), loc.
// Returns pileup reads (unpadded) or < 0 as smc_bam_pileup.
typedef void (*smc_aln_alloc)(void* ctx, int64_t n_aln, int64_t n_cig, int64_t n_seq, int64_t n_loci, void** out);
int64_t smc_bam_alignments(void* h, const char* chrom, int64_t start0, int64_t end0, int64_t max_reads, double mismatch_thr,
                           int nthreads, smc_aln_alloc alloc, void* alloc_ctx, int64_t* n_loci_done, int64_t* n_slots,
                           int32_t* n_bc_out, int32_t* n_pair_out, int32_t* status) {
    Bam& b = *(Bam*)h;
    b.io_threads = nthreads;
    *n_loci_done = *n_slots = 0; *status = 0;
    b.d_reads.clear(); b.d_bc_names.clear();
    int n_bc = 0, n_pair = 0;
    const auto t_a0 = std::chrono::steady_clock::now();
    { const int rc = collect_reads(b, chrom, start0, end0, b.d_reads, n_bc, n_pair, &b.d_bc_names); if (rc) return rc; }
    const auto t_a1 = std::chrono::steady_clock::now();
    const RawVec<Aln>& reads = b.d_reads;
    const int64_t span = end0 - start0;
    // one pass over the alignments: depth differences per position, and where each alignment's CIGAR words and bases go in the pools
    std::vector<int64_t> cov((size_t)span + 1, 0);
    // Where an alignment's bases go in the pool: so that reference position start0 + 64 t (the first locus of the device builder's
    // tile t) falls on a multiple of 64 bases = 128 bytes.  The 64 positions of an alignment under a tile are then ONE aligned
    // 128-byte line of the pool (k_bp_emit2.inc reads them with eight 16-byte loads per row) instead of parts of two lines that the
    // neighbouring tile fetches again; the gaps (31.5 bases on average) hold 'A' with quality 0 and are never under a covered locus.
    // (stretches of the alignments in parallel: a stretch's bases start on a multiple of 64, so where its alignments go relative to
    // that start does not depend on the stretches before it; the stretches' sizes are then summed up in order)
    std::vector<uint32_t> offc(reads.size() + 1, 0), offs(reads.size() + 1, 0);
    uint32_t pool_end = 0;
    const size_t NR = reads.size();
    const int NS = (int)std::min<size_t>(128, NR / 8192 + 1);
    std::vector<uint32_t> s_cig((size_t)NS + 1, 0), s_pool((size_t)NS + 1, 0);
    std::vector<int32_t> s_maxend((size_t)NS, INT32_MIN);                      // (for the windows below: the largest end of a stretch)
    {
        Pool::get().run(NS, nthreads, [&](int c) {
            const size_t lo_i = NR * (size_t)c / (size_t)NS, hi_i = NR * (size_t)(c + 1) / (size_t)NS;
            uint32_t nc = 0, pe = 0;
            int32_t me = INT32_MIN;
            for (size_t i = lo_i; i < hi_i; ++i) {
                const Aln& a = reads[i];
                const int64_t lo = std::max<int64_t>(a.pos, start0), hi = std::min<int64_t>(a.end, end0);
                if (lo < hi) { __atomic_fetch_add(&cov[(size_t)(lo - start0)], 1, __ATOMIC_RELAXED); __atomic_fetch_sub(&cov[(size_t)(hi - start0)], 1, __ATOMIC_RELAXED); }
                offc[i] = nc; nc += a.cigar.n;
                const uint32_t want = (uint32_t)((int64_t)a.pos - (int64_t)a.left_sp - start0) & 63u;
                offs[i] = pe + ((want - pe) & 63u);
                pe = offs[i] + a.l_seq;
                me = std::max<int32_t>(me, (int32_t)a.end);
            }
            s_cig[(size_t)c + 1] = nc; s_pool[(size_t)c + 1] = pe; s_maxend[(size_t)c] = me;
        });
        for (int c = 0; c < NS; ++c) {
            s_cig[(size_t)c + 1] += s_cig[(size_t)c];
            s_pool[(size_t)c + 1] = ((s_pool[(size_t)c] + 63u) & ~63u) + s_pool[(size_t)c + 1];   // (the stretch starts on a multiple of 64)
        }
        Pool::get().run(NS, nthreads, [&](int c) {
            const size_t lo_i = NR * (size_t)c / (size_t)NS, hi_i = NR * (size_t)(c + 1) / (size_t)NS;
            const uint32_t bc = s_cig[(size_t)c], bp = (s_pool[(size_t)c] + 63u) & ~63u;
            for (size_t i = lo_i; i < hi_i; ++i) { offc[i] += bc; offs[i] += bp; }
        });
        offc[NR] = s_cig[(size_t)NS];
        pool_end = s_pool[(size_t)NS];
    }
    int64_t nl = 0, total = 0, run = 0, slots = 0;
    std::vector<uint32_t> l_off, l_n;
    l_off.reserve((size_t)span); l_n.reserve((size_t)span);
    for (int64_t k = 0; k < span; ++k) {
        run += cov[(size_t)k];
        total += run;
        l_off.push_back((uint32_t)slots); l_n.push_back((uint32_t)run);
        slots += (run + 3) / 4 * 4;
        ++nl;
        if (total >= max_reads) break;
    }
    offs[reads.size()] = pool_end;
    const int64_t n_cig = offc[reads.size()], n_seq = pool_end;
    void* bufs[4] = {nullptr, nullptr, nullptr, nullptr};
    alloc(alloc_ctx, (int64_t)reads.size(), n_cig, n_seq, nl, bufs);
    smc_dev_aln* pa = (smc_dev_aln*)bufs[0]; uint32_t* pc = (uint32_t*)bufs[1];
    uint8_t* ps = (uint8_t*)bufs[2]; smc_dev_locus* pl = (smc_dev_locus*)bufs[3];
    if ((!reads.empty() && (!pa || !pc || !ps)) || (nl && !pl)) { b.err = "smc_bam_alignments: allocation callback returned no memory"; return -9; }
    // the records and pools are filled by the threads
    std::atomic<int> st_bits(0);
    const auto t_p0 = std::chrono::steady_clock::now();
    {
        static const char* const CODE = "=ACMGRSVTWYHKDBN";
        const int T = (int)(reads.size() / 1024 + 1);
        auto work = [&](int t) {
            const size_t lo = reads.size() * (size_t)t / (size_t)T, hi = reads.size() * (size_t)(t + 1) / (size_t)T;
            int st = 0;
            for (size_t i = lo; i < hi; ++i) {
                const Aln& a = reads[i];
                smc_dev_aln& d = pa[i];
                d.pos = a.pos; d.end = a.end;
                d.cig_off = offc[i]; d.seq_off = offs[i];
                if (a.cigar.n > 65535 || a.left_sp > 65535 || a.qalen > 65535 || a.l_seq > 65535) st |= 2;
                d.n_cig = (uint16_t)a.cigar.n;
                const int64_t mism = std::max<int64_t>(0, (int64_t)a.nm - (int64_t)a.n_ind);
                const double mm100 = a.l_seq > 0 ? 100.0 * (double)mism / (double)a.l_seq : 0.0;     // smCounter.py:352-356
                if (!(a.oflag & 3)) st |= 1;
                d.oflag = (uint8_t)((a.oflag & 7) | (mm100 <= mismatch_thr ? SMC_DA_MMOK : 0));
                d.mapq = a.mapq;
                d.left_sp = (uint16_t)a.left_sp; d.qalen = (uint16_t)a.qalen;
                d.l_seq = (uint16_t)a.l_seq; d.pad = 0;
                d.bc_gid = (uint32_t)a.bc_gid; d.pair_gid = (uint32_t)a.pair_gid;
                uint32_t* c = pc + offc[i];
                for (uint32_t w : a.cigar) *c++ = w;
                uint8_t* sq = ps + 2 * (size_t)offs[i];                          // (letter, quality) pairs
                for (size_t g = i ? (size_t)offs[i - 1] + reads[i - 1].l_seq : 0; g < (size_t)offs[i]; ++g) { ps[2 * g] = 'A'; ps[2 * g + 1] = 0; }   // (the gap before it)
                const uint8_t* packed = a.seq.p;
                const uint8_t* ql = a.qual.p;
                for (uint32_t k = 0; k + 1 < a.l_seq; k += 2) {
                    const uint8_t b2 = packed[k >> 1];
                    sq[2 * k] = (uint8_t)CODE[b2 >> 4]; sq[2 * k + 1] = ql[k]; sq[2 * k + 2] = (uint8_t)CODE[b2 & 15]; sq[2 * k + 3] = ql[k + 1];
                }
                if (a.l_seq & 1) { const uint32_t k = a.l_seq - 1; sq[2 * k] = (uint8_t)CODE[packed[k >> 1] >> 4]; sq[2 * k + 1] = ql[k]; }
            }
            if (st) st_bits.fetch_or(st);
        };
        Pool::get().run(T, nthreads, work);
    }
    const int st = st_bits.load();
    const auto t_p1 = std::chrono::steady_clock::now();
    // candidate window of every locus: [first alignment that ends behind it ... first alignment that starts behind it)
    // (stretches of loci in parallel; a stretch finds its first two indices from scratch - the stretches of alignments above know
    // their largest end, the positions are sorted - and then moves them along as the one-thread loop did; the packed records are read)
    {
        const int NL = (int)std::min<int64_t>(128, nl / 256 + 1);
        Pool::get().run(NL, nthreads, [&](int c) {
            const int64_t l_lo = nl * c / NL, l_hi = nl * (c + 1) / NL;
            if (l_lo >= l_hi) return;
            const int64_t pf = start0 + l_lo;
            size_t w0 = NR, w1;
            for (int k = 0; k < NS; ++k)
                if ((int64_t)s_maxend[(size_t)k] > pf) { w0 = NR * (size_t)k / (size_t)NS; break; }   // the first stretch with an alignment that ends behind pf
            while (w0 < NR && pa[w0].end <= pf) ++w0;
            { size_t lo = 0, hi = NR; while (lo < hi) { const size_t mid = (lo + hi) / 2; if (pa[mid].pos <= pf) lo = mid + 1; else hi = mid; } w1 = lo; }
            for (int64_t l = l_lo; l < l_hi; ++l) {
                const int64_t p0 = start0 + l;
                while (w0 < NR && pa[w0].end <= p0) ++w0;
                while (w1 < NR && pa[w1].pos <= p0) ++w1;
                pl[l].w0 = (uint32_t)w0; pl[l].w1 = (uint32_t)std::max(w0, w1);
                pl[l].slot_off = l_off[(size_t)l]; pl[l].n = l_n[(size_t)l];
            }
        });
    }
    *n_loci_done = nl; *n_slots = slots; *n_bc_out = n_bc; *n_pair_out = n_pair; *status = st;
    if (exp_env("SMC_BAM_TIMING")) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "smc_bam_alignments: %zu alignments, %lld loci, %lld reads: collect %.1f ms, depth + sizes %.1f ms, pack %.1f ms, windows %.1f ms\n",
                reads.size(), (long long)nl, (long long)total, ms(t_a0, t_a1), ms(t_a1, t_p0), ms(t_p0, t_p1), ms(t_p1, std::chrono::steady_clock::now()));
    }
    return total;
}

// key text of an allele the device builder saw first on alignment `ai` of the last smc_bam_alignments at query position
// qpos with `indel` (k_build_planes reports that triple for every allele beyond the six fixed ones): "INS|b|b<ins>",
// "D<len>|b" (the caller appends the deleted reference bases, smCounter.py:392-396) or the letter itself.
int smc_bam_allele_key(void* h, int64_t ai, int32_t qpos, int32_t indel, char* out, int cap) {
    Bam& b = *(Bam*)h;
    if (ai < 0 || (size_t)ai >= b.d_reads.size() || qpos < 0) return -1;
    const Aln& a = b.d_reads[(size_t)ai];
    if ((uint32_t)qpos >= a.l_seq) return -1;
    const char site = a.seq[(size_t)qpos];
    std::string key;
    if (indel > 0) key = std::string("INS|") + site + "|" + site + a.seq.substr((size_t)qpos + 1, (size_t)indel);
    else if (indel < 0) key = "D" + std::to_string(-indel) + "|" + site;
    else key = std::string(1, site);
    if ((int)key.size() + 1 > cap) return -2;
    memcpy(out, key.c_str(), key.size() + 1);
    return (int)key.size();
}
// text of run-wide barcode id `gid` of the last smc_bam_alignments (for the reference's down-sampling, :496-498)
const char* smc_bam_barcode_name(void* h, int32_t gid) {
    Bam& b = *(Bam*)h;
    return (gid >= 0 && (size_t)gid < b.d_bc_names.size()) ? b.d_bc_names[(size_t)gid].c_str() : "";
}

// a 64-bit identity of every run-wide barcode id of the last smc_bam_alignments - FNV-1a over its text: what the non-parity sampler
// keys on (smc_philox_marks: the sample of a locus then does not depend on how the file was cut into runs).  -> the number of ids
int64_t smc_bam_barcode_idents(void* h, uint64_t* out, int64_t cap) {
    Bam& b = *(Bam*)h;
    const int64_t n = (int64_t)b.d_bc_names.size();
    for (int64_t g = 0; g < n && g < cap; ++g) {
        uint64_t x = 1469598103934665603ull;
        for (unsigned char c : b.d_bc_names[(size_t)g]) { x ^= c; x *= 1099511628211ull; }
        out[g] = x;
    }
    return n;
}

}  // extern "C"
