// Read ingest of `platanus_b assemble` on the host: the reference's FASTA/FASTQ line rules (assemble.cpp:790-986), applied to
// pieces of a file in parallel.  Pure host code (no device calls): the oracle's test entry pbo_parse_file runs it on the CPU.
#pragma once
#include "kmer.hpp"
#include "errors.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>

namespace pbh {

// FASTA/FASTQ reader with the reference's line rules (assemble.cpp:790-986); bases are handed over as ASCII, the device
// applies Char2Bin (common.h:256).  Sequence lines are appended straight into `bases` (the reference's `read += oneLine`
// followed by the flush into the temp file).
//
// A plain file is parsed out of the page cache by several threads.  The reference's line rules make every line that starts
// with the record mark ('@' / '>') a restart point: there the sequential parser flushes the read collected so far if it is
// not empty and forgets everything else.  So the file is cut at such lines, every piece is parsed independently (flushing a
// pending read at its end only if it is not empty), and the pieces in file order give exactly the sequential result; only
// the last piece ends with the reference's unconditional trailing flush.
struct ParsedPiece {
    std::string bases;
    std::vector<u64> off{0};
};

inline void parse_piece(const char *p, const char *end, bool fastq, bool first_piece, bool last_piece, ParsedPiece &out)
{
    std::string &bases = out.bases;
    std::vector<u64> &off = out.off;
    const char *ls = nullptr;
    size_t len = 0;
    auto next_line = [&]() -> bool {            // std::getline: a last line without '\n' counts, nothing after the final '\n'
        if (p >= end) return false;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        ls = p;
        len = (size_t)((nl ? nl : end) - p);
        p = nl ? nl + 1 : end;
        return true;
    };
    bases.reserve((size_t)(end - p) / 2 + 64);
    u64 read_start = 0;                         // the current read = bases[read_start, bases.size())
    auto flush = [&]() {
        if (bases.size() - read_start >= 500000) throw read_too_long();   // common.h:465-467
        off.push_back(bases.size());
        read_start = bases.size();
    };
    if (first_piece) next_line();               // the format line has been looked at by the caller; the reference consumes it the same way
    if (fastq) {
        bool flag = true;
        while (next_line()) {
            if (len == 0) continue;
            if (ls[0] != '@') {
                if (flag && ls[0] != '+') bases.append(ls, len); else flag = false;
            } else {
                if (bases.size() != read_start) flush();
                flag = true;
            }
        }
    } else {
        while (next_line()) {
            if (len == 0 || ls[0] != '>') bases.append(ls, len);
            else if (bases.size() != read_start) flush();
        }
    }
    if (last_piece || bases.size() != read_start) flush();     // the file's trailing read is written even when empty
}

// BaseCommand::checkFileFormat (baseCommand.cpp:26-49) on the first lines of the (decompressed) file: line 2 may only hold "ACGTN"
// (upper case), a FASTQ file needs '+' at the start of line 3; anything else is "not FASTA/FASTQ" (assemble.cpp:798)
inline bool check_file_format(const char *data, size_t size, bool &fastq)
{
    std::string line[3];
    const char *p = data, *end = data + size;
    for (int i = 0; i < 3 && p < end; ++i) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        line[i].assign(p, (size_t)((nl ? nl : end) - p));
        line[i] = line[i].c_str();                                   // getlineFILE hands the line over as a C string (common.cpp:171)
        p = nl ? nl + 1 : end;
    }
    const bool seq_ok = strspn(line[1].c_str(), "ACGTN") == line[1].size();
    if (!line[0].empty() && line[0][0] == '>' && seq_ok) { fastq = false; return true; }
    if (!line[0].empty() && line[0][0] == '@' && seq_ok && !line[2].empty() && line[2][0] == '+') { fastq = true; return true; }
    return false;
}

// part / parts (N ranks on one run): the file is cut into `parts` byte ranges at record-mark lines -- the same rule as the thread
// pieces below, so the concatenation of all parts' pieces is the sequential result -- and only range `part` is parsed.  Every rank
// computes the same cuts from the same bytes; a compressed file is still decompressed by every rank (one serial stream).
// raw_sink (optional): instead of parsing, the pieces of this rank's range are handed over as bytes -- (data, size, fastq, first piece of
// the file, piece that owns the file's end), in file order, cut at record marks and no larger than raw_piece_max -- for the device-side
// parser (pb_reads_append_raw); `pieces` stays empty.
typedef std::function<void(const char *, size_t, bool, bool, bool)> RawSink;
inline void read_file(const std::string &path, std::vector<ParsedPiece> &pieces, unsigned max_threads, unsigned part = 0, unsigned parts = 1,
                      const RawSink &raw_sink = nullptr, size_t raw_piece_max = 1u << 30)
{
    // compression by content, as `file -bL` tells the reference (common.cpp:88-118), not by the name
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) throw file_error(path);                                 // common.cpp:140-141
    unsigned char magic[3] = {0, 0, 0};
    const size_t got_magic = fread(magic, 1, 3, fp);
    bool piped = false;
    if (got_magic >= 2 && magic[0] == 0x1f && magic[1] == 0x8b) { fclose(fp); fp = popen(("gzip -cd " + path).c_str(), "r"); piped = true; }
    else if (got_magic == 3 && magic[0] == 'B' && magic[1] == 'Z' && magic[2] == 'h') { fclose(fp); fp = popen(("bzip2 -cd " + path).c_str(), "r"); piped = true; }
    else rewind(fp);
    if (!fp) throw file_error(path);
    char *raw = nullptr;
    struct FreeRaw { char *&p; ~FreeRaw() { free(p); } } free_raw{raw};
    const char *data = nullptr;
    size_t size = 0;
    void *map = MAP_FAILED;
    if (!piped) {                                 // plain file: parse straight out of the page cache
        struct stat st;
        if (fstat(fileno(fp), &st) == 0 && st.st_size > 0) {
            map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fp), 0);
            if (map != MAP_FAILED) { data = (const char *)map; size = (size_t)st.st_size; madvise(map, size, MADV_WILLNEED); }
        }
    }
    if (map == MAP_FAILED) {
        // a pipe (or a file that cannot be mapped): read straight into the final buffer.  A gzip file says how large that is in its last
        // four bytes (ISIZE, modulo 2^32: a hint, not a promise), so the buffer does not grow by copying itself a dozen times
        size_t expect = 64u << 20;
        if (piped && magic[0] == 0x1f) {
            if (FILE *g = fopen(path.c_str(), "rb")) {
                unsigned char t[4];
                if (fseek(g, -4, SEEK_END) == 0 && fread(t, 1, 4, g) == 4) expect = ((size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24)) + (1u << 16);
                fclose(g);
            }
        }
#ifdef F_SETPIPE_SZ
        if (piped) (void)fcntl(fileno(fp), F_SETPIPE_SZ, 1 << 20);      // (fewer wake-ups of the decompressor)
#endif
        size_t cap = expect, have = 0, got;
        raw = (char *)malloc(cap);                                   // (not value-initialised: the pages are touched once, by fread)
        if (!raw) throw std::bad_alloc();
        for (;;) {
            if (have == cap) {
                cap += cap / 2;
                char *grown = (char *)realloc(raw, cap);
                if (!grown) { free(raw); raw = nullptr; throw std::bad_alloc(); }
                raw = grown;
            }
            got = fread(raw + have, 1, cap - have, fp);
            if (got == 0) break;
            have += got;
        }
        data = raw;
        size = have;
    }
    if (piped) pclose(fp); else fclose(fp);
    struct Unmap { void *m; size_t n; ~Unmap() { if (m != MAP_FAILED) munmap(m, n); } } unmap{map, size};
    bool fastq = false;
    if (!check_file_format(data, size, fastq)) throw read_error("Read file exception!!\nRead file is not FASTA/FASTQ format.");   // assemble.cpp:798
    // cut points: the first line at or after i * size / n that starts with the record mark
    const char mark = fastq ? '@' : '>';
    size_t piece_bytes = 8u << 20;
    if (const char *pb = getenv("PB_PARSE_PIECE_BYTES")) { const long v = atol(pb); if (v >= 64) piece_bytes = (size_t)v; }   // tests: tiny pieces
    auto next_mark_line = [&](size_t pos) -> size_t {       // first line start at or after pos (exclusive of pos's own line) that begins with the mark
        while (true) {
            const char *nl = (const char *)memchr(data + pos, '\n', size - pos);
            if (!nl) return size;
            pos = (size_t)(nl - data) + 1;
            if (pos >= size || data[pos] == mark) return std::min(pos, size);
        }
    };
    // this rank's byte range [lo, hi)
    size_t lo = 0, hi = size;
    bool owns_end = true;                         // the reference's unconditional trailing flush belongs to the last non-empty range
    if (parts > 1) {
        std::vector<size_t> rc{0};
        for (unsigned i = 1; i < parts; ++i) {
            size_t pos = size / parts * i;
            pos = pos <= rc.back() ? rc.back() : next_mark_line(pos);
            rc.push_back(std::max(pos, rc.back()));
        }
        rc.push_back(size);
        lo = rc[part];
        hi = rc[part + 1];
        unsigned last_owner = 0;
        for (unsigned r = 0; r < parts; ++r) if (rc[r] < size) last_owner = r;
        owns_end = part == last_owner;
        if (lo >= hi && !(owns_end && lo == 0)) { pieces.clear(); return; }
    }
    const size_t span = hi - lo;
    if (raw_sink) {
        pieces.clear();
        size_t at = lo;
        while (at < hi) {
            size_t end = hi;
            if (hi - at > raw_piece_max) { end = std::min(next_mark_line(at + raw_piece_max / 2), hi); if (end <= at) end = hi; }
            raw_sink(data + at, end - at, fastq, at == 0, owns_end && end == hi);
            at = end;
        }
        return;
    }
    unsigned n = (unsigned)std::min<size_t>(std::max<unsigned>(max_threads, 1), span / piece_bytes + 1);
    std::vector<size_t> cut{lo};
    for (unsigned i = 1; i < n; ++i) {
        size_t pos = lo + span / n * i;
        if (pos <= cut.back()) continue;
        pos = std::min(next_mark_line(pos), hi);
        if (pos < hi && pos > cut.back()) cut.push_back(pos);
    }
    cut.push_back(hi);
    const size_t np = cut.size() - 1;
    pieces.assign(np, ParsedPiece());
    std::vector<std::exception_ptr> err(np);
    std::vector<std::thread> th;
    for (size_t i = 0; i < np; ++i)
        th.emplace_back([&, i]() {
            try { parse_piece(data + cut[i], data + cut[i + 1], fastq, cut[i] == 0, owns_end && i + 1 == np, pieces[i]); }
            catch (...) { err[i] = std::current_exception(); }
        });
    for (auto &t : th) t.join();
    for (size_t i = 0; i < np; ++i) if (err[i]) std::rethrow_exception(err[i]);
}


} // namespace pbh
