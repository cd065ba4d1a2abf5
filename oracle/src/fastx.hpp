// TEST INFRASTRUCTURE ONLY -- oracle restatement of the reference's read ingest
// (assemble.cpp:790-986, baseCommand.cpp:29-50, common.h:460-477) for uncompressed FASTA/FASTQ.
#pragma once
#include "counter.hpp"
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

namespace pbo {

inline void put_read(ReadSet &rs, const std::string &s)
{
    if (s.length() >= MAX_READ_LEN) throw pbh::read_too_long();     // common.h:465-467
    std::vector<uint8_t> codes(s.size());
    for (size_t i = 0; i < s.size(); ++i) {
        const unsigned char c = char2bin(s[i]);
        codes[i] = c == 4 ? 4 : (c & 3);      // N kept as 4; anything else admitted only as ACGT by the format check
    }
    rs.add(codes.data(), codes.size());
}

// BaseCommand::checkFileFormat (baseCommand.cpp:26-49): '>' + a second line of ACGTN only = FASTA; '@' + such a line + '+' at the
// start of the third = FASTQ; everything else is refused (assemble.cpp:798).  A file that cannot be opened: FILEError (common.cpp:140).
inline void read_fastx(const std::string &path, ReadSet &rs)
{
    // whole (decompressed) text first: gzip / bzip2 data is recognised by its first bytes, as `file -bL` does for the reference
    // (common.cpp:88-144), and piped through `gzip -cd` / `bzip2 -cd`
    std::string text;
    {
        FILE *fp = fopen(path.c_str(), "rb");
        if (!fp) throw pbh::file_error(path);
        unsigned char m[3] = {0, 0, 0};
        const size_t got = fread(m, 1, 3, fp);
        if (got >= 2 && m[0] == 0x1f && m[1] == 0x8b) { fclose(fp); fp = popen(("gzip -cd " + path).c_str(), "r"); }
        else if (got == 3 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h') { fclose(fp); fp = popen(("bzip2 -cd " + path).c_str(), "r"); }
        else rewind(fp);
        if (!fp) throw pbh::file_error(path);
        char buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
        // (a pipe and a file are both closed by the matching call; the status is not looked at, as in the reference)
        if (got >= 2 && ((m[0] == 0x1f && m[1] == 0x8b) || (got == 3 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h'))) pclose(fp); else fclose(fp);
    }
    {
        std::istringstream probe(text);
        std::string l[3];
        for (int i = 0; i < 3; ++i) { if (!std::getline(probe, l[i])) break; l[i] = l[i].c_str(); }
        bool ok = true;
        for (char c : l[1]) ok = ok && (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N');
        const char c0 = l[0].empty() ? '\0' : l[0][0], c2 = l[2].empty() ? '\0' : l[2][0];
        if (!(ok && (c0 == '>' || (c0 == '@' && c2 == '+')))) throw pbh::read_error("Read file exception!!\nRead file is not FASTA/FASTQ format.");
    }
    std::istringstream ifs(text);
    std::string line, read;
    std::getline(ifs, line);
    const bool fastq = line[0] == '@';
    // skip to the first header (the line just read is one)
    if (fastq) {
        bool flag = true;                                       // assemble.cpp:902-940
        while (std::getline(ifs, line)) {
            if (line.length() == 0) continue;
            if (line[0] != '@') {
                if (flag && line[0] != '+') read += line; else flag = false;
            } else {
                if (read.length() != 0) { put_read(rs, read); read.clear(); }
                flag = true;
            }
        }
    } else {
        while (std::getline(ifs, line)) {                       // assemble.cpp:818-848
            if (line[0] != '>') read += line;
            else if (read.length() != 0) { put_read(rs, read); read.clear(); }
        }
    }
    put_read(rs, read);                                         // trailing read, written even when empty
}

} // namespace pbo
