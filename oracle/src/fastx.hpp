// TEST INFRASTRUCTURE ONLY -- oracle restatement of the reference's read ingest
// (assemble.cpp:790-986, baseCommand.cpp:29-50, common.h:460-477) for uncompressed FASTA/FASTQ.
#pragma once
#include "counter.hpp"
#include <fstream>
#include <string>

namespace pbo {

inline void put_read(ReadSet &rs, const std::string &s)
{
    if (s.length() >= MAX_READ_LEN) throw std::runtime_error("read too long");     // common.h:465-467
    std::vector<uint8_t> codes(s.size());
    for (size_t i = 0; i < s.size(); ++i) {
        const unsigned char c = char2bin(s[i]);
        codes[i] = c == 4 ? 4 : (c & 3);      // N kept as 4; anything else admitted only as ACGT by the format check
    }
    rs.add(codes.data(), codes.size());
}

// first character of the first line decides (baseCommand.cpp:29-50): '>' FASTA, '@' FASTQ
inline void read_fastx(const std::string &path, ReadSet &rs)
{
    std::ifstream ifs(path);
    if (!ifs) throw std::runtime_error("cannot open " + path);
    std::string line, read;
    if (!std::getline(ifs, line)) throw std::runtime_error("empty read file " + path);
    const bool fastq = line[0] == '@';
    if (!fastq && line[0] != '>') throw std::runtime_error("Read file is not FASTA/FASTQ format.");
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
