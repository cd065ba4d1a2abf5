// Errors of the reference's command line (common.h:55-56,86-246): an exception carries an id, `main` prints the message and returns
// the id as the exit status (main.cpp:118-127).  Same ids and message texts here, so that scripts that test `$?` keep working.
#pragma once
#include <stdexcept>
#include <string>

namespace pbh {

enum RefErrorId { ERR_IO = 0, ERR_FOPEN = 1, ERR_TMP = 2, ERR_FORMAT = 3, ERR_READ = 4, ERR_SUBBIN = 5, ERR_KMERDIST = 6 };

struct RefError : std::runtime_error {
    int id;
    RefError(int i, const std::string &m) : std::runtime_error(m), id(i) {}
};

inline RefError file_error(const std::string &name) { return RefError(ERR_FOPEN, "Error, File not open exception!!\n" + name + " is not open!!"); }   // common.h:113-118
inline RefError read_error(const std::string &mes) { return RefError(ERR_READ, "Error, Read file exception!!\n" + mes); }                              // common.h:134-139
inline RefError read_too_long() { return RefError(ERR_READ, "Error, Read file exception!!\nDNA read seq length is too long."); }                        // common.h:136
inline RefError kmer_dist_error()                                                                                                                      // common.h:149-153
{
    return RefError(ERR_KMERDIST, "Error, Kmer distribution exception!!\nKmer distribution cannot be caluculated correctly.");
}

} // namespace pbh
