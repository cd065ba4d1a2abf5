// `platanus_b kmer_divide` on the device path: host logic of kmer_divide.hpp, per-position look-ups through libpbgpu.
#include "kmer_divide.hpp"
#include <iostream>

extern "C" {
#include "../../include/pbgpu.h"
}

namespace pbh {

int run_kmer_divide_gpu(const DivideOptions &opt, int device, bool verbose)
{
    pb_ctx *ctx = nullptr;
    if (pb_create(device, &ctx) != PB_OK) throw std::runtime_error(std::string("libpbgpu: ") + pb_last_error(nullptr));
    auto chk = [&](int rc) { if (rc != PB_OK) { const std::string m = pb_last_error(ctx); pb_destroy(ctx); throw std::runtime_error("libpbgpu: " + m); } };
    unsigned k = 0;
    std::vector<u64> keys;
    std::vector<u16> counts;
    read_kmer_occ(opt.occ_file, k, keys, counts);
    ContigSet cs;
    u64 read_length = 100, max_k = 100;
    for (size_t i = 0; i < opt.files.size(); ++i) read_contig_fasta(opt.files[i], cs, i == 0, read_length, max_k);
    chk(pb_occ_table_load(ctx, k, keys.data(), counts.data(), counts.size()));
    std::string bases;
    std::vector<u64> off{0};
    for (const std::string &s : cs.seq) {
        if (s.size() < k) { pb_destroy(ctx); throw std::runtime_error("contig shorter than k"); }
        bases += s;
        off.push_back(bases.size());
    }
    std::vector<u16> flat(bases.size());
    chk(pb_occ_lookup(ctx, (const uint8_t *)bases.data(), off.data(), cs.size(), flat.data()));
    pb_destroy(ctx);
    std::vector<std::vector<u64>> occ(cs.size());
    for (size_t i = 0; i < cs.size(); ++i) {
        const u64 nk = cs.seq[i].size() - k + 1;
        occ[i].resize(nk);
        for (u64 j = 0; j < nk; ++j) occ[i][j] = flat[off[i] + j];
    }
    divide_and_write(opt, k, cs, occ, read_length, max_k, verbose);
    return 0;
}

} // namespace pbh
