// Options of `platanus_b assemble` (defaults: assemble.cpp:55-70) + the device to run on.
#pragma once
#include <string>
#include <utility>
#include <vector>

namespace pbh {

struct GpuAssembleOptions {
    std::string out_prefix = "out";
    std::vector<std::string> files;
    unsigned k0 = 32;
    double max_k_ratio = 0.5;            // -K
    unsigned step = 10;                  // -s
    unsigned n_cutoff = 0;               // -n (0 = auto)
    unsigned min_cov = 1;                // -c
    double safety = 10.0;                // -a
    std::string bubble = "0";            // -u (text: atoi() decides whether the final bubble crush runs)
    double branch = 0.5;                 // -d
    std::string expected_cov = "";       // -e
    unsigned long long memory_gb = 16;   // -m (only sizes the table written to PREFIX_kmer_occ.bin)
    bool kmer_occ_only = false;
    bool repeat = false;
    bool verbose = true;
    int device = 0;                      // not in the reference: which GPU
    bool device_parse = false;           // not in the reference: the FASTA / FASTQ line rules applied on the device (pb_reads_append_raw) instead of by host threads
    // not in the reference: N ranks (processes) on one run, reads range-partitioned over them
    int rank = 0, world = 1;
    void *shared = nullptr;              // pb_dist_shared_create's block, made before the ranks were forked
    bool rccl = true;                    // record exchange over RCCL (one GPU per rank); false: through shared host memory
    std::vector<std::pair<std::string, long long>> gpu_options;   // -gpu_opt NAME=INT: pb_set_option after pb_create
};

int run_assemble_gpu(const GpuAssembleOptions &opt);

} // namespace pbh
