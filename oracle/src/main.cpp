// TEST INFRASTRUCTURE ONLY -- `pb_oracle assemble ...`: CLI of the CPU restatement, same options and
// output files as `platanus_b assemble` (assemble.cpp:55-110), uncompressed input only.
#include "assemble.hpp"
#include "fastx.hpp"
#include <cstdlib>
#include <cstring>
#include <iostream>

int main(int argc, char **argv)
{
    using namespace pbo;
    if (argc < 2 || strcmp(argv[1], "assemble") != 0) {
        std::cerr << "usage: pb_oracle assemble [platanus_b assemble options]" << std::endl;
        return 1;
    }
    AssembleOptions o;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << a << std::endl; exit(1); } return argv[++i]; };
        if (a == "-o") o.out_prefix = val();
        else if (a == "-k") o.k0 = atoi(val().c_str());
        else if (a == "-K") o.max_k_ratio = atof(val().c_str());
        else if (a == "-s") o.step = atoi(val().c_str());
        else if (a == "-n") o.n_cutoff = atoi(val().c_str());
        else if (a == "-c") o.min_cov = atoi(val().c_str());
        else if (a == "-a") o.safety = atof(val().c_str());
        else if (a == "-u") o.bubble = val();
        else if (a == "-d") o.branch = atof(val().c_str());
        else if (a == "-e") o.expected_cov = val();
        else if (a == "-t") val();                       // single-thread schedule only
        else if (a == "-m") o.memory_gb = atoi(val().c_str());
        else if (a == "-tmp") val();
        else if (a == "-kmer_occ_only") o.kmer_occ_only = true;
        else if (a == "-repeat") o.repeat = true;
        else if (a == "-f") { while (i + 1 < argc && argv[i + 1][0] != '-') o.files.push_back(argv[++i]); }
        else { std::cerr << "unknown option " << a << std::endl; return 1; }
    }
    try {
        ReadSet reads;
        for (const std::string &f : o.files) read_fastx(f, reads);
        return run_assemble(o, reads);
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 2;
    }
}
