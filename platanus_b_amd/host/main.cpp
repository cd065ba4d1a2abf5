// platanus_b (MI355X build): same command line as the reference's `platanus_b assemble` (main.cpp:70-149,
// baseCommand.cpp:56-136, assemble.cpp:55-110).  Only the `assemble` command exists here.
#include "assemble_gpu.hpp"
#include "kmer_divide.hpp"
#include "merge.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "errors.hpp"
#include <iostream>
#include <csignal>
#include <execinfo.h>
#include <sys/mman.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>

extern "C" {
#include "../../include/pbgpu.h"
}

static void usage()
{
    std::cerr << "\nUsage platanus_b assemble [Options]\n"
              << "Options:\n"
              << "    -o STR               : prefix of output files (default out, length <= 200)\n"
              << "    -f FILE1 [FILE2 ...] : reads file (fasta or fastq, number <= 100)\n"
              << "    -k INT               : initial k-mer size (default 32)\n"
              << "    -K FLOAT             : maximum-k-mer factor (maximum-k = FLOAT*read-length, default  0.5)\n"
              << "    -s INT               : step size of k-mer extension (>= 1, default 10)\n"
              << "    -n INT               : initial k-mer coverage cutoff (default 0, 0 means auto)\n"
              << "    -c INT               : minimun k-mer coverage (default 1)\n"
              << "    -a FLOAT             : k-mer extension safety level (default 10.0)\n"
              << "    -u FLOAT             : maximum difference for bubble crush (identity, default 0)\n"
              << "    -d FLOAT             : maximum difference for branch cutting (coverage ratio, default 0.5)\n"
              << "    -e FLOAT             : k-mer coverage depth (k = initial k-mer size specified by -k) of homozygous region (default auto)\n"
              << "    -t INT               : number of threads (accepted, unused: the k-mer work runs on the GPU)\n"
              << "    -m INT               : memory limit for making kmer distribution (GB, >=1, default 16; sizes PREFIX_kmer_occ.bin only)\n"
              << "    -tmp DIR             : directory for temporary files (accepted, unused: no temporary files)\n"
              << "    -kmer_occ_only       : only output k-mer occurrence table (out_kmer_occ.bin; default off) \n"
              << "    -gpu INT             : device id (default 0; with -gpus N: rank r runs on device INT + r)\n"
              << "    -gpus INT            : number of GPUs = processes (default 1): every rank parses its byte range of the read files; records / k-mer tables\n"
              << "                           are exchanged by key owner and the retained sets gathered over RCCL before every graph stage; same output files\n"
              << "    -share_gpu           : with -gpus N: all ranks on device -gpu, records exchanged through shared host memory (tests)\n"
              << "    -comm STR            : with -gpus N: rccl (default) or host (shared host memory instead of RCCL)\n"
              << "    -device_parse        : uncompressed read files are parsed on the device (their bytes are uploaded as they are); default: by host threads\n"
              << "    -gpu_opt NAME=INT    : pb_set_option of libpbgpu (selects device code paths; results do not change)\n"
              << "    (-repeat is refused: the reference itself crashes in that mode, DESIGN.md section 1)\n"
              << "\n\nOutputs:\n    PREFIX_contig.fa\n    PREFIX_kmerFrq.tsv\n\nNote, PREFIX is specified by -o\n" << std::endl;
}

// main.cpp:49-63 (ShowVersion / ShowArgs) and main.cpp:128-147: the reference frames every run's stderr with its version, the command
// line, and the process's peak memory; kept so that logs look the same (1.3.2 = the version of the command line implemented here)
static void show_version_and_args(int argc, char **argv)
{
    std::cerr << "platanus_b version: 1.3.2" << std::endl;
    for (int i = 0; i < argc; ++i) std::cerr << argv[i] << " ";
    std::cerr << std::endl << std::endl;
}

static int process_information(int rc)
{
    unsigned long long vmpeak = 0, vmhwm = 0;
    FILE *f = fopen("/proc/self/status", "r");
    if (f) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            if (strncmp(line, "VmPeak", 6) == 0) vmpeak = (unsigned long long)atoi(line + 8);
            else if (strncmp(line, "VmHWM", 5) == 0) vmhwm = (unsigned long long)atoi(line + 8);
        }
        fclose(f);
    }
    fprintf(stderr, "\n#### PROCESS INFORMATION ####\n");
    fprintf(stderr, "VmPeak:      %10.3f GByte\n", vmpeak / 1048576.0);
    fprintf(stderr, "VmHWM:       %10.3f GByte\n", vmhwm / 1048576.0);
    return rc;
}

namespace pbh { int run_kmer_divide_gpu(const DivideOptions &opt, int device, bool verbose); int run_merge_gpu(const MergeOptions &opt, int device, bool verbose); }

static int main_merge(int argc, char **argv)             // options: merge.cpp:29-39
{
    pbh::MergeOptions o;
    int device = 0;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << a << std::endl; exit(1); } return argv[++i]; };
        if (a == "-o") o.out_prefix = val();
        else if (a == "-c") o.min_cov = atoi(val().c_str());
        else if (a == "-k") o.k_factor = atof(val().c_str());
        else if (a == "-l") o.len_factor = atof(val().c_str());
        else if (a == "-u") o.bubble = val();
        else if (a == "-d") o.branch = atof(val().c_str());
        else if (a == "-m") o.memory_gb = atoi(val().c_str());
        else if (a == "-tmp") val();
        else if (a == "-gpu") device = atoi(val().c_str());
        else if (a == "-f") { while (i + 1 < argc && argv[i + 1][0] != '-') o.files.push_back(argv[++i]); }
        else { std::cerr << "\nUsage: platanus_b merge [Options] (-o -f -c -k -l -u -d -m -tmp)" << std::endl; return 1; }
    }
    try {
        return process_information(pbh::run_merge_gpu(o, device, true));
    } catch (const pbh::RefError &e) {
        std::cerr << "Error(" << e.id << "): " << e.what() << std::endl;
        return e.id;
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 2;
    }
}

static int main_kmer_divide(int argc, char **argv)       // options: kmer_divide.cpp:28-38
{
    pbh::DivideOptions o;
    int device = 0;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << a << std::endl; exit(1); } return argv[++i]; };
        if (a == "-o") o.out_prefix = val();
        else if (a == "-k") o.occ_file = val();
        else if (a == "-r") o.rate = atof(val().c_str());
        else if (a == "-R") o.mask_rate = atof(val().c_str());
        else if (a == "-t" || a == "-tmp") val();
        else if (a == "-gpu") device = atoi(val().c_str());
        else if (a == "-recalc_cov") o.recalc_cov = true;
        else if (a == "-mask") o.mask = true;
        else if (a == "-f") { while (i + 1 < argc && argv[i + 1][0] != '-') o.files.push_back(argv[++i]); }
        else { std::cerr << "\nUsage: platanus_b kmer_divide [Options] (-o -f -k -r -R -recalc_cov -mask -t -tmp)" << std::endl; return 1; }
    }
    if (o.files.empty()) { std::cerr << "Error: not specified contig file!!" << std::endl; return 1; }
    if (o.occ_file.empty()) { std::cerr << "Error: not specified kmer occurrence file!!!" << std::endl; return 1; }
    try {
        return process_information(pbh::run_kmer_divide_gpu(o, device, true));
    } catch (const pbh::RefError &e) {
        std::cerr << "Error(" << e.id << "): " << e.what() << std::endl;
        return e.id;
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 2;
    }
}

// A crash says where: the frames go to stderr (async-signal-safe calls only), the other ranks of an N-rank run are released from their
// barriers (pb_dist_mark_failed: an atomic store and a futex wake), then the default action takes over, so the exit status is still the
// signal's (the reference has no such handler; its crashes are bare).  The handler runs on a stack of its own, so that a stack
// overflow leaves its frames too.
static void *g_shared = nullptr;
static void crash_trace(int sig)
{
    static const char msg[] = "\nplatanus_b: fatal signal, frames:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    void *frames[48];
    const int n = backtrace(frames, 48);
    backtrace_symbols_fd(frames, n, 2);
    if (g_shared) pb_dist_mark_failed(g_shared);
    signal(sig, SIG_DFL);
    raise(sig);
}

static void install_crash_trace()
{
    void *warm[4];
    (void)backtrace(warm, 4);                         // (loads libgcc's unwinder now, not inside the handler)
    static char alt[1 << 16];
    stack_t ss;
    ss.ss_sp = alt;
    ss.ss_size = sizeof(alt);
    ss.ss_flags = 0;
    sigaltstack(&ss, nullptr);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = crash_trace;
    sa.sa_flags = SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    for (int sig : {SIGSEGV, SIGBUS, SIGFPE, SIGILL, SIGABRT}) sigaction(sig, &sa, nullptr);
}

// Development: PB_HOST_SAMPLES=FILE -> the process's CPU time is sampled (SIGPROF every millisecond, a few return addresses per sample) and
// the raw addresses go to FILE with /proc/self/maps at exit; tools/host_samples.py turns that into a profile by function (there is no perf
// on the boxes, and gprof does not see libc / libstdc++).
static const int SAMPLE_DEPTH = 8;
static void **g_samples = nullptr;
static volatile long g_n_samples = 0;
static const long SAMPLE_CAP = 1L << 21;
static const char *g_samples_path = nullptr;
static void sample_tick(int)
{
    const long at = __sync_fetch_and_add(&g_n_samples, 1);
    if (at >= SAMPLE_CAP) return;
    void *fr[SAMPLE_DEPTH + 2];
    const int n = backtrace(fr, SAMPLE_DEPTH + 2);
    for (int i = 0; i < SAMPLE_DEPTH; ++i) g_samples[at * SAMPLE_DEPTH + i] = i + 2 < n ? fr[i + 2] : nullptr;      // (without the handler and the signal frame)
}
static void samples_write()
{
    struct itimerval off;
    memset(&off, 0, sizeof(off));
    setitimer(ITIMER_PROF, &off, nullptr);
    FILE *f = fopen(g_samples_path, "w");
    if (!f) return;
    if (FILE *m = fopen("/proc/self/maps", "r")) { char ln[1024]; while (fgets(ln, sizeof ln, m)) fputs(ln, f); fclose(m); }
    fputs("#### SAMPLES ####\n", f);
    const long n = g_n_samples < SAMPLE_CAP ? g_n_samples : SAMPLE_CAP;
    for (long i = 0; i < n; ++i) {
        for (int j = 0; j < SAMPLE_DEPTH; ++j) fprintf(f, j ? " %lx" : "%lx", (unsigned long)(uintptr_t)g_samples[i * SAMPLE_DEPTH + j]);
        fputc('\n', f);
    }
    fclose(f);
}
static void install_host_sampler()
{
    g_samples_path = getenv("PB_HOST_SAMPLES");
    if (!g_samples_path || !*g_samples_path) return;
    g_samples = (void **)calloc((size_t)SAMPLE_CAP * SAMPLE_DEPTH, sizeof(void *));
    if (!g_samples) return;
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = sample_tick;
    sa.sa_flags = SA_RESTART;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGPROF, &sa, nullptr);
    atexit(samples_write);
    struct itimerval it;
    it.it_interval.tv_sec = 0; it.it_interval.tv_usec = 1000;
    it.it_value = it.it_interval;
    setitimer(ITIMER_PROF, &it, nullptr);
}

int main(int argc, char **argv)
{
    install_crash_trace();
    install_host_sampler();
    show_version_and_args(argc, argv);
    if (argc >= 2 && strcmp(argv[1], "-v") == 0) return 0;                       // main.cpp:77-79
    if (argc >= 2 && strcmp(argv[1], "kmer_divide") == 0) return main_kmer_divide(argc, argv);
    if (argc >= 2 && strcmp(argv[1], "merge") == 0) return main_merge(argc, argv);
    if (argc < 2 || strcmp(argv[1], "assemble") != 0) {
        std::cerr << "Usage: platanus_b Command [options]\n\nCommand: assemble, kmer_divide, merge" << std::endl;
        return 1;
    }
    pbh::GpuAssembleOptions o;
    int n_ranks = 1;
    bool share_gpu = false;
    std::string comm = "rccl";
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { usage(); exit(1); } return argv[++i]; };
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
        else if (a == "-t") val();
        else if (a == "-m") o.memory_gb = atoi(val().c_str());
        else if (a == "-tmp") val();
        else if (a == "-gpu") o.device = atoi(val().c_str());
        else if (a == "-gpus") n_ranks = atoi(val().c_str());
        else if (a == "-share_gpu") share_gpu = true;
        else if (a == "-device_parse") o.device_parse = true;
        else if (a == "-comm") comm = val();
        else if (a == "-gpu_opt") {              // NAME=VALUE for pb_set_option (which kernels run, never what they return)
            const std::string nv = val();
            const size_t eq = nv.find('=');
            if (eq == std::string::npos) { usage(); return 1; }
            o.gpu_options.emplace_back(nv.substr(0, eq), atoll(nv.c_str() + eq + 1));
        }
        else if (a == "-kmer_occ_only") o.kmer_occ_only = true;
        else if (a == "-repeat") o.repeat = true;
        else if (a == "-f") { o.files.clear(); while (i + 1 < argc && argv[i + 1][0] != '-') o.files.push_back(argv[++i]); }   // (a repeated -f replaces the list, baseCommand.cpp:70)
        else { usage(); return 1; }              // unknown option / -h: usage, exit 1 (main.cpp:114-117)
    }
    if (o.files.empty()) { usage(); return 1; }
    if (n_ranks < 1 || n_ranks > 64 || (comm != "rccl" && comm != "host")) { usage(); return 1; }
    // what needs no GPU is checked before any rank exists: the message comes once and the exit status is the reference's
    for (const auto &nv : o.gpu_options)
        if (pb_check_option(nv.first.c_str(), nv.second) != PB_OK) {
            std::cerr << "Error: libpbgpu: pb_set_option: unknown option or value out of range: " << nv.first << std::endl;
            return 2;
        }
    if (n_ranks > 1 && o.kmer_occ_only) {
        std::cerr << "Error: -kmer_occ_only with -gpus > 1: the slot layout of PREFIX_kmer_occ.bin is defined by the order in which ONE process first sees "
                     "every k-mer (counter.h:458-476 at -t 1); run it on one GPU" << std::endl;
        return 2;
    }
    // N ranks: the processes are forked HERE (pb_dist_launch), before anything touches the GPU; what they share (barrier, headers,
    // outboxes) is set up first so that every child inherits it.  A child dies with the launcher; the launcher watches the children.
    void *shared = nullptr;
    if (n_ranks > 1) {
        if (pb_dist_shared_create(n_ranks, &shared) != PB_OK) { std::cerr << "Error: cannot set up the shared memory of " << n_ranks << " ranks" << std::endl; return 2; }
        g_shared = shared;
        o.world = n_ranks;
        o.shared = shared;
        o.rccl = comm == "rccl" && !share_gpu;
        if (pb_dist_launch(shared, &o.rank) != PB_OK) { std::cerr << "Error: fork" << std::endl; return 2; }
        if (o.rank) o.verbose = false;
        if (!share_gpu) o.device += o.rank;
    }
    int rc = 0;
    std::string fail;                          // the line for stderr when this rank's run ends in an exception
    try {
        rc = pbh::run_assemble_gpu(o);
    } catch (const pbh::RefError &e) {           // the reference's exceptions: message, exit status = error id (main.cpp:118-127, common.h:97-100)
        fail = "Error(" + std::to_string(e.id) + "): " + e.what();
        rc = e.id ? e.id : 2;
    } catch (const std::exception &e) {
        fail = std::string("Error: ") + e.what();
        rc = 2;
    }
    if (!shared) {
        if (!fail.empty()) { std::cerr << fail << std::endl; return rc; }
        return process_information(rc);
    }
    // N ranks.  A rank never signals another: a failing rank says why in the shared block and leaves; that stops the group (every
    // barrier returns), the others leave with "a peer has failed", and the launcher prints ONE message and returns ONE status.
    const bool secondary = fail.find("a peer rank has failed") != std::string::npos;
    if (o.rank) {
        if (!fail.empty()) pb_dist_report_failure(shared, o.rank, rc, fail.c_str(), secondary ? 1 : 0);
        fflush(nullptr);
        _exit(fail.empty() ? rc : (rc ? rc : 2));
    }
    int exit_code = 0;
    char msg[1024];
    pb_dist_finish(shared, fail.empty() ? 0 : rc, fail.c_str(), secondary ? 1 : 0, &exit_code, msg, sizeof(msg));
    if (exit_code) { std::cerr << (msg[0] ? msg : "Error: a rank did not finish") << std::endl; return exit_code; }
    return process_information(rc);
}
