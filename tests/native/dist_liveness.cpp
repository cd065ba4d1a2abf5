// Test driver (CPU only, no HIP call): the launcher side of an N-rank run -- pb_dist_shared_create / pb_dist_launch /
// pb_dist_shared_barrier / pb_dist_report_failure / pb_dist_finish of include/pbgpu.h -- with ranks that misbehave on request.
//   dist_liveness <world> <mode>
//     ok       every rank passes three barriers and leaves with 0
//     hang1    rank 1 passes one barrier, prints "ready <pid>" and sleeps; the others go on to the next barrier
//              (the test kills rank 1, or the launcher, from outside)
//     fail1    rank 1 reports a failure of its own after the first barrier and leaves with 7
//     segv1    rank 1 dereferences a null pointer after the first barrier
//     fail0    the launcher itself fails after the first barrier (status 5)
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>

extern "C" {
#include "../../include/pbgpu.h"
}

static void *g_shared = nullptr;
static void on_crash(int sig)
{
    if (g_shared) pb_dist_mark_failed(g_shared);
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, char **argv)
{
    if (argc < 3) return 64;
    const int world = atoi(argv[1]);
    const std::string mode = argv[2];
    void *sh = nullptr;
    if (pb_dist_shared_create(world, &sh) != PB_OK) return 65;
    g_shared = sh;
    signal(SIGSEGV, on_crash);
    int rank = -1;
    if (pb_dist_launch(sh, &rank) != PB_OK) return 66;
    int code = 0, secondary = 0;
    std::string msg;
    auto barrier = [&]() {
        if (code) return;
        if (pb_dist_shared_barrier(sh, rank) != PB_OK) { code = 2; secondary = 1; msg = "Error: rank " + std::to_string(rank) + ": a peer rank has failed"; }
    };
    barrier();
    if (rank == 1 && mode == "hang1") { printf("ready %d\n", (int)getpid()); fflush(stdout); for (;;) pause(); }
    if (rank == 1 && mode == "fail1" && !code) { code = 7; msg = "Error(7): rank 1 has a reason of its own"; }
    if (rank == 1 && mode == "segv1") { volatile int *p = nullptr; *p = 1; }
    if (rank == 0 && mode == "fail0" && !code) { code = 5; msg = "Error(5): the launcher has a reason of its own"; }
    if (rank == 0 && mode == "hang1") { printf("launcher %d\n", (int)getpid()); fflush(stdout); }
    barrier();
    barrier();
    if (rank) {
        if (code) pb_dist_report_failure(sh, rank, code, msg.c_str(), secondary);
        _exit(code);
    }
    int exit_code = 0;
    char out[1024];
    pb_dist_finish(sh, code, msg.c_str(), secondary, &exit_code, out, sizeof(out));
    if (exit_code) fprintf(stderr, "%s\n", out);
    return exit_code;
}
