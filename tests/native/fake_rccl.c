/* A stand-in for librccl.so for CPU tests of the launch path (tests/test_bench_launch.py): the entry points blueice_amd/comm.py
 * binds, with no GPU behind them.  Loaded through BLUEICE_AMD_RCCL.  Behaviour by environment:
 *   FAKE_RCCL_INIT = ok (default) | fail        what ncclCommInitRank returns on rank FAKE_RCCL_BAD_RANK (default: every rank)
 * ncclCommCount / ncclCommUserRank report what ncclCommInitRank was given; ncclCommCuDevice reports LOCAL_RANK.
 * The collectives only succeed (there is no device memory to move): callers in --dry mode gather through the bootstrap channel. */
#include <stdlib.h>
#include <string.h>

typedef struct { int world, rank, device; } fake_comm;
typedef struct { char internal[128]; } ncclUniqueId;

const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : "scripted failure (fake_rccl)"; }
int ncclGetVersion(int* v) { *v = 99999; return 0; }
int ncclGetUniqueId(ncclUniqueId* id) { memset(id, 7, sizeof *id); return 0; }

int ncclCommInitRank(void** comm, int world, ncclUniqueId id, int rank) {
    (void)id;
    const char* mode = getenv("FAKE_RCCL_INIT");
    const char* bad = getenv("FAKE_RCCL_BAD_RANK");
    if (mode && strcmp(mode, "fail") == 0 && (!bad || atoi(bad) == rank)) return 5;
    fake_comm* c = (fake_comm*)malloc(sizeof *c);
    if (!c) return 1;
    const char* lr = getenv("LOCAL_RANK");
    c->world = world; c->rank = rank; c->device = lr ? atoi(lr) : 0;
    *comm = c;
    return 0;
}
int ncclCommDestroy(void* comm) { free(comm); return 0; }
int ncclCommCount(void* comm, int* n) { *n = ((fake_comm*)comm)->world; return 0; }
int ncclCommUserRank(void* comm, int* r) { *r = ((fake_comm*)comm)->rank; return 0; }
int ncclCommCuDevice(void* comm, int* d) { *d = ((fake_comm*)comm)->device; return 0; }
int ncclAllGather(const void* s, void* r, size_t n, int t, void* comm, void* stream) { (void)s; (void)r; (void)n; (void)t; (void)comm; (void)stream; return 0; }
int ncclAllReduce(const void* s, void* r, size_t n, int t, int op, void* comm, void* stream) { (void)s; (void)r; (void)n; (void)t; (void)op; (void)comm; (void)stream; return 0; }
