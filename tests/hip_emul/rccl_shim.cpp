// rccl_shim.cpp — TEST INFRASTRUCTURE: a stand-in for librccl in the CPU emulation, so that the one-process-per-rank
// tensor-parallel DATA PATH of csrc/tp.hip (communicator bootstrap from a broadcast unique id, fp32 sum all-reduce of the
// o_proj / down_proj partials, byte all-gather of the logits shards) runs across real processes without a GPU
// (tests/test_multiproc_cpu.py).  Same seven entry points, same signatures as the ones tp.hip dlopens; the "device" buffers are
// host memory (emulated device memory) and streams are synchronous, so every collective is: copy into this rank's slot of a
// POSIX shared-memory segment, barrier, combine in RANK ORDER (the same sum on every rank), barrier.
// Selected with VLO_RCCL_LIBRARY=<this .so>.  Build: tests/hip_emul/build_emul.py::build_rccl_shim().
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kSlot = 8u << 20;
struct Shared {
    pthread_barrier_t bar;
    volatile int ready;
    int nranks;
    alignas(64) unsigned char slot[kMaxRanks][kSlot];
};
struct Comm { Shared *sh; int nranks, rank; char name[64]; };
struct Uid { char internal[128]; };
}  // namespace

extern "C" {
int ncclGetUniqueId(Uid *id) {
    static int counter = 0;
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/vlo_rccl_shim_%d_%d", (int)getpid(), counter++);
    return 0;
}

int ncclCommInitRank(void **comm, int nranks, Uid id, int rank) {
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;
    Comm *c = new Comm();
    c->nranks = nranks; c->rank = rank;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) return 2;
    } else {
        for (int i = 0; i < 60000 && fd < 0; ++i) { fd = shm_open(c->name, O_RDWR, 0600); if (fd < 0) usleep(1000); }
        if (fd < 0) return 2;
    }
    void *p = MAP_FAILED;
    for (int i = 0; i < 60000 && p == MAP_FAILED; ++i) {     // rank 0's ftruncate may not have happened yet
        p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (p == MAP_FAILED) usleep(1000);
    }
    close(fd);
    if (p == MAP_FAILED) return 2;
    c->sh = (Shared *)p;
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->sh->bar, &a, nranks);
        c->sh->nranks = nranks;
        __sync_synchronize();
        c->sh->ready = 1;
    } else {
        while (!c->sh->ready) usleep(200);
    }
    pthread_barrier_wait(&c->sh->bar);
    if (rank == 0) shm_unlink(c->name);          // every rank has it mapped: the name can go
    *comm = c;
    return 0;
}

int ncclCommDestroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return 0;
    munmap(c->sh, sizeof(Shared));
    delete c;
    return 0;
}
int ncclCommCount(void *comm, int *n) { *n = ((Comm *)comm)->nranks; return 0; }
int ncclCommUserRank(void *comm, int *r) { *r = ((Comm *)comm)->rank; return 0; }
const char *ncclGetErrorString(int code) { return code == 0 ? "success" : "rccl shim error"; }

// dtype 7 = float32, op 0 = sum (the only combination tp.hip issues)
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, void *) {
    Comm *c = (Comm *)comm;
    if (dtype != 7 || op != 0 || count * 4 > kSlot) return 4;
    memcpy(c->sh->slot[c->rank], send, count * 4);
    pthread_barrier_wait(&c->sh->bar);
    float *out = (float *)recv;
    for (size_t i = 0; i < count; ++i) {
        float s = 0.f;
        for (int r = 0; r < c->nranks; ++r) s += ((const float *)c->sh->slot[r])[i];
        out[i] = s;
    }
    pthread_barrier_wait(&c->sh->bar);
    return 0;
}

// dtype 0 = int8: `count` bytes per rank, concatenated in rank order
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, void *) {
    Comm *c = (Comm *)comm;
    if (dtype != 0 || count > kSlot) return 4;
    memcpy(c->sh->slot[c->rank], send, count);
    pthread_barrier_wait(&c->sh->bar);
    for (int r = 0; r < c->nranks; ++r) memcpy((char *)recv + (size_t)r * count, c->sh->slot[r], count);
    pthread_barrier_wait(&c->sh->bar);
    return 0;
}
}
