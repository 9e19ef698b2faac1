/* The 8-GPU evaluation's one collective from a torch-free host: plain C, one process per GPU, the library's RCCL communicator
 * bootstrapped through a file (no MPI, no torch.distributed).  Replaces the reference's init_process_group(backend="nccl")
 * (/root/reference/train.py:270-277) + its metric all_reduce (train.py:102-106).
 *
 *   gcc -std=c99 -Iinclude examples/metric_gather_from_c.c -o metric_gather -Le2e_multi_view_matching_amd -le2emv \
 *       -Wl,-rpath,$PWD/e2e_multi_view_matching_amd
 *   for r in 0 1 2 3 4 5 6 7; do ./metric_gather $r 8 /dev/shm/e2emv_id & done; wait      # rank r runs on GPU r
 *
 * Each rank contributes 4 "pose errors" (rank + 0.25 i), gathers everybody's, reduces their sum and checks both.  With one
 * argument-less call it runs as the only rank (what tests/test_c_example.py does on the 1-GPU box).  Exit code 0 = pass. */
#include <stdio.h>
#include <stdlib.h>

#include "e2emv.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != E2EMV_OK) {                                                                       \
            fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, ctx ? e2emv_last_error(ctx) : "");   \
            return 2;                                                                                \
        }                                                                                            \
    } while (0)

int main(int argc, char** argv) {
    const int rank = argc > 1 ? atoi(argv[1]) : 0, world = argc > 2 ? atoi(argv[2]) : 1;
    const char* id_file = argc > 3 ? argv[3] : "/tmp/e2emv_comm_id_example";
    enum { N = 4 };
    e2emv_ctx* ctx = NULL;
    CHECK(e2emv_create(&ctx, rank)); /* one GPU per rank */
    e2emv_comm* comm = NULL;
    if (rank == 0) remove(id_file);
    CHECK(e2emv_comm_init_file(ctx, id_file, rank, world, 120.0, &comm));

    float local[N], *d_local = NULL, *d_all = NULL, *all = (float*)malloc(sizeof(float) * N * world);
    for (int i = 0; i < N; ++i) local[i] = (float)rank + 0.25f * (float)i;
    CHECK(e2emv_malloc(ctx, (void**)&d_local, sizeof local));
    CHECK(e2emv_malloc(ctx, (void**)&d_all, sizeof(float) * N * world));
    CHECK(e2emv_h2d(ctx, d_local, local, sizeof local, NULL));
    CHECK(e2emv_metric_allgather(ctx, comm, d_local, N, d_all, NULL));
    CHECK(e2emv_d2h(ctx, all, d_all, sizeof(float) * N * world, NULL));
    CHECK(e2emv_metric_allreduce(ctx, comm, d_local, N, E2EMV_REDUCE_SUM, NULL));
    CHECK(e2emv_d2h(ctx, local, d_local, sizeof local, NULL));
    CHECK(e2emv_sync(ctx, NULL));
    int bad = 0;
    for (int r = 0; r < world; ++r)
        for (int i = 0; i < N; ++i) bad += all[r * N + i] != (float)r + 0.25f * (float)i;
    for (int i = 0; i < N; ++i) bad += local[i] != (float)(world * (world - 1) / 2) + 0.25f * (float)i * (float)world;
    int rk = -1, wd = -1;
    CHECK(e2emv_comm_rank(comm, &rk, &wd));
    bad += rk != rank || wd != world;
    printf("rank %d of %d: gathered %d values, %d wrong\n", rank, world, N * world, bad);
    CHECK(e2emv_comm_destroy(ctx, comm));
    e2emv_free(ctx, d_local);
    e2emv_free(ctx, d_all);
    e2emv_destroy(ctx);
    free(all);
    if (rank == 0) remove(id_file);
    return bad ? 1 : 0;
}
