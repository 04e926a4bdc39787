/* multi_ctx_test.c — plain C against include/madsim_hip.h: one process, one host thread, several GPU contexts.
 *
 * What a `cargo test` process on a multi-GPU node does through the Rust binding of INTEGRATION.md: create one
 * madsim_hip_ctx_t per GPU (argv[1] contexts; on a 1-GPU box they all sit on GPU 0, which exercises the same code),
 * run the seed loop of Builder::run (madsim/src/sim/runtime/builder.rs:129-150) sharded over them with
 * madsim_hip_run_batch_multi, and check that the result is bit-identical to the single-context run.
 * Exit code 0 = identical, 1 = mismatch, 2 = no GPU / library error. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/madsim_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s: %s: %s\n", #call, madsim_hip_strerror(rc_), madsim_hip_last_error()); return 2; } } while (0)

int main(int argc, char** argv) {
    int n_ctx = argc > 1 ? atoi(argv[1]) : 2;
    uint64_t count = argc > 2 ? strtoull(argv[2], NULL, 10) : 10000;
    double loss = argc > 3 ? atof(argv[3]) : 0.01;
    if (n_ctx < 1 || n_ctx > 16) return 2;
    madsim_node_t nodes[5]; madsim_prog_t progs[5]; madsim_sock_t socks[4]; madsim_insn_t insns[128];
    madsim_workload_t w;
    CHECK(madsim_workload_pingpong(4, 16, nodes, progs, socks, insns, 128, &w));
    madsim_config_t cfg; memset(&cfg, 0, sizeof cfg);
    cfg.packet_loss_rate = loss; cfg.lat_lo_ns = 1000000; cfg.lat_hi_ns = 10000000;
    /* a timer-heap capacity some seeds outgrow: exercises the compacted re-run of MADSIM_OVERFLOW seeds on both paths */
    madsim_limits_t lim; memset(&lim, 0, sizeof lim);
    lim.heap_lds_slots = 3; lim.heap_spill_slots = 0;

    int n_dev = argc > 4 ? atoi(argv[4]) : 1;          /* GPUs to spread the contexts over */
    madsim_hip_ctx_t* ctxs[16];
    for (int g = 0; g < n_ctx; g++) CHECK(madsim_hip_ctx_create(g % n_dev, &ctxs[g]));

    madsim_result_t* one = malloc(count * sizeof *one);
    madsim_result_t* many = malloc(count * sizeof *many);
    madsim_summary_t s1, sn;
    CHECK(madsim_hip_ctx_run_batch_auto(ctxs[0], &w, &cfg, 777, count, &lim, one, &s1, 5));
    CHECK(madsim_hip_run_batch_multi(ctxs, n_ctx, &w, &cfg, 777, count, &lim, many, &sn, 5));
    int same = memcmp(one, many, count * sizeof *one) == 0 && s1.first_failing_seed == sn.first_failing_seed &&
               s1.n_failed == sn.n_failed && s1.total_steps == sn.total_steps && s1.total_clock_ns == sn.total_clock_ns;
    uint64_t n_ovf = 0;
    for (uint64_t i = 0; i < count; i++) n_ovf += many[i].verdict == MADSIM_OVERFLOW || many[i].verdict == MADSIM_STEP_LIMIT;
    printf("multi_ctx_test: %d contexts, %llu seeds, first failing seed %llu, %llu failed, %llu runner verdicts left, %s\n",
           n_ctx, (unsigned long long)count, (unsigned long long)sn.first_failing_seed, (unsigned long long)sn.n_failed,
           (unsigned long long)n_ovf, same ? "identical to the single-context run" : "MISMATCH");
    for (int g = 0; g < n_ctx; g++) CHECK(madsim_hip_ctx_destroy(ctxs[g]));
    free(one); free(many);
    return same && n_ovf == 0 ? 0 : 1;
}
