/*
 * madsim_oracle.h — CPU ORACLE (test infrastructure, NOT the product; see madsim_oracle.c).
 * Same workload/config/result types as the product ABI so parity tests feed both the same bytes.
 */
#ifndef MADSIM_ORACLE_H
#define MADSIM_ORACLE_H

#include "../include/madsim_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* High-water marks of the reference's unbounded containers, used to size device capacities. */
typedef struct madsim_oracle_stats {
    uint32_t max_heap;   /* naive_timer BinaryHeap length */
    uint32_t max_ready;  /* ready Vec length */
    uint32_t max_tasks;  /* live futures */
    uint32_t max_msgs;   /* Mailbox.msgs per socket */
    uint32_t max_regs;   /* Mailbox.registered per socket */
    uint32_t max_conns;  /* live connections */
    uint32_t max_cq;     /* queued payloads per channel direction */
} madsim_oracle_stats_t;

/* Events in which a seed leaves the device runner's workload MODEL (not reference concepts; madsim_oracle.c model_event). */
#define MADSIM_ORACLE_ME_TASKS      1u   /* a 255th live task                                                              */
#define MADSIM_ORACLE_ME_REGS       2u   /* a 256th registration in one socket's mailbox                                   */
#define MADSIM_ORACLE_ME_REG_ALIAS  4u   /* a new registration whose 8-bit rxseq / generation bytes equal a dead one's      */
#define MADSIM_ORACLE_ME_CHAN_QUEUE 8u   /* a 16th payload queued in one channel direction                                 */
#define MADSIM_ORACLE_ME_ACCEPTQ    16u  /* a ninth connection waiting in one Endpoint's accept1 queue                     */
#define MADSIM_ORACLE_ME_IPVS       32u  /* a seventh server of one IPVS service                                           */
#define MADSIM_ORACLE_ME_PANIC_DYN  64u  /* a formatted panic value above madsim_workload_t.panic_dyn_max                  */
#define MADSIM_ORACLE_ME_EPH_REBIND 128u /* a port-0 entry bound again beside the live Endpoint of its previous bind       */
#define MADSIM_ORACLE_ME_EPH_STALE  256u /* an op through a port-0 entry that no longer names the socket its last bind made */
#define MADSIM_ORACLE_ME_GUARDS     512u /* a 128th connection end holding one Endpoint's BindGuard                          */
#define MADSIM_ORACLE_ME_EPH_PORTS  1024u /* an ephemeral bind whose port lies beyond the table's candidate ports of that (node, IP) */
#define MADSIM_ORACLE_ME_CONNS      2048u /* a 128th live connection                                                        */
#define MADSIM_ORACLE_ME_MSGS       4096u /* a 256th message queued in one mailbox                                          */

int madsim_oracle_run_batch(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0,
                            uint64_t count, const madsim_limits_t* lim, madsim_result_t* out,
                            madsim_summary_t* summary, madsim_oracle_stats_t* stats);

/* The same restatement with the model's ceilings switched OFF: unbounded containers throughout, `events[i]` = MADSIM_ORACLE_ME_*
 * mask of seed i (0 = the seed never met a ceiling, and its result equals madsim_oracle_run_batch's byte for byte). */
int madsim_oracle_run_batch_pure(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                                 const madsim_limits_t* lim, madsim_result_t* out, uint32_t* events);

/* CPU twin of madsim_hip_run_batch with the identical signature. */
int madsim_cpu_run_batch(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                         const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary);

int64_t madsim_oracle_trace_seed(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed,
                                 const madsim_limits_t* lim, uint8_t* log, uint64_t cap,
                                 madsim_result_t* out);

/* One seed plus the list of values it made observable (MS_OP_TRACE / MS_OP_TRACE_TIME, what obs_hash folds). */
int64_t madsim_oracle_observe_seed(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed,
                                   const madsim_limits_t* lim, uint64_t* obs, uint64_t cap, madsim_result_t* out);

/* Building blocks exposed for the known-answer tests. */
void     oracle_seed_from_u64(uint64_t seed, uint64_t s[4]);
uint64_t oracle_xoshiro_next(uint64_t s[4]);
uint64_t madsim_oracle_gen_range(uint64_t s[4], uint64_t lo, uint64_t hi, uint64_t* ncalls);
void     oracle_uniform_duration_params(uint64_t lo, uint64_t hi, int* mode, uint64_t* low,
                                        uint64_t* range, uint64_t* zone);

#ifdef __cplusplus
}
#endif
#endif
