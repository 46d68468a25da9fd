// ctk_device.h -- constants shared by the kernels and the host API (internal)
#pragma once
#include <stdint.h>

// device counters (uint32 each)
#define CTK_CNT_PAIRS      0
#define CTK_CNT_SEAMS      1
#define CTK_CNT_OVERFLOW   2
#define CTK_CNT_WROTE_ZERO 3
#define CTK_CNT_ALIVE      4
#define CTK_CNT_UPAIRS     5   /* co-occurrence records that bypassed the LDS hash table */
#define CTK_CNT_TICKET     6   /* workgroups of k_count_alive that have finished */
#define CTK_CNT_POISON     8   /* fused one-call path: nonzero = the speculative (host-sync-free) resolution must be repeated on the
                                  host-driven path (CTK_POISON_* bits, ctk_seam_dev.hip) */
#define CTK_CNT_NOPS       9   /* fused one-call path: relabel operations recorded by k_seam_driver */
#define CTK_CNT_N          16

// overflow bits
#define CTK_OVF_PAIRS 1u
#define CTK_OVF_SEAMS 2u
#define CTK_OVF_RUNS  4u
