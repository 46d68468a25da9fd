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
/* behind the mailed counters, zeroed with them at the start of a pass: sums over the timesteps that k_count_alive's workgroups add up
 * together (candidate records, operations, pair records: the fused pass' mail) -- one workgroup summing 438 000 timesteps took 1.5 ms */
#define CTK_CNT_SUM_NC     16
#define CTK_CNT_SUM_NOPS   17
#define CTK_CNT_SUM_NP     18
#define CTK_CNT_ZEROED     24
/* "a background value was written" (the + 1 of len(np.unique(flag)), contrack.py:793) is recorded by the write kernels in one of
 * CTK_ZF_SLOTS words, 64 bytes apart, picked by the workgroup index -- behind the counters, in the same buffer.  One word for all
 * workgroups was an atomicOr on ONE address from every workgroup of eight XCDs (each XCD's L2 keeps showing its stale zero, so the
 * look-before never helped): ~2.7 ns per workgroup, serialised -- it bounded k_relabel_v5 (123 us for 46 k workgroups at 1 degree,
 * 2.3 ms for 722 k at 2000 x 721 x 1440). */
#define CTK_ZF_OFF    64
#define CTK_ZF_SLOTS  1024
#define CTK_ZF_STRIDE 16
#define CTK_CNT_WORDS (CTK_ZF_OFF + CTK_ZF_SLOTS * CTK_ZF_STRIDE)

// overflow bits
#define CTK_OVF_PAIRS 1u
#define CTK_OVF_SEAMS 2u
#define CTK_OVF_RUNS  4u
