/* fdm_engine_debug.h — measurement / debugging entry points of libfdm_engine.so.
 *
 * NOT part of the drop-in boundary (include/fdm_engine.h): nothing here corresponds to a call of the reference, a
 * host that replaces FastDEM::integrate never needs them.  They exist for bench.py (device-side stopwatch around a
 * batch call), the timeline / soak scripts under scripts/ and the tests that have to see which pipeline a call took. */
#ifndef FDM_ENGINE_DEBUG_H
#define FDM_ENGINE_DEBUG_H

#include "fdm_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* fdm_engine_integrate_device_batch between fdm_engine_timer_start and fdm_engine_timer_stop, i.e. the held-back
 * update of the last scan is launched and fdm_engine_timer_ms returns the device-side duration of the whole batch. */
int fdm_engine_integrate_device_batch_timed(fdm_engine* e, uint32_t count, const fdm_device_scan* scans);

/* After everything enqueued has run: how many entries of the small-scan batch pipeline's scratch sets are not in
 * their clean state (keys, aux words, zero-sign words) — 0 0 0 in a healthy engine. */
int fdm_engine_debug_batch_dirty(fdm_engine* e, uint64_t out[3]);

/* Engine option "dbg_timeline" = 1: start / end time of every block of the last fused large-scan launch (update of
 * scan t | bin of scan t + 1), in ticks of the 100 MHz constant clock — ticks[2*b], ticks[2*b + 1] for block b;
 * blocks [0, *n_update_blocks) are tile-update groups, the rest bin blocks.  Waits for the stream. */
int fdm_engine_debug_timeline(fdm_engine* e, uint64_t* ticks, uint64_t cap_blocks, uint32_t* n_blocks,
                              uint32_t* n_update_blocks);

/* How many batch launches the engine has enqueued since it was created: out[0] small-scan batches (k_mbatch,
 * fdm_multi.hpp), out[1] always 0 (the tile batches of round 4 are gone).  Tests assert with it that a call really took the
 * pipeline they mean to check. */
int fdm_engine_debug_batch_launches(fdm_engine* e, uint64_t out[2]);

#ifdef __cplusplus
}
#endif
#endif
