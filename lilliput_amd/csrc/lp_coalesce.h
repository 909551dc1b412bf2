// lp_coalesce.h -- concurrent one-image Transform calls share device launches.
//
// The reference's deployment shape is many goroutines, each with its own ImageOps, each calling ops.Transform on one image
// (/root/reference/README.md:82-85, ops.go:83-91, 352-444). Served one by one, every such call pays its own launches, its own
// synchronisations and a decode that cannot fill the device (one 4096 x 4096 image is 9 workgroups' worth of entropy data per kernel),
// and N callers get N engines' worth of arenas. When several Transform calls are in flight at once, the calls that the batched path
// serves with identical results -- a static baseline / progressive JPEG source, JPEG output -- are handed to a small set of dispatcher
// threads instead: each dispatcher owns a batch object (lp_batch.cpp), takes whatever requests with equal options are waiting (up to
// a chunk's worth, no timer: a lone request leaves at once), runs them as ONE lilliput_hip_batch_transform -- the ingest pipeline, the
// fused planes -> thumbnail kernels, one encode launch, one packed D2H -- and wakes the callers. Results are the batched path's, i.e.
// byte-identical to the reference path's for integer scales and within the +-1 LSB contract otherwise, exactly like the direct route;
// an item the batch does not answer with LILLIPUT_OK is run again on the direct route, so error behaviour is the direct route's.
//   LILLIPUT_HIP_COALESCE          = n: coalesce once n Transform calls are in flight (default 3; 0 = never)
//   LILLIPUT_HIP_COALESCE_WORKERS  = dispatcher threads per device (default 4)
//   LILLIPUT_HIP_COALESCE_EXTRA    = dispatcher threads per device that take requests only while LILLIPUT_HIP_COALESCE_EXTRA_AT (64) are waiting (default 4)
//   LILLIPUT_HIP_COALESCE_MAX      = requests per dispatch (default 32)
//   LILLIPUT_HIP_COALESCE_IDLE_MS  = an idle dispatcher destroys its batch (engines, arenas) after this long (default 1000)
//   LILLIPUT_HIP_COALESCE_PINNED_MB = pinned staging slots the callers copy their sources into before queueing (default 0 = none: measured a loss
//                                    inside a 16-CPU container; for hosts with CPUs to spare)
#pragma once
#include <stddef.h>

#include "../../include/lilliput_hip.h"

// counts the Transform calls in flight (constructed at the top of lilliput_image_ops_transform)
struct LpTransformInFlight {
    LpTransformInFlight();
    ~LpTransformInFlight();
    int now;    // calls in flight including this one
};
// true when a call that sees `in_flight` concurrent calls should go through the dispatchers (and this thread is allowed to)
bool lp_coalesce_wanted(int in_flight);
// requests queued for, or being served by, the dispatchers right now (all devices). Deferred Part A serves a recorded chain on its caller's
// thread only while this is 0: once one request has gone to the dispatchers the ones arriving behind it follow, so the two routes do not mix
// under load (profiles/r06_part_a.md section 4).
int lp_coalesce_busy();
bool lp_coalesce_suppressed(); // this thread is inside a batch (LpCoalesceSuppress): its Transform calls stay on the direct route
// Hands one request over and waits for it. Returns true when the batched path served it (LILLIPUT_OK, *out_len set); false = take the
// direct route (not served, failed, or no device).
bool lp_coalesce_transform(int device, const void* src, size_t len, void* dst, size_t cap, const lilliput_batch_options& opt, size_t* out_len);
// The same with the batch item's own status (LILLIPUT_*) as the answer: what lilliput_hip_transform_one returns.
int lp_coalesce_transform_status(int device, const void* src, size_t len, void* dst, size_t cap, const lilliput_batch_options& opt, size_t* out_len);
// Transform calls made from inside a batch (its workers for non-JPEG items, its retry of short streams) must never queue behind the
// batch that issued them: a scope of this suppresses coalescing on the calling thread.
struct LpCoalesceSuppress { LpCoalesceSuppress(); ~LpCoalesceSuppress(); int prev; };
