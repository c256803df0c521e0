// Shared helpers for the gfx950 kernels (wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/sln_amodal.h"

#define SLN_WAVE 64

// Drop any stale (non-sticky) error another library left on this thread so the
// status returned below reflects THIS entry point's launches only.
static inline void sln_enter() { (void)hipGetLastError(); }

static inline int sln_launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return SLN_OK;
    if (getenv("SLN_DEBUG")) fprintf(stderr, "[sln] HIP error %d: %s\n", (int)e, hipGetErrorString(e));
    return SLN_ERR_LAUNCH;
}

static inline int sln_div_up(long a, long b) { return (int)((a + b - 1) / b); }
