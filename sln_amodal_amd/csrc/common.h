// Shared helpers for the gfx950 kernels (wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sln_amodal.h"

#define SLN_WAVE 64

static inline int sln_launch_status() {
    return hipGetLastError() == hipSuccess ? SLN_OK : SLN_ERR_LAUNCH;
}

static inline int sln_div_up(long a, long b) { return (int)((a + b - 1) / b); }
