// vpt_trace_direct.h -- the path states of direct_integrator (render_kernel.cu:1760), shared by the two tracers that run it:
// vpt_trace.hip (rays bound to lanes) and vpt_trace_pool.hip (rays in an LDS pool, waves pick phase-homogeneous batches).
#pragma once

#include "vpt_walk.h"

namespace vpt {

enum : uint32_t {
    PH_IDLE = 0,
    // walking phases
    PH_W_FIRST = 1,   // delta tracking: depth pass + first integrator walk, fused
    PH_W_TRACK = 2,   // delta tracking, integrator
    PH_W_SUN = 3,     // ratio tracking towards the sun
    PH_W_PL = 4,      // ratio tracking towards a point light
    PH_W_SPH = 5,     // ratio tracking after the sphere bounce
    PH_W_EMIT = 6,    // emission march
    PH_W_LAST = 6,
    // transition phases, in successor order
    PH_T_FIRST = 16,
    PH_T_FIRST_DONE = 16,
    PH_T_REPLAY = 17,       // history overflow: restart the integrator from the primary ray
    PH_T_TRACK_DONE = 18,
    PH_T_SUN_DONE = 19,
    PH_T_PL_DONE = 20,
    PH_T_PL_NEXT = 21,
    PH_T_EMIT_CHECK = 22,
    PH_T_EMIT_DONE = 23,
    PH_T_SPH_DONE = 24,
    PH_T_OUTER_SECOND = 25,
    PH_T_OUTER_TOP = 26,
    PH_T_FINISH = 27,
};

}  // namespace vpt
