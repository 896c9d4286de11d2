// pp_kernels.h — the gfx950 kernels of the hot path, one header per stage of the path.  Included once by ppgpu.hip.
#pragma once
#include "../../include/ppgpu.h"
#include "pp_device.h"
#include "pp_k_common.h"       // PPParams, PPEdgeSetup, clearance map, time grids, edge decoding, work queues
#include "pp_k_solve.h"        // pp_k_solve_edges
#include "pp_k_sweep.h"        // pp_k_plan_skips, pp_k_pose_sweep
#include "pp_k_cover.h"        // pp_k_approach_events, pp_k_cover_sweep, pp_k_cover_finish
#include "pp_k_heuristic.h"    // pp_k_heuristic*, pp_k_deferred_list
#include "pp_k_expand.h"       // pp_k_dubins_lengths, pp_k_select_nearest, the push-order pipeline, the round trip's pack / unpack
#include "pp_k_incumbent.h"    // pp_k_best_stage1/2, pp_k_key_min_n
