/*
 * ppgpu.h — C ABI of the MI355X (gfx950) implementation of the ASV planner's
 * per-iteration hot path: state sampling -> Dubins edge generation -> per-edge
 * cost evaluation (occupancy grid + dynamic obstacles + ribbon coverage + heuristic)
 * -> incumbent min-reduce.
 *
 * The reference has no FFI layer; its seam is the C++ virtual
 *   Planner::Stats Planner::plan(const RibbonManager&, const State&, PlannerConfig,
 *                                const DubinsPlan&, double timeRemaining)
 *   (/root/reference/path_planner/src/planner/Planner.h:50-51, called from
 *    path_planner/src/executive/executive.cpp:189-190),
 * with the arithmetic living in non-virtual members underneath it.  Each entry
 * point below names the reference function(s) whose work it replaces; the C++
 * host mirror of the reference classes (path_planner_amd/host) is built on
 * nothing but this header, and INTEGRATION.md shows the binding a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - every function returns PPGPU_OK (0) or a negative PPGPU_E* code and
 *     records a message retrievable with ppgpu_last_error(); the C++ host turns
 *     a non-zero code into std::runtime_error so Executive::planLoop's
 *     try/catch (executive.cpp:191-200) behaves as it does today;
 *   - plain pointers and sizes only; `h_` parameters are host memory, `d_`
 *     parameters are device (HBM) memory owned by the caller;
 *   - all floating point is IEEE double, exactly as in the reference
 *     (path_planner_common/include/path_planner_common/State.h:201-202);
 *   - nothing here falls back to the CPU: if no gfx950 device/code object is
 *     available ppgpu_create fails.
 */
#ifndef PATH_PLANNER_AMD_PPGPU_H
#define PATH_PLANNER_AMD_PPGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPGPU_OK          0
#define PPGPU_EINVAL     (-1)  /* bad argument / shape mismatch                      */
#define PPGPU_EHIP       (-2)  /* a HIP runtime call failed (message has the detail) */
#define PPGPU_ENODEV     (-3)  /* no usable gfx950 device                            */
#define PPGPU_ESTATE     (-4)  /* call order violated (e.g. cost before set_config)  */
#define PPGPU_ECAPACITY  (-5)  /* a fixed device-side capacity was exceeded          */
#define PPGPU_ERCCL      (-6)  /* an RCCL call failed                                */

typedef struct ppgpu_ctx ppgpu_ctx;

/* RibbonManager::Heuristic, same order
 * (path_planner/src/planner/utilities/RibbonManager.h:20-26). */
enum {
    PPGPU_H_MAX_DISTANCE = 0,
    PPGPU_H_TSP_POINT_ALL = 1,
    PPGPU_H_TSP_POINT_K = 2,
    PPGPU_H_TSP_DUBINS_ALL = 3,
    PPGPU_H_TSP_DUBINS_K = 4
};

/* DynamicObstaclesManager implementations behind PlannerConfig::obstaclesManager()
 * (path_planner/src/planner/PlannerConfig.h:98-104). */
enum {
    PPGPU_OBST_NONE = 0,    /* base class: collisionExists == 0 (DynamicObstaclesManager.h:23) */
    PPGPU_OBST_BINARY = 1,  /* BinaryDynamicObstaclesManager.cpp:4-22                           */
    PPGPU_OBST_GAUSSIAN = 2 /* GaussianDynamicObstaclesManager.cpp:3-13 (sum of bivariate normal pdfs, floor 1e-5) */
};

/* The scalars of PlannerConfig (PlannerConfig.h:179-207) plus the process-global
 * ribbon half-width (Ribbon.h:16, Ribbon.cpp:4), the two Edge constants
 * (Edge.h:151-152) and the RibbonManager heuristic settings
 * (RibbonManager.h:184-190).  Defaults are the reference's. */
typedef struct ppgpu_config {
    double max_speed;                   /* 2.5  */
    double slow_speed;                  /* 0.5 ; <= 0 means "same as max" (PlannerConfig.h:168-171) */
    double turning_radius;              /* 8    */
    double coverage_turning_radius;     /* 16   */
    double time_horizon;                /* 30   */
    double time_minimum;                /* 5    */
    double collision_checking_increment;/* 0.05 */
    double start_state_time;            /* PlannerConfig::startStateTime() */
    double ribbon_width;                /* 1.5  Ribbon::RibbonWidth (half width) */
    double collision_penalty_factor;    /* 600  Edge::collisionPenaltyFactor()   */
    double time_penalty_factor;         /* 1    Edge::timePenaltyFactor()        */
    double heuristic_turning_radius;    /* RibbonManager::m_TurningRadius (Dubins-TSP heuristics only) */
    int32_t heuristic;                  /* PPGPU_H_* */
    int32_t tsp_k;                      /* RibbonManager::m_K */
    int32_t branching_factor;           /* 9 */
    int32_t reserved;
} ppgpu_config;

/* One open vertex = what Edge::computeTrueCost reads from its start vertex
 * (Edge.cpp:88,98-99; Vertex.h:180-187): State, g, and the vertex's private
 * RibbonManager (list of ribbons + coverageCompletedTime). 64 bytes. */
typedef struct ppgpu_vertex {
    double x, y, heading, speed, time;  /* State (State.h:201-202)                        */
    double g;                           /* Vertex::currentCost()                          */
    double coverage_completed_time;     /* RibbonManager::coverageCompletedTime(), -1 unset */
    int32_t ribbon_offset;              /* first ribbon of this vertex in the ribbon pool */
    int32_t ribbon_count;
} ppgpu_vertex;

/* Bits of ppgpu_edge_result.flags */
#define PPGPU_F_INFEASIBLE   0x01u /* Edge::infeasible()                                          */
#define PPGPU_F_THROWS       0x02u /* the reference would throw out of computeTrueCost here
                                      (DubinsWrapper::sample on an uninitialised / out-of-range
                                      wrapper, Edge.cpp:178 / DubinsWrapper.cpp:29-35)            */
#define PPGPU_F_RIBBON_OVF   0x04u /* child ribbon list exceeded ribbon_stride or 64, or the heuristic's enumeration limit
                                      * (8 ribbons for the brute-force TSP heuristics; 12 for TspPointRobotNoSplitKRibbons
                                      * while its tree has fewer than 2^21 prefixes): h = 0, f = g, never silent */
#define PPGPU_F_RIBBON_LOST  0x40u /* the child's list outgrew the device's 64 ribbons per vertex while the edge was swept: pieces
                                      * were DROPPED, so the record's coverage state is not the reference's (which has no limit) and
                                      * must not be searched on.  Always together with PPGPU_F_RIBBON_OVF.  Without this bit,
                                      * PPGPU_F_RIBBON_OVF on a record whose child count fits ribbon_stride means only that the
                                      * heuristic was not enumerated: the list itself came back whole */
#define PPGPU_F_DUBINS_ERR   0x08u /* dubins_path_sample failed twice (DubinsWrapper.cpp:43-45)    */
#define PPGPU_F_GOAL         0x10u /* SamplingBasedPlanner::goalCondition(child)                   */
#define PPGPU_F_DONE         0x20u /* child->done()                                                */

/* What Vertex::connect + Edge::computeApproxCost + Edge::computeTrueCost leave
 * behind for one edge (Edge.cpp:68-206, Vertex.cpp:49-64,97-104). 128 bytes, so a
 * wavefront writes one record as a single 128-byte transaction. */
typedef struct ppgpu_edge_result {
    uint32_t flags;            /* PPGPU_F_*                                              */
    uint32_t info;             /* bits 0-7 DubinsPathType, bits 8-15 child ribbon count,
                                  bits 16-31 number of sweep steps executed              */
    double true_cost;          /* Edge::trueCost()                                       */
    double collision_penalty;  /* Edge::getSavedCollisionPenalty()                       */
    double approx_cost;        /* Edge::approxCost()                                     */
    double end_x, end_y, end_heading, end_speed, end_time; /* child State (Edge.cpp:177-178) */
    double g, h, f;            /* child currentCost, approxToGo, f                       */
    double coverage_completed_time; /* child RibbonManager::coverageCompletedTime()      */
    double param[3];           /* DubinsPath::param of the edge's curve                  */
} ppgpu_edge_result;

/* Edge descriptor for the list form of ppgpu_cost_edges: which open vertex,
 * which target state, which (radius, speed) configuration
 * (SamplingBasedPlanner.cpp:58-63,69-79,134-148). */
#define PPGPU_EDGE_COVERAGE 0x1u /* use coverage_turning_radius, coverageAllowed = true */
#define PPGPU_EDGE_SLOW     0x2u /* end state's speed = slow_speed instead of max_speed */
static inline uint64_t ppgpu_edge_pack(uint32_t vertex, uint32_t target, uint32_t cfg) {
    return ((uint64_t)(cfg & 0xffu) << 56) | ((uint64_t)(vertex & 0xffffffu) << 32) | (uint64_t)target;
}

/* An edge whose curve already exists: Vertex::connect(start, DubinsWrapper, coverageAllowed)
 * (Vertex.cpp:28-36) — how AStarPlanner::plan re-costs the previous plan (AStarPlanner.cpp:46-59).
 * Fields are DubinsWrapper's (DubinsWrapper.h:118-120) as serialised by NodeBase.h:201-220. */
typedef struct ppgpu_wrapper_edge {
    int32_t vertex;            /* open vertex the edge starts from                        */
    int32_t coverage_allowed;  /* p.getRho() == config.coverageTurningRadius()            */
    double qi[3];              /* DubinsPath::qi (x, y, yaw)                              */
    double param[3];           /* DubinsPath::param                                       */
    double rho;                /* DubinsPath::rho                                         */
    int32_t type;              /* DubinsPathType                                          */
    int32_t reserved;
    double speed;              /* DubinsWrapper::getSpeed()                               */
    double start_time;         /* start time of the curve (m_StartTime); later than the
                                * vertex's first step: that sample throws, the edge is
                                * infeasible with 0 steps (Edge.cpp:126-133)             */
    double end_time;           /* getEndTime(), possibly truncated by updateEndTime()     */
} ppgpu_wrapper_edge;

/* ------------------------------------------------------------------ lifecycle */

/* Process-level handle: device context, stream, persistent buffers.  The reference
 * constructs a new planner every cycle (executive.cpp:85-90); the handle outlives it. */
int ppgpu_create(int device, ppgpu_ctx** out);
int ppgpu_destroy(ppgpu_ctx* ctx);
const char* ppgpu_last_error(void);
/* Launch on a caller-owned hipStream_t (NULL = the handle's own stream). */
int ppgpu_set_stream(ppgpu_ctx* ctx, void* hip_stream);
int ppgpu_synchronize(ppgpu_ctx* ctx);
/* Size the buffers that grow with the sample set — the sample store, the sampler's scratch, the length table and candidate lists
 * of ppgpu_expand_host for batches of up to max_vertices open vertices — for max_samples samples now, so that an anytime planner
 * that doubles its sample set every iteration (AStarPlanner.cpp:101-102) never meets a device allocation inside its time budget.
 * Optional: without it the buffers grow on demand (powers of two) and stay for the life of the handle.  8 million samples and
 * 16 vertices take 2.3 GB of the 288 GB. */
int ppgpu_reserve_samples(ppgpu_ctx* ctx, int64_t max_samples, int32_t max_vertices);
/* How many times the library has grown a device or pinned-host buffer in this process so far, and the wall time those
 * allocations took.  A caller with a time contract (Planner.h:42: plan() returns before timeRemaining) reads it before and after
 * a cycle: a growth inside the budget costs milliseconds and is what ppgpu_reserve_samples is there to avoid.  Either may be NULL. */
int ppgpu_growth_stats(ppgpu_ctx* ctx, uint64_t* count, double* seconds);
/* Measurement aid: with timing on, every costing launch records HIP events on the handle's stream between its kernels;
 * ppgpu_last_timing waits for the last launch and returns, in milliseconds: ms_solve = pp_k_solve_edges; ms_pose = the pose
 * sweep with its chunk-skip planner (pp_k_plan_skips + pp_k_pose_sweep); ms_cover = pp_k_cover_sweep alone; ms_heuristic =
 * everything else of the launch (pp_k_approach_events, pp_k_deferred_list, the heuristic kernels).  The four add up to the
 * launch.  For a launch that ran as several workspace slices each figure is summed over the slices.  On small launches, with
 * the point heuristics and the binary obstacle model, the cover sweep's wavefront computes the edge's heuristic itself.
 * The events cost a few microseconds per launch: leave timing off in production.
 * ppgpu_past_timing reads an earlier launch: back = 0 is the last one, up to 7 launches back (a ring of event sets), so a
 * caller can time several launches in a row without a host wait between them and collect the durations afterwards. */
int ppgpu_enable_timing(ppgpu_ctx* ctx, int32_t on);
int ppgpu_last_timing(ppgpu_ctx* ctx, double* ms_solve, double* ms_pose, double* ms_cover, double* ms_heuristic);
int ppgpu_past_timing(ppgpu_ctx* ctx, int32_t back, double* ms_solve, double* ms_pose, double* ms_cover, double* ms_heuristic);
/* Measurement aid: how many edges of the last costing launch pp_k_cover_sweep visited.  On large launches the approach prepass
 * finishes the edges whose coverage state machine has nothing to do and hands the cover sweep a packed list of the others; on
 * small launches it is every edge.  (Sliced launches: exact with timing on, the last slice's share otherwise.)  Waits for the launch. */
int ppgpu_last_cover_edges(ppgpu_ctx* ctx, int64_t* n_edges);

/* Device memory for callers that have no HIP toolchain of their own (the C++ host library is built with g++ against this header
 * only): buffers for the `d_` parameters below, on the handle's device, and a blocking copy back to the host that first waits
 * for the handle's stream. */
int ppgpu_device_alloc(ppgpu_ctx* ctx, uint64_t bytes, void** d_out);
int ppgpu_device_free(ppgpu_ctx* ctx, void* d_ptr);
int ppgpu_device_read(ppgpu_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes);
/* The records' way home while the next batch is being costed (SURVEY.md 8 e: "overlap D2H with the next batch"): an asynchronous
 * device-to-host copy on an SDMA engine (hsa_amd_memory_async_copy) instead of on the CUs — hipMemcpyAsync into pinned memory runs
 * as a blit kernel on this stack and takes wave slots from the fp64-bound costing kernels.  The caller makes sure the source is
 * complete (ppgpu_synchronize) and that h_pinned_dst is pinned host memory (hipHostMalloc, torch pin_memory).  One copy in flight
 * per handle: a second call first waits for the first; ppgpu_copy_engine_wait blocks until the copy has landed. */
int ppgpu_copy_engine_read(ppgpu_ctx* ctx, void* h_pinned_dst, const void* d_src, uint64_t bytes);
int ppgpu_copy_engine_wait(ppgpu_ctx* ctx);

/* ---------------------------------------------------------------- world state */

/* PlannerConfig setters + Ribbon::RibbonWidth + RibbonManager heuristic. */
int ppgpu_set_config(ppgpu_ctx* ctx, const ppgpu_config* cfg);

/* Map::isBlocked source.  rows*cols bytes, row 0 = y in [0,res) (i.e. AFTER the
 * row reversal GridWorldMap's loader applies, GridWorldMap.cpp:25), non-zero =
 * blocked; queries outside [0,cols*res) x [0,rows*res) are blocked
 * (GridWorldMap.cpp:84-93).  rows == 0 selects the base Map: nothing is ever
 * blocked and the extremes are +-DBL_MAX (Map.cpp:4-6, Map.h:34). */
int ppgpu_set_grid(ppgpu_ctx* ctx, const uint8_t* h_cells, int32_t rows, int32_t cols, double resolution);

/* BinaryDynamicObstaclesManager contents: n rows of
 * {x, y, heading, speed, time, width, length} exactly as passed to update()
 * (BinaryDynamicObstaclesManager.cpp:24-35, constructor .h:17-19). */
int ppgpu_set_obstacles(ppgpu_ctx* ctx, int32_t model, int32_t n, const double* h_obstacles7);

/* GaussianDynamicObstaclesManager contents (GaussianDynamicObstaclesManager.h:19-49, .cpp:16-47): n rows of
 * {x, y, heading, speed, time} as passed to update(mmsi, x, y, heading, speed, time), followed — when
 * covariance_given != 0 — by the row-major 2x2 covariance {c00, c01, c10, c11} of the second update() overload
 * (rows of 9 doubles); otherwise rows of 5 doubles and the reference's default covariance [[30,10],[10,30]].
 * collisionExists then is the sum of the obstacles' pdfs at the pose, 0 when below 1e-5, and the edge's collision
 * penalty the sum over steps of that value times collision_penalty_factor (Edge.cpp:150-151).
 * Obstacles whose pdf is below 1e-13 over a whole 64-step chunk are skipped (relative effect < 1e-7). */
int ppgpu_set_gaussian_obstacles(ppgpu_ctx* ctx, int32_t n, const double* h_obstacles, int32_t covariance_given);

/* The open-vertex array: n vertices and the pool of their ribbons
 * (4 doubles each: startX, startY, endX, endY; Ribbon.h:126).  Also rebuilds the
 * per-vertex collision-check time grids (Edge.cpp:114-120,173). */
int ppgpu_set_vertices(ppgpu_ctx* ctx, int32_t n, const ppgpu_vertex* h_vertices,
                       int32_t n_ribbons, const double* h_ribbons4);

/* -------------------------------------------------------------------- sampling */

/* StateGenerator(minX,maxX,minY,maxY,minSpeed,maxSpeed,seed,ribbonManager)
 * (StateGenerator.cpp:5-13,33-38).  bounds6 = {minX,maxX,minY,maxY,minSpeed,maxSpeed};
 * n_ribbons < 0 selects the ribbon-less constructor.  Clears the sample store. */
int ppgpu_sampler_init(ppgpu_ctx* ctx, const double* bounds6, uint64_t seed,
                       int32_t n_ribbons, const double* h_ribbons4);

/* SamplingBasedPlanner::addSamples(generator, n) (SamplingBasedPlanner.cpp:157-164):
 * draw n_attempts states from the generator's stream, keep those whose cell is
 * free, append them (in stream order) to the device sample store.
 * *n_total_out = resulting m_Samples.size(). */
int ppgpu_sampler_add(ppgpu_ctx* ctx, int64_t n_attempts, int64_t* n_total_out);

/* Advance the generator by n_attempts states without storing them (rank r of a sharded
 * batch skips the r * batch attempts that belong to lower ranks; SURVEY.md 8 e).
 * Asynchronous: launches only.  Where the stream resumes depends on the skipped draws (a sample projected onto a ribbon
 * consumes a sixth draw, StateGenerator.cpp:21-28), but the position is kept in device memory and the next
 * ppgpu_sampler_add / ppgpu_sampler_skip starts from it there; an error of the skip is reported by the next ppgpu_sampler_add. */
int ppgpu_sampler_skip(ppgpu_ctx* ctx, int64_t n_attempts);

/* Replace the sample (target-state) store with caller data; x/y/heading arrays of n. */
int ppgpu_set_samples(ppgpu_ctx* ctx, int64_t n, const double* h_x, const double* h_y, const double* h_heading);
/* Explicit target states that are not samples — the nearest ribbon endpoint of the vertex being expanded
 * (SamplingBasedPlanner.cpp:66-79), Brown-path seeds (AStarPlanner.cpp:150-162).  They are stored BEHIND the samples
 * (indices *first_index .. *first_index + n - 1 for edge descriptors), are not seen by ppgpu_dubins_lengths /
 * ppgpu_select_nearest, and are dropped by the next ppgpu_sampler_add / ppgpu_set_samples / ppgpu_set_extra_targets. */
int ppgpu_set_extra_targets(ppgpu_ctx* ctx, int32_t n, const double* h_x, const double* h_y, const double* h_heading,
                            int64_t* first_index);

/* Copy samples [first, first+n) out as States {x,y,heading,speed,time} (5 doubles each). */
int ppgpu_get_samples(ppgpu_ctx* ctx, int64_t first, int64_t n, double* h_states5);
int64_t ppgpu_num_samples(ppgpu_ctx* ctx);

/* ------------------------------------------------------------- edge generation */

/* Edge::computeApproxCost for every (vertex in [v0,v0+nv), sample, radius):
 * the Dubins length DubinsWrapper::length() (Edge.cpp:11-20, DubinsWrapper.cpp:9-22).
 * d_lengths receives nv * n_samples * 2 doubles, index ((v-v0)*n_samples + s)*2 + r,
 * r = 0 turning_radius, r = 1 coverage_turning_radius; -1 for pairs closer than
 * collision_checking_increment (SamplingBasedPlanner.cpp:111). */
int ppgpu_dubins_lengths(ppgpu_ctx* ctx, int32_t v0, int32_t nv, double* d_lengths);

/* The k best samples by Dubins length per (vertex, radius) — the result of the
 * lazy scan in SamplingBasedPlanner::expand (SamplingBasedPlanner.cpp:85-133).
 * h_sample_index / h_length receive nv*2*k entries (index -1 = fewer than k). */
int ppgpu_select_nearest(ppgpu_ctx* ctx, int32_t v0, int32_t nv, int32_t k,
                         int32_t* h_sample_index, double* h_length);

/* The same k samples per (vertex, radius), in the ORDER SamplingBasedPlanner::expand pushes their children: the front-to-back
 * order of the reference's `bestSamples` heap array when its nearest-first scan of the samples ends (std::push_heap / std::pop_heap
 * on approximate cost, SamplingBasedPlanner.cpp:82-133,134-149).  Which child std::pop_heap surfaces first among children of
 * exactly equal f depends on it.  Vertices [0, nv) (v0 must be 0); h_sample_index receives nv*2*k entries (-1 = fewer than k).
 * A list the device cannot replay (k >= 64, more candidates than its scratch holds, two candidates of exactly equal cost inside
 * the heap) keeps ascending length and is counted in *h_fallbacks (may be NULL) and in ppgpu_order_fallbacks(). */
int ppgpu_expand_order(ppgpu_ctx* ctx, int32_t v0, int32_t nv, int32_t k, int32_t* h_sample_index, uint32_t* h_fallbacks);
uint64_t ppgpu_order_fallbacks(ppgpu_ctx* ctx);

/* --------------------------------------------------------------- edge costing */

/* Dense form: every vertex in [v0,v0+nv) x every sample in [s0,s0+ns) x the
 * (radius,speed) configurations enabled in cfg_mask (bit c set = configuration c,
 * c = PPGPU_EDGE_* bits, so 0xF = all four).  Edge e of the launch is
 *   e = ((v-v0)*ns + (s-s0))*popcount(cfg_mask) + rank of c in cfg_mask.
 * Results go to d_results[e]; child ribbon lists (ribbon_stride*4 doubles per edge)
 * to d_child_ribbons, which may be NULL.  Asynchronous on the handle's stream. */
int ppgpu_cost_edges_dense(ppgpu_ctx* ctx, int32_t v0, int32_t nv, int64_t s0, int64_t ns,
                           uint32_t cfg_mask, ppgpu_edge_result* d_results,
                           double* d_child_ribbons, int32_t ribbon_stride);

/* List form: n packed descriptors (ppgpu_edge_pack) in device memory. */
int ppgpu_cost_edges_list(ppgpu_ctx* ctx, int64_t n, const uint64_t* d_edges,
                          ppgpu_edge_result* d_results,
                          double* d_child_ribbons, int32_t ribbon_stride);

/* SamplingBasedPlanner::expand (SamplingBasedPlanner.cpp:52-151) for nv open vertices in ONE round trip: upload the
 * vertices (as ppgpu_set_vertices), find the k samples of smallest Dubins length per vertex and radius (:85-133), build every
 * vertex's edges in the order expand() pushes them —
 *     the vertex's nearest-point-to-cover target (h_nearest, 3 doubles {x, y, heading} per vertex, x = NaN: none; :64-81)
 *       at each speed {max, slow if distinct} and each radius {turning, coverage if distinct},
 *     then, per radius, its k winners in the order of the reference's heap array (ppgpu_expand_order), each at each speed (:134-149)
 * — cost them (Vertex::connect + Edge::computeTrueCost + computeApproxToGo) and return descriptors, records and child ribbons
 * compacted vertex by vertex.  Output arrays hold ppgpu_expand_capacity(nv, k) entries; *n_edges receives the count.
 * Synchronous; replaces the open-vertex array and the explicit targets of the handle. */
int64_t ppgpu_expand_capacity(int32_t nv, int32_t k);
int ppgpu_expand_host(ppgpu_ctx* ctx, int32_t nv, const ppgpu_vertex* h_vertices, int32_t n_ribbons, const double* h_ribbons,
                      const double* h_nearest, int32_t k, int64_t* n_edges, uint64_t* h_edges, ppgpu_edge_result* h_results,
                      double* h_child_ribbons, int32_t ribbon_stride);

/* Vertex::computeApproxToGo (Vertex.cpp:49-64) on its own: h of n poses {x, y, heading}, pose i with its own ribbon list
 * (h_ribbon_counts[i] ribbons, 4 doubles each, concatenated in h_ribbons), with the configured heuristic.  The root of a
 * search has no parent edge; this gives it the same arithmetic as every other vertex.  h_out[i] = distance / max_speed *
 * time_penalty_factor; h_flags[i] (may be NULL) receives PPGPU_F_RIBBON_OVF where the list exceeds the heuristic's limit. */
int ppgpu_heuristic_host(ppgpu_ctx* ctx, int32_t n, const double* h_poses3, const int32_t* h_ribbon_counts, const double* h_ribbons,
                         double* h_out, uint32_t* h_flags);

/* Convenience for small batches (the host planner's <= 40 edges per expansion):
 * host descriptors in, host results out, synchronous. */
int ppgpu_cost_edges_host(ppgpu_ctx* ctx, int64_t n, const uint64_t* h_edges,
                          ppgpu_edge_result* h_results,
                          double* h_child_ribbons, int32_t ribbon_stride);

/* Wrapper edges (previous-plan re-costing): host descriptors in, host results out, synchronous.
 * The caller must route a wrapper whose rho differs from the radius its coverage flag implies
 * through ppgpu_cost_edges_* instead (Edge.cpp:78-80 re-solves the curve in that case). */
int ppgpu_cost_wrapper_edges_host(ppgpu_ctx* ctx, int64_t n, const ppgpu_wrapper_edge* h_edges,
                                  ppgpu_edge_result* h_results,
                                  double* h_child_ribbons, int32_t ribbon_stride);

/* Number of edges a dense launch with these arguments produces. */
int64_t ppgpu_dense_edge_count(int32_t nv, int64_t ns, uint32_t cfg_mask);

/* ---------------------------------------------------------- incumbent selection */

/* Best (smallest f, ties -> smallest edge index) feasible edge of the last costing
 * launch — the batch analogue of the incumbent update AStarPlanner.cpp:109-117.
 * goal_only != 0 restricts to children satisfying goalCondition.
 * d_key2 (device, 2 x uint64) = { bit pattern of f (monotone for f >= 0), edge index },
 * UINT64_MAX/UINT64_MAX when no edge qualifies.  Asynchronous. */
int ppgpu_best_edge(ppgpu_ctx* ctx, int64_t n, const ppgpu_edge_result* d_results,
                    int32_t goal_only, uint64_t edge_index_base, uint64_t* d_key2);

/* Lexicographic min of n keys {f bits, edge index} resident on the device into d_key2 — the
 * combine step after an all-gather issued on the caller's own communicator. Asynchronous. */
int ppgpu_key_min(ppgpu_ctx* ctx, int32_t n, const uint64_t* d_keys, uint64_t* d_key2);

/* The communicator of a sharded iteration (SURVEY.md 8 e: the sample batch is split over the GPUs of a node, every rank
 * costs the edges to its own samples, the incumbents are combined once per iteration).  The reference has no counterpart:
 * its incumbent update is AStarPlanner.cpp:109-117 on one thread.  RCCL is loaded at the first call (dlopen), so a
 * single-GPU user never needs it.
 *   one process per GPU:  rank 0 calls ppgpu_comm_unique_id and hands the 128 bytes to the other ranks by any means (a
 *     file, a TCP store, MPI); every rank then calls ppgpu_comm_init_rank on its own handle (collective: returns when all
 *     world ranks have called it).
 *   one process, several GPUs:  ppgpu_comm_init_all on the n handles (one per device, rank i = ctxs[i]); each handle's
 *     ppgpu_allreduce_best must then be issued from its own host thread (the call blocks until every rank has joined).
 * ppgpu_comm_info reads the size and this handle's rank back from RCCL (ncclCommCount / ncclCommUserRank).
 * The handle owns its communicator; ppgpu_destroy releases it.  ppgpu_comm_init_all either gives every handle a communicator
 * or none (on failure the communicators already made are destroyed again). */
#define PPGPU_COMM_ID_BYTES 128
int ppgpu_comm_unique_id(uint8_t* h_id128);
int ppgpu_comm_init_rank(ppgpu_ctx* ctx, int32_t world, int32_t rank, const uint8_t* h_id128);
int ppgpu_comm_init_all(ppgpu_ctx** ctxs, int32_t n);
int ppgpu_comm_info(ppgpu_ctx* ctx, int32_t* world, int32_t* rank);
int ppgpu_comm_destroy(ppgpu_ctx* ctx);
/* ncclCommAbort on the handle's communicator: gives it up WITHOUT waiting for outstanding collectives, which releases a rank
 * that is waiting in a collective another rank never joined (its own failure came first).  Callable from any host thread. */
int ppgpu_comm_abort(ppgpu_ctx* ctx);

/* Global incumbent across the ranks of one node: lexicographic min of the
 * per-rank keys with one RCCL collective over xGMI (all-gather of 16 bytes per rank, then ppgpu_key_min; RCCL has no
 * MINLOC and 64 bits cannot carry a full-precision f and an index).  rccl_comm is an ncclComm_t created by the caller,
 * or NULL for the handle's own communicator (ppgpu_comm_init_*).  In place on d_key2, asynchronous on the handle's stream. */
int ppgpu_allreduce_best(ppgpu_ctx* ctx, void* rccl_comm, uint64_t* d_key2);

#ifdef __cplusplus
}
#endif

#endif
