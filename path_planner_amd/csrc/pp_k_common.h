// pp_k_common.h — what every costing kernel shares: the launch parameters (PPParams), the per-edge setup record, the clearance map
// and time-grid kernels, edge decoding, the work queues of the resident grids.  Included by pp_kernels.h.
#pragma once
#include "../../include/ppgpu.h"
#include "pp_device.h"

// Everything a costing launch needs, passed by value (kernarg segment, scalar loads).
struct PPParams {
    // PlannerConfig / Edge constants / RibbonManager settings
    double max_speed, slow_speed, rho, rho_cov, horizon, tmin, inc_d, sst, ribw, cpf, tpf;
    double inv_inc_d;                    // 1 / inc_d (host division): first guess of a quotient that is then verified
    int heuristic, tsp_k;
    int fuse_h;                          // the cover sweep's wave goes straight on to the edge's heuristic (see PP_FUSE_HEUR)
    int quiet_finish;                    // pp_k_approach_events finishes the edges whose cover sweep has nothing to do
    int lane_split;                      // ... and splits the ribbon an edge enters itself (the wave then starts inside the corridor run)
    int defer_h;                         // ... unless the edge left its ribbons untouched: then pp_k_heuristic_lanes does it (large launches)
    double h_rho;                        // RibbonManager::m_TurningRadius of the Dubins-TSP heuristics
    // world
    PPGrid grid;
    const PPObst* obst; int n_obst; int obst_model;   // PPGPU_OBST_BINARY, or PPGPU_OBST_GAUSSIAN (then obst points at PPGauss records)
    // open vertices
    const ppgpu_vertex* verts; const double* ribbons; const double* tgrid; int ng; int nverts;
    // targets
    const double* sx; const double* sy; const double* sh; long long n_samples;
    // edges: explicit list, or dense enumeration when edges == nullptr; wedges: edges whose curve is given
    // (Vertex::connect(start, DubinsWrapper, coverageAllowed), Vertex.cpp:28-36) instead of solved
    const unsigned long long* edges; long long n_edges;   // n_edges: of the current slice; total_edges: of the whole launch
    long long total_edges;
    const ppgpu_wrapper_edge* wedges;
    int v0, nv; long long s0, ns; unsigned cfg_mask; int per;
    // outputs
    ppgpu_edge_result* out; double* child; int stride;
    // A launch may be cut into slices of consecutive edges (n_edges = slice size): e_base = first edge of the slice in
    // the caller's list, ws_base = where the slice's workspace starts.
    long long e_base, ws_base;
    // workspace: one PPEdgeSetup per edge from pp_k_solve_edges, then what the pose sweep leaves for the cover sweep
    struct PPEdgeSetup* setup;
    unsigned short* track_hits;          // [edge][ngp]  dynamic-obstacle boxes hit at step k
    unsigned long long* track_eq;        // [edge][nch]  bit k & 63 of word k >> 6: heading(k) == heading(k - 1).  Until the pose sweep has sampled
                                         //              a chunk its word holds, as a double, the heading of the step before the chunk
                                         //              (pp_k_plan_skips leaves it for chunks it does not skip; round 4: one dense row
                                         //              instead of two half-empty ones)
    unsigned* track_chunk_hits;          // [edge][nch]  hits summed over the chunk's executable steps
    double* track_pen;                   // Gaussian model only: [edge][ngp] collisionExists(step k) ...
    double* track_chunk_pen;             // ... and [edge][nch] its sum times the penalty factor over the chunk's executable steps
    struct PPTrackSummary* track_summary;
    unsigned char* track_skip;           // [edge][nch]  1: the pose sweep skips this 64-step chunk (pp_k_plan_skips); NULL: no skipping
    int2* track_far;                     // [edge] {first event the cover sweep's wave has to visit, last event before it} (pp_k_approach_events)
    unsigned long long* work;            // work-queue heads of the per-edge kernels (PP_Q_*), zeroed by pp_k_solve_edges
    unsigned* live_list; unsigned* live_count;     // {workspace slot, list position} of the edges the cover sweep still has to visit (pp_k_approach_events)
    unsigned* defer_list; unsigned* defer_count;   // edges whose heuristic the cover sweep left to pp_k_heuristic_lanes
    unsigned* need_big;                  // set by the cover sweep when some child has 9..12 ribbons (pp_k_heuristic_big then has work)
    struct PPCoverState* cover_state;    // [edge] what the cover sweep's wave hands to pp_k_cover_finish (NULL: every wave finishes its own edges)
    unsigned* hw_list; unsigned* hw_count;   // edges pp_k_cover_finish leaves to pp_k_heuristic_listed (a TSP enumeration of 7 or 8 ribbons)
    int ngp, nch;                        // steps per edge rounded up to whole 64-step chunks, and that many chunks
};

// Phase 0 of an edge (Vertex::connect + Edge::computeApproxCost: which vertex/target/configuration, the Dubins word and
// the constants of its curve), solved with one LANE per edge by pp_k_solve_edges and consumed with scalar loads by the
// one-WAVE-per-edge sweep.  256 bytes (two 128-byte lines; 384 up to round 3: the three segments carried their tprime intervals and
// offsets, which follow from p0 / p1, and a never-used clear-after parameter), device-only.
#define PP_SETUP_MALFORMED 1u   // descriptor out of range
#define PP_SETUP_COLOCATED 2u   // State::isCoLocated(start, end): the reference throws
struct PPEdgeSetupBody {
    PPSegBase seg[3];                      // the bases of the curve's three segments (pp_seg_load_uniform / pp_setup_seg_pose make the rest)
    double p0, p1, p2, hi1;                // DubinsPath::param; hi1 = p0 + p1 as the solver rounded it (where segment 2 begins)
    double qx, qy, rho, rho_inv, length;   // DubinsPath::qi (position), rho, path length
    double wStart, wEnd, speed;            // DubinsWrapper start / end time and speed
    double approx;                         // Edge::approxCost (only read when the record is written)
    unsigned long long omask;              // bit j: obstacle j can come near this edge at all (all ones with more than 64 obstacles)
    int type;                              // DubinsPathType, -1 = no path
    unsigned vi, cbits, sflags;
};
struct __attribute__((aligned(128))) PPEdgeSetup : PPEdgeSetupBody {};
static_assert(sizeof(PPEdgeSetup) == 256 && sizeof(PPEdgeSetupBody) == 248, "PPEdgeSetup is sized for two 128-byte lines");
// this lane's pose on segment i of a record (dubins_path_sample on that segment, un-normalised yaw): the lane-per-edge kernels
// (p0, p1, word: the record's own, which a caller that samples several poses keeps in registers — per pose only the 40-byte base
// is then read, and the three bases share the record's first 128-byte line)
template <bool TAB = false>
__device__ __forceinline__ void pp_setup_seg_pose(const PPEdgeSetupBody* S, int i, double tprime, double p0, double p1, int word, double& ux, double& uy, double& uth) {
    const PPSegBase* g = &S->seg[i];
    pp_curve_seg<TAB>(pp_word_seg_type(word, i), (tprime - pp_seg_o1(i, p0)) - pp_seg_o2(i, p1), g->bx, g->by, g->bth, g->sb, g->cb, ux, uy, uth);
}
template <bool TAB = false>
__device__ __forceinline__ void pp_setup_seg_pose(const PPEdgeSetupBody* S, int i, double tprime, double& ux, double& uy, double& uth) {
    pp_setup_seg_pose<TAB>(S, i, tprime, S->p0, S->p1, S->type, ux, uy, uth);
}
// The lane-per-edge prepasses read a record per LANE.  Straight from memory that is one 64-line gather per field; they stage the
// records of their workgroup in LDS instead (contiguous, coalesced 8-byte-per-lane loads) and read the fields from there.  The
// LDS copy holds the 31 doubles that carry data, at a stride of 31: odd in 8-byte units, so lanes reading one field of
// consecutive records fall on different banks.
#define PP_SETUP_GLOBAL_WORDS 32
#define PP_SETUP_WORDS 31
#define PP_SETUP_LDS_STRIDE 31
static_assert(sizeof(PPEdgeSetup) == 8 * PP_SETUP_GLOBAL_WORDS && sizeof(PPEdgeSetupBody) == 8 * PP_SETUP_WORDS, "the LDS staging of the skip planner copies whole records");

// ------------------------------------------------------------------------------------------
// Clearance map of the occupancy grid (PPGrid::clearance), built whenever a grid is set: chessboard (L-infinity) distance in cells
// to the nearest cell that is blocked or outside the grid, capped at PP_CLEAR_CAP.  The L-infinity distance separates: with
// r(x, y') = distance along row y' from column x to the nearest blocked-or-outside cell, d(x, y) = min over dy of max(|dy|, r(x, y + dy)).
__global__ __launch_bounds__(256) void pp_k_grid_row_clear(const uint32_t* bits, int rows, int cols, int wpr, unsigned char* rowclear) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    const uint32_t* row = bits + (size_t)r * wpr;
    int d = 0;
    for (; d < PP_CLEAR_CAP; d++) {
        const int a = c - d, b = c + d;
        if (a < 0 || b >= cols) break;                                      // the grid's edge is as good as a blocked cell
        if (((row[a >> 5] >> (a & 31)) | (row[b >> 5] >> (b & 31))) & 1u) break;
    }
    rowclear[i] = (unsigned char)d;
}
__global__ __launch_bounds__(256) void pp_k_grid_clear(const unsigned char* rowclear, int rows, int cols, unsigned char* clearance) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    int best = rowclear[i];
    for (int dy = 1; dy < best; dy++) {                                     // a row |dy| away cannot give less than |dy|
        const int up = r + dy, dn = r - dy;
        int m = dy;                                                         // rows outside the grid: blocked at distance |dy|
        if (up < rows && dn >= 0) {
            const int ru = rowclear[(size_t)up * cols + c], rd = rowclear[(size_t)dn * cols + c];
            m = max(dy, min(ru, rd));
        }
        best = min(best, m);
    }
    clearance[i] = (unsigned char)best;
}

// ------------------------------------------------------------------------------------------
// Collision-check time grid, one row per open vertex (Edge.cpp:114-120,173): the reference
// advances `intermediate.time() += timeIncrement` once per step, so step times are a running
// sum, not t0 + k*inc; they depend only on the source vertex's time, hence one table per vertex
// (ng entries), built sequentially by one lane per vertex.
// One wavefront per vertex.  The running sum cannot be reassociated, but it can be GUESSED and CHECKED in parallel: while the
// times stay in one binade, adding the increment to a representable time moves it by the same whole number of ulps every
// step, so from an exact anchor t_s the row is t_s + (k - s) * c with c = fl(t_s + inc) - t_s.  Every lane then verifies the
// reference's own recurrence on its entries, fl(t_k + inc) == t_(k+1): the anchor is exact, so by induction everything before
// the first failing k is the sequential result bit for bit.  At a failure (a binade boundary: the ulp doubles) the next time is
// computed the reference's way and becomes the new anchor.  An increment that falls exactly between two ulps (ties to even
// alternate) would fail every other step: after a few restarts the rest of the row is done by the sequential chain (lane 0 into
// LDS, 2 048 steps at a time, copied out coalesced).  19 us -> 2 us for a 1 500-step row on the planner's 16-vertex round trips.
#define PP_TG_SEG 2048
__global__ __launch_bounds__(64) void pp_k_time_grid(const ppgpu_vertex* verts, int nverts, double sst, double inc_d, double max_speed,
                                                    int ng, double* tgrid) {
    __shared__ double seg[PP_TG_SEG];
    const int v = blockIdx.x;
    if (v >= nverts) return;
    const int lane = threadIdx.x;
    double timeIncrement = inc_d / max_speed;                 // Edge.cpp:114
    double t = verts[v].time;
    double timeSinceStart = t - sst;                          // :117
    double timeNudge = fmod(timeSinceStart, timeIncrement);   // :118
    t += timeNudge;                                           // :119
    double* row = tgrid + (size_t)v * ng;
    int s = 0;                                                // row[s] = t is exact
    for (int tries = 0; s < ng && tries < 8; tries++) {
        const double c = (t + timeIncrement) - t;             // what one step adds on this ulp grid
        int firstBad = ng;
        for (int k0 = s; k0 < ng; k0 += 64) {
            const int k = k0 + lane;
            const double val = t + (double)(k - s) * c;
            const double nxt = t + (double)(k + 1 - s) * c;
            if (k < ng) row[k] = val;
            const bool bad = (k < ng - 1) && !(val + timeIncrement == nxt);      // :173, checked
            const unsigned long long m = __ballot(bad);
            if (m != 0ull) { firstBad = k0 + (int)__builtin_ctzll(m); break; }
        }
        if (firstBad == ng) { s = ng; break; }
        const double tf = t + (double)(firstBad - s) * c;     // verified entry
        t = tf + timeIncrement;                               // the reference's own step across the boundary
        s = firstBad + 1;
    }
    for (int k0 = s; k0 < ng; k0 += PP_TG_SEG) {              // only after repeated failures: the dependent chain
        const int m = (ng - k0) < PP_TG_SEG ? (ng - k0) : PP_TG_SEG;
        if (lane == 0) {
            for (int k = 0; k < m; k++) {
                seg[k] = t;
                t += timeIncrement;                           // :173
            }
        }
        __syncthreads();
        for (int k = lane; k < m; k += 64) row[k0 + k] = seg[k];
        __syncthreads();
        t = __shfl(t, 0, 64);
    }
}

// ------------------------------------------------------------------------------------------
// Work item w of a launch -> position in the caller's edge list.  Explicit lists are taken in order.  The dense enumeration
// is walked configuration-major, highest configuration first: the slow-speed configurations are the long edges (most
// collision-check steps), so the long work is dispatched first and the grid drains on short edges, and the wavefronts of one
// workgroup get edges of similar length.  Records still land at the position the C ABI documents.
__device__ __forceinline__ long long pp_edge_position(const PPParams& p, long long w) {
    if (p.wedges || p.edges) return w;
    const long long Q = p.total_edges / p.per;          // (vertex, sample) pairs
    const long long r = w / Q;
    return (w - r * Q) * p.per + (p.per - 1 - r);
}

// Which (vertex, target, configuration) edge `e` of the launch is: wrapper list, explicit list or dense enumeration.
__device__ __forceinline__ void pp_edge_decode(const PPParams& p, long long e, unsigned& vi, unsigned& target, unsigned& cbits) {
    if (p.wedges) {
        vi = (unsigned)p.wedges[e].vertex;
        target = 0;
        cbits = p.wedges[e].coverage_allowed ? PPGPU_EDGE_COVERAGE : 0u;
    } else if (p.edges) {
        unsigned long long d = p.edges[e];
        target = (unsigned)(d & 0xffffffffull);
        vi = (unsigned)((d >> 32) & 0xffffffull);
        cbits = (unsigned)(d >> 56);
    } else {
        long long q = e / p.per;
        int rank = (int)(e - q * p.per);
        long long vv = q / p.ns;
        target = (unsigned)(p.s0 + (q - vv * p.ns));
        vi = (unsigned)(p.v0 + vv);
        unsigned m = p.cfg_mask;
        for (int i = 0; i < rank; i++) m &= m - 1;   // drop `rank` lowest set bits
        cbits = (unsigned)(__ffs((int)m) - 1);
    }
}

// (Rounds 1-2 also worked out, per edge, the curve parameter beyond which the curve stays clear of every ribbon of its source vertex, so
// that the event walkers could stop there: pp_curve_clear_after, commit c2b5b03.  Round 3 measured it again — 38 us of the solve kernel
// for events the approach lane walks at ~60 instructions each — and switched it off; round 4 removed it.)
// A per-edge kernel can be launched as a resident grid whose waves pull edges from queues, in launch order, instead of one
// workgroup per PP_WPB edges.  Edges differ in length by two orders of magnitude (blocked at the first step ... the full
// horizon); the cover sweep runs 4 waves per SIMD (128 VGPRs) and with dispatcher-placed workgroups the counters show 3.15 of
// those 4 slots occupied on average - a wave that takes its next edge itself leaves none empty (cover sweep: -9 %).  The pose
// sweep (6 waves per SIMD, VALU 97 % busy either way) and the heuristic (3 us of work per edge, about the latency of the
// atomic) measure 4 % and 8 % SLOWER that way and keep the plain launch (tools/ablate.py q0 / q2 / q7).
// One queue head would serialise: a device-scope atomic on one address completes every ~12.5 ns on this part (measured: 236 140
// of them stretch any kernel to 3 ms), so the edges are dealt round-robin onto PP_NQ queues whose heads sit in different memory
// channels; a workgroup works on queue (blockIdx mod PP_NQ) and, when that is empty, on the next ones (a plain look first: an
// exhausted queue stays exhausted).
#define PP_Q_POSE 0
#define PP_Q_COVER 1
#define PP_Q_HEUR 2
#define PP_Q_BIG 3
#define PP_NQ 32
#define PP_QSTRIDE 544               // unsigned long longs between queue heads: 4 KiB + 256 B
#define PP_WORK_WORDS (4 * PP_NQ * PP_QSTRIDE)
#ifndef PP_QUEUE_MASK
#define PP_QUEUE_MASK 2              // which kernels pull from queues: 1 pose sweep, 2 cover sweep, 4 heuristics (others: one workgroup per PP_WPB edges)
#endif
#ifndef PP_Q_CHUNK_COVER
#define PP_Q_CHUNK_COVER 1           // edges a cover-sweep wave takes per atomic
#endif
#ifndef PP_Q_CHUNK_HEUR
#define PP_Q_CHUNK_HEUR 4            // edges a heuristic wave takes per atomic (its edges are short: see pp_next_edge)
#endif
struct PPQueue { int q, dry, left; unsigned long long k; };   // queue drawn from, empty queues seen in a row, rest of the chunk in hand
__device__ __forceinline__ PPQueue pp_queue_init() {
    PPQueue s; s.q = (int)(blockIdx.x % PP_NQ); s.dry = 0; s.left = 0; s.k = 0; return s;
}
// for (PP_EACH_EDGE(idx, kernel bit, queue, n, chunk)) body;  -- either this wave's one edge, or edges from the queues until they are dry
#define PP_EACH_EDGE(idx, bit, kern, n, chunk)                                                                              \
    long long idx = ((PP_QUEUE_MASK) & (bit)) ? pp_next_edge<chunk>(p, kern, qs, n)                                         \
                                              : (long long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); \
    idx < (n);                                                                                                              \
    idx = ((PP_QUEUE_MASK) & (bit)) ? pp_next_edge<chunk>(p, kern, qs, n) : (n)
// next edge of kernel `kern` for this wave, or n when every queue is empty (every wave gets there: the grid always drains).
// Queue q holds the edges q, q + NQ, q + 2 NQ, ...; one atomic takes CHUNK consecutive ones of them.
template <int CHUNK>
__device__ __forceinline__ long long pp_next_edge(const PPParams& p, int kern, PPQueue& s, long long n) {
    if (CHUNK > 1 && s.left > 0) {
        const long long idx = (long long)(s.k * PP_NQ) + __builtin_amdgcn_readfirstlane(s.q);
        if (idx < n) { s.left--; s.k++; return idx; }
        s.left = 0;
    }
    while (s.dry < PP_NQ) {
        unsigned long long* head = p.work + (size_t)(kern * PP_NQ + s.q) * PP_QSTRIDE;
        unsigned long long k = 0;
        if (pp_lane() == 0) {
            k = (s.dry > 0) ? __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;   // someone else's queue: look first
            if ((long long)(k * PP_NQ) + s.q < n) k = atomicAdd(head, (unsigned long long)CHUNK);
        }
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)k);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(k >> 32));
        const int q = __builtin_amdgcn_readfirstlane(s.q);
        const unsigned long long kk = ((unsigned long long)hi << 32) | lo;
        const long long idx = (long long)(kk * PP_NQ) + q;
        if (idx < n) { s.dry = 0; s.left = CHUNK - 1; s.k = kk + 1; return idx; }
        s.q = (s.q + 1 == PP_NQ) ? 0 : s.q + 1;
        s.dry++;
    }
    return n;
}
