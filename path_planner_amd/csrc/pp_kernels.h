// pp_kernels.h — the gfx950 kernels of the hot path.  Included once by ppgpu.hip.
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
    unsigned long long* track_eq;        // [edge][nch]  bit k & 63 of word k >> 6: heading(k) == heading(k - 1)
    unsigned* track_chunk_hits;          // [edge][nch]  hits summed over the chunk's executable steps
    double* track_pen;                   // Gaussian model only: [edge][ngp] collisionExists(step k) ...
    double* track_chunk_pen;             // ... and [edge][nch] its sum times the penalty factor over the chunk's executable steps
    struct PPTrackSummary* track_summary;
    unsigned char* track_skip;           // [edge][nch]  1: the pose sweep skips this 64-step chunk (pp_k_plan_skips); NULL: no skipping
    double* track_carry;                 // [edge][nch]  heading of the step before the chunk, for edges that may not cover while turning
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
// one-WAVE-per-edge sweep.  384 bytes, device-only.
#define PP_SETUP_MALFORMED 1u   // descriptor out of range
#define PP_SETUP_COLOCATED 2u   // State::isCoLocated(start, end): the reference throws
struct PPEdgeSetupBody {
    PPSeg seg[3];                          // what the sweep keeps one of in registers at a time
    double qx, qy, rho, rho_inv, length;   // DubinsPath::qi (position), rho, path length
    double wStart, wEnd, speed;            // DubinsWrapper start / end time and speed
    double approx, p0, p1, p2;             // Edge::approxCost, DubinsPath::param (only read when the record is written)
    double tfar;                           // curve parameter (arc length / rho) beyond which the curve stays clear of every ribbon of
                                           // the source vertex (pp_curve_clear_after); +inf: unknown
    int type;                              // DubinsPathType, -1 = no path
    unsigned vi, cbits, sflags;
};
struct __attribute__((aligned(128))) PPEdgeSetup : PPEdgeSetupBody {};
static_assert(sizeof(PPEdgeSetup) == 384 && sizeof(PPEdgeSetupBody) == 360, "PPEdgeSetup is sized for three 128-byte lines");
// The lane-per-edge prepasses read a record per LANE.  Straight from memory that is one 64-line gather per field; they stage the
// records of their workgroup in LDS instead (contiguous, coalesced 8-byte-per-lane loads) and read the fields from there.  The
// LDS copy holds the 45 doubles that carry data, at a stride of 45: odd in 8-byte units, so lanes reading one field of
// consecutive records fall on different banks.
#define PP_SETUP_GLOBAL_WORDS 48
#define PP_SETUP_WORDS 45
#define PP_SETUP_LDS_STRIDE 45

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

// PPEdgeSetup::tfar (pp_curve_clear_after below): built only with -DPP_TFAR since round 3.  Measured on config 3: computing it costs
// pp_k_solve_edges 38 us (148 -> 110) and saves the event walkers nothing any more (cover sweep 1057 -> 1050 us WITHOUT it, approach
// kernel 68 -> 71): the events it spares are the sparse far ones, which the lane-per-edge approach kernel walks at ~60 instructions
// each.  Without it tfar = +inf and both walkers' tests never fire; records are the same either way.
#if !defined(PP_TFAR) && !defined(PP_NO_TFAR)
#define PP_NO_TFAR
#endif
// Phase 0 for every edge of a launch, one lane per edge (Vertex::connect -> Edge::computeApproxCost ->
// DubinsWrapper::set, Edge.cpp:14-18,73-76; Edge::setEnd(wrapper), Edge.cpp:208-216 for wrapper edges).
// From which curve parameter on does a solved curve stay clear of every ribbon of its source vertex?  "Clear" is what
// pp_ribbons_event's bounding-box path needs: outside each ribbon's bounding box grown by the ribbon width (and 2 mm).  cover()
// only ever shortens or splits ribbons, so the pieces an edge is left with lie inside the boxes of the ribbons it started
// with; past that parameter every coverage event finds nothing to contain the vehicle and changes nothing, whatever happened
// before, and the cover sweep stops visiting them (it checks for itself that no piece is short enough to be erased).
// Conservative geometry, one lane per edge: an arc can touch a box only if the bounding square of its circle and (for sweeps up
// to half a turn) the box of its chord grown by the sagitta overlap it; such an arc is cut into eight pieces, each inside its
// own chord-plus-sagitta box, and counts up to the end of the last piece that touches; the straight piece is clipped against
// the box (slab test) and counts up to where it leaves it.
#ifndef PP_CLEAR_SUB
#define PP_CLEAR_SUB 8
#endif
__device__ inline double pp_curve_clear_after(const PPCurve& c, const double* ribbons4, int n, double w) {
    if (n <= 0) return INFINITY;                         // a finished vertex: its events do other things (Edge.cpp:162-170)
    const double rho = c.rho, g = w + 2e-3;
    const double lo[3] = {0.0, c.p0, c.p0 + c.p1}, len[3] = {c.p0, c.p1, c.p2};
    const int typ[3] = {c.t0, c.t1, c.t2};
    // segment end points in world coordinates: start, after segment 0, after segment 1, end of the curve
    double ex2, ey2, eth2;
    pp_curve_seg(c.t2, c.p2, c.b2x, c.b2y, c.b2th, c.s2, c.c2, ex2, ey2, eth2);
    const double px[4] = {c.qx, c.b1x * rho + c.qx, c.b2x * rho + c.qx, ex2 * rho + c.qx};
    const double py[4] = {c.qy, c.b1y * rho + c.qy, c.b2y * rho + c.qy, ey2 * rho + c.qy};
    const double sb[3] = {c.s0, c.s1, c.s2}, cb[3] = {c.c0, c.c1, c.c2};
    double ccx[3], ccy[3], sag[3];
    const double ubx[3] = {0.0, c.b1x, c.b2x}, uby[3] = {0.0, c.b1y, c.b2y};          // segment bases, unit radius, origin at qi
    for (int s = 0; s < 3; s++) {
        ccx[s] = ((typ[s] == 0) ? (ubx[s] - sb[s]) : (ubx[s] + sb[s])) * rho + c.qx;   // centre of the segment's circle (pp_curve_seg)
        ccy[s] = ((typ[s] == 0) ? (uby[s] + cb[s]) : (uby[s] - cb[s])) * rho + c.qy;
        sag[s] = (len[s] <= 3.14159) ? rho * (len[s] * len[s] * 0.125) * (1.0 + 1e-9) + 1e-6 : INFINITY;   // 1 - cos(a/2) <= a^2/8
    }
    const double rr = rho * (1.0 + 1e-9) + 1e-6;
    double tfar = 0.0;
    for (int s = 2; s >= 0 && tfar == 0.0; s--) {        // from the end of the curve: the last segment that touches decides
        if (!(len[s] > 0.0)) continue;
        if (typ[s] != 1) {
            // which boxes can this arc touch at all?
            unsigned long long may = 0ull;
            for (int i = 0; i < n; i++) {
                const double sx = ribbons4[4 * i], sy = ribbons4[4 * i + 1], ex = ribbons4[4 * i + 2], ey = ribbons4[4 * i + 3];
                const double bx0 = fmin(sx, ex) - g, bx1 = fmax(sx, ex) + g, by0 = fmin(sy, ey) - g, by1 = fmax(sy, ey) + g;
                bool touch = (ccx[s] + rr >= bx0) & (ccx[s] - rr <= bx1) & (ccy[s] + rr >= by0) & (ccy[s] - rr <= by1);
                if (touch && sag[s] < INFINITY) {
                    const double x0 = fmin(px[s], px[s + 1]) - sag[s], x1 = fmax(px[s], px[s + 1]) + sag[s];
                    const double y0 = fmin(py[s], py[s + 1]) - sag[s], y1 = fmax(py[s], py[s + 1]) + sag[s];
                    touch = (x1 >= bx0) & (x0 <= bx1) & (y1 >= by0) & (y0 <= by1);
                }
                if (touch) may |= 1ull << i;
            }
            if (may == 0ull) continue;
            // the arc in PP_CLEAR_SUB pieces (each inside its chord's box grown by its sagitta): the last piece that touches a box
            double qx_[PP_CLEAR_SUB + 1], qy_[PP_CLEAR_SUB + 1];
            qx_[0] = px[s]; qy_[0] = py[s]; qx_[PP_CLEAR_SUB] = px[s + 1]; qy_[PP_CLEAR_SUB] = py[s + 1];
            const double sub = len[s] / PP_CLEAR_SUB;
            {
                // the division points by rotating the radius vector (their error, ~1e-15, disappears in the margins)
                double sd, cd;
                pp_sincos_bounded((typ[s] == 0) ? sub : -sub, &sd, &cd);
                double rx = px[s] - ccx[s], ry = py[s] - ccy[s];
                for (int j = 1; j < PP_CLEAR_SUB; j++) {
                    const double nx = rx * cd - ry * sd, ny = rx * sd + ry * cd;
                    rx = nx; ry = ny;
                    qx_[j] = ccx[s] + rx; qy_[j] = ccy[s] + ry;
                }
            }
            const double sg = rho * (sub * sub * 0.125) * (1.0 + 1e-9) + 1e-5;          // sagitta bound, and room for the rotations' rounding
            int last = -1;
            for (int i = 0; i < n; i++) {
                if (!((may >> i) & 1ull)) continue;
                const double sx = ribbons4[4 * i], sy = ribbons4[4 * i + 1], ex = ribbons4[4 * i + 2], ey = ribbons4[4 * i + 3];
                const double bx0 = fmin(sx, ex) - g, bx1 = fmax(sx, ex) + g, by0 = fmin(sy, ey) - g, by1 = fmax(sy, ey) + g;
                for (int j = PP_CLEAR_SUB - 1; j > last; j--) {
                    const double x0 = fmin(qx_[j], qx_[j + 1]) - sg, x1 = fmax(qx_[j], qx_[j + 1]) + sg;
                    const double y0 = fmin(qy_[j], qy_[j + 1]) - sg, y1 = fmax(qy_[j], qy_[j + 1]) + sg;
                    if ((x1 >= bx0) & (x0 <= bx1) & (y1 >= by0) & (y0 <= by1)) { last = j; break; }
                }
            }
            if (last >= 0) tfar = fmax(tfar, lo[s] + fmin(len[s], sub * (double)(last + 1) * (1.0 + 1e-12)));
        } else {
            for (int i = 0; i < n; i++) {
                const double sx = ribbons4[4 * i], sy = ribbons4[4 * i + 1], ex = ribbons4[4 * i + 2], ey = ribbons4[4 * i + 3];
                const double bx0 = fmin(sx, ex) - g, bx1 = fmax(sx, ex) + g, by0 = fmin(sy, ey) - g, by1 = fmax(sy, ey) + g;
                // clip P(u) = P0 + u * d, u in [0, L], against the box (slabs); d = (cos, sin) of the base heading
                const double L = len[s] * rho, dx = cb[s], dy = sb[s];
                double u0 = -1e-6, u1 = L + 1e-6;
                bool miss = false;
                if (fabs(dx) > 1e-12) {
                    const double a = (bx0 - px[s]) / dx, b = (bx1 - px[s]) / dx;
                    u0 = fmax(u0, fmin(a, b) - 1e-6); u1 = fmin(u1, fmax(a, b) + 1e-6);
                } else miss |= (px[s] < bx0 - 1e-6) | (px[s] > bx1 + 1e-6);
                if (fabs(dy) > 1e-12) {
                    const double a = (by0 - py[s]) / dy, b = (by1 - py[s]) / dy;
                    u0 = fmax(u0, fmin(a, b) - 1e-6); u1 = fmin(u1, fmax(a, b) + 1e-6);
                } else miss |= (py[s] < by0 - 1e-6) | (py[s] > by1 + 1e-6);
                if (!miss && u0 <= u1) tfar = fmax(tfar, lo[s] + fmin(u1, L) / rho);
            }
        }
    }
    return tfar + 1e-9;
}

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
#ifndef PP_CR_SOLVE
#define PP_CR_SOLVE true     // the edges' curves with correctly rounded atan2 / acos / sin / cos (pp_cr.h)
#endif
#ifndef PP_SOLVE_MIN_WAVES
#define PP_SOLVE_MIN_WAVES 1
#endif
__global__ __launch_bounds__(256, PP_SOLVE_MIN_WAVES) void pp_k_solve_edges(PPParams p) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    // the queue heads of the kernels that follow (all of them start after this kernel has ended, in stream order)
    if (e < 4 * PP_NQ) p.work[(size_t)e * PP_QSTRIDE] = 0ull;
    if (e == 0 && p.live_count) *p.live_count = 0u;
    if (e == 0 && p.e_base == 0) { *p.need_big = 0u; if (p.defer_count) for (int i = 0; i <= PP_HL_MAX_N; i++) p.defer_count[i] = 0u; if (p.hw_count) *p.hw_count = 0u; }            // raised by the cover sweeps of this launch, read by pp_k_heuristic_big
    if (e >= p.n_edges) return;
    unsigned vi, target, cbits;
    const long long eg = pp_edge_position(p, p.e_base + e);   // position in the caller's edge list; e = position in this slice
    pp_edge_decode(p, eg, vi, target, cbits);
    PPEdgeSetup* __restrict__ O = p.setup + p.ws_base + e;
    PPCurve cv;
    struct { double approx, wStart, wEnd, speed; int type; unsigned vi, cbits, sflags; } S;
    S.vi = vi; S.cbits = cbits; S.sflags = 0; S.type = -1;
    S.approx = S.wStart = S.wEnd = 0; S.speed = 1;
    PPDubins dub;
    dub.p0 = dub.p1 = dub.p2 = 0; dub.type = -1;
    if (vi >= (unsigned)p.nverts || (!p.wedges && (long long)target >= p.n_samples)) {
        S.sflags = PP_SETUP_MALFORMED;
        pp_curve_init<false>(cv, 0, 0, 0, 1.0, dub);
    } else {
        const ppgpu_vertex* V = p.verts + vi;
        const double srcX = V->x, srcY = V->y, srcH = V->heading, srcT = V->time;
        double rho = (cbits & PPGPU_EDGE_COVERAGE) ? p.rho_cov : p.rho;             // Edge.cpp:73-76
        double speed = (cbits & PPGPU_EDGE_SLOW) ? p.slow_speed : p.max_speed;
        if (p.wedges) {
            // the wrapper comes with the edge: DubinsWrapper::fill semantics, start time of ITS curve, possibly truncated end
            const ppgpu_wrapper_edge* W = p.wedges + eg;
            dub.p0 = W->param[0]; dub.p1 = W->param[1]; dub.p2 = W->param[2]; dub.type = W->type;
            if (dub.type < 0 || dub.type > 5) dub.type = -1;
            rho = W->rho; speed = W->speed;
            pp_curve_init<PP_CR_SOLVE>(cv, W->qi[0], W->qi[1], W->qi[2], rho, dub);
            S.wStart = W->start_time; S.wEnd = W->end_time;
            S.approx = (S.wEnd - srcT) * 1.0;                         // Edge::setEnd(wrapper), Edge.cpp:208-216
        } else {
            const double tgtX = p.sx[target], tgtY = p.sy[target], tgtH = p.sh[target];
            if ((srcX == tgtX) && (srcY == tgtY) && (srcH == tgtH)) S.sflags |= PP_SETUP_COLOCATED;   // State::isCoLocated
            pp_dubins_shortest<PP_CR_SOLVE>(srcX, srcY, pp_yaw(srcH), tgtX, tgtY, pp_yaw(tgtH), rho, dub);
            pp_curve_init<PP_CR_SOLVE>(cv, srcX, srcY, pp_yaw(srcH), rho, dub);
            S.approx = cv.length / speed * 1.0;                     // Edge.cpp:17
            S.wStart = srcT;
            S.wEnd = srcT + cv.length / speed;                      // DubinsWrapper::setEndTime
        }
        // the sweeps take sin/cos of (segment base heading +- arc) with the bounded-argument routine: refuse curves whose
        // angles leave its range (a heading of tens of thousands of radians, or NaN) instead of sampling them wrongly
        {
            const double bound = fabs(cv.qth) + cv.p0 + (cv.t1 == 1 ? 0.0 : cv.p1) + cv.p2;
            if (!(bound < 9.0e4)) dub.type = -1;
        }
        S.type = dub.type;
        S.speed = speed;
    }
    pp_curve_segments(cv, O->seg);
    O->qx = cv.qx; O->qy = cv.qy; O->rho = cv.rho; O->rho_inv = cv.rho_inv; O->length = cv.length;
    O->p0 = cv.p0; O->p1 = cv.p1; O->p2 = cv.p2;
    O->approx = S.approx; O->wStart = S.wStart; O->wEnd = S.wEnd; O->speed = S.speed;
    O->type = S.type; O->vi = S.vi; O->cbits = S.cbits; O->sflags = S.sflags;
    double tfar = INFINITY;
#ifndef PP_NO_TFAR
    if (S.type >= 0 && !(S.sflags & PP_SETUP_MALFORMED)) {
        const ppgpu_vertex* V = p.verts + vi;
        tfar = pp_curve_clear_after(cv, p.ribbons + 4 * (size_t)V->ribbon_offset, V->ribbon_count, p.ribw);
    }
#endif
    O->tfar = tfar;
    // Which obstacles can come near this edge at all?  Every sampled pose lies within `travel` (arc length) of the curve's first
    // point and an obstacle moves at most |Speed| * duration during the sweep (the bound the pose sweep applies once per edge);
    // pp_k_plan_skips only looks at these.  Bit j = obstacle j, all ones when there are more than 64.
    unsigned long long omask = 0ull;
    if (p.n_obst > PP_WAVE) omask = ~0ull;
    else if (p.n_obst > 0 && S.type >= 0 && !(S.sflags & PP_SETUP_MALFORMED) && p.ng > 0) {
        const double t0 = p.tgrid[(size_t)vi * p.ng];
        const double endTime = fmin(p.horizon + 1e-12 + p.sst, S.wEnd);
        const double chunkTime = 64.0 * (p.inc_d / p.max_speed);
        const double duration = fmax(endTime - t0, 0.0) + chunkTime;
        const double travel = fmin(cv.length, fmax(endTime - S.wStart, 0.0) * S.speed) + 1e-3;
        for (int j = 0; j < p.n_obst; j++) {
            const PPObst& o = p.obst[j];
            const double dt = t0 - o.Time;
            const double X = o.X + o.Speed * dt * o.cosYaw, Y = o.Y + o.Speed * dt * o.sinYaw;
            const double R = o.reach + travel + fabs(o.Speed) * duration + 1e-3;
            const double dx = cv.qx - X, dy = cv.qy - Y;
            if (!(dx * dx + dy * dy > R * R)) omask |= 1ull << j;
        }
    }
    O->seg[0].pad = (int)(unsigned)(omask & 0xffffffffull);
    O->seg[1].pad = (int)(unsigned)(omask >> 32);
}

// ------------------------------------------------------------------------------------------
// Edge costing = four launches over the same edge list (the fourth, pp_k_heuristic, further down), one wavefront-sized piece
// of work each:
//
//   pp_k_solve_edges  (lane per edge)  phase 0: Vertex::connect + Edge::computeApproxCost: Dubins solve, curve constants,
//                                      and from where on the curve is clear of the vertex's ribbons (pp_curve_clear_after)
//   pp_k_pose_sweep   (wave per edge)  phase A: 64 consecutive collision-check steps at a time: closed-form pose,
//                                      occupancy lookup, dynamic-obstacle box tests  ->  the edge's "track"
//   pp_k_cover_sweep  (wave per edge)  phase B: the sequential coverage state machine of Edge.cpp:153-171, visited only
//                                      at its event steps (ribbon per lane); phase C: end state, last cover, cost, g,
//                                      one 128-byte record per edge
//
// Fused in one kernel the state machine's registers and the pose pipeline's registers are live together and the loop
// spills; apart, the pose sweep is a spill-free streaming kernel.  What it leaves for the cover sweep (the "track") is small:
// per 64-step chunk a word of heading-unchanged bits and a hit count, per edge where the sweep stopped and why.  The poses
// themselves are not stored: the cover sweep recomputes them (pp_window_pose, the same code) for the few windows it visits.
#ifndef PP_WPB
#define PP_WPB 4   // wavefronts (= edges) per workgroup of the per-edge kernels
#endif
#ifndef PP_MIN_WAVES
#define PP_MIN_WAVES 4   // cover sweep: waves per SIMD the register allocator must leave room for (4 = 128 VGPRs: no spills;
                         // 6 measures 5 % faster but turns 43 spilled registers into 5 GB of scratch traffic per launch)
#endif
#define PP_SF64(field) (pp_const_f64(&S->field)[0])
#define PP_SI32(field) (pp_const_i32(&S->field)[0])

// per-edge result of the pose sweep
struct PPTrackSummary {
    int limit;      // steps [0, limit) can execute: the first blocked step, or the first step at/after the edge's end time
    int blocked;    // 1: step `limit` exists and is blocked (Edge.cpp:144-147); 2: sampling step 0 threw (limit = 0, :126-133)
    int dub_err;    // some sampled arc length fell outside the curve even after the reference's 1e-5 retry
    int hits;       // -DPP_SUMMARY_HITS only (measured, not taken: DESIGN.md Appendix B): boxes hit, summed over steps [0, limit); else 0
};

// What the cover sweep's wave knows when its event loop (Edge.cpp:153-171) is over, for pp_k_cover_finish (one LANE per edge) to go
// on from: the rest of computeTrueCost (Edge.cpp:177-205) is scalar work per edge — where the loop stopped, two poses, the last
// cover, the hit sums, the cost, the record — that a whole wave used to do for one edge at a time (287 of the sweep's 1 047 us at
// config 3).  The ribbons as the loop left them travel in the edge's child-ribbon slot.  nrib < 0: the wave finished the edge itself.
#define PP_FINISH_MAX 8              // ribbons a lane takes over at most (it keeps them in registers; longer lists stay with the wave)
struct PPCoverState {
    double cct, endTime;             // RibbonManager::coverageCompletedTime, the edge's (possibly shortened) end time
    int nrib, lastEv, rdt;           // ribbons left, last event visited, `ribbonsDoneTime` (an int: Edge.cpp:92)
    unsigned flags;                  // PPGPU_F_* collected so far
};
// DubinsWrapper::sample (DubinsWrapper.cpp:29-49) -> dubins_path_sample for the 64 steps of one window, one step per lane:
// x, y and the un-normalised yaw.  Used by BOTH sweeps with the same arithmetic, so the cover sweep sees exactly the poses the
// pose sweep tested (it recomputes them for the few windows that hold coverage events instead of reading them back from HBM).
// The constants of the segment the caller is on (cur / cs) live in scalar registers and are swapped when the window moved on.
struct PPCurveHot { double wStart, speed, length, rho, rho_inv, qx, qy; };
__device__ __forceinline__ PPCurveHot pp_curve_hot(const PPEdgeSetup* S) {
    PPCurveHot h;
    h.wStart = PP_SF64(wStart); h.speed = PP_SF64(speed); h.length = PP_SF64(length); h.rho = PP_SF64(rho); h.rho_inv = PP_SF64(rho_inv);
    h.qx = PP_SF64(qx); h.qy = PP_SF64(qy);
    return h;
}
template <bool TAB = false>
__device__ __forceinline__ void pp_window_pose(const PPEdgeSetup* S, const PPCurveHot& c, int& cur, PPSeg& cs, double t, double tFirst, bool valid,
                                               double& x, double& y, double& uth, bool& dubErr) {
    // lanes past the end of the sweep redo lane 0's step (benign arithmetic, uniform control flow); the caller masks them
    const double tl = valid ? t : tFirst;
    double dist = (tl - c.wStart) * c.speed;                          // DubinsWrapper.cpp:36
    if (__ballot((dist < 0) | (dist > c.length)) != 0ull) {           // rare: the first / last step of a curve
        if (dist < 0 || dist > c.length) dist = dist - 1e-5;          // EDUBPARAM retry, :39-42
        if (dist < 0 || dist > c.length) { dubErr = true; dist = fmin(fmax(dist, 0.0), c.length); }
    }
    // dubins_path_sample(): 64 consecutive arc lengths almost always fall on one segment, which is then advanced with
    // wave-uniform constants
    const double tprime = (c.rho_inv != 0.0) ? dist * c.rho_inv : dist / c.rho;
    double ux, uy;
    bool uniformSeg = __ballot(!((tprime >= cs.lo) & (tprime < cs.hi))) == 0ull;
    if (!uniformSeg) {
        const double hi0 = PP_SF64(seg[0].hi), hi1 = PP_SF64(seg[1].hi);
        const int mine = pp_seg_of(tprime, hi0, hi1);
        const int firstSeg = __builtin_amdgcn_readfirstlane(mine);
        const int lastSeg = __builtin_amdgcn_readlane(mine, 63 - __clzll((long long)__ballot(valid)));
        if (__ballot(mine != firstSeg) != 0ull) {
            // the window straddles a junction: every lane takes its own segment's constants from memory
            const PPSeg* g = &S->seg[mine];
            pp_curve_seg<TAB>(g->type, (tprime - g->o1) - g->o2, g->bx, g->by, g->bth, g->sb, g->cb, ux, uy, uth);
        } else {
            uniformSeg = true;
            if (cur != firstSeg) { cur = firstSeg; cs = pp_seg_load_uniform(&S->seg[cur]); }
        }
        if (cur != lastSeg && !uniformSeg) { cur = lastSeg; cs = pp_seg_load_uniform(&S->seg[cur]); }
    }
    if (uniformSeg) pp_curve_seg<TAB>(cs.type, (tprime - cs.o1) - cs.o2, cs.bx, cs.by, cs.bth, cs.sb, cs.cb, ux, uy, uth);
    x = ux * c.rho + c.qx;
    y = uy * c.rho + c.qy;
}

// The cover sweep samples poses only when it loads a window: it re-reads the curve constants there (scalar loads, kept
// inside the loop by laundering the pointer) rather than carrying 33 scalar registers of them through the event loop.
#ifndef PP_COVER_SINCOS_TAB
#define PP_COVER_SINCOS_TAB true    // the cover sweep takes the sine / cosine constants from memory (see pp_sincos_bounded)
#endif
#define PP_WINDOW_POSE(S, t, t0, valid, x, y) do {                                                         \
        const PPEdgeSetup* _S = (S);                                                                       \
        asm volatile("" : "+s"(_S));                                                                       \
        const PPCurveHot _hot = pp_curve_hot(_S);                                                          \
        int _cur = -1;                                                                                     \
        PPSeg _cs = PPSeg{0, 0, 0, 0, 0, INFINITY, -INFINITY, 0, 0, 1, 0};   /* matches nothing: the first use loads a segment */ \
        double _u; bool _e = false;                                                                        \
        pp_window_pose<PP_COVER_SINCOS_TAB>(_S, _hot, _cur, _cs, t, t0, valid, x, y, _u, _e);              \
    } while (0)
// Which 64-step chunks of an edge's sweep can be skipped?  One THREAD per (edge, chunk), in a kernel of its own ahead of the pose
// sweep (inside the sweep the test's registers pushed the per-step loop into spills).  A chunk is skipped when it provably changes
// nothing the sweep records:
//   * all 64 steps exist and lie before the edge's end time, on the curve proper (no retry at the ends);
//   * every pose of the chunk lies within `hs` (arc length from the chunk's middle step, so also Euclidean distance) of the middle
//     pose, and the clearance map says every cell within that distance of the middle pose's cell is free and inside the grid
//     (+2 cells for the pose's place inside its cell and the rounding of the cell index): no step can be blocked;
//   * no obstacle can hold any pose of the chunk: seen from the middle step, the pose stays outside the obstacle's box grown by
//     the distance pose and obstacle can drift apart within the chunk (Gaussian model: outside the 1e-13 radius grown likewise);
//   * on edges that may not cover while turning (Edge.cpp:159) the heading-unchanged bits are known without sampling: the step
//     before the chunk and its last step lie on the same segment of the curve — a straight (the heading is the same expression
//     at every step: all bits set) or an arc whose steps are more than 1e-9 rad apart (no two headings equal: all bits clear).
// A skipped chunk's outputs are stored here (no hits; the heading bits); for a chunk that is NOT skipped on such an edge the
// heading of the step before it is stored (`lastHeading`, Edge.cpp:96,174: the sweep needs it when the chunk before was skipped).
// Everything is the arithmetic the sweep itself would do (pp_window_pose's expressions, one lane's worth).
#ifndef PP_NO_CHUNK_SKIP
#define PP_CHUNK_SKIP 1
#else
#define PP_CHUNK_SKIP 0
#endif
// Can obstacle o hold any pose of a chunk whose middle pose is (x, y) at time tM, when no pose is farther than hs and no time
// farther than ht from the middle step?  The box test of pp_obstacle_hit with both half-extents grown by the distance pose and
// obstacle can drift apart (Gaussian model: the 1e-13 radius grown likewise).  true = certainly not.
template <bool GAUSSIAN>
__device__ __forceinline__ bool pp_chunk_clear_of(const PPObst& o, double x, double y, double tM, double hs, double ht) {
    const double dt = tM - o.Time;
    const double X = o.X + o.Speed * dt * o.cosYaw, Y = o.Y + o.Speed * dt * o.sinYaw;
    const double slack = hs + fabs(o.Speed) * ht + 1e-3;
    const double dx = x - X, dy = y - Y;
    if (GAUSSIAN) {
        const double R = o.reach + slack;
        return dx * dx + dy * dy > R * R;
    }
    const double rx = dx * o.cosYaw - dy * o.sinYaw, ry = dx * o.sinYaw + dy * o.cosYaw;
    return (fabs(rx) > o.halfL + slack) | (fabs(ry) > o.halfW + slack);
}
#define PP_SKIP_ALL 1     // track_skip bits: the chunk is not sampled at all
#define PP_SKIP_GRID 2    // sampled, but no pose of it can lie on a blocked cell
#define PP_SKIP_OBST 4    // sampled, but no pose of it can lie inside an obstacle
#define PP_SKIP_HITS 8    // with PP_SKIP_ALL: every pose of the chunk lies inside some obstacle box (the chunk's hit count is not zero)
#define PP_PLAN_EDGES_MAX 32          // edges a workgroup of the skip planner stages at most (12.3 KB of LDS)
template <bool GAUSSIAN, bool OBST_LDS>
__device__ __forceinline__ void pp_plan_skips_chunk(const PPParams& p, const PPEdgeSetupBody* S, const PPObst* OB, const long long e, const int chunk) {
    {
    const int k0 = chunk * PP_WAVE;
    unsigned char* skipb = p.track_skip + (size_t)e * p.nch + chunk;
    const bool sane = !(S->sflags & (PP_SETUP_MALFORMED | PP_SETUP_COLOCATED)) && S->type >= 0;
    const bool whole = k0 + PP_WAVE - 1 < p.ng;                // only whole chunks can be skipped ...
    const double endTime = fmin(p.horizon + 1e-12 + p.sst, S->wEnd);
    const double* tg = p.tgrid + (size_t)(sane ? S->vi : 0) * p.ng;
    // ... but a chunk cut by the end of the time grid is still sampled, and if the chunk before it is skipped the sweep takes
    // `lastHeading` from here like for any other chunk (tools/fuzz_parity.py seed 17 round 3: an edge of 291 steps on a 300-step grid)
    const double tF = (sane && k0 < p.ng) ? tg[k0] : INFINITY;
    if (!(tF < endTime)) { *skipb = 0; return; }               // the sweep never reaches this chunk: most threads of a short edge
    const bool cov = (S->cbits & PPGPU_EDGE_COVERAGE) != 0;
    const double wStart = S->wStart, speed = S->speed, length = S->length, rho = S->rho, rho_inv = S->rho_inv;
    const double tM = whole ? tg[k0 + PP_WAVE / 2] : tF, tL = whole ? tg[k0 + PP_WAVE - 1] : tF;
    const double tP = (k0 > 0) ? tg[k0 - 1] : 0.0;
    const double dP = (tP - wStart) * speed, dF = (tF - wStart) * speed, dM = (tM - wStart) * speed, dL = (tL - wStart) * speed;
    const bool okGeom = whole && tL < endTime && (dF >= 0.0) && (dL <= length);   // 64 steps, all before the end time, on the curve proper
    const double hs = fmax(dL - dM, dM - dF) * (1.0 + 1e-12) + 1e-9;      // how far (arc length) a step of the chunk is from the middle step
    const double ht = fmax(tL - tM, tM - tF);
    const double hi0 = S->seg[0].hi, hi1 = S->seg[1].hi;
    unsigned long long eqWord = ~0ull;
    double tpP = 0.0;
    int segP = 0;
    bool okHead = true;                                        // the heading-unchanged bits of the chunk are known without sampling
    if (!cov) {
        // the step before the chunk: its heading is what the first step of the chunk is compared with
        tpP = (rho_inv != 0.0) ? dP * rho_inv : dP / rho;
        const double tpL = (rho_inv != 0.0) ? dL * rho_inv : dL / rho;
        segP = pp_seg_of(tpP, hi0, hi1);
        const int segL = pp_seg_of(tpL, hi0, hi1);
        const bool straight = S->seg[segL].type == 1;
        eqWord = straight ? ~0ull : 0ull;
        if (k0 > 0) {
            okHead = (dP >= 0.0) && (segP == segL) && (straight || (tpL - tpP) > 65.0 * 1e-9);
        } else {
            // the first chunk: its first step is compared with the source vertex's heading (`lastHeading` starts there, Edge.cpp:96),
            // which is one evaluation of the sweep's own heading expression — no sine or cosine in it
            const double tpF = (rho_inv != 0.0) ? dF * rho_inv : dF / rho;
            const int segF = pp_seg_of(tpF, hi0, hi1);
            okHead = (dF >= 0.0) && (segF == segL) && (straight || (tpL - tpF) > 64.0 * 1e-9);
            const PPSeg* g = &S->seg[segF];
            const double tt = (tpF - g->o1) - g->o2;
            const double uth0 = (g->type == 1) ? (0.0 + g->bth) : ((g->type == 0) ? (tt + g->bth) : (-tt + g->bth));
            const bool same0 = pp_heading_from_yaw(pp_mod2pi(uth0)) == p.verts[S->vi].heading;
            eqWord = (eqWord & ~1ull) | (same0 ? 1ull : 0ull);
        }
    }
    // Two separate answers: no pose of the chunk can be on a blocked cell; no pose can be inside an obstacle.  Both, with the
    // heading bits known, skip the chunk; one alone still spares the sweep that half of its per-step work (PP_SKIP_* bits).
    bool gridClear = false, obstClear = false;
    int nInside = 0;                                           // obstacles that hold EVERY pose of the chunk (binary model)
    bool decided = false;                                      // every obstacle either holds all poses or none
    if (okGeom) {
#ifndef PP_PLAN_BALL
        // The chunk's poses against the chord between its first and its last pose.  The vehicle moves at constant speed on a curve
        // of curvature <= 1/rho and an obstacle at constant velocity, both linear in the step time: relative to an obstacle's box
        // the pose at time t is within dev = L^2 / (8 rho) of the point of the chord at the same time fraction (a function that
        // vanishes at both ends with second derivative bounded by 1/rho), L = the chunk's arc length.  A box is convex, so
        // both ends inside it shrunk by dev => every pose inside (64 hits per step, known without sampling); both ends beyond one
        // face grown by dev => no pose inside.  At config 3 dev is 0.08 .. 0.16 m where the ball around the middle pose needed 3.5 m.
        const PPSeg* gF = &S->seg[pp_seg_of((rho_inv != 0.0) ? dF * rho_inv : dF / rho, hi0, hi1)];
        const PPSeg* gL = &S->seg[pp_seg_of((rho_inv != 0.0) ? dL * rho_inv : dL / rho, hi0, hi1)];
        double uxF, uyF, uxL, uyL, uthU;
        pp_curve_seg(gF->type, (((rho_inv != 0.0) ? dF * rho_inv : dF / rho) - gF->o1) - gF->o2, gF->bx, gF->by, gF->bth, gF->sb, gF->cb, uxF, uyF, uthU);
        pp_curve_seg(gL->type, (((rho_inv != 0.0) ? dL * rho_inv : dL / rho) - gL->o1) - gL->o2, gL->bx, gL->by, gL->bth, gL->sb, gL->cb, uxL, uyL, uthU);
        const double xF = uxF * rho + S->qx, yF = uyF * rho + S->qy, xL = uxL * rho + S->qx, yL = uyL * rho + S->qy;
        const double Lc = dL - dF;
        const double dev = Lc * Lc / (8.0 * rho) * (1.0 + 1e-9) + 1e-3;
        gridClear = true;
        if (p.grid.rows != 0) {
            // two balls around the quarter points of the chord: every chord point is within L/4 of one of them, every pose within dev
            // of the chord
            const int need = (int)((0.25 * Lc + dev) * p.grid.inv_res) + 2;
            for (int h = 0; h < 2; h++) {
                const double f = h ? 0.75 : 0.25;
                const double x = xF + f * (xL - xF), y = yF + f * (yL - yF);
                const double cx = x * p.grid.inv_res, cy = y * p.grid.inv_res;
                const bool inside = (x >= 0.0) & (y >= 0.0) & (cx < (double)p.grid.cols) & (cy < (double)p.grid.rows);
                int clear = 0;
                if (inside) clear = (int)p.grid.clearance[(size_t)(unsigned)cy * p.grid.cols + (unsigned)cx];
                gridClear = gridClear && inside && (need < PP_CLEAR_CAP) && (clear > need);
            }
        }
        decided = true;
        auto against = [&](const PPObst& o) {
            if (GAUSSIAN) {
                // the 1e-13 radius around the chord's midpoint (no "inside": the density varies)
                if (!pp_chunk_clear_of<true>(o, 0.5 * (xF + xL), 0.5 * (yF + yL), tM, 0.5 * Lc + dev, ht)) decided = false;
                return;
            }
#ifndef PP_PLAN_NO_PRETEST
            {
                // most boxes on an edge's list are nowhere near this chunk: every pose lies within Lc/2 + dev of the chord's midpoint,
                // the box within its own reach of its centre, which moves at most |Speed| ht around where it is at the middle step
                const double dtM = tM - o.Time;
                const double ddx = 0.5 * (xF + xL) - (o.X + o.Speed * dtM * o.cosYaw), ddy = 0.5 * (yF + yL) - (o.Y + o.Speed * dtM * o.sinYaw);
                const double R = o.reach + 0.5 * Lc + dev + fabs(o.Speed) * ht + 1e-3;
                if (ddx * ddx + ddy * ddy > R * R) return;
            }
#endif
            const double dtF = tF - o.Time, dtL = tL - o.Time;
            const double txF = xF - (o.X + o.Speed * dtF * o.cosYaw), tyF = yF - (o.Y + o.Speed * dtF * o.sinYaw);
            const double txL = xL - (o.X + o.Speed * dtL * o.cosYaw), tyL = yL - (o.Y + o.Speed * dtL * o.sinYaw);
            const double rxF = txF * o.cosYaw - tyF * o.sinYaw, ryF = txF * o.sinYaw + tyF * o.cosYaw;
            const double rxL = txL * o.cosYaw - tyL * o.sinYaw, ryL = txL * o.sinYaw + tyL * o.cosYaw;
            const bool out = (fmin(rxF, rxL) > o.halfL + dev) | (fmax(rxF, rxL) < -o.halfL - dev) | (fmin(ryF, ryL) > o.halfW + dev) | (fmax(ryF, ryL) < -o.halfW - dev);
            const bool in = (fmax(fabs(rxF), fabs(rxL)) < o.halfL - dev) & (fmax(fabs(ryF), fabs(ryL)) < o.halfW - dev);
            if (in) nInside++;
            else if (!out) decided = false;
        };
        // only the obstacles that can come near this edge at all (pp_k_solve_edges left the list in the setup record)
        unsigned long long m = ((unsigned long long)(unsigned)S->seg[1].pad << 32) | (unsigned long long)(unsigned)S->seg[0].pad;
        if (p.n_obst > PP_WAVE) m = 0ull;
        while (decided && m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            against(OB[j]);
        }
        if (p.n_obst > PP_WAVE)
            for (int j = 0; j < p.n_obst && decided; j++) against(OB[j]);
        obstClear = decided && nInside == 0;
#else
        const double tpM = (rho_inv != 0.0) ? dM * rho_inv : dM / rho;
        const PPSeg* g = &S->seg[pp_seg_of(tpM, hi0, hi1)];
        double ux, uy, uth;
        pp_curve_seg(g->type, (tpM - g->o1) - g->o2, g->bx, g->by, g->bth, g->sb, g->cb, ux, uy, uth);
        const double x = ux * rho + S->qx, y = uy * rho + S->qy;
        gridClear = true;
        if (p.grid.rows != 0) {
            const double cx = x * p.grid.inv_res, cy = y * p.grid.inv_res;
            const bool inside = (x >= 0.0) & (y >= 0.0) & (cx < (double)p.grid.cols) & (cy < (double)p.grid.rows);
            const int need = (int)(hs * p.grid.inv_res) + 2;
            int clear = 0;
            if (inside) clear = (int)p.grid.clearance[(size_t)(unsigned)cy * p.grid.cols + (unsigned)cx];
            gridClear = inside && (need < PP_CLEAR_CAP) && (clear > need);
        }
        unsigned long long m = ((unsigned long long)(unsigned)S->seg[1].pad << 32) | (unsigned long long)(unsigned)S->seg[0].pad;
        if (p.n_obst > PP_WAVE) m = 0ull;
        obstClear = true;
        while (obstClear && m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            obstClear = pp_chunk_clear_of<GAUSSIAN>(OB[j], x, y, tM, hs, ht);
        }
        if (obstClear && p.n_obst > PP_WAVE)
            for (int j = 0; j < p.n_obst && obstClear; j++) obstClear = pp_chunk_clear_of<GAUSSIAN>(OB[j], x, y, tM, hs, ht);
        decided = obstClear;
#endif
    }
    const bool ok = okGeom && okHead && gridClear && decided;
    *skipb = ok ? (unsigned char)(PP_SKIP_ALL | (nInside > 0 ? PP_SKIP_HITS : 0)) : (unsigned char)((gridClear ? PP_SKIP_GRID : 0) | (obstClear ? PP_SKIP_OBST : 0));
    if (ok) {
        // (the per-step counts of a skipped chunk are not stored: every step is inside the same nInside boxes, and the one reader
        // that can stop inside a skipped chunk — the cover sweep, when coverage completes there — divides the chunk's sum by 64;
        // round 2 wrote them, 128 bytes per such chunk: half of this kernel's 205 MB of writes)
        p.track_chunk_hits[(size_t)e * p.nch + chunk] = (unsigned)(PP_WAVE * nInside);
        if (!cov) p.track_eq[(size_t)e * p.nch + chunk] = eqWord;
        if (GAUSSIAN) p.track_chunk_pen[(size_t)e * p.nch + chunk] = 0.0;
    } else if (!cov && k0 > 0 && dP >= 0.0 && dP <= length) {
        // not skipped: if the chunk before this one is, the sweep takes `lastHeading` from here
        const PPSeg* g = &S->seg[segP];
        double ux, uy, uth;
        pp_curve_seg(g->type, (tpP - g->o1) - g->o2, g->bx, g->by, g->bth, g->sb, g->cb, ux, uy, uth);
        p.track_carry[(size_t)e * p.nch + chunk] = pp_heading_from_yaw(pp_mod2pi(uth));
    }
    }
}
// One workgroup per `epw` consecutive edges (host: as many as give it 256 (edge, chunk) pairs, at most PP_PLAN_EDGES_MAX), one
// THREAD per (edge, chunk) — measured against one lane per edge walking its chunks (0.29 ms at config 3: 24 dependent iterations
// on 3 700 wavefronts) this mapping took 0.21 ms, most threads of a short edge leaving after two loads.  Round 3: the workgroup
// first copies its edges' setup records (contiguous in the workspace) and the obstacle table into LDS.  A thread reads some 40
// fields of its record and ten doubles per obstacle it tests; from memory every one of those was a vector load whose lanes hit
// two or three different lines, ≈ 200 per thread, and the kernel ran at the rate the L1 serves such loads, not at the VALU's.
template <bool GAUSSIAN, bool OBST_LDS>
__device__ __forceinline__ void pp_plan_skips_thread(const PPParams& p, int epw) {
    __shared__ double s_setup[PP_PLAN_EDGES_MAX * PP_SETUP_LDS_STRIDE];
    __shared__ PPObst s_obst[OBST_LDS ? PP_WAVE : 1];
    const int tid = (int)threadIdx.x;
    const long long el0 = (long long)blockIdx.x * epw;
    const int ne = (int)((p.n_edges - el0 < (long long)epw) ? (p.n_edges - el0) : (long long)epw);
    {
        const double* src = reinterpret_cast<const double*>(p.setup + p.ws_base + el0);
        for (int i = tid; i < ne * PP_SETUP_GLOBAL_WORDS; i += 256) {
            const int ed = i / PP_SETUP_GLOBAL_WORDS, w = i - ed * PP_SETUP_GLOBAL_WORDS;
            if (w < PP_SETUP_WORDS) s_setup[ed * PP_SETUP_LDS_STRIDE + w] = src[i];
        }
        if (OBST_LDS) {
            const double* os = reinterpret_cast<const double*>(p.obst);
            double* od = reinterpret_cast<double*>(s_obst);
            for (int i = tid; i < p.n_obst * (int)(sizeof(PPObst) / sizeof(double)); i += 256) od[i] = os[i];
        }
    }
    __syncthreads();
    // (blockIdx.y: further tiles of 256 chunks when one edge alone has more than 256 of them)
    const int t = (int)blockIdx.y * 256 + tid;
    if (t >= ne * p.nch) return;
    const int el = (int)((unsigned)t / (unsigned)p.nch);
    const int chunk = t - el * p.nch;
    const PPEdgeSetupBody* S = reinterpret_cast<const PPEdgeSetupBody*>(&s_setup[el * PP_SETUP_LDS_STRIDE]);
    pp_plan_skips_chunk<GAUSSIAN, OBST_LDS>(p, S, OBST_LDS ? s_obst : p.obst, p.ws_base + el0 + el, chunk);
}
#ifndef PP_PLAN_MIN_WAVES
#define PP_PLAN_MIN_WAVES 8   // 62 VGPRs, no spills; 0.28 -> 0.27 ms against the compiler's own choice (6 waves)
#endif
// the obstacle table in LDS (up to 64 obstacles) / read from memory (more)
__global__ __launch_bounds__(256, PP_PLAN_MIN_WAVES) void pp_k_plan_skips(PPParams p, int epw) { pp_plan_skips_thread<false, true>(p, epw); }
__global__ __launch_bounds__(256) void pp_k_plan_skips_many(PPParams p, int epw) { pp_plan_skips_thread<false, false>(p, epw); }
__global__ __launch_bounds__(256) void pp_k_plan_skips_gaussian(PPParams p, int epw) { pp_plan_skips_thread<true, true>(p, epw); }
__global__ __launch_bounds__(256) void pp_k_plan_skips_gaussian_many(PPParams p, int epw) { pp_plan_skips_thread<true, false>(p, epw); }

// e = the edge's slot in the workspace.  GAUSSIAN: the dynamic obstacles are GaussianDynamicObstaclesManager's (its own
// instantiation: exp() and the density bookkeeping would otherwise cost the common kernel registers).
template <bool GAUSSIAN>
__device__ __forceinline__ void pp_pose_sweep_edge(const PPParams& p, const long long e, uint32_t* gridTile = nullptr) {
    const int lane = pp_lane();
    const PPEdgeSetup* S = p.setup + e;
    PPTrackSummary* sum = p.track_summary + e;
    const unsigned sflags = (unsigned)PP_SI32(sflags);
    const int dubType = PP_SI32(type);
    if ((sflags & (PP_SETUP_MALFORMED | PP_SETUP_COLOCATED)) || dubType < 0) {
        if (lane == 0) { sum->limit = 0; sum->blocked = 0; sum->dub_err = 0; sum->hits = 0; }
        return;
    }
    const unsigned vi = (unsigned)PP_SI32(vi);
    const bool cov = (((unsigned)PP_SI32(cbits)) & PPGPU_EDGE_COVERAGE) != 0;
    const ppgpu_vertex* V = p.verts + vi;
    const double srcH = pp_sgpr(V->heading);
    const PPCurveHot hot = pp_curve_hot(S);
    const double wEnd = PP_SF64(wEnd), wStart = hot.wStart, speed = hot.speed, cvLength = hot.length, cvQx = hot.qx, cvQy = hot.qy;
    const double endTime = fmin(p.horizon + 1e-12 + p.sst, wEnd);    // Edge.cpp:90 (the cover sweep may end the edge earlier)
    const double* tg = p.tgrid + (size_t)vi * p.ng;
    if (p.wedges && p.ng > 0) {
        // a given curve that starts after the vertex's first step: DubinsWrapper::sample throws at that step, the loop
        // catches it, marks the edge infeasible and stops without counting the step (Edge.cpp:126-133)
        const double t0 = pp_const_f64(tg)[0];
        if (t0 < endTime && t0 < wStart) {
            if (lane == 0) { sum->limit = 0; sum->blocked = 2; sum->dub_err = 0; sum->hits = 0; }
            return;
        }
    }
    // the segment of the curve the sweep is on: its constants live in scalar registers, the other two stay in memory
    int cur = 0;
    PPSeg cs = pp_seg_load_uniform(&S->seg[0]);

    unsigned short* thits = p.track_hits + (size_t)e * p.ngp;
    unsigned long long* teq = p.track_eq + (size_t)e * p.nch;
    unsigned* tch = p.track_chunk_hits + (size_t)e * p.nch;
    const bool gaussian = GAUSSIAN;
    // bounds used by the obstacle culling: how far the vehicle / time advance over one 64-step chunk
    const double chunkTime = 64.0 * (p.inc_d / p.max_speed);
    const double chunkSpan = 64.0 * (p.inc_d / p.max_speed) * speed;
    double carryHeading = srcH;                                       // `lastHeading`, Edge.cpp:96
    bool dubErr = false;
    int limit = 0, blocked = 0;
    int totalHits;                                                    // wave-uniform, but kept in a VECTOR register: the loop below has no scalar
    asm volatile("v_mov_b32 %0, 0" : "=v"(totalHits));                // register to spare (every one more is a v_readlane / v_writelane pair per use)
    // Can any obstacle come near this edge at all?  Every sampled pose lies within `travel` (arc length from the start of
    // the curve) of the curve's first point, and an obstacle moves at most |Speed| * duration during the sweep: the same
    // kind of exact bound as the per-chunk culling, applied once.
    bool anyObstacle = false;
    // Up to 64 obstacles: lane i keeps obstacle i's motion for the whole sweep (position at the first step's time, velocity,
    // squared culling radius), so the per-chunk culling below is a dozen instructions and no loads.  The bound is the one
    // pp_obstacle_hits_chunk uses (reach + chunk span + |Speed| * chunk time + slack); it only has to be conservative.
    const bool laneCull = p.n_obst <= PP_WAVE;
    double oX0 = 0, oY0 = 0, oVx = 0, oVy = 0, oR2 = -1.0, cullT0 = 0;
#ifndef PP_ABL_NO_OBST
    if (p.n_obst > 0 && p.ng > 0) {
        const double t0 = pp_const_f64(tg)[0];
        cullT0 = t0;
        const double duration = fmax(endTime - t0, 0.0) + chunkTime;
        const double travel = fmin(cvLength, fmax(endTime - wStart, 0.0) * speed) + 1e-3;
        for (int b = 0; b < p.n_obst && !anyObstacle; b += PP_WAVE) {
            bool near = false;
            if (b + lane < p.n_obst) {
                const PPObst o = p.obst[b + lane];
                const double dt = t0 - o.Time;
                const double X = o.X + o.Speed * dt * o.cosYaw, Y = o.Y + o.Speed * dt * o.sinYaw;
                const double R = o.reach + travel + fabs(o.Speed) * duration + 1e-3;
                const double dx = cvQx - X, dy = cvQy - Y;
                near = !(dx * dx + dy * dy > R * R);
                if (laneCull) {
                    oX0 = X; oY0 = Y; oVx = o.Speed * o.cosYaw; oVy = o.Speed * o.sinYaw;
                    const double Rc = o.reach + chunkSpan + fabs(o.Speed) * chunkTime + 2e-3;
                    oR2 = Rc * Rc;
                }
            }
            anyObstacle = __ballot(near) != 0ull;
        }
    }
#endif

    // Chunks of 64 steps that provably touch neither a blocked cell nor an obstacle are not sampled at all (pp_k_plan_skips decided
    // which, one thread per chunk); the others go through the per-step code below, one step per lane.
    const unsigned char* skipb = p.track_skip ? p.track_skip + (size_t)e * p.nch : nullptr;
    const double* carry = p.track_carry + (size_t)e * p.nch;
    bool afterSkip = false, stop = false;
    for (int g0 = 0; !stop; g0 += PP_WAVE) {
        const unsigned sbits = (skipb && g0 + lane < p.nch) ? (unsigned)skipb[g0 + lane] : 0u;
        const unsigned long long skips = __ballot((sbits & PP_SKIP_ALL) != 0u);
#ifndef PP_NO_PARTIAL_SKIP
        const unsigned long long gclear = __ballot((sbits & PP_SKIP_GRID) != 0u), oclear = __ballot((sbits & PP_SKIP_OBST) != 0u);
#else
        const unsigned long long gclear = 0ull, oclear = 0ull;
#endif
        int ci = 0;
        for (; ci < PP_WAVE; ci++) {
            const int base = (g0 + ci) * PP_WAVE;
#ifdef PP_DBG_TRACE
            if (pp_edge_position(p, p.e_base + (e - p.ws_base)) == (long long)(PP_DBG_TRACE) && lane == 0 && base < 400)
                printf("[pose] chunk at %d: skip %d (eq word %llx)\n", base, (int)((skips >> ci) & 1ull), (unsigned long long)teq[base >> 6]);
#endif
            if ((skips >> ci) & 1ull) { limit = base + PP_WAVE; afterSkip = true; continue; }
            const bool gridClear = ((gclear >> ci) & 1ull) != 0ull, obstClear = ((oclear >> ci) & 1ull) != 0ull;
            const int k = base + lane;
            const double t = (k < p.ng) ? tg[k] : INFINITY;
            const double tFirst = pp_readlane(t, 0);
            if (!(tFirst < endTime)) { limit = base; stop = true; break; }   // `while (intermediate.time() < endTime)`
            // `lastHeading` (Edge.cpp:96,174) of the step before this chunk: the chunks in between were skipped, pp_k_plan_skips left it
            if (!cov && afterSkip) carryHeading = pp_const_f64(carry + (base >> 6))[0];
            afterSkip = false;
            const bool valid = t < endTime;
            double x, y, heading;
            bool blk = false;
            int hits = 0;
            {
                double uth;
                pp_window_pose(S, hot, cur, cs, t, tFirst, valid, x, y, uth, dubErr);
                // the heading itself (:47) only matters for "unchanged since the last step" (Edge.cpp:159), which only matters
                // on edges that may not cover while turning
                heading = cov ? 0.0 : pp_heading_from_yaw(pp_mod2pi(uth));
    #ifndef PP_ABL_NO_GRID
#ifdef PP_GRID_LDS
                if (!gridClear) blk = pp_is_blocked_lds(p.grid, x, y, valid, gridTile);
#else
                if (!gridClear) blk = valid & pp_is_blocked(p.grid, x, y);   // Edge.cpp:144 (pp_k_plan_skips may have ruled it out for the whole chunk)
#endif
    #endif
            }
            double dens = 0;
    #ifndef PP_ABL_NO_OBST
            if (anyObstacle && obstClear) {
                // pp_k_plan_skips: no obstacle can hold a pose of this chunk
            } else if (anyObstacle && laneCull) {                         // :150-151
                // which obstacles can come near this chunk: lane i answers for obstacle i from its registers
                const double dtc = tFirst - cullT0;
                const double ddx = pp_readlane(x, 0) - (oX0 + oVx * dtc), ddy = pp_readlane(y, 0) - (oY0 + oVy * dtc);
                unsigned long long m = __ballot(!(ddx * ddx + ddy * ddy > oR2));      // oR2 = -1 in lanes without an obstacle
                while (m) {
                    const int j = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    if (!gaussian) { if (valid) hits += pp_obstacle_hit(p.obst[j], x, y, t); }
                    else dens += pp_obstacle_pdf(reinterpret_cast<const PPGauss*>(p.obst)[j], x, y, t);
                }
                if (gaussian) { if (dens < 1e-5) dens = 0; if (!valid) dens = 0; }   // GaussianDynamicObstaclesManager.cpp:11
            } else if (anyObstacle) {
                if (!gaussian)
                    hits = pp_obstacle_hits_chunk(p.obst, p.n_obst, x, y, t, valid, pp_readlane(x, 0), pp_readlane(y, 0), tFirst, chunkSpan, chunkTime);
                else
                    dens = pp_obstacle_density_chunk(reinterpret_cast<const PPGauss*>(p.obst), p.n_obst, x, y, t, valid, pp_readlane(x, 0),
                                                     pp_readlane(y, 0), tFirst, chunkSpan, chunkTime);
            }
    #endif
            unsigned long long eqMask = ~0ull;
            if (!cov) {
                double prevHeading = __shfl_up(heading, 1, PP_WAVE);
                if (lane == 0) prevHeading = carryHeading;
                eqMask = __ballot(prevHeading == heading);
#ifdef PP_DBG_TRACE
                if (pp_edge_position(p, p.e_base + (e - p.ws_base)) == (long long)(PP_DBG_TRACE) && lane == 0 && base < 400)
                    printf("[pose] chunk at %d sampled: carry %.17g heading0 %.17g heading1 %.17g eq %llx\n", base, prevHeading, heading, pp_readlane(heading, 1), (unsigned long long)eqMask);
#endif
                carryHeading = pp_readlane(heading, 63);
            }

            const unsigned long long bm = __ballot(blk);
            const int fb = bm ? (__ffsll((long long)bm) - 1) : PP_WAVE;
            const int nvalid = __popcll(__ballot(valid));
            const int nlim = fb < nvalid ? fb : nvalid;

            int chunkHits = 0;
            if (__ballot(hits != 0) != 0ull) {
                // per-step counts are only ever read for a chunk whose sum is not zero
                chunkHits = pp_wave_sum_i(lane < nlim ? hits : 0);
                thits[k] = (unsigned short)(hits > 65535 ? 65535 : hits);
#ifdef PP_SUMMARY_HITS
                totalHits += chunkHits;
#endif
            }
            if (gaussian) {
                double chunkPen = 0;
                if (__ballot(dens != 0.0) != 0ull) {
                    chunkPen = pp_wave_sum_d(lane < nlim ? dens * p.cpf : 0.0);
                    p.track_pen[(size_t)e * p.ngp + k] = dens;
                }
                if (lane == 0) p.track_chunk_pen[(size_t)e * p.nch + (base >> 6)] = chunkPen;
            }
            if (lane == 0) {
                tch[base >> 6] = (unsigned)chunkHits;
                if (!cov) teq[base >> 6] = eqMask;                        // only read for edges that may not cover while turning
            }

            if (fb < nvalid) { limit = base + fb; blocked = 1; stop = true; break; }
            if (nvalid < PP_WAVE) { limit = base + nvalid; stop = true; break; }
            limit = base + PP_WAVE;
        }
        // skipped chunks the sweep passed whose every pose lies inside some box: the planner's count, 64 per box (rare; kept out of
        // the per-chunk path above, which runs 3.5 million times per launch)
#ifdef PP_SUMMARY_HITS
        // (the skip bytes are read again rather than kept across the loop: the loop has no scalar register to spare)
        const unsigned sb2 = (skipb && g0 + lane < p.nch) ? (unsigned)skipb[g0 + lane] : 0u;
        unsigned long long hm = __ballot((sb2 & (PP_SKIP_ALL | PP_SKIP_HITS)) == (PP_SKIP_ALL | PP_SKIP_HITS)) & ((ci >= PP_WAVE) ? ~0ull : ((1ull << ci) - 1ull));
        while (hm) {
            const int cj = __ffsll((long long)hm) - 1;
            hm &= hm - 1;
            totalHits += (int)pp_const_i32(tch + (g0 + cj))[0];
        }
#endif
    }
    const int anyErr = (__ballot(dubErr) != 0ull) ? 1 : 0;
    if (lane == 0) { sum->limit = limit; sum->blocked = blocked; sum->dub_err = anyErr; sum->hits = totalHits; }
}

// How many steps pass before the next coverage event: the loop of Edge.cpp:153-154 subtracts the increment from toCoverDistance
// once per step while it is above the increment, so after an event that measured D the next one is m + 1 steps on, m = the
// number of subtractions.  m is guessed as ceil(D / inc - 1) and accepted when the remainder is clearly inside (0, inc); within
// 1e-12 of a boundary the subtraction runs literally.
__device__ __forceinline__ int pp_event_stride(double D, double inc_d, double inv_inc_d, int ng) {
    int m = 0;
    if (D > inc_d) {
        const double qd = D * inv_inc_d;                   // a guess good to an ulp or two; m0 is verified below
        if (qd > (double)(ng + 2)) {
            m = ng + 1;                                    // beyond the grid: never again
        } else {
            const int m0 = (int)ceil(qd - 1.0);
            const double r = fma(-(double)m0, inc_d, D);   // D - m0*inc, one rounding
            const double margin = (double)m0 * D * 5e-16 + 1e-12;
            if (m0 >= 1 && r > margin && r < inc_d - margin) {
                m = m0;                                    // the running subtraction cannot differ
            } else {
                double tc = D;                             // too close to call: do it the long way
                while (tc > inc_d && m <= ng) { tc -= inc_d; m++; }
            }
        }
    }
    return m;
}

// The approach to the ribbons, one LANE per edge.  Until the vehicle first comes within reach of a ribbon (inside some
// ribbon's bounding box grown by the ribbon width: the test of pp_ribbons_event's fast path) a coverage event changes nothing
// and only yields the index of the next one, from the distance to the nearest ribbon endpoint.  That chain is sequential per
// edge but independent across edges; walked by the edge's own wavefront it costs a 64-lane window of poses per event to use one
// pose (2.5 of the 3.85 one-at-a-time events per edge at config 3).  Here 64 edges walk their chains side by side — pose,
// boxes and distance per lane with the expressions of pp_window_pose / pp_ribbons_event, so every number is the one the
// wavefront would have computed — and each hands over {next event, last event visited} where its chain meets a ribbon, runs
// past the sweep's limit or end time, or passes the point from which the curve stays clear of all ribbons (PPEdgeSetup::tfar).
// The cover sweep starts its state machine there instead of at step 0.
//
// Quiet edges.  When the chain ends without meeting a ribbon (past the sweep's limit, or past PPEdgeSetup::tfar) the cover sweep's
// event loop has nothing to do for this edge, and unless the last cover (Edge.cpp:182-191) happens within reach of a ribbon
// the rest of computeTrueCost is scalar work: where the loop stopped, two poses, the hit sums, the cost, the record, a copy of the
// vertex's ribbons.  The lane does that too (pp_finish_quiet_edge: phase C of pp_cover_sweep_edge, the same expressions, for the
// case "no event changed anything") and marks the edge PP_FAR_DONE; the cover sweep's wave then drops it at once.  Nearly half
// the edges of config 3.
// One ribbon's part of a coverage event at (x, y), lane form (the expressions of pp_k_cover_finish / pp_ribbons_event): does the ribbon
// contain the point (RibbonManager::minDistanceFrom then returns 0) and does it contain it strictly (cover() would split it)?
// Only called for a ribbon whose grown bounding box holds the point; outside it neither can be.
__device__ __forceinline__ void pp_lane_ribbon_contains(double sx, double sy, double ex, double ey, double x, double y, double w, bool& inside, bool& strict) {
    const double T = PP_RIBBON_TOL;
    const double dxr = ex - sx, dyr = ey - sy;
    const double sqL = dxr * dxr + dyr * dyr;
    const double dot = (x - sx) * dxr + (y - sy) * dyr;
    const double px = dxr * dot / sqL + sx;                  // Ribbon::getProjection (Ribbon.cpp:72-78)
    const double py = dyr * dot / sqL + sy;
    const double a1 = px - sx, a2 = px - ex, b1 = py - sy, b2 = py - ey;
    const bool outx = ((a1 < -T) & (a2 < -T)) | ((a1 > T) & (a2 > T));
    const bool outy = ((b1 < -T) & (b2 < -T)) | ((b1 > T) & (b2 > T));
    const bool cp = !(outx | outy);                          // Ribbon::containsProjection (:90-95)
    const double num = dyr * x - dxr * y + ex * sy - ey * sx;
    const double ld = fabs(num) / sqrt(sqL);                 // Ribbon::distance (Ribbon.h:118-121)
    inside = cp && (ld < w);
    strict = cp && (ld < (w / 2.0));
}
#define PP_FAR_DONE (-2)
__device__ __forceinline__ void pp_lane_pose(const PPEdgeSetupBody* S, double t, double wStart, double speed, double length, double rho, double rho_inv,
                                             double qx, double qy, double hi0, double hi1, double& x, double& y, double& uth, bool& err) {
    double dist = (t - wStart) * speed;                                     // DubinsWrapper.cpp:36
    if (dist < 0 || dist > length) dist = dist - 1e-5;                      // EDUBPARAM retry, :39-42
    if (dist < 0 || dist > length) { err = true; dist = fmin(fmax(dist, 0.0), length); }
    const double tprime = (rho_inv != 0.0) ? dist * rho_inv : dist / rho;
    const PPSeg* g = &S->seg[pp_seg_of(tprime, hi0, hi1)];
    double ux, uy;
    pp_curve_seg(g->type, (tprime - g->o1) - g->o2, g->bx, g->by, g->bth, g->sb, g->cb, ux, uy, uth);
    x = ux * rho + qx;
    y = uy * rho + qy;
}
// -> true: the edge's record and child ribbons are written.  false: nothing was written, the wave does the edge.
// `stage` = this lane's 16 doubles of LDS (stride PP_REC_STRIDE): the record goes there, and the wave then stores the records of its
// lanes together, 4 records of 128 contiguous bytes per store instruction instead of 64 different lines per field.
#define PP_REC_STRIDE 17
__device__ __forceinline__ bool pp_finish_quiet_edge(const PPParams& p, const PPEdgeSetupBody* S, const ppgpu_vertex* V, long long e, long long eg,
                                                     int limit, int lastEv, const double* rp, const double* tg, double* stage, bool rpUniform) {
    const int nrib = V->ribbon_count;                                       // > 0, no piece short enough to be erased
    const PPTrackSummary* sum = p.track_summary + e;
    if (sum->dub_err) return false;
    if (nrib > p.stride || nrib > PP_TSP_MAX) return false;
    if (p.n_obst > 0 && p.obst_model == PPGPU_OBST_GAUSSIAN) return false;
    // the heuristic must not need this edge's wave either
    const bool deferH = p.defer_h && pp_lane_tsp_ok(p.heuristic, p.tsp_k, nrib);
    if (p.fuse_h && !deferH && p.heuristic != PPGPU_H_MAX_DISTANCE) return false;
    const double wStart = S->wStart, wEnd = S->wEnd, speed = S->speed, length = S->length, rho = S->rho, rho_inv = S->rho_inv, qx = S->qx, qy = S->qy;
    const double hi0 = S->seg[0].hi, hi1 = S->seg[1].hi;
    const double srcT = V->time;
    const bool cov = (S->cbits & PPGPU_EDGE_COVERAGE) != 0;
    const double endTime = fmin(p.horizon + 1e-12 + p.sst, wEnd);           // Edge.cpp:90; no event shortened it
    bool infeasible = (srcT >= endTime);                                    // :102-110
    const int stopKind = sum->blocked;
    // where the loop of Edge.cpp:143-175 stopped (no event: every step below `limit` ran)
    int steps, hexec, lastIdx;
    double tfinal;
    bool coverFinal = true;
    const int nexec = limit;                                                // max(cnt, lastEv + 1), cnt = limit
    (void)lastEv;
    if (stopKind == 1 && tg[limit] < endTime) {                             // `break` at :146
        infeasible = true;
        lastIdx = limit;
        coverFinal = cov || (((p.track_eq[(size_t)e * p.nch + (limit >> 6)] >> (limit & 63)) & 1ull) != 0ull);
        tfinal = tg[limit];
        steps = limit + 1;
        hexec = limit;
    } else {
        if (stopKind == 2 && tg[0] < endTime) infeasible = true;
        lastIdx = nexec - 1;
        tfinal = (nexec < p.ng) ? tg[nexec] : INFINITY;
        steps = nexec;
        hexec = nexec;
    }
    (void)tfinal;                                                           // only used when the ribbons run out: they do not here
    if (!(wStart <= endTime && wEnd >= endTime)) return false;              // DubinsWrapper::containsTime: the reference throws
    double ix = V->x, iy = V->y, uth;
    bool perr = false, ignored = false;
    if (lastIdx >= 0) pp_lane_pose(S, tg[lastIdx], wStart, speed, length, rho, rho_inv, qx, qy, hi0, hi1, ix, iy, uth, ignored);
    double endX, endY;
    pp_lane_pose(S, endTime, wStart, speed, length, rho, rho_inv, qx, qy, hi0, hi1, endX, endY, uth, perr);
    if (perr) return false;
    const double endHeading = pp_heading_from_yaw(pp_mod2pi(uth));
    if (cov || coverFinal) {                                                // the last cover (:182-191): only if it cannot touch a ribbon
        const double grow = p.ribw + 1e-3;
        bool inBox = false;
        if (rpUniform) {                                                    // one vertex for the whole wave: its ribbons through scalar loads
            const PP_AS4 double* ru = pp_const_f64(rp);
            for (int i = 0; i < nrib; i++) {
                const double sx = ru[4 * i], sy = ru[4 * i + 1], ex = ru[4 * i + 2], ey = ru[4 * i + 3];
                inBox |= (ix >= fmin(sx, ex) - grow) & (ix <= fmax(sx, ex) + grow) & (iy >= fmin(sy, ey) - grow) & (iy <= fmax(sy, ey) + grow);
            }
        } else
        for (int i = 0; i < nrib; i++) {
            const double sx = rp[4 * i], sy = rp[4 * i + 1], ex = rp[4 * i + 2], ey = rp[4 * i + 3];
            inBox |= (ix >= fmin(sx, ex) - grow) & (ix <= fmax(sx, ex) + grow) & (iy >= fmin(sy, ey) - grow) & (iy <= fmax(sy, ey) + grow);
        }
        if (inBox) return false;
    }
    int hitsTotal = 0;
#ifdef PP_SUMMARY_HITS
    if (p.n_obst > 0 && hexec == limit) {
        hitsTotal = sum->hits;                                              // every step below `limit` ran: the pose sweep's own total
    } else
#endif
    if (p.n_obst > 0) {
        const unsigned* tch = p.track_chunk_hits + (size_t)e * p.nch;
        const int cfull = hexec >> 6;
        for (int c = 0; c < cfull; c++) hitsTotal += (int)tch[c];
        if ((hexec & 63) != 0 && tch[cfull] != 0u) {
            if (p.track_skip && (p.track_skip[(size_t)e * p.nch + cfull] & PP_SKIP_ALL) != 0) {
                hitsTotal += (hexec & 63) * (int)(tch[cfull] >> 6);           // a skipped chunk: the same boxes at every step (no per-step counts)
            } else {
                const unsigned short* thits = p.track_hits + (size_t)e * p.ngp;
                for (int i = cfull << 6; i < hexec; i++) hitsTotal += (int)thits[i];
            }
        }
    }
    const double penalty = (double)hitsTotal * p.cpf;
    const double netTime = endTime - srcT;
    const double tc = fmax(netTime - 0, 0);                                 // :197 with ribbons left
    const double trueCost = tc * p.tpf + penalty;
    const double g = V->g + trueCost;
    unsigned flags = infeasible ? PPGPU_F_INFEASIBLE : 0u;
    if (endTime >= p.sst + p.horizon) flags |= PPGPU_F_GOAL;
    double h = 0;
    if (deferH) h = PP_H_DEFERRED;
    else if (p.fuse_h) {                                                    // MaxDistance (RibbonManager.cpp:234-248), as pp_h_max_distance
        double sumLength = 0, mn = PP_DBL_MAX, mx = 0;
        for (int i = 0; i < nrib; i++) {
            const double sx = rp[4 * i], sy = rp[4 * i + 1], ex = rp[4 * i + 2], ey = rp[4 * i + 3];
            sumLength += sqrt(pp_sq_len(sx, sy, ex, ey)) - 2 * p.ribw;
            const double dStart = pp_dist(sx, sy, endX, endY);
            const double dEnd = pp_dist(ex, ey, endX, endY);
            mn = fmin(fmin(mn, dEnd), dStart);
            mx = fmax(fmax(mx, dEnd), dStart);
        }
        h = fmax(sumLength + mn, mx) / p.max_speed * p.tpf;
    }
    double* r = stage;
    const unsigned info = (unsigned)(S->type & 0xff) | ((unsigned)(nrib & 0xff) << 8) | ((unsigned)(steps & 0xffff) << 16);
    r[0] = __hiloint2double((int)info, (int)flags);
    r[1] = trueCost; r[2] = penalty; r[3] = S->approx;
    r[4] = endX; r[5] = endY; r[6] = endHeading; r[7] = speed; r[8] = endTime;
    r[9] = g; r[10] = h; r[11] = (h == PP_H_DEFERRED) ? g : g + h;
    r[12] = V->coverage_completed_time; r[13] = S->p0; r[14] = S->p1; r[15] = S->p2;
    double* c = p.child + (size_t)eg * p.stride * 4;
    if (rpUniform) {
        const PP_AS4 double* ru = pp_const_f64(rp);
        for (int i = 0; i < 4 * nrib; i++) c[i] = ru[i];
    } else {
        for (int i = 0; i < 4 * nrib; i++) c[i] = rp[i];
    }
    return true;
}
#ifndef PP_APPROACH_MIN_WAVES
#define PP_APPROACH_MIN_WAVES 1
#endif
#ifndef PP_APPROACH_LDS
#define PP_APPROACH_LDS 0            // 1: stage the workgroup's setup records in LDS (measured slower: see below)
#endif
#ifndef PP_APPROACH_THREADS
#define PP_APPROACH_THREADS 256
#endif
#ifndef PP_LANE_NEAR_MAX
#define PP_LANE_NEAR_MAX 8
#endif
#ifndef PP_APPROACH_MAX_EVENTS
#define PP_APPROACH_MAX_EVENTS 0     // > 0: a lane hands its edge to the wave after this many approach events (bounds the kernel's tail)
#endif
__global__ __launch_bounds__(PP_APPROACH_THREADS, PP_APPROACH_MIN_WAVES) void pp_k_approach_events(PPParams p) {
    __shared__ double s_rec[PP_APPROACH_THREADS * PP_REC_STRIDE];      // quiet edges' records, transposed through LDS (34 KB: four workgroups per CU still fit)
    const long long e0 = (long long)blockIdx.x * PP_APPROACH_THREADS;
    const long long e = e0 + threadIdx.x;
    const bool valid = e < p.n_edges;
#if PP_APPROACH_LDS
    // Every lane reads ITS edge's setup record — 17 scalars up front, then eight fields of a segment per event; from memory each
    // of those loads touches 64 different lines.  Staging the workgroup's records (contiguous in the workspace) in LDS with
    // coalesced loads and reading them there was measured in round 3: 84 -> 109 us (128 threads per workgroup; 113 with 64, 134
    // with 256).  The kernel is bound by its longest per-lane chain of events, each a dependent load -> pose -> distance, and
    // hides that latency only with all its 14 waves per CU resident; 45 KB of LDS per 128 lanes leaves room for 6.
    __shared__ double s_setup[PP_APPROACH_THREADS * PP_SETUP_LDS_STRIDE];
    {
        const int ne = (int)((p.n_edges - e0 < (long long)PP_APPROACH_THREADS) ? (p.n_edges - e0) : (long long)PP_APPROACH_THREADS);
        const double2* src = reinterpret_cast<const double2*>(p.setup + p.ws_base + e0);
        for (int i = (int)threadIdx.x; i < ne * (PP_SETUP_GLOBAL_WORDS / 2); i += PP_APPROACH_THREADS) {
            const int ed = i / (PP_SETUP_GLOBAL_WORDS / 2), w = 2 * (i - ed * (PP_SETUP_GLOBAL_WORDS / 2));
            if (w < PP_SETUP_WORDS) {
                const double2 v = src[i];
                s_setup[ed * PP_SETUP_LDS_STRIDE + w] = v.x;
                if (w + 1 < PP_SETUP_WORDS) s_setup[ed * PP_SETUP_LDS_STRIDE + w + 1] = v.y;
            }
        }
    }
    __syncthreads();
    const PPEdgeSetupBody* S = reinterpret_cast<const PPEdgeSetupBody*>(&s_setup[(valid ? (int)threadIdx.x : 0) * PP_SETUP_LDS_STRIDE]);
#else
    const PPEdgeSetupBody* S = p.setup + p.ws_base + (valid ? e : 0);
#endif
    int2 out; out.x = 0; out.y = -1;
    const unsigned sflags = S->sflags;
    const int dubType = S->type;
    // do all lanes of this wave start from the same open vertex?
    const unsigned viMine = S->vi;
    const unsigned viFirst = (unsigned)__builtin_amdgcn_readfirstlane((int)viMine);
    const bool oneVertex = viFirst < (unsigned)p.nverts && __ballot(valid && viMine != viFirst) == 0ull;
    const double* rpU = p.ribbons + 4 * (size_t)pp_const_i32(&p.verts[oneVertex ? viFirst : 0].ribbon_offset)[0];
    const int nribU = pp_const_i32(&p.verts[oneVertex ? viFirst : 0].ribbon_count)[0];
    if (valid && !(sflags & (PP_SETUP_MALFORMED | PP_SETUP_COLOCATED)) && dubType >= 0) {
        const ppgpu_vertex* V = p.verts + S->vi;
        const int nrib = V->ribbon_count;
        const int limit = p.track_summary[p.ws_base + e].limit;
        if (nrib > 0 && nrib <= PP_WAVE && limit > 0) {
            const double* rp = p.ribbons + 4 * (size_t)V->ribbon_offset;
            const double* tg = p.tgrid + (size_t)S->vi * p.ng;
            const double wStart = S->wStart, speed = S->speed, length = S->length, rho = S->rho, rho_inv = S->rho_inv, qx = S->qx, qy = S->qy;
            const double hi0 = S->seg[0].hi, hi1 = S->seg[1].hi;
#ifndef PP_NO_TFAR
            const double tfar = S->tfar;
#endif
            const double endTime0 = fmin(p.horizon + 1e-12 + p.sst, S->wEnd);
            const double w = p.ribw, grow = w + 1e-3, minLength0 = 2 * w;
            bool tiny = false;
            if (oneVertex) {
                const PP_AS4 double* ru = pp_const_f64(rpU);
                for (int i = 0; i < nribU; i++) tiny |= pp_sq_len(ru[4 * i], ru[4 * i + 1], ru[4 * i + 2], ru[4 * i + 3]) < minLength0 * minLength0 / (2.0 * 2.0);
            } else
            for (int i = 0; i < nrib; i++) tiny |= pp_sq_len(rp[4 * i], rp[4 * i + 1], rp[4 * i + 2], rp[4 * i + 3]) < minLength0 * minLength0 / (2.0 * 2.0);
            int k = 0, lastEv = -1;
            bool handOver = tiny;               // the wave has events to visit (or an error to flag)
#ifndef PP_NO_LANE_NEAR
            // Round 3: an event within reach of ONE ribbon is no longer handed over at once.  The lane takes it exactly (does that
            // ribbon contain the point, strictly or not: the reference's own expressions) and goes on while it changes nothing — the
            // vehicle passes near a ribbon, or travels inside a corridor before it reaches the strict one, or may not cover while it
            // turns: what the wave used to start with (one event and one quiet run, two of a slow edge's nine operations).  At most
            // PP_LANE_NEAR_MAX such events per edge (a crawl along a corridor is the wave's, 64 steps at a time); within reach of
            // two ribbons at once the wave takes over as before.
            int nearBudget = PP_LANE_NEAR_MAX;
            const bool covEdge = (S->cbits & PPGPU_EDGE_COVERAGE) != 0;
#endif
#if PP_APPROACH_MAX_EVENTS > 0
            int budget = PP_APPROACH_MAX_EVENTS;
#endif
            // a piece short enough to be erased makes every event a real one (Ribbon::covered is checked wherever the vehicle is)
            while (!tiny) {
                if (k >= limit) break;
                const double t = tg[k];
                if (!(t < endTime0)) break;
#ifndef PP_NO_TFAR
                if ((t - wStart) * speed / rho > tfar) { k = 0x3fffffff; break; }       // the rest of the curve is clear: no event is visited
#endif
#if PP_APPROACH_MAX_EVENTS > 0
                if (--budget < 0) { handOver = true; break; }                           // a long chain: the wave goes on from event k
#endif
                double dist = (t - wStart) * speed;                                     // DubinsWrapper.cpp:36
                if (dist < 0 || dist > length) dist = dist - 1e-5;                      // EDUBPARAM retry, :39-42
                if (dist < 0 || dist > length) { handOver = true; break; }              // the wavefront's code flags the error
                const double tprime = (rho_inv != 0.0) ? dist * rho_inv : dist / rho;
                const PPSeg* g = &S->seg[pp_seg_of(tprime, hi0, hi1)];
                double ux, uy, uth;
                pp_curve_seg(g->type, (tprime - g->o1) - g->o2, g->bx, g->by, g->bth, g->sb, g->cb, ux, uy, uth);
                const double x = ux * rho + qx, y = uy * rho + qy;
                bool inBox = false;
                int boxCount = 0, boxIdx = 0;
                double q = PP_DBL_MAX;
#ifndef PP_APPROACH_NO_UNIFORM_RIBBONS
                if (oneVertex) {
                    // every lane of the wave starts from the same vertex (a dense launch from one open vertex): its ribbons come through
                    // scalar loads instead of twenty vector loads per event
                    const PP_AS4 double* ru = pp_const_f64(rpU);
                    for (int i = 0; i < nribU; i++) {
                        const double sx = ru[4 * i], sy = ru[4 * i + 1], ex = ru[4 * i + 2], ey = ru[4 * i + 3];
                        const bool in = (x >= fmin(sx, ex) - grow) & (x <= fmax(sx, ex) + grow) & (y >= fmin(sy, ey) - grow) & (y <= fmax(sy, ey) + grow);
                        inBox |= in; boxCount += in ? 1 : 0; boxIdx = in ? i : boxIdx;
                        const double qS = (sx - x) * (sx - x) + (sy - y) * (sy - y);
                        const double qE = (ex - x) * (ex - x) + (ey - y) * (ey - y);
                        q = fmin(q, fmin(qE, qS));
                    }
                } else
#endif
                for (int i = 0; i < nrib; i++) {
                    const double sx = rp[4 * i], sy = rp[4 * i + 1], ex = rp[4 * i + 2], ey = rp[4 * i + 3];
                    const bool in = (x >= fmin(sx, ex) - grow) & (x <= fmax(sx, ex) + grow) & (y >= fmin(sy, ey) - grow) & (y <= fmax(sy, ey) + grow);
                    inBox |= in; boxCount += in ? 1 : 0; boxIdx = in ? i : boxIdx;
                    const double qS = (sx - x) * (sx - x) + (sy - y) * (sy - y);
                    const double qE = (ex - x) * (ex - x) + (ey - y) * (ey - y);
                    q = fmin(q, fmin(qE, qS));
                }
#ifdef PP_DBG_TRACE
                if (pp_edge_position(p, p.e_base + e) == (long long)(PP_DBG_TRACE)) printf("[lane] event %d: inBox %d q %.17g x %.17g y %.17g\n", k, (int)inBox, q, x, y);
#endif
                double D = fmin(PP_DBL_MAX, sqrt(q));
                if (inBox) {
#ifndef PP_NO_LANE_NEAR
                    if (boxCount == 1 && nearBudget-- > 0) {
                        bool inside, strict;
                        pp_lane_ribbon_contains(rp[4 * boxIdx], rp[4 * boxIdx + 1], rp[4 * boxIdx + 2], rp[4 * boxIdx + 3], x, y, w, inside, strict);
                        // Edge.cpp:159: cover() runs when coverage is allowed on this edge or the heading did not change since the last step
                        const bool coverOn = covEdge || (((p.track_eq[(size_t)(p.ws_base + e) * p.nch + (k >> 6)] >> (k & 63)) & 1ull) != 0ull);
                        if (!(strict && coverOn)) {
                            if (inside) D = 0;                                          // RibbonManager::minDistanceFrom: contained
                            lastEv = k;
                            k = k + pp_event_stride(D, p.inc_d, p.inv_inc_d, p.ng) + 1;
                            continue;
                        }
                    }
#endif
                    handOver = true; break;                                             // a ribbon changes here (or two are in reach, or the budget is spent): the wavefront takes over
                }
                lastEv = k;
                k = k + pp_event_stride(D, p.inc_d, p.inv_inc_d, p.ng) + 1;
            }
            out.x = k; out.y = lastEv;
#ifndef PP_NO_QUIET_FINISH
            if (!handOver && k >= limit && p.quiet_finish &&
                pp_finish_quiet_edge(p, S, V, p.ws_base + e, pp_edge_position(p, p.e_base + e), limit, lastEv, oneVertex ? rpU : rp, tg, s_rec + (size_t)threadIdx.x * PP_REC_STRIDE, oneVertex))
                out.x = PP_FAR_DONE;
#endif
#ifdef PP_DBG_QUIET
            atomicAdd(p.need_big + 8 + (out.x == PP_FAR_DONE ? 0 : (!handOver && k >= limit) ? 1 : 2), 1u);
#endif
        }
    }
    if (valid) p.track_far[p.ws_base + e] = out;
#ifndef PP_NO_QUIET_FINISH
    {
        // the records of this wave's quiet edges, from LDS: lanes 16g .. 16g+15 store the 16 doubles of record 4 it + g
        const int lane = threadIdx.x & 63, wbase = (int)threadIdx.x - lane;
        const unsigned long long doneMask = __ballot(valid && out.x == PP_FAR_DONE);
        if (doneMask) {
            pp_wave_lds_fence();
            const long long myEg = valid ? pp_edge_position(p, p.e_base + e) : 0;
            const int g = lane >> 4, slot = lane & 15;
            for (int it = 0; it < 16; it++) {
                const int src = 4 * it + g;
                const long long egs = __shfl(myEg, src, PP_WAVE);
                if ((doneMask >> src) & 1ull) reinterpret_cast<double*>(p.out + egs)[slot] = s_rec[(size_t)(wbase + src) * PP_REC_STRIDE + slot];
            }
        }
    }
#endif
    // the edges the cover sweep's waves still have to visit, packed (one atomic per workgroup; the order of the launch — long
    // edges first — survives up to the order in which workgroups get here)
    if (p.live_list) {
        __shared__ unsigned s_cnt[PP_APPROACH_THREADS / 64];
        __shared__ unsigned s_base;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const bool live = valid && out.x != PP_FAR_DONE;
        const unsigned long long m = __ballot(live);
        if (lane == 0) s_cnt[wave] = (unsigned)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tot = 0;
            for (int w = 0; w < PP_APPROACH_THREADS / 64; w++) tot += s_cnt[w];
            s_base = tot ? atomicAdd(p.live_count, tot) : 0u;
        }
        __syncthreads();
        if (live) {
            unsigned at = s_base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; w++) at += s_cnt[w];
            p.live_list[2 * at] = (unsigned)e;                                             // slot in the workspace ...
            p.live_list[2 * at + 1] = (unsigned)pp_edge_position(p, p.e_base + e);         // ... and position in the caller's list (a 64-bit division the wave is spared)
        }
    }
}

// e = the edge's slot in the workspace, eg = its position in the caller's edge list, lds = 256 doubles private to the wave
#ifndef PP_LANE_HEUR
#define PP_LANE_HEUR 1   // untouched ribbon lists: heuristic by pp_k_heuristic_lanes
#endif
#ifndef PP_FUSE_HEUR
#define PP_FUSE_HEUR 1
#endif
template <int MAXN>
__device__ __forceinline__ double pp_h_point_from_pts(int heuristic, int tsp_k, double ribw, double* lds_wave, int nrib, unsigned passFirst = 0u, unsigned passStride = 1u);   // further down, with the heuristics

template <bool GAUSSIAN>
__device__ __forceinline__ void pp_cover_sweep_edge(const PPParams& p, const long long e, const long long eg, double* lds) {
    const int lane = pp_lane();

#ifndef PP_NO_APPROACH
    if (p.track_far && pp_const_i32(&p.track_far[e].x)[0] == PP_FAR_DONE) return;   // a quiet edge: pp_k_approach_events finished it
#endif
    // ---- phase 0 was done by pp_k_solve_edges: everything here is wave-uniform and arrives through scalar loads
    const PPEdgeSetup* S = p.setup + e;
    unsigned flags = 0;
    ppgpu_edge_result* rec = p.out + eg;
    const unsigned sflags = (unsigned)PP_SI32(sflags);
#if !defined(PP_NO_LANE_FINISH) && !defined(PP_DBG_COUNTS) && !defined(PP_DBG_EVENTS) && !defined(PP_ABL_ONLY_EVENTS)
    const bool laneFinish = !GAUSSIAN && p.cover_state != nullptr;
#else
    const bool laneFinish = false;
#endif
    if (laneFinish && lane == 0) p.cover_state[e].nrib = -1;          // until the hand-over below says otherwise: finished here
    if (sflags & PP_SETUP_MALFORMED) {
        // malformed descriptor: fail loudly in the record, touch nothing else
        if (lane == 0) { rec->flags = PPGPU_F_INFEASIBLE | PPGPU_F_THROWS | PPGPU_F_DUBINS_ERR; rec->info = 0; }
        return;
    }
    const unsigned vi = (unsigned)PP_SI32(vi), cbits = (unsigned)PP_SI32(cbits);
    const ppgpu_vertex* V = p.verts + vi;
    const double srcX = pp_sgpr(V->x), srcY = pp_sgpr(V->y), srcT = pp_sgpr(V->time), srcG = pp_sgpr(V->g);
    double cct = pp_sgpr(V->coverage_completed_time);
    int nrib = __builtin_amdgcn_readfirstlane(V->ribbon_count);
    const bool cov = (cbits & PPGPU_EDGE_COVERAGE) != 0;

    // this vertex's ribbons, one per lane (Vertex::connect copies the parent's RibbonManager, Vertex.cpp:24)
    PPRibbon rib = {0, 0, 0, 0};
    if (nrib > PP_WAVE) { nrib = PP_WAVE; flags |= PPGPU_F_RIBBON_OVF | PPGPU_F_RIBBON_LOST; }
    if (lane < nrib) {
        const double* rp = p.ribbons + 4 * ((size_t)V->ribbon_offset + lane);
        rib.sx = rp[0]; rib.sy = rp[1]; rib.ex = rp[2]; rib.ey = rp[3];
    }
    const bool startedDone = (nrib == 0);                             // Edge.cpp:93

    const int dubType = PP_SI32(type);
    const double wEnd = PP_SF64(wEnd), wStart = PP_SF64(wStart), speed = PP_SF64(speed);
    const double endTime0 = fmin(p.horizon + 1e-12 + p.sst, wEnd);    // Edge.cpp:90
    double endTime = endTime0;
    bool infeasible = (srcT >= endTime);                              // :102-110
    bool throwsRef = ((sflags & PP_SETUP_COLOCATED) != 0) || (dubType < 0);
    if (dubType < 0) flags |= PPGPU_F_DUBINS_ERR;

    // ---- the pose sweep's track of this edge
    const PPTrackSummary* sum = p.track_summary + e;
    const int limit = pp_const_i32(&sum->limit)[0];
    const int stopKind = pp_const_i32(&sum->blocked)[0];
    const bool blockedAtLimit = stopKind == 1;
    if (pp_const_i32(&sum->dub_err)[0]) flags |= PPGPU_F_DUBINS_ERR;
    const unsigned long long* teq = p.track_eq + (size_t)e * p.nch;
    const double* tg = p.tgrid + (size_t)vi * p.ng;

#ifdef PP_DBG_EVENTS
    int dbgEvents = 0;
#endif
#ifdef PP_DBG_COUNTS
    int dbgWindows = 0, dbgCorr = 0, dbgQuiet = 0, dbgGeneric = 0, dbgCorrLen = 0, dbgQuietLen = 0, dbgFar = 0, dbgNoChange = 0, dbgInPlace = 0, dbgRestFar = 0;
#define PP_CNT(x) x
#else
#define PP_CNT(x)
#endif
#ifdef PP_DBG_TRACE
#define PP_TRACE(...) do { if (eg == (long long)(PP_DBG_TRACE) && lane == 0) printf(__VA_ARGS__); } while (0)
#else
#define PP_TRACE(...)
#endif
    int rdt = -1;                       // `auto ribbonsDoneTime = -1;` is an int (Edge.cpp:92)
    int nextEvent = 0;                  // toCoverDistance starts at 0: step 0 is an event
    int lastEv = -1;
#ifndef PP_NO_APPROACH
    if (p.track_far) {                  // the approach was walked by pp_k_approach_events: start where it handed over
        nextEvent = pp_const_i32(&p.track_far[e].x)[0];
        lastEv = pp_const_i32(&p.track_far[e].y)[0];
    }
#endif
    const double w = p.ribw;
    const double inc_d = p.inc_d;
    const double runSpan = 64.0 * (p.inc_d / p.max_speed) * speed * (1.0 + 1e-9) + 1e-6;   // how far 64 steps can take the vehicle

    // long runs (pp_corridor_run / pp_quiet_run with ell > 0): one sample every longStride steps — as many steps as make
    // PP_LONG_REACH of travel (a slow edge of config 3: 5 steps of 1 cm; at full speed a step is 5 cm and nothing changes; measured 0.02 .. 0.4 m: the wider the margins, the more attempts fail) — with margins of that travel
#ifndef PP_LONG_REACH
#define PP_LONG_REACH 0.05
#endif
#ifndef PP_LONG_MAX_STRIDE
#define PP_LONG_MAX_STRIDE 16
#endif
#define PP_STEP_LEN() ((p.inc_d / p.max_speed) * speed * (1.0 + 1e-9) + 1e-9)      /* arc length of one step, from above */
#ifndef PP_NO_LONG_RUN
    int longStride = 1;
    {
        const double stepLen = PP_STEP_LEN();
        const double sd = PP_LONG_REACH / stepLen;
        longStride = sd >= (double)PP_LONG_MAX_STRIDE ? PP_LONG_MAX_STRIDE : (sd > 1.0 ? (int)sd : 1);
        // the half that vanishes between two samples must be shorter than the minimum length; the curve must not turn much between them
        if (!((double)longStride * stepLen + 1e-6 < 0.5 * w) || !(((double)longStride * stepLen + 1e-6) / PP_SF64(rho) < 0.5)) longStride = 1;
    }
#else
    const int longStride = 1;
#endif

    // ---- phase B: coverage events among steps [0, limit)
#ifdef PP_ABL_NO_EVENTS
    nextEvent = 1 << 30;
#endif
    if (!throwsRef) {
        bool ended = false;
        int cont = 0, contPiece = 0;        // 1 / 2: the last window ended inside a corridor / quiet run (of piece contPiece); + 4: it was
                                            // one run from end to end (a long run is worth trying)
        bool contMoveEnd = false;
#if !defined(PP_NO_APPROACH) && defined(PP_HANDOVER_QUIET)      // measured in round 3: 1056 -> 1063 us (the attempts that fail cost more than the events saved): off
        // the approach kernel handed over at an event inside some ribbon's grown bounding box: nearly always inside that ribbon's
        // corridor, where this event and the following ones change nothing until the strict corridor is reached — a quiet run is tried
        // from the window's first step (its guarded checks decide; if the first step does not pass, the one-at-a-time code takes it)
        if (p.track_far && nextEvent > 0) cont = 2;
#endif
        while (!ended) {
            if (nextEvent >= limit) break;
#ifndef PP_NO_TFAR
            if (nrib > 0) {
                // past the point from which the curve stays clear of every ribbon this vertex had (PPEdgeSetup::tfar), every
                // further event only measures a distance: none of them is visited, provided no piece is waiting to be erased
                const double tpn = (pp_const_f64(tg + nextEvent)[0] - wStart) * speed / pp_const_f64(&S->rho)[0];
                if (tpn > pp_const_f64(&S->tfar)[0]) {
                    const double minLength0 = 2 * w;
                    const bool tiny = (lane < nrib) & (pp_sq_len(rib.sx, rib.sy, rib.ex, rib.ey) < minLength0 * minLength0 / (2.0 * 2.0));
                    if (__ballot(tiny) == 0ull) { PP_TRACE("[wave] event %d: past tfar (tpn %.9g > %.9g): stop\n", nextEvent, tpn, pp_const_f64(&S->tfar)[0]); PP_CNT(dbgRestFar++); break; }
                }
            }
#endif
            // a window of 64 steps of the track starting AT the next event, one step per lane (stretches without events are
            // never loaded)
            const int base = nextEvent;
            // ... or, after a window that was one run from end to end, 64 SAMPLES longStride steps apart (a long run)
            const int stride = ((cont & 4) != 0 && longStride > 1 && base + 2 * longStride < limit) ? longStride : 1;
            PP_TRACE("[wave] window at %d stride %d (limit %d, lastEv %d, nrib %d)\n", base, stride, limit, lastEv, nrib);
            PP_CNT(dbgWindows++);
            const int k = base + lane * stride;
            const double t = (k < p.ng) ? tg[k] : INFINITY;
            // the poses of the window, recomputed with the pose sweep's own arithmetic (pp_window_pose)
            double2 q;
            {
                PP_WINDOW_POSE(S, t, pp_readlane(t, 0), k < limit, q.x, q.y);
                if (k >= limit) { q.x = 0.0; q.y = 0.0; }
            }
            // Edge.cpp:159: cover only when coverage is allowed on this edge or the heading did not change since the last step
            unsigned long long coverMask = ~0ull, coverAny = ~0ull;
            if (!cov) {
                if (stride == 1) {
                    const int c0 = base >> 6, sh = base & 63;
                    const unsigned long long lo = pp_const_u64(teq + c0)[0];
                    const unsigned long long hi = (sh != 0 && c0 + 1 < p.nch) ? pp_const_u64(teq + c0 + 1)[0] : 0ull;
                    coverMask = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
                    coverAny = coverMask;
                } else {
                    // a sample vouches for the steps (previous sample, itself]: cover() enabled at all of them / at any of them
                    const int lo = (lane == 0) ? base : (k - stride + 1);
                    const int cnt = (lane == 0) ? 1 : stride;
                    unsigned long long bits = 0ull;
                    if (k < limit) {
                        const int w0 = lo >> 6, sh = lo & 63;
                        bits = teq[w0] >> sh;
                        if (sh + cnt > 64 && w0 + 1 < p.nch) bits |= teq[w0 + 1] << (64 - sh);
                    }
                    const unsigned long long mask = (1ull << cnt) - 1ull;
                    bits &= mask;
                    coverMask = __ballot(bits == mask);
                    coverAny = __ballot(bits != 0ull);
                }
            }
            const int climit = (stride == 1) ? ((limit - base) < PP_WAVE ? (limit - base) : PP_WAVE) : __popcll(__ballot(k < limit));
            bool runFailed = false, quietFailed = false;
            bool tryQuiet = false;              // a corridor run has just ended inside this window
            while (true) {
                const int j = nextEvent - base;
                if (j >= climit) break;
                const double tj = pp_readlane(t, j);
                if (!(tj < endTime)) { ended = true; break; }         // `while (intermediate.time() < endTime)`
#if !defined(PP_NO_CORRIDOR_RUN) && !defined(PP_NO_QUIET_RUN) && defined(PP_RUN_THEN_QUIET)   // measured in round 3: 1057 -> 1123 us (one more inlined run in a kernel whose code already fills the instruction cache): off
                if (tryQuiet) {
                    // the step a corridor run stopped at: very often the vehicle has left the piece's strict corridor sideways and
                    // travels on inside its outer corridor — events that change nothing.  The quiet run's guarded checks decide from
                    // this very step on; if it does not pass them, the one-at-a-time code below takes it as before.
                    tryQuiet = false;
                    if (stride == 1 && nrib > 0 && j + 1 < climit && !quietFailed) {
                        const int L = pp_quiet_run(rib, nrib, w, q.x, q.y, (lane < climit) & (t < endTime), coverMask, j, runSpan);
                        PP_CNT(dbgQuiet++; dbgQuietLen += L);
                        PP_TRACE("[wave]   quiet run tried where the corridor run ended, from %d: L %d\n", base + j, L);
                        if (L > 0) {
                            lastEv = base + j + L - 1;
                            nextEvent = lastEv + 1;
                            if (j + L >= climit) cont = 2 | 4;
                            continue;
                        }
                    }
                }
#endif
#ifndef PP_NO_CORRIDOR_RUN
                if (j == 0 && (cont & 3) != 0) {
                    // the previous window ended inside a run: this step is an event of the same kind, very likely the whole
                    // window is.  The run's own guarded checks decide; if its first step does not pass, the step goes
                    // through the one-at-a-time code below like any other.
                    int L = 0;
                    double nsx = 0, nsy = 0;
                    const bool stepOk = (lane < climit) & (t < endTime);
                    const int kind = cont & 3;
                    PP_CNT(if (kind == 1) dbgCorr++; else dbgQuiet++);
                    // a long run: margins of the travel between two samples
                    const double ell = (stride > 1) ? ((double)stride * PP_STEP_LEN() + 1e-6) : 0.0;
                    const double span = (stride > 1) ? 64.0 * ell : runSpan;
                    if (kind == 1) L = pp_corridor_run(rib, nrib, w, contPiece, contMoveEnd, q.x, q.y, stepOk, coverMask, 0, span, nsx, nsy, ell, ell / PP_SF64(rho));
#ifndef PP_NO_QUIET_RUN
                    else L = pp_quiet_run(rib, nrib, w, q.x, q.y, stepOk, coverAny, 0, span, ell);
#endif
                    PP_TRACE("[wave]   continued run (kind %d, stride %d) from %d: L %d\n", kind, stride, base, L);
                    if (L > 0) {
                        if (kind == 1 && lane == contPiece) {
                            if (contMoveEnd) { rib.ex = nsx; rib.ey = nsy; } else { rib.sx = nsx; rib.sy = nsy; }
                        }
                        PP_CNT(if (kind == 1) dbgCorrLen += (L - 1) * stride + 1; else dbgQuietLen += (L - 1) * stride + 1);
                        lastEv = base + (L - 1) * stride;
                        nextEvent = lastEv + 1;
                    }
                    if (stride > 1) {
                        // a window of samples is only ever this one attempt: whatever it absorbed, ordinary windows (or, if every
                        // sample held, another long run) go on from there
                        cont = kind | ((L == PP_WAVE) ? 4 : 0);
                        break;
                    }
                    if (L > 0) {
                        cont = (L < climit) ? 0 : (kind | 4);  // ended inside the window: something else happens next / filled it: a long run next
                        tryQuiet = (kind == 1) && (L < climit);
                        continue;
                    }
                    cont = 0;
                }
#endif
                const double xj = pp_readlane(q.x, j), yj = pp_readlane(q.y, j);
                double D;                                             // Edge.cpp:158-161
                int adv;
                PP_CNT(dbgGeneric++);
                nrib = pp_ribbons_event(rib, nrib, w, xj, yj, ((coverMask >> j) & 1ull) != 0ull, lds, D, adv);
                PP_CNT(if (adv == -3) dbgFar++; else if (adv == -2) dbgNoChange++; else if (adv >= 0) dbgInPlace++);
                PP_TRACE("[wave]   event %d: adv %d D %.17g nrib %d cover %d x %.17g y %.17g\n", base + j, adv, D, nrib, (int)((coverMask >> j) & 1ull), xj, yj);
                if (nrib > PP_WAVE) { nrib = PP_WAVE; flags |= PPGPU_F_RIBBON_OVF | PPGPU_F_RIBBON_LOST; }

#ifndef PP_NO_CORRIDOR_RUN
                bool guessed = false;
#ifndef PP_NO_SPLIT_GUESS
                if (adv <= -4 && j + 1 < climit && !runFailed && nrib <= PP_WAVE) {
                    // this event split one piece in two (the vehicle has just entered its strict corridor): the next step will move the
                    // inner endpoint of the half the vehicle travels into — guess which from the direction of travel and try the run
                    // at once instead of learning it from one more one-at-a-time event (the run's own checks decide: a wrong guess
                    // gives L = 0 and costs one attempt)
                    const int front = -4 - adv;
                    const double dxp = pp_readlane(rib.ex, front + 1) - pp_readlane(rib.sx, front), dyp = pp_readlane(rib.ey, front + 1) - pp_readlane(rib.sy, front);
                    const bool towardsEnd = ((pp_readlane(q.x, j + 1) - xj) * dxp + (pp_readlane(q.y, j + 1) - yj) * dyp) > 0.0;
                    adv = towardsEnd ? (front + 1) : (front | 0x100);
                    guessed = true;
                }
#endif
                if (adv >= 0 && j + 1 < climit && !runFailed) {
                    // this event only moved one piece's endpoint: the following steps very likely do the same
                    double nsx, nsy;
                    const bool moveEnd = (adv & 0x100) != 0;
                    const int piece = adv & 0xff;
                    const int L = pp_corridor_run(rib, nrib, w, piece, moveEnd, q.x, q.y, (lane < climit) & (t < endTime), coverMask, j + 1, runSpan, nsx, nsy);
                    PP_CNT(dbgCorr++; dbgCorrLen += L);
                    runFailed = (L == 0) && !guessed;      // do not keep paying for attempts that do not start
                    PP_TRACE("[wave]   corridor run from %d: L %d\n", base + j + 1, L);
                    if (L > 0) {
                        if (lane == piece) {
                            if (moveEnd) { rib.ex = nsx; rib.ey = nsy; } else { rib.sx = nsx; rib.sy = nsy; }
                        }
                        lastEv = base + j + L;
                        nextEvent = base + j + L + 1;      // inside the corridor minDistanceFrom is 0: the next step is an event too
                        if (j + L + 1 >= climit) { cont = 1 | 4; contPiece = piece; contMoveEnd = moveEnd; }   // cut by the window, not by a guard
                        else tryQuiet = true;
                        continue;
                    }
                }
#ifndef PP_NO_QUIET_RUN
                else if (adv == -2 && D == 0 && nrib > 0 && j + 1 < climit && !quietFailed) {
                    // inside a corridor, nothing changed: the following steps are very likely the same kind of event
                    const int L = pp_quiet_run(rib, nrib, w, q.x, q.y, (lane < climit) & (t < endTime), coverMask, j + 1, runSpan);
                    PP_CNT(dbgQuiet++; dbgQuietLen += L);
                    PP_TRACE("[wave]   quiet run from %d: L %d\n", base + j + 1, L);
                    quietFailed = (L == 0);
                    if (L > 0) {
                        lastEv = base + j + L;
                        nextEvent = base + j + L + 1;
                        if (j + L + 1 >= climit) cont = 2 | 4;
                        continue;
                    }
                }
#endif
#endif
                if (nrib == 0) {                                      // :162-170
                    if (cct == -1) cct = tj;
                    rdt = (int)tj;
                    endTime = fmin(endTime, cct + p.tmin);
                }
                lastEv = base + j;
#ifdef PP_DBG_EVENTS
                dbgEvents++;
#endif
                // steps until toCoverDistance <= increment again (:153-154): m subtractions
                const int m = pp_event_stride(D, inc_d, p.inv_inc_d, p.ng);
                nextEvent = base + j + m + 1;
            }
        }
    }

#ifdef PP_ABL_ONLY_EVENTS
    if (lane == 0) p.out[eg].flags = (unsigned)(nrib + lastEv + rdt + (int)cct);     // (timing experiment: keep phase B's results alive, skip the rest)
    return;
#endif
    // ---- the rest is scalar work per edge: pp_k_cover_finish does it with one lane per edge, from what this wave knows now
    if (laneFinish && !throwsRef && (wStart <= endTime && wEnd >= endTime) && nrib <= PP_FINISH_MAX && nrib <= p.stride) {
        if (lane == 0) {
            PPCoverState* st = p.cover_state + e;
            st->cct = cct; st->endTime = endTime; st->lastEv = lastEv; st->rdt = rdt; st->flags = flags | (infeasible ? PPGPU_F_INFEASIBLE : 0u);
            st->nrib = nrib;                                          // (same lane, program order: after the -1 above)
        }
        if (lane < nrib) {
            double* c = p.child + ((size_t)eg * p.stride + lane) * 4;
            c[0] = rib.sx; c[1] = rib.sy; c[2] = rib.ex; c[3] = rib.ey;
        }
        return;
    }
    // ---- where the loop of Edge.cpp:143-175 stopped
    int steps = 0;
    double ix = srcX, iy = srcY;        // `intermediate` position
    double tfinal = (p.ng > 0) ? pp_const_f64(tg)[0] : INFINITY;
    bool coverFinal = true;             // `lastHeading == intermediate.heading()` unless the loop broke at a blocked step
    int hexec = 0;                      // steps whose obstacle hits count
    int lastIdx = -1;                   // the step whose pose `intermediate` holds when the loop stops (-1: the source pose)
    if (!throwsRef) {
        // cnt = steps k < limit with t_k < endTime (the time grid is non-decreasing)
        int cnt = limit;
        if (endTime != endTime0) {
            int lo = 0, hi = limit;
            while (hi > lo) {
                const int span = hi - lo, stride = (span + 63) >> 6;
                const int k = lo + lane * stride;
                const bool lt = (k < hi) && (tg[k] < endTime);
                const int c = __popcll(__ballot(lt));
                if (c == 0) { hi = lo; break; }
                const int nlo = lo + (c - 1) * stride + 1;
                const int nhi = lo + c * stride;
                hi = nhi < hi ? nhi : hi;
                lo = nlo;
            }
            cnt = lo;
        }
        int nexec = cnt > lastEv + 1 ? cnt : lastEv + 1;
        nexec = nexec < limit ? nexec : limit;
        // the blocked step is reached only if every step before it ran AND its own time still passes `while (t < endTime)`
        // (endTime may have shrunk at an event before it, Edge.cpp:169)
        if (blockedAtLimit && nexec == limit && pp_const_f64(tg + limit)[0] < endTime) {   // `break` at :146
            infeasible = true;
            lastIdx = limit;
            coverFinal = cov || (((pp_const_u64(teq + (limit >> 6))[0] >> (limit & 63)) & 1ull) != 0ull);
            tfinal = pp_const_f64(tg + limit)[0];
            steps = limit + 1;
            hexec = limit;
        } else {                                            // loop condition failed, or the first sample threw
            if (stopKind == 2 && pp_const_f64(tg)[0] < endTime) infeasible = true;   // `intermediate` still holds the source pose
            lastIdx = nexec - 1;
            tfinal = (nexec < p.ng) ? pp_const_f64(tg + nexec)[0] : INFINITY;
            steps = nexec;
            hexec = nexec;
        }
    }

    // ---- phase C
    // end()->state().time() = endTime; wrapper.sample(end state)  (Edge.cpp:177-178)
    if (!throwsRef && !(wStart <= endTime && wEnd >= endTime)) throwsRef = true;  // DubinsWrapper::containsTime
    double endX = 0, endY = 0, endHeading = 0;
    int hitsTotal = 0;
    if (!throwsRef) {
        // two samples of the curve in one pass: lane 1 takes the end state's time, every other lane the time of the step
        // `intermediate` stopped on (its position is needed for the last cover below)
        {
            const double tl = (lastIdx >= 0) ? pp_const_f64(tg + lastIdx)[0] : endTime;
            const PPEdgeSetup* S2 = S;
            asm volatile("" : "+s"(S2));
            const PPCurveHot hot = pp_curve_hot(S2);
            int cur = -1;
            PPSeg cs = PPSeg{0, 0, 0, 0, 0, INFINITY, -INFINITY, 0, 0, 1, 0};
            double px, py, puth;
            bool perr = false;
            pp_window_pose<PP_COVER_SINCOS_TAB>(S2, hot, cur, cs, lane == 1 ? endTime : tl, tl, true, px, py, puth, perr);
            if (lastIdx >= 0) { ix = pp_readlane(px, 0); iy = pp_readlane(py, 0); }
            endX = pp_readlane(px, 1);
            endY = pp_readlane(py, 1);
            endHeading = pp_heading_from_yaw(pp_mod2pi(pp_readlane(puth, 1)));
            if ((__ballot(perr) >> 1) & 1ull) flags |= PPGPU_F_DUBINS_ERR;
        }
        // cover the last little bit (:182-191)
        if (cov || coverFinal) {
            double Dunused;
            int advUnused;
            nrib = pp_ribbons_event(rib, nrib, w, ix, iy, true, lds, Dunused, advUnused);
            if (nrib > PP_WAVE) { nrib = PP_WAVE; flags |= PPGPU_F_RIBBON_OVF | PPGPU_F_RIBBON_LOST; }
        }
        if (nrib == 0) {
            if (cct == -1) cct = tfinal;
            rdt = (int)tfinal;
        }
        // obstacle hits of the executed steps (:150-151 summed): whole chunks from the pose sweep's per-chunk sums, the
        // last partial chunk step by step
        const unsigned* tch = p.track_chunk_hits + (size_t)e * p.nch;
        const unsigned short* thits = p.track_hits + (size_t)e * p.ngp;
        const int cfull = hexec >> 6;
        int acc = 0;
#ifdef PP_SUMMARY_HITS
        if (!GAUSSIAN && p.n_obst > 0 && hexec == limit) {
            hitsTotal = pp_const_i32(&sum->hits)[0];                        // every step below `limit` ran: the pose sweep's own total
        } else
#endif
        if (p.n_obst > 0) {
            for (int c = lane; c < cfull; c += PP_WAVE) acc += (int)tch[c];
            if ((hexec & 63) != 0 && tch[cfull] != 0u && (cfull << 6) + lane < hexec) {
                // a chunk the pose sweep skipped has no per-step counts: all of its 64 steps are inside the same boxes
                const bool skipped = p.track_skip && (p.track_skip[(size_t)e * p.nch + cfull] & PP_SKIP_ALL) != 0;
                acc += skipped ? (int)(tch[cfull] >> 6) : (int)thits[(cfull << 6) + lane];
            }
            hitsTotal = pp_wave_sum_i(acc);
        }
    }
    double penalty = (double)hitsTotal * p.cpf;                                   // :150-151 summed
    if (GAUSSIAN && !throwsRef && p.n_obst > 0) {
        // Gaussian model: the per-step values are doubles; whole chunks from the pose sweep's sums, the rest step by step
        const double* cpn = p.track_chunk_pen + (size_t)e * p.nch;
        const int cfull = hexec >> 6;
        double acc = 0;
        for (int c = lane; c < cfull; c += PP_WAVE) acc += cpn[c];
        if ((hexec & 63) != 0 && cpn[cfull] != 0.0 && (cfull << 6) + lane < hexec) acc += p.track_pen[(size_t)e * p.ngp + (cfull << 6) + lane] * p.cpf;
        penalty = pp_wave_sum_d(acc);
    }
    const double netTime = endTime - srcT;                                        // Edge::netTime
    double tc = fmax(netTime - ((nrib == 0) ? (endTime - (double)rdt) : 0), 0);  // :197
    if (startedDone) tc = 0;                                                      // :198
    const double trueCost = tc * p.tpf + penalty;                                 // :199
    const double g = srcG + trueCost;                                             // Vertex::setCurrentCost

    // h and f are filled in after the record is stored: by this wave from the ribbons it still holds (PP_FUSE_HEUR, below), or
    // by pp_k_heuristic* from the child ribbons
    const double h = 0;

    if (infeasible) flags |= PPGPU_F_INFEASIBLE;
    if (throwsRef) flags |= PPGPU_F_THROWS | PPGPU_F_INFEASIBLE;
    if (!throwsRef) {
        if (nrib == 0) flags |= PPGPU_F_DONE;
        // SamplingBasedPlanner::goalCondition (SamplingBasedPlanner.cpp:42-50)
        const double coverageDoneTime = cct + p.tmin;
        const double nonCoverageDoneTime = p.sst + p.horizon;
        if (endTime >= nonCoverageDoneTime || (nrib == 0 && endTime >= coverageDoneTime)) flags |= PPGPU_F_GOAL;
    }

    // ---- one 128-byte record, lanes 0..15 write one 8-byte slot each
    {
#ifdef PP_DBG_EVENTS
        steps = dbgEvents;
#endif
        const unsigned info = (unsigned)((dubType < 0 ? 0 : dubType) & 0xff) | ((unsigned)(nrib & 0xff) << 8) |
                              ((unsigned)(steps & 0xffff) << 16);
        double v;
        switch (lane) {
            case 0: v = __hiloint2double((int)info, (int)flags); break;   // {flags (low), info (high)}
            case 1: v = trueCost; break;
            case 2: v = penalty; break;
            case 3: v = PP_SF64(approx); break;
            case 4: v = endX; break;
            case 5: v = endY; break;
            case 6: v = endHeading; break;
            case 7: v = speed; break;
            case 8: v = endTime; break;
            case 9: v = g; break;
            case 10: v = h; break;
            case 11: v = g + h; break;
#ifdef PP_DBG_COUNTS
            case 12: v = (double)dbgRestFar * 1e9 + (double)dbgFar * 1e6 + (double)dbgNoChange * 1e3 + (double)dbgInPlace; break;
            case 13: v = (double)dbgWindows * 1e6 + (double)dbgGeneric; break;
            case 14: v = (double)dbgCorr * 1e6 + (double)dbgCorrLen; break;
            default: v = (double)dbgQuiet * 1e6 + (double)dbgQuietLen; break;
#else
            case 12: v = cct; break;
            case 13: v = PP_SF64(p0); break;
            case 14: v = PP_SF64(p1); break;
            default: v = PP_SF64(p2); break;
#endif
        }
        if (throwsRef && lane != 0) v = 0;
        if (lane < 16) reinterpret_cast<double*>(rec)[lane] = v;
    }
    if (!throwsRef) {
        if (nrib > PP_TSP_MAX && lane == 0) atomicOr(p.need_big, 1u);
        if (nrib > p.stride && lane == 0) rec->flags = flags | PPGPU_F_RIBBON_OVF;   // after the record store above
        if (lane < nrib && lane < p.stride) {
            double* c = p.child + ((size_t)eg * p.stride + lane) * 4;
            c[0] = rib.sx; c[1] = rib.sy; c[2] = rib.ex; c[3] = rib.ey;
        }
    }
#if PP_FUSE_HEUR
    // The edge's heuristic, by this wave, from the ribbons it still holds in registers (point heuristics; the record and the child
    // ribbons are stored, so nothing of the sweep is live any more): what pp_heuristic_edge<false, PP_TSP_MAX> would do.
    // large launches: the TSP enumeration of a short list is pp_k_heuristic_lanes' (a few lanes instead of this wave)
    const bool deferred = !GAUSSIAN && p.defer_h && !throwsRef && nrib <= p.stride && pp_lane_tsp_ok(p.heuristic, p.tsp_k, nrib);
    if (deferred && lane == 0) rec->h = PP_H_DEFERRED;
    if (!GAUSSIAN && p.fuse_h && !deferred && !throwsRef && nrib > 0 && nrib <= p.stride) {          // = pp_heuristic_edge<false, PP_TSP_MAX>
        const bool tsp = p.heuristic != PPGPU_H_MAX_DISTANCE;
        bool leaveToBigPass = false;
        const unsigned flags0 = flags;
        double hdist = 0;
        if (tsp && nrib > PP_TSP_MAX) {
            if (pp_tsp_big_ok(p.heuristic, p.tsp_k, nrib)) leaveToBigPass = true;    // pp_k_heuristic_big fills it in
            else flags |= PPGPU_F_RIBBON_OVF;
        } else if (!tsp && nrib > 31) {
            // MaxDistance over a long list (RibbonManager.cpp:234-248), ribbon by ribbon in list order
            double sumLength = 0, mn = PP_DBL_MAX, mx = 0;
            for (int i = 0; i < nrib; i++) {
                const double sx = pp_readlane(rib.sx, i), sy = pp_readlane(rib.sy, i), ex = pp_readlane(rib.ex, i), ey = pp_readlane(rib.ey, i);
                sumLength += sqrt(pp_sq_len(sx, sy, ex, ey)) - 2 * p.ribw;
                const double dStart = pp_dist(sx, sy, endX, endY);
                const double dEnd = pp_dist(ex, ey, endX, endY);
                mn = fmin(fmin(mn, dEnd), dStart);
                mx = fmax(fmax(mx, dEnd), dStart);
            }
            hdist = fmax(sumLength + mn, mx);
        } else {
            pp_wave_lds_fence();                                   // the event machinery is done with this scratch
            if (lane == 0) { lds[0] = endX; lds[1] = endY; }
            if (lane < nrib) {
                lds[2 * (1 + 2 * lane)] = rib.sx; lds[2 * (1 + 2 * lane) + 1] = rib.sy;
                lds[2 * (2 + 2 * lane)] = rib.ex; lds[2 * (2 + 2 * lane) + 1] = rib.ey;
            }
            pp_wave_lds_fence();
            hdist = pp_h_point_from_pts<PP_TSP_MAX>(p.heuristic, p.tsp_k, p.ribw, lds, nrib, 0u, 1u);
            pp_wave_lds_fence();
        }
        if (!leaveToBigPass) {
            const double hh = hdist / p.max_speed * p.tpf;
#ifndef PP_ABL_NO_HPATCH
            if (lane == 0) { rec->h = hh; rec->f = g + hh; if (flags != flags0) rec->flags = flags | ((nrib > p.stride) ? PPGPU_F_RIBBON_OVF : 0u); }
#else
            if (hh < 0) rec->h = hh;
#endif
        }
    }
#endif

}

#ifndef PP_POSE_MIN_WAVES
#define PP_POSE_MIN_WAVES 6
#endif
// n_edges = slice size (ppgpu.hip: launch_cost)
__global__ __launch_bounds__(PP_WPB * 64, PP_POSE_MIN_WAVES) void pp_k_pose_sweep(PPParams p) {
    PPQueue qs = pp_queue_init();
#ifdef PP_GRID_LDS
    __shared__ uint32_t s_tile[PP_WPB * PP_GRID_TILE_WORDS];
    uint32_t* tile = s_tile + (threadIdx.x >> 6) * PP_GRID_TILE_WORDS;
#else
    uint32_t* tile = nullptr;
#endif
    for (PP_EACH_EDGE(idx, 1, PP_Q_POSE, p.n_edges, 1))
        pp_pose_sweep_edge<false>(p, p.ws_base + idx, tile);
}
__global__ __launch_bounds__(PP_WPB * 64, 4) void pp_k_pose_sweep_gaussian(PPParams p) {
    PPQueue qs = pp_queue_init();
#ifdef PP_GRID_LDS
    __shared__ uint32_t s_tile[PP_WPB * PP_GRID_TILE_WORDS];
    uint32_t* tile = s_tile + (threadIdx.x >> 6) * PP_GRID_TILE_WORDS;
#else
    uint32_t* tile = nullptr;
#endif
    for (PP_EACH_EDGE(idx, 1, PP_Q_POSE, p.n_edges, 1))
        pp_pose_sweep_edge<true>(p, p.ws_base + idx, tile);
}
// The wave that finished an edge's cover sweep goes straight on to the edge's heuristic (point heuristics, binary-obstacle
// sweep; the Dubins heuristics and the 12-ribbon pass keep their own kernels): the child ribbons and the record it needs were
// just written by the same wave, there is no second launch, and the two phases' stalls fall at different times in the four
// waves of a SIMD.  Cover sweep + heuristic 2.26 -> 2.18 ms (tools/ablate.py fuse0).
#define PP_COVER_LDS ((PP_FUSE_HEUR) ? (PPTsp<PP_TSP_MAX>::LDS > PP_WAVE * 4 ? PPTsp<PP_TSP_MAX>::LDS : PP_WAVE * 4) : PP_WAVE * 4)
__global__ __launch_bounds__(PP_WPB * 64, PP_MIN_WAVES) void pp_k_cover_sweep(PPParams p) {
    __shared__ double lds_all[PP_WPB][PP_COVER_LDS];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    PPQueue qs = pp_queue_init();
    const long long n = p.live_list ? (long long)(unsigned)pp_const_i32(p.live_count)[0] : p.n_edges;
    for (PP_EACH_EDGE(i, 2, PP_Q_COVER, n, PP_Q_CHUNK_COVER)) {
        const long long idx = p.live_list ? (long long)(unsigned)pp_const_i32(p.live_list + 2 * i)[0] : i;
        const long long eg = p.live_list ? (long long)(unsigned)pp_const_i32(p.live_list + 2 * i)[1] : pp_edge_position(p, p.e_base + idx);
        pp_cover_sweep_edge<false>(p, p.ws_base + idx, eg, lds_all[wave]);
    }
}
__global__ __launch_bounds__(PP_WPB * 64, PP_MIN_WAVES) void pp_k_cover_sweep_gaussian(PPParams p) {
    __shared__ double lds_all[PP_WPB][PP_WAVE * 4];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    PPQueue qs = pp_queue_init();
    const long long n = p.live_list ? (long long)(unsigned)pp_const_i32(p.live_count)[0] : p.n_edges;
    for (PP_EACH_EDGE(i, 2, PP_Q_COVER, n, PP_Q_CHUNK_COVER)) {
        const long long idx = p.live_list ? (long long)(unsigned)pp_const_i32(p.live_list + 2 * i)[0] : i;
        const long long eg = p.live_list ? (long long)(unsigned)pp_const_i32(p.live_list + 2 * i)[1] : pp_edge_position(p, p.e_base + idx);
        pp_cover_sweep_edge<true>(p, p.ws_base + idx, eg, lds_all[wave]);
    }
}

// ------------------------------------------------------------------------------------------
// Vertex::computeApproxToGo (Vertex.cpp:49-64) for every costed edge: h = heuristic(child pose,
// child ribbons) / maxSpeed, f = g + h, patched into the edge's record.  Its own kernel so that the
// sweep kernel's register budget is not set by the TSP enumeration.  One wavefront per edge.
#ifndef PP_H_MIN_WAVES
#define PP_H_MIN_WAVES 6   // measured: 1 (87 VGPRs, 5 waves) 1.12 ms, 6 (72 VGPRs) 1.06 ms, 8 (63 VGPRs) 1.09 ms
#endif
// DUBINS = the two Dubins-TSP heuristics (RibbonManager.cpp:97-140): the same enumeration over a table of Dubins
// distances between oriented ribbon endpoints.  A separate instantiation so that the six-word solve does not set the
// register budget of the common kernel.  MAXN = 8: every edge; MAXN = 12: a second pass that only touches the edges whose
// 9..12 child ribbons the first pass left for it (pp_tsp_big_ok).
// MaxDistance / TspPointRobotNoSplit{All,K}Ribbons from the points staged in the wave's LDS (x,y of the query point, then
// start / end of every ribbon): distance table, nearest-endpoint table, enumeration.  nrib <= MAXN for the TSP variants.
template <int MAXN>
__device__ __forceinline__ double pp_h_point_from_pts(int heuristic, int tsp_k, double ribw, double* lds_wave, int nrib, unsigned passFirst, unsigned passStride) {
    typedef PPTsp<MAXN> TS;
    const int lane = pp_lane();
    double* pts = lds_wave;
    double* T = lds_wave + PP_WAVE * 2;
    double* KM = T + TS::PTS * (TS::PTS - 1);
    if (heuristic == PPGPU_H_MAX_DISTANCE) return pp_h_max_distance(pts, nrib, ribw);
    const int npts = 2 * nrib + 1;
    const int ncol = npts - 1;
    for (int idx = lane; idx < npts * ncol; idx += PP_WAVE) {      // all distances, once
        const int pp = (int)pp_udiv_small((unsigned)idx, (unsigned)ncol), qq = 1 + (idx - pp * ncol);
        T[pp * (TS::PTS - 1) + (qq - 1)] = pp_dist(pts[2 * pp], pts[2 * pp + 1], pts[2 * qq], pts[2 * qq + 1]);
    }
    pp_wave_lds_fence();
    for (int idx = lane; idx < npts * nrib; idx += PP_WAVE) {
        const int pp = (int)pp_udiv_small((unsigned)idx, (unsigned)nrib), ri = idx - pp * nrib;
        KM[pp * MAXN + ri] = fmin(pp_h_T<MAXN>(T, pp, 1 + 2 * ri), pp_h_T<MAXN>(T, pp, 2 + 2 * ri));
    }
    pp_wave_lds_fence();
    if (heuristic == PPGPU_H_TSP_POINT_ALL) return pp_h_tsp_point<MAXN>(T, KM, nrib, ribw, MAXN, false, nullptr, passFirst, passStride);
    return pp_h_tsp_point<MAXN>(T, KM, nrib, ribw, tsp_k, true, nullptr, passFirst, passStride);
}


template <bool DUBINS, int MAXN>
__device__ __forceinline__ void pp_heuristic_edge(const PPParams& p, const long long e, double* lds_wave) {
    typedef PPTsp<MAXN> TS;
    const bool bigPass = MAXN > PP_TSP_MAX;
    const int lane = pp_lane();
    double* pts = lds_wave;                      // x,y of the query point, then start/end of every child ribbon
    double* T = lds_wave + PP_WAVE * 2;          // distance table of the TSP heuristics
    double* KM = T + TS::PTS * (TS::PTS - 1);    // KM[p][i] = distance from point p to the nearer endpoint of ribbon i
    ppgpu_edge_result* rec = p.out + e;
    unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)rec->flags);
    if (flags & PPGPU_F_THROWS) return;
    int nrib = (int)((__builtin_amdgcn_readfirstlane((int)rec->info) >> 8) & 0xff);
    const bool tsp = p.heuristic != PPGPU_H_MAX_DISTANCE;
    if (bigPass && !(tsp && nrib <= p.stride && pp_tsp_big_ok(p.heuristic, p.tsp_k, nrib))) return;
    const double endX = rec->end_x, endY = rec->end_y, g = rec->g;
    double hdist = 0;
    if (nrib > 0 && nrib <= p.stride) {          // a truncated list (already flagged) carries no heuristic
        if (tsp && nrib > MAXN) {
            if (pp_tsp_big_ok(p.heuristic, p.tsp_k, nrib)) return;            // the MAXN = 12 pass fills it in
            flags |= PPGPU_F_RIBBON_OVF;
        } else if (!tsp && nrib > 31) {
            // MaxDistance over a long list: straight from global memory, no table
            double sumLength = 0, mn = PP_DBL_MAX, mx = 0;
            for (int i = 0; i < nrib; i++) {
                const double* c = p.child + ((size_t)e * p.stride + i) * 4;
                sumLength += sqrt(pp_sq_len(c[0], c[1], c[2], c[3])) - 2 * p.ribw;
                double dStart = pp_dist(c[0], c[1], endX, endY);
                double dEnd = pp_dist(c[2], c[3], endX, endY);
                mn = fmin(fmin(mn, dEnd), dStart);
                mx = fmax(fmax(mx, dEnd), dStart);
            }
            hdist = fmax(sumLength + mn, mx);
        } else {
            if (lane == 0) { pts[0] = endX; pts[1] = endY; }
            if (lane < nrib) {
                const double* c = p.child + ((size_t)e * p.stride + lane) * 4;
                pts[2 * (1 + 2 * lane)] = c[0]; pts[2 * (1 + 2 * lane) + 1] = c[1];
                pts[2 * (2 + 2 * lane)] = c[2]; pts[2 * (2 + 2 * lane) + 1] = c[3];
            }
            pp_wave_lds_fence();
            const int npts = 2 * nrib + 1;
            const int ncol = npts - 1;
            if (!tsp || !DUBINS) {
                hdist = pp_h_point_from_pts<MAXN>(p.heuristic, p.tsp_k, p.ribw, lds_wave, nrib);
            } else {
                // Oriented endpoints (Ribbon::startAsState / endAsState, Ribbon.cpp:60-70: at one end, heading towards the
                // other); the query pose passes the child's HEADING where the callee says yaw (Vertex.cpp:51) — kept.
                double* YAW = KM;                        // yaw of point q (q >= 1), then the ribbon lengths
                double* LEN = KM + TS::PTS;
                if (lane < nrib) {
                    const double sx = pts[2 * (1 + 2 * lane)], sy = pts[2 * (1 + 2 * lane) + 1];
                    const double ex = pts[2 * (2 + 2 * lane)], ey = pts[2 * (2 + 2 * lane) + 1];
                    YAW[1 + 2 * lane] = pp_yaw(pp_heading_to(sx, sy, ex, ey));
                    YAW[2 + 2 * lane] = pp_yaw(pp_heading_to(ex, ey, sx, sy));
                    LEN[lane] = sqrt(pp_sq_len(sx, sy, ex, ey));                       // Ribbon::length()
                }
                if (lane == 0) YAW[0] = rec->end_heading;
                pp_wave_lds_fence();
                for (int idx = lane; idx < npts * ncol; idx += PP_WAVE) {      // RibbonManager::dubinsDistance for every ordered pair
                    const int pp = (int)pp_udiv_small((unsigned)idx, (unsigned)ncol), qq = 1 + (idx - pp * ncol);
                    PPDubins d;
                    pp_dubins_shortest(pts[2 * pp], pts[2 * pp + 1], YAW[pp], pts[2 * qq], pts[2 * qq + 1], YAW[qq], p.h_rho, d);
                    T[pp * (TS::PTS - 1) + (qq - 1)] = pp_dubins_length(d, p.h_rho);
                }
                pp_wave_lds_fence();
                // K variant: its comparator compares r1 with r1 (:121-122), so the sort changes nothing, and its counter is
                // never incremented (:128), so every ribbon is branched: the All enumeration, unless K <= 0 (nothing runs)
                const int K = (p.heuristic == PPGPU_H_TSP_DUBINS_K && p.tsp_k <= 0) ? 0 : MAXN;
                hdist = pp_h_tsp_point<MAXN>(T, KM, nrib, p.ribw, K, false, LEN);
            }
        }
    }
    const double h = hdist / p.max_speed * p.tpf;
    if (lane == 0) { rec->h = h; rec->f = g + h; rec->flags = flags; }
}
#ifndef PP_H_WPB
#define PP_H_WPB PP_WPB   // wavefronts per workgroup of the heuristic kernels
#endif
__global__ __launch_bounds__(PP_H_WPB * 64, PP_H_MIN_WAVES) void pp_k_heuristic(PPParams p) {
    __shared__ double lds_all[PP_H_WPB][PPTsp<PP_TSP_MAX>::LDS];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    PPQueue qs = pp_queue_init();
    for (PP_EACH_EDGE(e, 4, PP_Q_HEUR, p.n_edges, PP_Q_CHUNK_HEUR))
        pp_heuristic_edge<false, PP_TSP_MAX>(p, e, lds_all[wave]);
}
__global__ __launch_bounds__(PP_H_WPB * 64) void pp_k_heuristic_dubins(PPParams p) {
    __shared__ double lds_all[PP_H_WPB][PPTsp<PP_TSP_MAX>::LDS];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    PPQueue qs = pp_queue_init();
    for (PP_EACH_EDGE(e, 4, PP_Q_HEUR, p.n_edges, PP_Q_CHUNK_HEUR))
        pp_heuristic_edge<true, PP_TSP_MAX>(p, e, lds_all[wave]);
}
// TspPointRobotNoSplitKRibbons on child lists of 9..12 ribbons (rare: a vertex whose ribbons were split many times)
#define PP_BIG_GRID 1024
__global__ __launch_bounds__(PP_H_WPB * 64) void pp_k_heuristic_big(PPParams p) {
    __shared__ double lds_all[PP_H_WPB][PPTsp<PP_TSP_MAX_BIG>::LDS];
    if (pp_const_i32(p.need_big)[0] == 0) return;            // almost always: no child list beyond 8 ribbons in this launch
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // a modest grid whose waves stride over the edges (ppgpu.hip: at most PP_BIG_GRID workgroups): the launch that finds nothing to do —
    // nearly every one — used to start a workgroup per four edges to learn it, 14 us at config 3
    for (long long e = (long long)blockIdx.x * PP_H_WPB + wave; e < p.n_edges; e += (long long)gridDim.x * PP_H_WPB)
        pp_heuristic_edge<false, PP_TSP_MAX_BIG>(p, e, lds_all[wave]);
}

// ------------------------------------------------------------------------------------------
// Phase C of the edges the cover sweep's waves handed over (PPCoverState), one LANE per edge: Edge.cpp:177-205 with the expressions
// of pp_cover_sweep_edge's own phase C (which stays, for the edges a wave keeps: a list longer than PP_FINISH_MAX, the Gaussian
// model, a curve the reference throws on) — pp_lane_pose for pp_window_pose, the reference's strict cover() ribbon by ribbon in list
// order for pp_ribbons_event (Ribbon::split / covered, Ribbon.cpp:9-25,39-58: the same projection, containsProjection and distance
// expressions; the wave decides the distance test on squares and falls back to this very quotient when it is close).
// Heuristic: a list the lane kernel enumerates is marked PP_H_DEFERRED as the wave would; MaxDistance is computed here; the rare
// list of 7 or 8 ribbons under a TSP heuristic goes to pp_k_heuristic_listed (a wave per such edge).
#ifndef PP_FINISH_THREADS
#define PP_FINISH_THREADS 64
#endif
__global__ __launch_bounds__(PP_FINISH_THREADS) void pp_k_cover_finish(PPParams p) {
    const unsigned nlive = (unsigned)pp_const_i32(p.live_count)[0];
    const unsigned li = blockIdx.x * PP_FINISH_THREADS + threadIdx.x;
    if (li >= nlive) return;
    const long long e = p.ws_base + (long long)p.live_list[2 * (size_t)li];
    const long long eg = (long long)p.live_list[2 * (size_t)li + 1];
    const PPCoverState st = p.cover_state[e];
    if (st.nrib < 0) return;                                   // its wave finished it
    const PPEdgeSetupBody* S = p.setup + e;
    const unsigned vi = S->vi;
    const ppgpu_vertex* V = p.verts + vi;
    const bool cov = (S->cbits & PPGPU_EDGE_COVERAGE) != 0;
    const double wStart = S->wStart, wEnd = S->wEnd, speed = S->speed, length = S->length, rho = S->rho, rho_inv = S->rho_inv, qx = S->qx, qy = S->qy;
    const double hi0 = S->seg[0].hi, hi1 = S->seg[1].hi;
    const double srcT = V->time;
    const double endTime0 = fmin(p.horizon + 1e-12 + p.sst, wEnd);    // Edge.cpp:90
    const double endTime = st.endTime;
    double cct = st.cct;
    int nrib = st.nrib, rdt = st.rdt;
    const int lastEv = st.lastEv;
    unsigned flags = st.flags;
    bool infeasible = (flags & PPGPU_F_INFEASIBLE) != 0;
    const bool startedDone = V->ribbon_count == 0;             // Edge.cpp:93
    const PPTrackSummary* sum = p.track_summary + e;
    const int limit = sum->limit, stopKind = sum->blocked;
    const double* tg = p.tgrid + (size_t)vi * p.ng;
    const double w = p.ribw;

    // ---- where the loop of Edge.cpp:143-175 stopped
    int cnt = limit;                                           // steps k < limit with t_k < endTime (the time grid is non-decreasing)
    if (endTime != endTime0) {
        int lo = 0, hi = limit;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tg[mid] < endTime) lo = mid + 1; else hi = mid;
        }
        cnt = lo;
    }
    int nexec = cnt > lastEv + 1 ? cnt : lastEv + 1;
    nexec = nexec < limit ? nexec : limit;
    int steps, hexec, lastIdx;
    double tfinal;
    bool coverFinal = true;
    if (stopKind == 1 && nexec == limit && tg[limit] < endTime) {   // `break` at :146
        infeasible = true;
        lastIdx = limit;
        coverFinal = cov || (((p.track_eq[(size_t)e * p.nch + (limit >> 6)] >> (limit & 63)) & 1ull) != 0ull);
        tfinal = tg[limit];
        steps = limit + 1;
        hexec = limit;
    } else {
        if (stopKind == 2 && p.ng > 0 && tg[0] < endTime) infeasible = true;
        lastIdx = nexec - 1;
        tfinal = (nexec < p.ng) ? tg[nexec] : INFINITY;
        steps = nexec;
        hexec = nexec;
    }
    // ---- end state (:177-178) and the pose `intermediate` stopped on
    double ix = V->x, iy = V->y, uth;
    bool ignored = false, perr = false;
    if (lastIdx >= 0) pp_lane_pose(S, tg[lastIdx], wStart, speed, length, rho, rho_inv, qx, qy, hi0, hi1, ix, iy, uth, ignored);
    double endX, endY;
    pp_lane_pose(S, endTime, wStart, speed, length, rho, rho_inv, qx, qy, hi0, hi1, endX, endY, uth, perr);
    if (perr) flags |= PPGPU_F_DUBINS_ERR;
    const double endHeading = pp_heading_from_yaw(pp_mod2pi(uth));
    // ---- cover the last little bit (:182-191): RibbonManager::cover(x, y, strict) over the list in order
    double* c = p.child + (size_t)eg * p.stride * 4;
    if ((cov || coverFinal) && nrib > 0) {
        double rsx[PP_FINISH_MAX], rsy[PP_FINISH_MAX], rex[PP_FINISH_MAX], rey[PP_FINISH_MAX];
#pragma unroll
        for (int i = 0; i < PP_FINISH_MAX; i++) {
            const bool have = i < nrib;
            rsx[i] = have ? c[4 * i] : 0.0; rsy[i] = have ? c[4 * i + 1] : 0.0; rex[i] = have ? c[4 * i + 2] : 0.0; rey[i] = have ? c[4 * i + 3] : 0.0;
        }
        const double minLength = 2 * w;                                  // Ribbon::minLength (Ribbon.cpp:52-58)
        const double thr = minLength * minLength / (2.0 * 2.0);          // covered(strict): c_StrictModifier^2
        const double T = PP_RIBBON_TOL;
        int nout = 0;
#pragma unroll
        for (int i = 0; i < PP_FINISH_MAX; i++) {
            if (i < nrib) {
                const double sx = rsx[i], sy = rsy[i], ex = rex[i], ey = rey[i];
                const double dxr = ex - sx, dyr = ey - sy;
                const double sqL = dxr * dxr + dyr * dyr;
                const double dot = (ix - sx) * dxr + (iy - sy) * dyr;
                const double px = dxr * dot / sqL + sx;                  // Ribbon::getProjection (Ribbon.cpp:72-78)
                const double py = dyr * dot / sqL + sy;
                const double a1 = px - sx, a2 = px - ex, b1 = py - sy, b2 = py - ey;
                const bool outx = ((a1 < -T) & (a2 < -T)) | ((a1 > T) & (a2 > T));
                const bool outy = ((b1 < -T) & (b2 < -T)) | ((b1 > T) & (b2 > T));
                const bool cp = !(outx | outy);                          // Ribbon::containsProjection (:90-95)
                const double num = dyr * ix - dxr * iy + ex * sy - ey * sx;
                const bool stc = cp && ((fabs(num) / sqrt(sqL)) < (w / 2.0));   // Ribbon::contains(strict): distance (Ribbon.h:118-121) < w / 2
                const bool keepF = stc && !(pp_sq_len(sx, sy, px, py) < thr);
                const bool keepR = stc ? !(pp_sq_len(px, py, ex, ey) < thr) : !(sqL < thr);
                if (keepF) {
                    if (nout < p.stride) { c[4 * nout] = sx; c[4 * nout + 1] = sy; c[4 * nout + 2] = px; c[4 * nout + 3] = py; }
                    nout++;
                }
                if (keepR) {
                    if (nout < p.stride) { c[4 * nout] = stc ? px : sx; c[4 * nout + 1] = stc ? py : sy; c[4 * nout + 2] = ex; c[4 * nout + 3] = ey; }
                    nout++;
                }
            }
        }
        // (the slots the handed-over list filled beyond the final one go back to zero: a wave that finishes its own edge never
        // writes them, and callers hand in zeroed buffers)
        for (int i = nout; i < nrib; i++) { c[4 * i] = 0.0; c[4 * i + 1] = 0.0; c[4 * i + 2] = 0.0; c[4 * i + 3] = 0.0; }
        nrib = nout;
    }
    if (nrib == 0) {
        if (cct == -1) cct = tfinal;
        rdt = (int)tfinal;
    }
    // ---- obstacle hits of the executed steps (:150-151 summed)
    int hitsTotal = 0;
    if (p.n_obst > 0) {
        const unsigned* tch = p.track_chunk_hits + (size_t)e * p.nch;
        const int cfull = hexec >> 6;
        for (int ch = 0; ch < cfull; ch++) hitsTotal += (int)tch[ch];
        if ((hexec & 63) != 0 && tch[cfull] != 0u) {
            if (p.track_skip && (p.track_skip[(size_t)e * p.nch + cfull] & PP_SKIP_ALL) != 0) {
                hitsTotal += (hexec & 63) * (int)(tch[cfull] >> 6);       // a skipped chunk: the same boxes at every step (no per-step counts)
            } else {
                const unsigned short* thits = p.track_hits + (size_t)e * p.ngp;
                for (int i = cfull << 6; i < hexec; i++) hitsTotal += (int)thits[i];
            }
        }
    }
    const double penalty = (double)hitsTotal * p.cpf;
    const double netTime = endTime - srcT;                                        // Edge::netTime
    double tc = fmax(netTime - ((nrib == 0) ? (endTime - (double)rdt) : 0), 0);  // :197
    if (startedDone) tc = 0;                                                      // :198
    const double trueCost = tc * p.tpf + penalty;                                 // :199
    const double g = V->g + trueCost;                                             // Vertex::setCurrentCost
    if (infeasible) flags |= PPGPU_F_INFEASIBLE;
    if (nrib == 0) flags |= PPGPU_F_DONE;
    {   // SamplingBasedPlanner::goalCondition (SamplingBasedPlanner.cpp:42-50)
        const double coverageDoneTime = cct + p.tmin;
        const double nonCoverageDoneTime = p.sst + p.horizon;
        if (endTime >= nonCoverageDoneTime || (nrib == 0 && endTime >= coverageDoneTime)) flags |= PPGPU_F_GOAL;
    }
    if (nrib > PP_TSP_MAX) atomicOr(p.need_big, 1u);
    if (nrib > p.stride) flags |= PPGPU_F_RIBBON_OVF;
    // ---- h: Vertex::computeApproxToGo, as the wave decides it
    double h = 0;
    bool listed = false;
    if (p.defer_h && nrib <= p.stride && pp_lane_tsp_ok(p.heuristic, p.tsp_k, nrib)) {
        h = PP_H_DEFERRED;
    } else if (p.fuse_h && nrib > 0 && nrib <= p.stride) {
        const bool tsp = p.heuristic != PPGPU_H_MAX_DISTANCE;
        if (tsp && nrib > PP_TSP_MAX) {
            if (!pp_tsp_big_ok(p.heuristic, p.tsp_k, nrib)) flags |= PPGPU_F_RIBBON_OVF;      // else pp_k_heuristic_big fills it in
        } else if (!tsp) {                                      // MaxDistance (RibbonManager.cpp:234-248), ribbon by ribbon in list order
            double sumLength = 0, mn = PP_DBL_MAX, mx = 0;
            for (int i = 0; i < nrib; i++) {
                const double sx = c[4 * i], sy = c[4 * i + 1], ex = c[4 * i + 2], ey = c[4 * i + 3];
                sumLength += sqrt(pp_sq_len(sx, sy, ex, ey)) - 2 * p.ribw;
                const double dStart = pp_dist(sx, sy, endX, endY);
                const double dEnd = pp_dist(ex, ey, endX, endY);
                mn = fmin(fmin(mn, dEnd), dStart);
                mx = fmax(fmax(mx, dEnd), dStart);
            }
            h = fmax(sumLength + mn, mx) / p.max_speed * p.tpf;
        } else {
            listed = true;                                      // a TSP enumeration the lanes do not take: a wave's work
        }
    }
    ppgpu_edge_result* rec = p.out + eg;
    double* r = reinterpret_cast<double*>(rec);
    const unsigned info = (unsigned)(S->type & 0xff) | ((unsigned)(nrib & 0xff) << 8) | ((unsigned)(steps & 0xffff) << 16);
    r[0] = __hiloint2double((int)info, (int)flags);
    r[1] = trueCost; r[2] = penalty; r[3] = S->approx;
    r[4] = endX; r[5] = endY; r[6] = endHeading; r[7] = speed; r[8] = endTime;
    r[9] = g; r[10] = h; r[11] = (h == PP_H_DEFERRED) ? g : g + h;
    r[12] = cct; r[13] = S->p0; r[14] = S->p1; r[15] = S->p2;
    if (listed) p.hw_list[atomicAdd(p.hw_count, 1u)] = (unsigned)eg;
}
// The edges pp_k_cover_finish listed (a TSP enumeration of 7 or 8 child ribbons: up to 32 768 leaves): Vertex::computeApproxToGo from
// the record and the child ribbons, as pp_heuristic_edge does it, by a whole WORKGROUP per edge — its four waves build the
// same tables, take every fourth pass of 64 prefixes each, and the smallest of their four minima is the minimum (exact).  One wave per
// edge was ~170 us of a single wave's time for each of config 3's ~1 100 such edges.  A grid as large as pp_k_heuristic keeps
// resident, striding over the list; usually the list is short or empty.
__global__ __launch_bounds__(PP_H_WPB * 64, PP_H_MIN_WAVES) void pp_k_heuristic_listed(PPParams p) {
    __shared__ double lds_all[PP_H_WPB][PPTsp<PP_TSP_MAX>::LDS];
    __shared__ double s_part[PP_H_WPB];
    const unsigned n = (unsigned)pp_const_i32(p.hw_count)[0];
    if (n == 0 || blockIdx.x >= n) return;
#ifndef PP_LISTED_NO_PRIO
    __builtin_amdgcn_s_setprio(3);           // few, long waves sharing their SIMDs with pp_k_heuristic_lanes' many short ones: issue first
#endif
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = pp_lane();
    double* pts = lds_all[wave];
    for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
        const long long e = (long long)(unsigned)pp_const_i32(p.hw_list + i)[0];
        ppgpu_edge_result* rec = p.out + e;
        const int nrib = (int)((__builtin_amdgcn_readfirstlane((int)rec->info) >> 8) & 0xff);      // 7 or 8 (<= PP_TSP_MAX, <= p.stride)
        const double endX = rec->end_x, endY = rec->end_y, g = rec->g;
        if (lane == 0) { pts[0] = endX; pts[1] = endY; }
        if (lane < nrib) {
            const double* c = p.child + ((size_t)e * p.stride + lane) * 4;
            pts[2 * (1 + 2 * lane)] = c[0]; pts[2 * (1 + 2 * lane) + 1] = c[1];
            pts[2 * (2 + 2 * lane)] = c[2]; pts[2 * (2 + 2 * lane) + 1] = c[3];
        }
        pp_wave_lds_fence();
        const double part = pp_h_point_from_pts<PP_TSP_MAX>(p.heuristic, p.tsp_k, p.ribw, pts, nrib, (unsigned)wave, (unsigned)PP_H_WPB);
        if (lane == 0) s_part[wave] = part;
        __syncthreads();
        if (threadIdx.x == 0) {
            double hdist = s_part[0];
            for (int w = 1; w < PP_H_WPB; w++) hdist = fmin(hdist, s_part[w]);
            const double h = hdist / p.max_speed * p.tpf;
            rec->h = h; rec->f = g + h;
        }
        __syncthreads();
    }
}
// ------------------------------------------------------------------------------------------
// The point-robot TSP heuristics with a few LANES per edge instead of a wave (large launches: the cover sweep marks the edge by
// h = PP_H_DEFERRED and goes on to its next edge).  The enumeration of RibbonManager.cpp:53-94 is a tree walk of lookups, adds and
// compares; run by a whole wave for one edge (pp_h_tsp_point) most of its instructions are the bookkeeping of spreading prefixes
// over lanes, and the table is rebuilt per edge by 64 lanes that mostly idle.  Here PP_HL_SPLIT adjacent lanes share an edge and
// take the root's branches in turn, each walking its subtree depth-first with control flow that is uniform across the wave for
// equal n (the data differ, the trip counts do not).  Distances between ribbon endpoints live in the edge's own triangle of LDS
// (pp_dist(a, b) == pp_dist(b, a) bit for bit: one entry per pair), the distances from the child's end position in registers.
// Same expressions as pp_h_tsp_point, fmin / fmax taken in another (exact) order: the same bits.
#ifndef PP_HL_SPLIT
#define PP_HL_SPLIT 4
#endif
#define PP_HL_TRI(MAXN) ((MAXN) * (2 * (MAXN) - 1) + 1)     // doubles per edge: pairs of 2n endpoints (+1: odd stride)
#define PP_HL_PTS(MAXN) (4 * (MAXN) + 1)
struct PPLaneTsp { const double* T; const double* CB; double twoW; int K; bool sortK; unsigned* cnt; };
// Branch and bound (round 3), exact.  Whatever order the remaining ribbons are visited in, the tour still has to add, for every
// one of them, its length - 2w and a transition INTO one of its endpoints from an endpoint of another ribbon, which is at least
// mind[r] = the smallest such distance in the edge's triangle; the fmax(., 0) clamps only raise a sum.  So every leaf below a node
// is at least sf + rb with rb = the sum of CB[r] = len[r] - 2w + mind[r] over the node's remaining ribbons (real arithmetic; the
// rounding of at most 3 x 6 additions of values below 1e6 is below 1e-8).  A subtree whose bound exceeds the best tour the edge
// has seen so far (`gbest`, shared by the edge's lanes) by more than PP_HL_MARGIN cannot hold the minimum and is not walked —
// when that is so for EVERY lane of the wave (control flow stays uniform: one ballot per node).  The minimum over the leaves that
// are visited is the minimum over all leaves: the same bits.  The K ribbons branched on are taken last-first (the nearer of the K
// farthest first: finds short tours, hence a tight gbest, earlier; a minimum does not depend on the order).
#ifndef PP_HL_MARGIN
#define PP_HL_MARGIN 1e-6
#endif
#ifndef PP_HL_PRUNE_MIN_REM
#define PP_HL_PRUNE_MIN_REM 2        // nodes with at least this many ribbons left are tested
#endif
#ifndef PP_HL_NO_PRUNE
#define PP_HL_PRUNE 1
#else
#define PP_HL_PRUNE 0
#endif
__device__ __forceinline__ double pp_quad_min(double v) {           // over the PP_HL_SPLIT adjacent lanes of an edge
#pragma unroll
    for (int m = 1; m < PP_HL_SPLIT; m <<= 1) v = fmin(v, __shfl_xor(v, m));
    return v;
}
#ifndef PP_HL_SHARE_MIN_REM
#define PP_HL_SHARE_MIN_REM 3        // nodes with at least this many ribbons left first take the best tour of the edge's other lanes (2: 160 us, 3: 153)
#endif
template <int REM>
__device__ __forceinline__ bool pp_lane_tsp_pruned(double sf, double rb, double& gbest) {
#if PP_HL_PRUNE
    if constexpr (REM >= PP_HL_SHARE_MIN_REM) gbest = pp_quad_min(gbest);
    return __ballot(!(sf + rb > gbest + PP_HL_MARGIN)) == 0ull;
#else
    return false;
#endif
}
__device__ __forceinline__ int pp_tri(int a, int b) {            // endpoints 0 .. 2n-1 (start / end of ribbon i = 2i / 2i + 1), a != b
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return ((hi * (hi - 1)) >> 1) + lo;
}
template <int REM>
__device__ __forceinline__ unsigned pp_lane_tsp_order(const double (&key)[REM > 0 ? REM : 1], unsigned ord) {
    unsigned o = 0;                                              // pp_tsp_sort_n's ranks (stable, descending)
#pragma unroll
    for (int i = 0; i < REM; i++) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < REM; j++)
            if (j != i) rank += ((key[j] > key[i]) | ((key[j] == key[i]) & (j < i))) ? 1 : 0;
        o |= ((ord >> (4 * i)) & 0xfu) << (4 * rank);
    }
    return o;
}
__device__ __forceinline__ unsigned pp_lane_tsp_drop(unsigned srt, int c) {        // remove position c of the 4-bit list
    const unsigned lowmask = (c == 0) ? 0u : ((1u << (4 * c)) - 1u);
    return (srt & lowmask) | ((srt >> 4) & ~lowmask);
}
// The last two levels in one piece when both remaining ribbons are branched on (K >= 2, or the All variant): eight leaves from
// ten table entries that do not depend on one another, instead of two nested loops of dependent lookups.  The order in which
// the two ribbons are tried does not matter for a minimum, so their sort is skipped.
// the two entries (pt, start of ribbon r) and (pt, end of ribbon r) of the triangle, pt not an endpoint of r: adjacent when pt is the
// larger index, one row apart otherwise (a third of pp_tri's arithmetic, and the lookups are most of this kernel)
__device__ __forceinline__ void pp_tri_pair(int pt, int r, int& i0, int& i1) {
    const bool above = pt > 2 * r;
    const int rowp = (pt * (pt - 1)) >> 1;
    i0 = above ? rowp + 2 * r : r * (2 * r - 1) + pt;
    i1 = above ? rowp + 2 * r + 1 : r * (2 * r + 1) + pt;
}
__device__ __forceinline__ double pp_lane_tsp_last2(const PPLaneTsp& c, double sf, unsigned ord, int pt) {
    const int a = (int)(ord & 0xfu), b = (int)((ord >> 4) & 0xfu);
    const double la = c.T[a * (2 * a + 1) + 2 * a], lb = c.T[b * (2 * b + 1) + 2 * b];
    int ia0, ia1, ib0, ib1;
    pp_tri_pair(pt, a, ia0, ia1);
    pp_tri_pair(pt, b, ib0, ib1);
    const double pas = c.T[ia0], pae = c.T[ia1], pbs = c.T[ib0], pbe = c.T[ib1];
    // the four distances between an endpoint of a and an endpoint of b: rows 2B and 2B + 1 of the larger ribbon B, columns 2A, 2A + 1
    const int A = a < b ? a : b, B = a < b ? b : a;
    const int rowE = B * (2 * B - 1) + 2 * A, rowO = B * (2 * B + 1) + 2 * A;
    const double xss = c.T[rowE], xee = c.T[rowO + 1];
    const double u = c.T[rowE + 1], v = c.T[rowO];           // (end of A, start of B), (start of A, end of B)
    const double xes = a < b ? u : v, xse = a < b ? v : u;
    const double baseA = sf + la - c.twoW, baseB = sf + lb - c.twoW;
    const double a0 = fmax(baseA + pas, 0) + lb - c.twoW;          // a from its start: now at a's end, b to go
    const double a1 = fmax(baseA + pae, 0) + lb - c.twoW;          // a from its end: now at a's start
    const double b0 = fmax(baseB + pbs, 0) + la - c.twoW;
    const double b1 = fmax(baseB + pbe, 0) + la - c.twoW;
    const double m0 = fmin(fmax(a0 + xes, 0), fmax(a0 + xee, 0));  // from a's end to b's start / end
    const double m1 = fmin(fmax(a1 + xss, 0), fmax(a1 + xse, 0));  // from a's start
    const double m2 = fmin(fmax(b0 + xse, 0), fmax(b0 + xee, 0));  // from b's end to a's start / end
    const double m3 = fmin(fmax(b1 + xss, 0), fmax(b1 + xes, 0));  // from b's start
    return fmin(fmin(m0, m1), fmin(m2, m3));
}
template <int REM>
__device__ __forceinline__ double pp_lane_tsp(const PPLaneTsp& c, double sf, unsigned ord, int pt, double rb, double& gbest) {
    if constexpr (REM == 0) {
        return sf;
    } else {
        if constexpr (REM >= PP_HL_PRUNE_MIN_REM) {
            if (pp_lane_tsp_pruned<REM>(sf, rb, gbest)) return PP_DBL_MAX;
        }
        if constexpr (REM == 2) {
#ifdef PP_HL_COUNT
            if (pp_lane() == 0) atomicAdd(c.cnt, 1u);
#endif
            if (c.K >= 2) { const double v = pp_lane_tsp_last2(c, sf, ord, pt); gbest = fmin(gbest, v); return v; }
        }
        unsigned srt = ord;
        if (REM > 1 && c.sortK && REM > c.K) {               // with K >= REM every ribbon is branched on: their order is immaterial
            double key[REM];
#pragma unroll
            for (int i = 0; i < REM; i++) {
                const int r = (int)((ord >> (4 * i)) & 0xfu);
                int i0, i1;
                pp_tri_pair(pt, r, i0, i1);
                key[i] = fmin(c.T[i0], c.T[i1]);
            }
            srt = pp_lane_tsp_order<REM>(key, ord);
        }
        const int nb = REM < c.K ? REM : c.K;                 // ribbons branched on, each entered from both ends
        double best = PP_DBL_MAX;
        for (int cc = nb - 1; cc >= 0; cc--) {
            const int rid = (int)((srt >> (4 * cc)) & 0xfu);
            const double len = c.T[rid * (2 * rid + 1) + 2 * rid];         // = pp_tri(2 rid, 2 rid + 1)
            const double base = sf + len - c.twoW;
            int i0, i1;
            pp_tri_pair(pt, rid, i0, i1);
            const double fromStart = fmax(base + c.T[i0], 0);                         // enter at the start, leave from the end
            const double fromEnd = fmax(base + c.T[i1], 0);
            if constexpr (REM == 1) {
                best = fmin(best, fmin(fromStart, fromEnd));
            } else {
                const unsigned nord = pp_lane_tsp_drop(srt, cc);
                const double nrb = PP_HL_PRUNE ? rb - c.CB[rid] : 0.0;
#pragma unroll 1
                for (int dir = 0; dir < 2; dir++)
                    best = fmin(best, pp_lane_tsp<REM - 1>(c, dir == 0 ? fromStart : fromEnd, nord, 2 * rid + 1 - dir, nrb, gbest));
            }
        }
        if constexpr (REM == 1) gbest = fmin(gbest, best);
        return best;
    }
}
// The root: the current point is the child's end position.
template <int N>
__device__ __forceinline__ double pp_lane_tsp_root(const PPLaneTsp& c, const double* P, double qx, double qy, int sub) {
    double d0[2 * N];
#pragma unroll
    for (int q = 0; q < 2 * N; q++) d0[q] = pp_dist(qx, qy, P[2 * q], P[2 * q + 1]);
    unsigned srt = 0x76543210u;
    if (N > 1 && c.sortK && N > c.K) {
        double key[N];
#pragma unroll
        for (int i = 0; i < N; i++) key[i] = fmin(d0[2 * i], d0[2 * i + 1]);
        srt = pp_lane_tsp_order<N>(key, srt);
    }
    const int b = 2 * (N < c.K ? N : c.K);
    double best = PP_DBL_MAX, gbest = PP_DBL_MAX, rbAll = 0;
#pragma unroll
    for (int r = 0; r < N; r++) rbAll += PP_HL_PRUNE ? c.CB[r] : 0.0;
    for (int u0 = 0; u0 < b; u0 += PP_HL_SPLIT) {
        const bool act = u0 + sub < b;
        const int u = act ? u0 + sub : 0;
        const int cc = u >> 1, dir = u & 1;
        const int rid = (int)((srt >> (4 * cc)) & 0xfu);
        const double len = c.T[rid * (2 * rid + 1) + 2 * rid];
        const int qi = 2 * rid + dir;                           // the endpoint entered: start (dir 0) or end
        double dd = d0[0];
#pragma unroll
        for (int j = 1; j < 2 * N; j++) dd = (qi == j) ? d0[j] : dd;
        const double nsf = fmax(0.0 + len - c.twoW + dd, 0);
        double v = nsf;
        // (a lane past the last branch repeats branch 0: its gbest is a tour of this edge too)
        if constexpr (N > 1) v = pp_lane_tsp<N - 1>(c, nsf, pp_lane_tsp_drop(srt, cc), 2 * rid + 1 - dir, PP_HL_PRUNE ? rbAll - c.CB[rid] : 0.0, gbest);
        if (act) best = fmin(best, v);
    }
#pragma unroll
    for (int m = 1; m < PP_HL_SPLIT; m <<= 1) best = fmin(best, __shfl_xor(best, m));
    return best;
}

// The edges the cover sweeps deferred, packed into one list per ribbon count (a wave of pp_k_heuristic_lanes then holds lists of
// one length: uniform control flow); one atomic per workgroup and list reserves its run.  defer_count[n], defer_list[(n-1) * total ..].
#define PP_DL_PER 4          // edges per thread of pp_k_deferred_list
__global__ __launch_bounds__(256) void pp_k_deferred_list(PPParams p) {
    __shared__ unsigned s_cnt[PP_HL_MAX_N][4 * PP_DL_PER];
    __shared__ unsigned s_base[PP_HL_MAX_N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nn[PP_DL_PER];                                   // ribbon count of a deferred edge, 0: not deferred
#pragma unroll
    for (int j = 0; j < PP_DL_PER; j++) {
        const long long e = ((long long)blockIdx.x * PP_DL_PER + j) * 256 + tid;
        nn[j] = 0;
        if (e < p.total_edges) {
            const ppgpu_edge_result* rec = p.out + e;
            if (rec->h == PP_H_DEFERRED && !(rec->flags & PPGPU_F_THROWS)) nn[j] = (int)((rec->info >> 8) & 0xffu);
        }
        for (int n = 1; n <= PP_HL_MAX_N; n++) {
            const unsigned long long m = __ballot(nn[j] == n);
            if (lane == 0) s_cnt[n - 1][j * 4 + wave] = (unsigned)__popcll(m);
        }
    }
    __syncthreads();
    if (tid < PP_HL_MAX_N) {
        unsigned tot = 0;
        for (int i = 0; i < 4 * PP_DL_PER; i++) { const unsigned c = s_cnt[tid][i]; s_cnt[tid][i] = tot; tot += c; }   // -> offsets
        s_base[tid] = tot ? atomicAdd(p.defer_count + 1 + tid, tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PP_DL_PER; j++) {
        const int n = nn[j];
        // every lane votes (n = 0: in no list), so the ballots below are taken by whole waves
        for (int q = 1; q <= PP_HL_MAX_N; q++) {
            const unsigned long long m = __ballot(n == q);
            if (n == q)
                p.defer_list[(size_t)(q - 1) * (size_t)p.total_edges + s_base[q - 1] + s_cnt[q - 1][j * 4 + wave] +
                             (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (unsigned)(((long long)blockIdx.x * PP_DL_PER + j) * 256 + tid);
        }
    }
}
#ifndef PP_HL_THREADS
#define PP_HL_THREADS 64
#endif
// (occupancy is set by LDS: 11.8 KB per 16 edges, 13 waves per CU)
#ifndef PP_HL_MIN_WAVES
#define PP_HL_MIN_WAVES 3
#endif
// One workgroup's edges: slots [blk * PER, (blk + 1) * PER) of the list of edges with n child ribbons.  MAXN sizes the tables.
template <int MAXN>
__device__ __forceinline__ void pp_heuristic_lanes_block(const PPParams& p, const int n, const unsigned blk, const unsigned count,
                                                         double* Tall, double* Pall, double* CBall) {
    const int tid = threadIdx.x;
    const unsigned slot = blk * (unsigned)(PP_HL_THREADS / PP_HL_SPLIT) + (unsigned)tid / PP_HL_SPLIT;
    const int sub = tid & (PP_HL_SPLIT - 1);
    const bool have = slot < count;
    const long long e = have ? (long long)p.defer_list[(size_t)(n - 1) * (size_t)p.total_edges + slot] : 0;
    ppgpu_edge_result* rec = p.out + e;
    double* T = Tall + (tid / PP_HL_SPLIT) * PP_HL_TRI(MAXN);
    double* P = Pall + (tid / PP_HL_SPLIT) * PP_HL_PTS(MAXN);
    double qx = 0, qy = 0, g = 0;
    if (have) {
        qx = rec->end_x; qy = rec->end_y; g = rec->g;
        const double* cr = p.child + (size_t)e * p.stride * 4;    // ribbon i = 4 doubles = endpoints 2i, 2i + 1
        for (int j = sub; j < 4 * n; j += PP_HL_SPLIT) P[j] = cr[j];
    }
    __syncthreads();
    if (have)
        for (int hi = 1 + sub; hi < 2 * n; hi += PP_HL_SPLIT)
            for (int lo = 0; lo < hi; lo++)
                T[((hi * (hi - 1)) >> 1) + lo] = pp_dist(P[2 * lo], P[2 * lo + 1], P[2 * hi], P[2 * hi + 1]);
    __syncthreads();
    double* CB = CBall + (tid / PP_HL_SPLIT) * MAXN;
#if PP_HL_PRUNE
    if (have)
        for (int r = sub; r < n; r += PP_HL_SPLIT) {              // CB[r] = len[r] - 2w + the shortest way into ribbon r from another ribbon
            double m = (n > 1) ? PP_DBL_MAX : 0.0;
            for (int q = 0; q < 2 * n; q++)
                if ((q >> 1) != r) m = fmin(m, fmin(T[pp_tri(q, 2 * r)], T[pp_tri(q, 2 * r + 1)]));
            CB[r] = T[r * (2 * r + 1) + 2 * r] - 2 * p.ribw + m;
        }
    __syncthreads();
#endif
    if (have) {
        PPLaneTsp c;
        c.T = T; c.CB = CB; c.twoW = 2 * p.ribw; c.cnt = p.need_big + 14;
#ifdef PP_HL_COUNT
        if ((threadIdx.x & 63) == 0) atomicAdd(p.need_big + 15, n == 6 ? 64u : (n == 5 ? 16u : (n == 4 ? 4u : 1u)));   // last-two-level calls per lane without pruning (K = 2)
#endif
        c.sortK = p.heuristic != PPGPU_H_TSP_POINT_ALL;
        c.K = c.sortK ? p.tsp_k : PP_TSP_MAX;
        double hdist = 0;
        {
            switch (n) {
                case 1: hdist = pp_lane_tsp_root<1>(c, P, qx, qy, sub); break;
                case 2: hdist = pp_lane_tsp_root<2>(c, P, qx, qy, sub); break;
                case 3: hdist = pp_lane_tsp_root<3>(c, P, qx, qy, sub); break;
                case 4: hdist = pp_lane_tsp_root<4>(c, P, qx, qy, sub); break;
                case 5: hdist = pp_lane_tsp_root<5>(c, P, qx, qy, sub); break;
                default: hdist = pp_lane_tsp_root<PP_HL_MAX_N>(c, P, qx, qy, sub); break;
            }
        }
        const double hh = hdist / p.max_speed * p.tpf;
        if (sub == 0) { rec->h = hh; rec->f = g + hh; }
    }
}
__global__ __launch_bounds__(PP_HL_THREADS, PP_HL_MIN_WAVES) void pp_k_heuristic_lanes(PPParams p) {
    constexpr int PER = PP_HL_THREADS / PP_HL_SPLIT;              // edges per workgroup
    __shared__ double Tall[PER * PP_HL_TRI(PP_HL_MAX_N)];
    __shared__ double Pall[PER * PP_HL_PTS(PP_HL_MAX_N)];
    __shared__ double CBall[PER * PP_HL_MAX_N];
    // which list this workgroup serves: the lists follow one another in whole workgroups
    // (longest lists first: an edge with 6 ribbons takes four times as long as one with 5, and the workgroups dispatched last
    // decide how the kernel drains)
    unsigned blk = blockIdx.x, count = 0;
    int n = PP_HL_MAX_N;
    for (; n >= 1; n--) {
        count = (unsigned)pp_const_i32(p.defer_count + n)[0];
        const unsigned nblk = (count + (unsigned)PER - 1u) / (unsigned)PER;
        if (blk < nblk) break;
        blk -= nblk;
    }
    if (n < 1) return;                                            // the grid is sized for "every edge deferred"
    pp_heuristic_lanes_block<PP_HL_MAX_N>(p, n, blk, count, Tall, Pall, CBall);
}

// ------------------------------------------------------------------------------------------
// Dubins lengths from open vertices to every sample, both radii (Edge::computeApproxCost for the
// k-nearest selection in SamplingBasedPlanner::expand, SamplingBasedPlanner.cpp:109-119).
// Thread per (vertex, sample); sample loads are coalesced, the vertex is a scalar load.
__global__ __launch_bounds__(256) void pp_k_dubins_lengths(const ppgpu_vertex* verts, int v0, const double* sx,
                                                           const double* sy, const double* sh, long long ns, double rho,
                                                           double rho_cov, double inc_d, double* out) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y;
    if (s >= ns) return;
    const ppgpu_vertex* V = verts + v0 + v;
    const double ax = V->x, ay = V->y, ayaw = pp_yaw(V->heading);
    const double bx = sx[s], by = sy[s], byaw = pp_yaw(sh[s]);
    double l0 = -1, l1 = -1;
    if (sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by)) > inc_d) {   // State::distanceTo, :111
        PPDubins d;
        pp_dubins_shortest(ax, ay, ayaw, bx, by, byaw, rho, d);
        l0 = pp_dubins_length(d, rho);
        pp_dubins_shortest(ax, ay, ayaw, bx, by, byaw, rho_cov, d);
        l1 = pp_dubins_length(d, rho_cov);
    }
    double2 o; o.x = l0; o.y = l1;
    reinterpret_cast<double2*>(out)[(size_t)v * ns + s] = o;
}

// k smallest (length, index) per (vertex, radius), in ascending (length, index) order; one 256-thread workgroup each.
// Two passes over the lengths instead of k: the k-th smallest of the 256 per-thread minima bounds the k-th smallest overall,
// so the second pass keeps the few entries not above that bound and ranks them.  (Fewer than k threads with an entry, or more
// survivors than the list holds - many equal lengths - fall back to k successive minimum scans.)
#define PP_SEL_CAP 1024
__device__ __forceinline__ bool pp_sel_less(double l1, long long i1, double l2, long long i2) {   // (l1, i1) < (l2, i2); i < 0 = none, after everything
    if (i1 < 0) return false;
    if (i2 < 0) return true;
    return l1 < l2 || (l1 == l2 && i1 < i2);
}
// Round j finds the lexicographic successor of round j-1's winner, so no exclusion list is needed.
__device__ __noinline__ void pp_select_rounds(const double* L, long long ns, int k, int* oi, double* ol, double* sl, long long* si) {
    double prevL = -INFINITY;
    long long prevI = -1;
    for (int j = 0; j < k; j++) {
        double bl = INFINITY;
        long long bi = -1;
        for (long long s = threadIdx.x; s < ns; s += 256) {
            double l = L[s * 2];
            if (l < 0) continue;                                           // closer than the increment: skipped
            bool after = (l > prevL) || (l == prevL && s > prevI);
            if (after && (l < bl || (l == bl && (bi < 0 || s < bi)))) { bl = l; bi = s; }
        }
        sl[threadIdx.x] = bl; si[threadIdx.x] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                double l2 = sl[threadIdx.x + o]; long long i2 = si[threadIdx.x + o];
                double l1 = sl[threadIdx.x]; long long i1 = si[threadIdx.x];
                if (i2 >= 0 && (i1 < 0 || l2 < l1 || (l2 == l1 && i2 < i1))) { sl[threadIdx.x] = l2; si[threadIdx.x] = i2; }
            }
            __syncthreads();
        }
        prevL = sl[0]; prevI = si[0];
        if (threadIdx.x == 0) { oi[j] = (int)prevI; ol[j] = prevI >= 0 ? prevL : -1.0; }
        __syncthreads();
        if (prevI < 0) {                                                    // fewer than k candidates
            for (int jj = j + 1; jj < k; jj++) if (threadIdx.x == 0) { oi[jj] = -1; ol[jj] = -1.0; }
            break;
        }
    }
}
__global__ __launch_bounds__(256) void pp_k_select_nearest(const double* lengths, long long ns, int k, int* out_idx,
                                                           double* out_len) {
    __shared__ double sl[PP_SEL_CAP];
    __shared__ long long si[PP_SEL_CAP];
    __shared__ double boundL;
    __shared__ long long boundI;
    __shared__ int count;
    const int vr = blockIdx.x;               // vertex * 2 + radius
    const int v = vr >> 1, r = vr & 1;
    const double* L = lengths + ((size_t)v * ns) * 2 + r;
    int* oi = out_idx + (size_t)vr * k;
    double* ol = out_len + (size_t)vr * k;
    const int tid = (int)threadIdx.x;
    // pass 1: this thread's smallest entry
    double bl = INFINITY;
    long long bi = -1;
    for (long long s = tid; s < ns; s += 256) {
        const double l = L[s * 2];
        if (l >= 0 && (bi < 0 || l < bl)) { bl = l; bi = s; }   // ascending s: the first of equal lengths stays
    }
    sl[tid] = bl; si[tid] = bi;
    if (tid == 0) { boundI = -1; boundL = 0; count = 0; }
    __syncthreads();
    // the k-th smallest of the 256 minima (rank by counting; every thread reads the same LDS word at a time)
    if (k <= 256) {
        int rank = 0;
        for (int j = 0; j < 256; j++) rank += pp_sel_less(sl[j], si[j], bl, bi) ? 1 : 0;
        if (bi >= 0 && rank == k - 1) { boundL = bl; boundI = bi; }
    }
    __syncthreads();
    const double bL = boundL;
    const long long bI = boundI;
    __syncthreads();
    if (bI < 0) {                            // fewer than k threads hold an entry (a short sample list)
        pp_select_rounds(L, ns, k, oi, ol, sl, si);
        return;
    }
    // pass 2: the entries not above the bound
    for (long long s = tid; s < ns; s += 256) {
        const double l = L[s * 2];
        if (l >= 0 && !pp_sel_less(bL, bI, l, s)) {
            const int slot = atomicAdd(&count, 1);
            if (slot < PP_SEL_CAP) { sl[slot] = l; si[slot] = s; }
        }
    }
    __syncthreads();
    const int m = count;
    if (m > PP_SEL_CAP) {                    // uniform: the list overflowed
        __syncthreads();
        pp_select_rounds(L, ns, k, oi, ol, sl, si);
        return;
    }
    // rank the survivors (at least k of them: the k minima themselves)
    for (int c = tid; c < m; c += 256) {
        const double l = sl[c];
        const long long i = si[c];
        int rank = 0;
        for (int j = 0; j < m; j++) rank += pp_sel_less(sl[j], si[j], l, i) ? 1 : 0;
        if (rank < k) { oi[rank] = (int)i; ol[rank] = l; }
    }
}

// ------------------------------------------------------------------------------------------
// The ORDER in which SamplingBasedPlanner::expand pushes the k winners of a radius (SamplingBasedPlanner.cpp:82-149): it visits
// the samples nearest-first by Euclidean distance, keeps a max-heap of the k best by approximate cost (std::push_heap, and
// std::pop_heap once the heap holds k + 1), stops once the heap is full and its worst LENGTH is not above the next distance, and
// then walks the heap ARRAY front to back.  Which vertex std::pop_heap later surfaces among children of exactly equal f depends
// on that order, so it is replayed here: one 256-thread workgroup per (vertex, radius).
//   1. candidates = valid samples (farther than the increment) with distance <= an upper bound of the k-th smallest length:
//      everything the scan can visit before it stops (a winner's distance is at most its length);
//   2. bitonic sort by (distance, sample index) in LDS;
//   3. wave 0 replays the scan with the heap held one slot per lane (parent/child moves are v_readlane and a lane-select, no memory),
//      following libstdc++'s __push_heap / __adjust_heap step for step.  A candidate whose cost is strictly above the heap's
//      root, pushed onto a full heap of pairwise distinct costs and popped again, leaves the array exactly as it was (the hole
//      sinks along the path the push shifted down and every element returns to its slot), so only the candidates at or below
//      the current root — a few dozen of the hundreds to thousands — are taken through the exact steps.
// Equal costs are handled exactly: the heap steps are libstdc++'s, and a candidate above the root is only skipped while no pair of
// equal costs sits where the pop would take another way down than the push came up (`safe`, in the kernel).  More than PP_ORD_CAP
// candidates (8 192 after the ring filter), k above 63, or such a pair turning up while the ring filter has dropped
// candidates fall back to ascending length and raise *fallbacks (the caller reports it).
#define PP_ORD_CAP 8192
struct PPOrdHeap { double cost, len; int idx; };
__device__ __forceinline__ void pp_ord_set(PPOrdHeap& h, int slot, double cost, double len, int idx) {   // slot and values are wave-uniform
    const bool mine = pp_lane() == slot;
    h.cost = mine ? cost : h.cost;
    h.len = mine ? len : h.len;
    h.idx = mine ? idx : h.idx;
}
__device__ __forceinline__ void pp_ord_move(PPOrdHeap& h, int to, int from) {
    pp_ord_set(h, to, pp_readlane(h.cost, from), pp_readlane(h.len, from), pp_readlane_i(h.idx, from));
}
// std::__push_heap(first, holeIndex, topIndex = 0, value, comp = cost <): the value climbs from `hole` past every ancestor whose
// cost is below it, until the first one that is not; the ancestors it passes move one step down the path.  Round 4: the whole
// climb at once instead of a loop of read-lane / compare / move per level (the replay's exact heap steps were 0.8 us each, most of
// pp_k_expand_order's 117 us on a planner round trip).  Lane a is an ancestor of `hole` iff (hole + 1) >> (level difference) == a + 1;
// ancestors have smaller indices the nearer the root, so "the first ancestor, seen from the hole, that is not below the value" is
// the HIGHEST lane among the ancestors that are not below it, and the ones passed are the ancestors above that lane.
__device__ __forceinline__ void pp_ord_sift_up(PPOrdHeap& h, int hole, double cost, double len, int idx) {
    const int lane = pp_lane();
    const int lh = 31 - __clz(hole + 1), ll = 31 - __clz(lane + 1);
    const bool onPath = (ll <= lh) && (((hole + 1) >> (lh - ll)) == lane + 1);          // the hole and its ancestors
    const bool isAnc = onPath && lane != hole;
    const unsigned long long anc = __ballot(isAnc);
    const unsigned long long stays = __ballot(isAnc && !(h.cost < cost));                // `comp(first + parent, value)` false
    unsigned long long passed = anc;
    if (stays) passed &= ~((2ull << (63 - __clzll((long long)stays))) - 1ull);            // only the ancestors between the hole and the first that stays
    // every lane of the path whose parent is passed takes its parent's entry
    const int par = lane > 0 ? ((lane - 1) >> 1) : 0;
    const double pc = __shfl(h.cost, par, PP_WAVE), pl = __shfl(h.len, par, PP_WAVE);
    const int pi = __shfl(h.idx, par, PP_WAVE);
    const bool recv = onPath && lane > 0 && ((passed >> par) & 1ull) != 0ull;
    h.cost = recv ? pc : h.cost; h.len = recv ? pl : h.len; h.idx = recv ? pi : h.idx;
    const int fin = passed ? (__ffsll((long long)passed) - 1) : hole;                     // the topmost ancestor passed, or the hole itself
    pp_ord_set(h, fin, cost, len, idx);
}
// std::pop_heap on n + 1 elements: the last one is taken out as `value`, the root leaves, std::__adjust_heap(first, 0, n, value):
// the hole sinks from the root to a leaf — at every node to the larger child, the RIGHT one on equal costs, a lone left child when
// n is even — the children on that path move up one step, and the value climbs back from the leaf (__push_heap).  Every node's
// choice is made at once (two shuffles), the path is then a handful of read-lanes.
__device__ __forceinline__ void pp_ord_pop(PPOrdHeap& h, int n) {
    const int lane = pp_lane();
    const double vc = pp_readlane(h.cost, n), vl = pp_readlane(h.len, n);
    const int vi = pp_readlane_i(h.idx, n);
    const int left = 2 * lane + 1, right = 2 * lane + 2;
    const double cl = __shfl(h.cost, left < PP_WAVE ? left : 0, PP_WAVE), cr = __shfl(h.cost, right < PP_WAVE ? right : 0, PP_WAVE);
    int pick = -1;
    if (right < n) pick = (cr < cl) ? left : right;          // `if (comp(first + secondChild, first + (secondChild - 1))) secondChild--`
    else if (left < n) pick = left;                          // `(len & 1) == 0 && secondChild == (len - 2) / 2`
    unsigned long long path = 1ull;
    int bottom = 0;
    for (;;) {
        const int nx = pp_readlane_i(pick, bottom);
        if (nx < 0) break;
        path |= 1ull << nx;
        bottom = nx;
    }
    // every node of the path but the last takes the entry of the child the hole went to
    const int src = pick >= 0 ? pick : 0;
    const double sc = __shfl(h.cost, src, PP_WAVE), sl = __shfl(h.len, src, PP_WAVE);
    const int si = __shfl(h.idx, src, PP_WAVE);
    const bool recv = ((path >> lane) & 1ull) != 0ull && lane != bottom;
    h.cost = recv ? sc : h.cost; h.len = recv ? sl : h.len; h.idx = recv ? si : h.idx;
    pp_ord_sift_up(h, bottom, vc, vl, vi);
}
// The four steps of the replay, each as parallel as its data allows (the first version did everything in the one workgroup of a
// (vertex, radius): two serial passes over all samples and a gather per 64 candidates made it the slowest kernel of the planner's
// round trip):
//   pp_k_lengths_minima     thread per (vertex, sample): both Dubins lengths (as pp_k_dubins_lengths) and, per 256-sample block, the
//                           smallest valid length of each radius and the number of valid samples;
//   pp_k_expand_bound       workgroup per (vertex, radius): U = the k-th smallest of the block minima (of groups of blocks when there
//                           are more than 512).  At least k samples are not longer than U, so U bounds the k-th smallest length from
//                           above, and it is nearly always equal to it (the k best samples seldom share a block).  Fewer than k valid
//                           samples, or fewer than k blocks with one: U = +inf (the scan visits everything);
//   pp_k_expand_candidates  thread per (vertex, sample): the samples the scan can visit — valid, distance <= U — appended to the
//                           (vertex, radius) list {distance, index, length} with one atomic per wavefront;
//   pp_k_expand_order       workgroup per (vertex, radius): sort the list by (distance, index), replay (above).
__global__ __launch_bounds__(256) void pp_k_lengths_minima(const ppgpu_vertex* verts, const double* sx, const double* sy, const double* sh, long long ns,
                                                           double rho, double rho_cov, double inc_d, double* out, double* blockmin, int* blockcnt) {
    __shared__ double m0[4], m1[4];
    __shared__ int nv[4];
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    const int v = blockIdx.y;
    double l0 = -1, l1 = -1;
    if (s < ns) {
        const ppgpu_vertex* V = verts + v;
        const double ax = V->x, ay = V->y, ayaw = pp_yaw(V->heading);
        const double bx = sx[s], by = sy[s], byaw = pp_yaw(sh[s]);
        if (sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by)) > inc_d) {   // State::distanceTo, SamplingBasedPlanner.cpp:111
            PPDubins d;
            pp_dubins_shortest(ax, ay, ayaw, bx, by, byaw, rho, d);
            l0 = pp_dubins_length(d, rho);
            pp_dubins_shortest(ax, ay, ayaw, bx, by, byaw, rho_cov, d);
            l1 = pp_dubins_length(d, rho_cov);
        }
        double2 o; o.x = l0; o.y = l1;
        reinterpret_cast<double2*>(out)[(size_t)v * ns + s] = o;
    }
    const double a0 = pp_wave_min(l0 >= 0 ? l0 : INFINITY), a1 = pp_wave_min(l1 >= 0 ? l1 : INFINITY);
    const int cnt = __popcll(__ballot(l0 >= 0));
    const int w = (int)(threadIdx.x >> 6);
    if (pp_lane() == 0) { m0[w] = a0; m1[w] = a1; nv[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t b = (size_t)v * gridDim.x + blockIdx.x;
        blockmin[2 * b] = fmin(fmin(m0[0], m0[1]), fmin(m0[2], m0[3]));
        blockmin[2 * b + 1] = fmin(fmin(m1[0], m1[1]), fmin(m1[2], m1[3]));
        blockcnt[b] = nv[0] + nv[1] + nv[2] + nv[3];
    }
}
#define PP_BOUND_CAP 512             // values the bound kernel ranks: block minima, merged into groups of consecutive blocks when there are more
__global__ __launch_bounds__(256) void pp_k_expand_bound(const double* blockmin, const int* blockcnt, int nblk, int k, double* bound, int* cand_count) {
    __shared__ double vals[PP_BOUND_CAP];
    __shared__ int valid;
    __shared__ double U;
    const int vr = blockIdx.x, v = vr >> 1, r = vr & 1;
    const int tid = (int)threadIdx.x;
    if (tid == 0) { valid = 0; U = INFINITY; cand_count[vr] = 0; }
    __syncthreads();
    const int per = (nblk + PP_BOUND_CAP - 1) / PP_BOUND_CAP;            // blocks per ranked value
    const int nval = (nblk + per - 1) / per;
    int c = 0;
    for (int j = tid; j < nval; j += 256) {
        double m = INFINITY;
        for (int b = j * per; b < (j + 1) * per && b < nblk; b++) {
            m = fmin(m, blockmin[2 * ((size_t)v * nblk + b) + r]);
            c += blockcnt[(size_t)v * nblk + b];
        }
        vals[j] = m;
    }
    atomicAdd(&valid, c);
    __syncthreads();
    // the k-th smallest of the minima by rank counting (ties ranked by position): k groups hold a sample not longer than it
    if (valid >= k) {
        for (int j = tid; j < nval; j += 256) {
            const double x = vals[j];
            if (!(x < INFINITY)) continue;
            int rank = 0;
            for (int i = 0; i < nval; i++) rank += ((vals[i] < x) | ((vals[i] == x) & (i < j))) ? 1 : 0;
            if (rank == k - 1) U = x;
        }
    }
    __syncthreads();
    if (tid == 0) bound[vr] = U;
}
__global__ __launch_bounds__(256) void pp_k_expand_candidates(const double* lengths, const ppgpu_vertex* verts, const double* sx, const double* sy,
                                                              long long ns, int two_radii, const double* bound, double* g_key, int* g_val, double* g_len,
                                                              long long g_cap, int* cand_count) {
    __shared__ int wcount[2][4], wbase[2][4];
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    const int v = blockIdx.y, w = (int)(threadIdx.x >> 6), lane = pp_lane();
    double l0 = -1, l1 = -1, d = 0;
    if (s < ns) {
        const double2 L = reinterpret_cast<const double2*>(lengths)[(size_t)v * ns + s];
        l0 = L.x; l1 = L.y;
        const double vx = verts[v].x, vy = verts[v].y;
        d = sqrt((sx[s] - vx) * (sx[s] - vx) + (sy[s] - vy) * (sy[s] - vy));       // State::distanceTo of the sample to the source
    }
    const int nr = two_radii ? 2 : 1;
    bool take[2] = {false, false};
    unsigned long long m[2] = {0ull, 0ull};
    for (int r = 0; r < nr; r++) {
        const double len = r ? l1 : l0;
        take[r] = (len >= 0) && !(d > bound[2 * v + r] * (1.0 + 1e-9));
        m[r] = __ballot(take[r]);
        if (lane == 0) wcount[r][w] = __popcll(m[r]);
    }
    __syncthreads();
    // one atomic per workgroup and radius (a device-scope atomic on one address completes every ~12 ns: per wavefront they took
    // longer than everything else in this kernel)
    if (threadIdx.x < (unsigned)nr) {
        const int r = (int)threadIdx.x;
        const int tot = wcount[r][0] + wcount[r][1] + wcount[r][2] + wcount[r][3];
        int base = tot ? atomicAdd(&cand_count[2 * v + r], tot) : 0;
        for (int i = 0; i < 4; i++) { wbase[r][i] = base; base += wcount[r][i]; }
    }
    __syncthreads();
    for (int r = 0; r < nr; r++) {
        if (!take[r]) continue;
        const long long slot = wbase[r][w] + __popcll(m[r] & ((1ull << lane) - 1ull));
        if (slot < g_cap) {
            const size_t at = (size_t)(2 * v + r) * g_cap + slot;
            g_key[at] = d; g_val[at] = (int)s; g_len[at] = r ? l1 : l0;
        }
    }
}
#define PP_ORD_INNER 1024            // candidates of the inner ring whose costs set the filter threshold
__global__ __launch_bounds__(256) void pp_k_expand_order(long long ns, int k, double max_speed, double tpf, int two_radii, const double* bound,
                                                         const double* g_key, const int* g_val, const double* g_len, long long g_cap,
                                                         const int* cand_count, const double* lengths, int* out_idx, unsigned* fallbacks) {
    __shared__ double cd[PP_ORD_CAP];        // 96 KB of the CU's 160 KB LDS: distance and list position (later sample index); the lengths
    __shared__ int ci[PP_ORD_CAP];           // stay in the list in memory and are fetched by position when the replay gets there
    __shared__ double inner[PP_ORD_INNER];
    __shared__ int nInner, nKept;
    __shared__ double threshold;
    const int vr = blockIdx.x, r = vr & 1;
    const int tid = (int)threadIdx.x;
    int* out = out_idx + (size_t)vr * k;
    const int M = cand_count[vr];
#ifdef PP_DBG_ORD
    long long tk0 = wall_clock64(), tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0; int nexact = 0;
#endif
    if ((r == 1 && !two_radii) || M <= 0) {                  // radius not in use (:60-63,97-100), or no sample farther than the increment
        for (int j = tid; j < k; j += 256) out[j] = -1;
        return;
    }
    const double* gk = g_key + (size_t)vr * g_cap;
    const int* gv = g_val + (size_t)vr * g_cap;
    const double* gl = g_len + (size_t)vr * g_cap;
    bool fallback = k >= PP_WAVE || (long long)M > g_cap;     // the heap holds k + 1 entries for a moment, one per lane
    // Most candidates cannot change the heap: a candidate beyond the inner ring (distance > U/4, a sixteenth of the disc) whose cost
    // is above the k-th smallest cost INSIDE that ring finds the heap full of k cheaper entries when its turn comes, whatever the
    // order inside the ring.  Only the ring and the cheaper ones outside it are sorted and replayed (a few hundred of thousands).
    const double U = bound[vr];
    if (tid == 0) { nInner = 0; nKept = 0; threshold = INFINITY; }
    __syncthreads();
    const bool filter = !fallback && (U < INFINITY) && M > 512;
    double dq = 0.25 * U;
    if (filter) {
        int n = 0;
        for (int attempt = 0; attempt < 12; attempt++) {
            for (int i = tid; i < M; i += 256)
                if (gk[i] <= dq) {
                    const int slot = atomicAdd(&nInner, 1);
                    if (slot < PP_ORD_INNER) inner[slot] = gl[i] / max_speed * tpf;
                }
            __syncthreads();
            n = nInner;
            if (n <= PP_ORD_INNER) break;
            // a crowded ring (tens of thousands of samples under a loose bound): any ring with at least k candidates will do,
            // halving the radius leaves about a quarter of them
            __syncthreads();
            if (tid == 0) nInner = 0;
            dq *= 0.5;
            __syncthreads();
        }
        if (n >= k && n <= PP_ORD_INNER)
            for (int i = tid; i < n; i += 256) {
                const double x = inner[i];
                int rank = 0;
                for (int j = 0; j < n; j++) rank += ((inner[j] < x) | ((inner[j] == x) & (j < i))) ? 1 : 0;
                if (rank == k - 1) threshold = x;
            }
        __syncthreads();
    }
#ifdef PP_DBG_ORD
    tk1 = wall_clock64();
#endif
    const double T = threshold;                               // +inf: keep everything
    for (int i = tid; i < M && !fallback; i += 256) {
        const double d = gk[i], len = gl[i];
        if (d <= dq || !(len / max_speed * tpf > T)) {
            const int slot = atomicAdd(&nKept, 1);
            if (slot < PP_ORD_CAP) { cd[slot] = d; ci[slot] = i; }
        }
    }
    __syncthreads();
    const int Mk = nKept;
#ifdef PP_DBG_ORD
    tk2 = wall_clock64();
#endif
    fallback = fallback || Mk > PP_ORD_CAP;
    if (!fallback) {
        int n2 = 64;
        while (n2 < Mk) n2 <<= 1;
        for (int i = Mk + tid; i < n2; i += 256) { cd[i] = INFINITY; ci[i] = 0x7fffffff; }
        // by (distance, sample index): list positions are not in sample order, so equal distances compare their samples
        for (int size = 2; size <= n2; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                __syncthreads();
                for (int t = tid; t < (n2 >> 1); t += 256) {
                    const int i = ((t / stride) * stride << 1) + (t % stride), j = i + stride;
                    const double ki = cd[i], kj = cd[j];
                    const int vi = ci[i], vj = ci[j];
                    bool gt = ki > kj;
                    if (ki == kj) gt = (vi == 0x7fffffff) ? (vj != 0x7fffffff) : (vj != 0x7fffffff && gv[vi] > gv[vj]);
                    if (gt == ((i & size) == 0)) { cd[i] = kj; cd[j] = ki; ci[i] = vj; ci[j] = vi; }
                }
            }
        __syncthreads();
    }
#ifdef PP_DBG_ORD
    tk3 = wall_clock64();
#endif
    if (tid >= PP_WAVE) return;
    // the replay (wave 0)
    const int lane = tid;
    PPOrdHeap h;
    h.cost = INFINITY; h.len = INFINITY; h.idx = -1;
    int hsize = 0;
    bool unsafeFiltered = false, stopped = false;
    // Skipping a candidate above the root is only right if pushing and popping it would put every element back (see above).  The
    // push shifts the ancestors of slot k down along their path and the pop's hole walks down again choosing the larger child,
    // the RIGHT one on equal costs: it retraces the path unless some ancestor whose path child is its left child has a right
    // child of EQUAL cost, or k is a right child whose left sibling equals their parent (then the two equal entries trade
    // places).  `safe` says that no such pair exists in the full heap as it stands; it changes only when the heap does.
    bool safe = true, anyEqual = false;
    auto heapSafe = [&]() -> bool {
        bool ok = true;
        for (int c = k; c > 0; c = (c - 1) >> 1) {
            const int par = (c - 1) >> 1;
            if (c & 1) { if (c + 1 < k) ok = ok && (pp_readlane(h.cost, c + 1) < pp_readlane(h.cost, par)); }
            else if (c == k) ok = ok && (pp_readlane(h.cost, k - 1) < pp_readlane(h.cost, par));
        }
        return ok;
    };
    if (!fallback) {
        for (int base = 0; base < Mk && !stopped && !unsafeFiltered; base += PP_WAVE) {
            const int c = base + lane;
            const bool have = c < Mk;
            const double d = have ? cd[c] : INFINITY;
            const int pos = have ? ci[c] : 0;                           // position in the candidate list
            const double len = have ? gl[pos] : INFINITY;
            const int idx = have ? gv[pos] : -1;
            const double cost = len / max_speed * tpf;                 // Edge::computeApproxCost (Edge.cpp:17)
            // the candidates of this chunk that can change the heap: every one while it is not full (or not `safe`), afterwards
            // those at or below the root's cost as it stands at the start of the chunk (the root only ever gets cheaper)
            unsigned long long rest = __ballot(have);
            const unsigned long long low = __ballot(have & (cost <= pp_readlane(h.cost, 0)));
            if (hsize >= k && safe && low == 0ull) {
                // nobody enters the heap; the scan still ends at the first distance the worst kept length does not exceed
                if (__ballot(have & !(pp_readlane(h.len, 0) > d)) != 0ull) stopped = true;
                continue;
            }
            while (true) {
                const unsigned long long cand = (hsize < k || !safe) ? rest : (rest & low);
                if (!cand) break;
                const int j = __ffsll((long long)cand) - 1;
                rest &= ~((2ull << j) - 1ull);                          // j and the candidates before it (no-ops in this state) are done
                const double dj = pp_readlane(d, j), lj = pp_readlane(len, j), cj = pp_readlane(cost, j);
                const int ij = pp_readlane_i(idx, j);
                if (hsize >= k) {
                    if (!(pp_readlane(h.len, 0) > dj)) { stopped = true; break; }        // :104-106, else branch :130-132
                    if (safe && cj > pp_readlane(h.cost, 0)) continue;                    // the root moved since the chunk began: a no-op
                }
#ifdef PP_DBG_ORD
                nexact++;
#endif
                // equal costs are what can make the heap unsafe: until one has been pushed onto an equal entry there is nothing to check
                anyEqual = anyEqual || (__ballot((lane < hsize) & (h.cost == cj)) != 0ull);
                pp_ord_sift_up(h, hsize, cj, lj, ij);                                     // push_back + std::push_heap
                hsize++;
                if (hsize > k) { hsize--; pp_ord_pop(h, hsize); }                         // std::pop_heap + pop_back
                if (hsize >= k && anyEqual) {
                    safe = heapSafe();
                    // the ring filter dropped candidates on the strength of "a no-op whenever its turn comes": not in this state
                    if (!safe && Mk < M) { unsafeFiltered = true; break; }
                }
            }
        }
    }
#ifdef PP_DBG_ORD
    tk4 = wall_clock64();
    if (lane == 0 && vr < 4) printf("[ord] vr %d M %d inner %d kept %d exact %d | filter %lld keep %lld sort %lld replay %lld (x10ns)\n", vr, M, nInner, Mk, nexact, tk1 - tk0, tk2 - tk1, tk3 - tk2, tk4 - tk3);
#endif
    if (fallback || unsafeFiltered) {
        // keep what a plain selection gives: the k cheapest of the list, ascending by (length, sample); only the push order is lost
        if (lane == 0) atomicAdd(fallbacks, 1u);
#ifdef PP_DBG_ORD
        if (lane == 0) printf("[ord] FALLBACK vr %d: k %d M %d cap %lld kept %d unsafeFiltered %d hsize %d | U %g dq %g inner %d threshold %g\n", vr, k, M, g_cap, Mk, (int)unsafeFiltered, hsize, U, dq, nInner, threshold);
#endif
        // More candidates within the bound than the list holds (M > g_cap: slots beyond it were dropped in the order the atomics
        // happened to arrive): the truncated list is not a set anyone can name, so the selection runs over the vertex's whole row
        // of lengths instead (-1 = closer than the increment, never a candidate: SamplingBasedPlanner.cpp:111)
        const bool fullRow = (long long)M > g_cap;
        const long long Mc = fullRow ? ns : (long long)M;
        const double* row = lengths + (size_t)(vr >> 1) * (size_t)ns * 2 + r;
        double prevL = -INFINITY; int prevI = -1;
        for (int j = 0; j < k; j++) {
            double bl = INFINITY; int bi = 0x7fffffff;
            for (long long c = lane; c < Mc; c += PP_WAVE) {
                const double l = fullRow ? row[2 * c] : gl[c];
                const int i = fullRow ? (int)c : gv[c];
                if (fullRow && !(l >= 0)) continue;
                const bool after = (l > prevL) || (l == prevL && i > prevI);
                if (after && (l < bl || (l == bl && i < bi))) { bl = l; bi = i; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const double l2 = __shfl_xor(bl, o, PP_WAVE); const int i2 = __shfl_xor(bi, o, PP_WAVE);
                if (l2 < bl || (l2 == bl && i2 < bi)) { bl = l2; bi = i2; }
            }
            if (lane == 0) out[j] = (bi == 0x7fffffff) ? -1 : bi;
            prevL = bl; prevI = bi;
            if (bi == 0x7fffffff) { for (int jj = j + 1 + lane; jj < k; jj += PP_WAVE) out[jj] = -1; break; }
        }
        return;
    }
    if (lane < k) out[lane] = (lane < hsize) ? h.idx : -1;      // the heap array, front to back
}

// ------------------------------------------------------------------------------------------
// ppgpu_expand_host uploads one block {vertices | ribbons | explicit target x, y, heading per vertex | has-target flags} in one
// copy; this puts its parts where the other kernels expect them (vertex array, ribbon pool, the slots behind the stored
// samples, flags).  Everything is 8-byte words except the flags.
__global__ __launch_bounds__(256) void pp_k_expand_unpack(const unsigned char* blk, int nv, int n_ribbons, ppgpu_vertex* verts, double* ribbons,
                                                        double* ex, double* ey, double* eh, unsigned char* flags, unsigned* zero_word) {
    const size_t wv = (size_t)nv * (sizeof(ppgpu_vertex) / 8), wr = (size_t)n_ribbons * 4;
    const unsigned long long* src = (const unsigned long long*)blk;
    const size_t total = wv + wr + 3 * (size_t)nv;
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0u;      // the push-order fallback counter of this round trip
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total + (size_t)nv; i += (size_t)gridDim.x * 256) {
        if (i < wv) ((unsigned long long*)verts)[i] = src[i];
        else if (i < wv + wr) ((unsigned long long*)ribbons)[i - wv] = src[i];
        else if (i < wv + wr + nv) ((unsigned long long*)ex)[i - wv - wr] = src[i];
        else if (i < wv + wr + 2 * (size_t)nv) ((unsigned long long*)ey)[i - wv - wr - nv] = src[i];
        else if (i < total) ((unsigned long long*)eh)[i - wv - wr - 2 * (size_t)nv] = src[i];
        else flags[i - total] = blk[total * 8 + (i - total)];
    }
}

// ------------------------------------------------------------------------------------------
// The edge list of SamplingBasedPlanner::expand for every open vertex, in its push order (see ppgpu_expand_host): E slots
// per vertex, unused slots hold an all-ones descriptor (vertex index out of range: the costing kernels skip it).
__global__ __launch_bounds__(64) void pp_k_build_expand_edges(int nverts, int k, const int* nearest_idx /* [nv][2][k] */, const unsigned char* has_extra,
                                                             long long first_extra, int two_speeds, int two_radii, int E,
                                                             unsigned long long* edges, const unsigned* fallbacks, unsigned long long* header) {
    const int v = blockIdx.x * 64 + threadIdx.x;
    // (the push-order fallback count of this round trip travels home in the block's header: one download instead of two)
    if (v == 0 && header) header[0] = fallbacks ? (unsigned long long)*fallbacks : 0ull;
    if (v >= nverts) return;
    unsigned long long* out = edges + (size_t)v * E;
    int n = 0;
    const int nsp = two_speeds ? 2 : 1, nrad = two_radii ? 2 : 1;
    // coverageAllowed = (radius == coverageTurningRadius): with one radius that single radius IS the coverage radius
    if (has_extra[v]) {
        for (int si = 0; si < nsp; si++)
            for (int ri = 0; ri < nrad; ri++) {
                const unsigned cov = (two_radii ? ri == 1 : 1) ? PPGPU_EDGE_COVERAGE : 0u;
                out[n++] = ((unsigned long long)(cov | (si == 1 ? PPGPU_EDGE_SLOW : 0u)) << 56) | ((unsigned long long)v << 32) |
                           (unsigned long long)(unsigned)(first_extra + v);
            }
    }
    if (nearest_idx) {
        for (int ri = 0; ri < nrad; ri++) {
            const unsigned cov = (two_radii ? ri == 1 : 1) ? PPGPU_EDGE_COVERAGE : 0u;
            const int slot = (ri == 1) ? 1 : 0;           // slot 1 of the selection is always the coverage radius
            for (int j = 0; j < k; j++) {
                const int s = nearest_idx[((size_t)v * 2 + slot) * k + j];
                if (s < 0) break;
                for (int si = 0; si < nsp; si++)
                    out[n++] = ((unsigned long long)(cov | (si == 1 ? PPGPU_EDGE_SLOW : 0u)) << 56) | ((unsigned long long)v << 32) |
                               (unsigned long long)(unsigned)s;
            }
        }
    }
    for (; n < E; n++) out[n] = ~0ull;
}

// ------------------------------------------------------------------------------------------
// Incumbent selection: lexicographic min of (bits of f, edge index) over feasible edges — the
// batch form of `if (!best || v->f() < best->f()) best = v` (AStarPlanner.cpp:109-117).
// Stage 1: wave shuffle-reduce + LDS across the 4 waves -> one partial per workgroup;
// stage 2: one workgroup over the partials.  Deterministic (no atomics).
__device__ __forceinline__ void pp_key_min(unsigned long long& f, unsigned long long& i, unsigned long long f2, unsigned long long i2) {
    if (f2 < f || (f2 == f && i2 < i)) { f = f2; i = i2; }
}
__device__ __forceinline__ void pp_key_wave_min(unsigned long long& f, unsigned long long& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long f2 = __shfl_xor(f, o, PP_WAVE), i2 = __shfl_xor(i, o, PP_WAVE);
        pp_key_min(f, i, f2, i2);
    }
}
__global__ __launch_bounds__(256) void pp_k_best_stage1(const ppgpu_edge_result* res, long long n, int goal_only,
                                                        unsigned long long base, unsigned long long* partial) {
    __shared__ unsigned long long sf[4], si[4];
    unsigned long long f = ~0ull, idx = ~0ull;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        unsigned fl = res[e].flags;
        bool ok = !(fl & PPGPU_F_INFEASIBLE) && (!goal_only || (fl & PPGPU_F_GOAL));
        if (ok) {
            unsigned long long fb = (unsigned long long)__double_as_longlong(res[e].f);
            pp_key_min(f, idx, fb, base + (unsigned long long)e);
        }
    }
    pp_key_wave_min(f, idx);
    if (pp_lane() == 0) { sf[threadIdx.x >> 6] = f; si[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) pp_key_min(f, idx, sf[w], si[w]);
        partial[2 * blockIdx.x] = f; partial[2 * blockIdx.x + 1] = idx;
    }
}
__global__ __launch_bounds__(256) void pp_k_best_stage2(const unsigned long long* partial, int nparts, unsigned long long* key2) {
    __shared__ unsigned long long sf[4], si[4];
    unsigned long long f = ~0ull, idx = ~0ull;
    for (int i = threadIdx.x; i < nparts; i += 256) pp_key_min(f, idx, partial[2 * i], partial[2 * i + 1]);
    pp_key_wave_min(f, idx);
    if (pp_lane() == 0) { sf[threadIdx.x >> 6] = f; si[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) pp_key_min(f, idx, sf[w], si[w]);
        key2[0] = f; key2[1] = idx;
    }
}
// after an all-gather of per-rank keys: lexicographic min of `n` (f, idx) pairs
__global__ void pp_k_key_min_n(const unsigned long long* keys, int n, unsigned long long* key2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long f = ~0ull, idx = ~0ull;
        for (int i = 0; i < n; i++) pp_key_min(f, idx, keys[2 * i], keys[2 * i + 1]);
        key2[0] = f; key2[1] = idx;
    }
}
