// pp_k_sweep.h — the collision sweep of Edge::computeTrueCost (Edge.cpp:143-152,172-174): pp_window_pose, the chunk-skip planner
// (pp_k_plan_skips) and pp_k_pose_sweep.  Included by pp_kernels.h.
#pragma once
// ------------------------------------------------------------------------------------------
// Edge costing = four launches over the same edge list (the fourth, pp_k_heuristic, further down), one wavefront-sized piece
// of work each:
//
//   pp_k_solve_edges  (lane per edge)  phase 0: Vertex::connect + Edge::computeApproxCost: Dubins solve, curve constants,
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
    int pad;        // (round 3 measured a per-edge hit total here: pose sweep +33 us for -10 elsewhere, not taken)
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
        const double hi0 = PP_SF64(p0), hi1 = PP_SF64(hi1);
        const int mine = pp_seg_of(tprime, hi0, hi1);
        const int firstSeg = __builtin_amdgcn_readfirstlane(mine);
        const int lastSeg = __builtin_amdgcn_readlane(mine, 63 - __clzll((long long)__ballot(valid)));
        if (__ballot(mine != firstSeg) != 0ull) {
            // the window straddles a junction: every lane takes its own segment's constants from memory
            pp_setup_seg_pose<TAB>(S, mine, tprime, ux, uy, uth);
        } else {
            uniformSeg = true;
            if (cur != firstSeg) { cur = firstSeg; cs = pp_seg_load_uniform(&S->seg[cur], cur, hi0, PP_SF64(p1), hi1, PP_SI32(type)); }
        }
        if (cur != lastSeg && !uniformSeg) { cur = lastSeg; cs = pp_seg_load_uniform(&S->seg[cur], cur, hi0, PP_SF64(p1), hi1, PP_SI32(type)); }
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
        PPSeg _cs = PPSeg{0, 0, 0, 0, 0, INFINITY, -INFINITY, 0, 0, 1};   /* matches nothing: the first use loads a segment */ \
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
    const double hi0 = S->p0, hi1 = S->hi1;
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
        const bool straight = pp_word_seg_type(S->type, segL) == 1;
        eqWord = straight ? ~0ull : 0ull;
        if (k0 > 0) {
            okHead = (dP >= 0.0) && (segP == segL) && (straight || (tpL - tpP) > 65.0 * 1e-9);
        } else {
            // the first chunk: its first step is compared with the source vertex's heading (`lastHeading` starts there, Edge.cpp:96),
            // which is one evaluation of the sweep's own heading expression — no sine or cosine in it
            const double tpF = (rho_inv != 0.0) ? dF * rho_inv : dF / rho;
            const int segF = pp_seg_of(tpF, hi0, hi1);
            okHead = (dF >= 0.0) && (segF == segL) && (straight || (tpL - tpF) > 64.0 * 1e-9);
            const PPSegBase* g = &S->seg[segF];
            const int gtype = pp_word_seg_type(S->type, segF);
            const double tt = (tpF - pp_seg_o1(segF, S->p0)) - pp_seg_o2(segF, S->p1);
            const double uth0 = (gtype == 1) ? (0.0 + g->bth) : ((gtype == 0) ? (tt + g->bth) : (-tt + g->bth));
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
        // The chunk's poses against the chord between its first and its last pose.  The vehicle moves at constant speed on a curve
        // of curvature <= 1/rho and an obstacle at constant velocity, both linear in the step time: relative to an obstacle's box
        // the pose at time t is within dev = L^2 / (8 rho) of the point of the chord at the same time fraction (a function that
        // vanishes at both ends with second derivative bounded by 1/rho), L = the chunk's arc length.  A box is convex, so
        // both ends inside it shrunk by dev => every pose inside (64 hits per step, known without sampling); both ends beyond one
        // face grown by dev => no pose inside.  At config 3 dev is 0.08 .. 0.16 m where the ball around the middle pose needed 3.5 m.
        const double tpF2 = (rho_inv != 0.0) ? dF * rho_inv : dF / rho, tpL2 = (rho_inv != 0.0) ? dL * rho_inv : dL / rho;
        double uxF, uyF, uxL, uyL, uthU;
        pp_setup_seg_pose(S, pp_seg_of(tpF2, hi0, hi1), tpF2, uxF, uyF, uthU);
        pp_setup_seg_pose(S, pp_seg_of(tpL2, hi0, hi1), tpL2, uxL, uyL, uthU);
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
            {
                // most boxes on an edge's list are nowhere near this chunk: every pose lies within Lc/2 + dev of the chord's midpoint,
                // the box within its own reach of its centre, which moves at most |Speed| ht around where it is at the middle step
                const double dtM = tM - o.Time;
                const double ddx = 0.5 * (xF + xL) - (o.X + o.Speed * dtM * o.cosYaw), ddy = 0.5 * (yF + yL) - (o.Y + o.Speed * dtM * o.sinYaw);
                const double R = o.reach + 0.5 * Lc + dev + fabs(o.Speed) * ht + 1e-3;
                if (ddx * ddx + ddy * ddy > R * R) return;
            }
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
        unsigned long long m = S->omask;
        if (p.n_obst > PP_WAVE) m = 0ull;
        while (decided && m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            against(OB[j]);
        }
        if (p.n_obst > PP_WAVE)
            for (int j = 0; j < p.n_obst && decided; j++) against(OB[j]);
        obstClear = decided && nInside == 0;
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
        double ux, uy, uth;
        pp_setup_seg_pose(S, segP, tpP, ux, uy, uth);
        p.track_eq[(size_t)e * p.nch + chunk] = (unsigned long long)__double_as_longlong(pp_heading_from_yaw(pp_mod2pi(uth)));   // (the sweep replaces it by the chunk's bits)
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
__device__ __forceinline__ void pp_pose_sweep_edge(const PPParams& p, const long long e) {
    const int lane = pp_lane();
    const PPEdgeSetup* S = p.setup + e;
    PPTrackSummary* sum = p.track_summary + e;
    const unsigned sflags = (unsigned)PP_SI32(sflags);
    const int dubType = PP_SI32(type);
    if ((sflags & (PP_SETUP_MALFORMED | PP_SETUP_COLOCATED)) || dubType < 0) {
        if (lane == 0) { sum->limit = 0; sum->blocked = 0; sum->dub_err = 0; sum->pad = 0; }
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
            if (lane == 0) { sum->limit = 0; sum->blocked = 2; sum->dub_err = 0; sum->pad = 0; }
            return;
        }
    }
    // the segment of the curve the sweep is on: its constants live in scalar registers, the other two stay in memory
    int cur = 0;
    PPSeg cs = pp_seg_load_uniform(&S->seg[0], 0, PP_SF64(p0), PP_SF64(p1), PP_SF64(hi1), PP_SI32(type));

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
    // Can any obstacle come near this edge at all?  Every sampled pose lies within `travel` (arc length from the start of
    // the curve) of the curve's first point, and an obstacle moves at most |Speed| * duration during the sweep: the same
    // kind of exact bound as the per-chunk culling, applied once.
    bool anyObstacle = false;
    // Up to 64 obstacles: lane i keeps obstacle i's motion for the whole sweep (position at the first step's time, velocity,
    // squared culling radius), so the per-chunk culling below is a dozen instructions and no loads.  The bound is the one
    // pp_obstacle_hits_chunk uses (reach + chunk span + |Speed| * chunk time + slack); it only has to be conservative.
    const bool laneCull = p.n_obst <= PP_WAVE;
    double oX0 = 0, oY0 = 0, oVx = 0, oVy = 0, oR2 = -1.0, cullT0 = 0;
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

    // Chunks of 64 steps that provably touch neither a blocked cell nor an obstacle are not sampled at all (pp_k_plan_skips decided
    // which, one thread per chunk); the others go through the per-step code below, one step per lane.
    const unsigned char* skipb = p.track_skip ? p.track_skip + (size_t)e * p.nch : nullptr;
    const double* carry = reinterpret_cast<const double*>(p.track_eq + (size_t)e * p.nch);   // (a chunk's word before the sweep gets to it)
    bool afterSkip = false, stop = false;
    for (int g0 = 0; !stop; g0 += PP_WAVE) {
        const unsigned sbits = (skipb && g0 + lane < p.nch) ? (unsigned)skipb[g0 + lane] : 0u;
        const unsigned long long skips = __ballot((sbits & PP_SKIP_ALL) != 0u);
        const unsigned long long gclear = __ballot((sbits & PP_SKIP_GRID) != 0u), oclear = __ballot((sbits & PP_SKIP_OBST) != 0u);
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
                if (!gridClear) blk = valid & pp_is_blocked(p.grid, x, y);   // Edge.cpp:144 (pp_k_plan_skips may have ruled it out for the whole chunk)
            }
            double dens = 0;
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
    }
    const int anyErr = (__ballot(dubErr) != 0ull) ? 1 : 0;
    if (lane == 0) { sum->limit = limit; sum->blocked = blocked; sum->dub_err = anyErr; sum->pad = 0; }
}

#ifndef PP_POSE_MIN_WAVES
#define PP_POSE_MIN_WAVES 6
#endif
// n_edges = slice size (ppgpu.hip: launch_cost)
__global__ __launch_bounds__(PP_WPB * 64, PP_POSE_MIN_WAVES) void pp_k_pose_sweep(PPParams p) {
    PPQueue qs = pp_queue_init();
    for (PP_EACH_EDGE(idx, 1, PP_Q_POSE, p.n_edges, 1))
        pp_pose_sweep_edge<false>(p, p.ws_base + idx);
}
__global__ __launch_bounds__(PP_WPB * 64, 4) void pp_k_pose_sweep_gaussian(PPParams p) {
    PPQueue qs = pp_queue_init();
    for (PP_EACH_EDGE(idx, 1, PP_Q_POSE, p.n_edges, 1))
        pp_pose_sweep_edge<true>(p, p.ws_base + idx);
}
