// pp_kernels.h — the gfx950 kernels of the hot path.  Included once by ppgpu.hip.
#pragma once
#include "pp_device.h"
#include "../../include/ppgpu.h"

// Everything a costing launch needs, passed by value (kernarg segment, scalar loads).
struct PPParams {
    // PlannerConfig / Edge constants / RibbonManager settings
    double max_speed, slow_speed, rho, rho_cov, horizon, tmin, inc_d, sst, ribw, cpf, tpf;
    int heuristic, tsp_k;
    // world
    PPGrid grid;
    const PPObst* obst; int n_obst;
    // open vertices
    const ppgpu_vertex* verts; const double* ribbons; const double* tgrid; int ng; int nverts;
    // targets
    const double* sx; const double* sy; const double* sh; long long n_samples;
    // edges: explicit list, or dense enumeration when edges == nullptr; wedges: edges whose curve is given
    // (Vertex::connect(start, DubinsWrapper, coverageAllowed), Vertex.cpp:28-36) instead of solved
    const unsigned long long* edges; long long n_edges;
    const ppgpu_wrapper_edge* wedges;
    int v0, nv; long long s0, ns; unsigned cfg_mask; int per;
    // outputs
    ppgpu_edge_result* out; double* child; int stride;
};

// ------------------------------------------------------------------------------------------
// Collision-check time grid, one row per open vertex (Edge.cpp:114-120,173): the reference
// advances `intermediate.time() += timeIncrement` once per step, so step times are a running
// sum, not t0 + k*inc; they depend only on the source vertex's time, hence one table per vertex
// (ng entries), built sequentially by one lane per vertex.
__global__ void pp_k_time_grid(const ppgpu_vertex* verts, int nverts, double sst, double inc_d, double max_speed,
                               int ng, double* tgrid) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nverts) return;
    double timeIncrement = inc_d / max_speed;                 // Edge.cpp:114
    double t = verts[v].time;
    double timeSinceStart = t - sst;                          // :117
    double timeNudge = fmod(timeSinceStart, timeIncrement);   // :118
    t += timeNudge;                                           // :119
    double* row = tgrid + (size_t)v * ng;
    for (int k = 0; k < ng; k++) {
        row[k] = t;
        t += timeIncrement;                                   // :173
    }
}

// ------------------------------------------------------------------------------------------
// Edge costing: one wavefront per edge, 4 edges per 256-thread workgroup.
//
//   phase 0 (uniform)   Vertex::connect + Edge::computeApproxCost: Dubins solve, curve constants
//   phase A (64 lanes)  64 consecutive collision-check steps at a time: closed-form pose,
//                       occupancy lookup, dynamic-obstacle box tests
//   phase B (uniform + ribbon-per-lane)  the sequential coverage state machine of
//                       Edge.cpp:153-171, visited only at its event steps
//   phase C             end state, last cover, cost, g/h/f, one 128-byte record per edge
#ifndef PP_WPB
#define PP_WPB 4   // wavefronts (= edges) per workgroup of the two per-edge kernels
#endif
#ifndef PP_MIN_WAVES
#define PP_MIN_WAVES 6   // 2nd __launch_bounds__ argument: waves per SIMD the register allocator must leave room for
                         // (measured on MI355X, config 3: 1 -> 10.7 ms, 5 -> 10.6 ms, 6 -> 10.3 ms, 8 -> slower: spills)
#endif
__global__ __launch_bounds__(PP_WPB * 64, PP_MIN_WAVES) void pp_k_cost_edges(PPParams p) {
    __shared__ double lds_all[PP_WPB][PP_WAVE * 4];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = pp_lane();
    const long long e = (long long)blockIdx.x * PP_WPB + wave;
    if (e >= p.n_edges) return;
    double* lds = lds_all[wave];

    // ---- which edge
    unsigned vi, target, cbits;
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
    vi = (unsigned)__builtin_amdgcn_readfirstlane((int)vi);
    target = (unsigned)__builtin_amdgcn_readfirstlane((int)target);
    cbits = (unsigned)__builtin_amdgcn_readfirstlane((int)cbits);

    unsigned flags = 0;
    ppgpu_edge_result* rec = p.out + e;
    if (vi >= (unsigned)p.nverts || (!p.wedges && (long long)target >= p.n_samples)) {
        // malformed descriptor: fail loudly in the record, touch nothing else
        if (lane == 0) { rec->flags = PPGPU_F_INFEASIBLE | PPGPU_F_THROWS | PPGPU_F_DUBINS_ERR; rec->info = 0; }
        return;
    }

    const ppgpu_vertex* V = p.verts + vi;
    const double srcX = pp_sgpr(V->x), srcY = pp_sgpr(V->y), srcH = pp_sgpr(V->heading), srcT = pp_sgpr(V->time), srcG = pp_sgpr(V->g);
    double cct = pp_sgpr(V->coverage_completed_time);
    int nrib = __builtin_amdgcn_readfirstlane(V->ribbon_count);
    const bool cov = (cbits & PPGPU_EDGE_COVERAGE) != 0;
    double rho = cov ? p.rho_cov : p.rho;                             // Edge.cpp:73-76
    double speed = (cbits & PPGPU_EDGE_SLOW) ? p.slow_speed : p.max_speed;
    double tgtX = 0, tgtY = 0, tgtH = 0;
    if (!p.wedges) { tgtX = p.sx[target]; tgtY = p.sy[target]; tgtH = p.sh[target]; }

    // this vertex's ribbons, one per lane (Vertex::connect copies the parent's RibbonManager, Vertex.cpp:24)
    PPRibbon rib = {0, 0, 0, 0};
    if (nrib > PP_WAVE) { nrib = PP_WAVE; flags |= PPGPU_F_RIBBON_OVF; }
    if (lane < nrib) {
        const double* rp = p.ribbons + 4 * ((size_t)V->ribbon_offset + lane);
        rib.sx = rp[0]; rib.sy = rp[1]; rib.ex = rp[2]; rib.ey = rp[3];
    }
    const bool startedDone = (nrib == 0);                             // Edge.cpp:93

    // ---- phase 0: the curve (Edge::computeApproxCost -> DubinsWrapper::set)
    bool colocated = false;
    PPDubins dub;
    PPCurve cv;
    double approx, wEnd, wStart;
    if (p.wedges) {
        // the wrapper comes with the edge: DubinsWrapper::fill semantics, start time of ITS curve, possibly truncated end
        const ppgpu_wrapper_edge* W = p.wedges + e;
        dub.p0 = W->param[0]; dub.p1 = W->param[1]; dub.p2 = W->param[2]; dub.type = W->type;
        if (dub.type < 0 || dub.type > 5) dub.type = -1;
        rho = W->rho; speed = W->speed;
        pp_curve_init_wave(cv, W->qi[0], W->qi[1], W->qi[2], rho, dub);
        wStart = W->start_time; wEnd = W->end_time;
        approx = (wEnd - srcT) * 1.0;                                 // Edge::setEnd(wrapper), Edge.cpp:208-216
    } else {
        colocated = (srcX == tgtX) && (srcY == tgtY) && (srcH == tgtH);              // State::isCoLocated
        pp_dubins_shortest_wave(srcX, srcY, pp_yaw(srcH), tgtX, tgtY, pp_yaw(tgtH), rho, dub);
        pp_curve_init_wave(cv, srcX, srcY, pp_yaw(srcH), rho, dub);
        approx = cv.length / speed * 1.0;                             // Edge.cpp:17
        wStart = srcT;
        wEnd = srcT + cv.length / speed;                              // DubinsWrapper::setEndTime
    }
    // everything computed so far is identical in all 64 lanes: move it to scalar registers
    pp_curve_scalarize(cv);
    dub.p0 = pp_sgpr(dub.p0); dub.p1 = pp_sgpr(dub.p1); dub.p2 = pp_sgpr(dub.p2); dub.type = __builtin_amdgcn_readfirstlane(dub.type);
    approx = pp_sgpr(approx); wEnd = pp_sgpr(wEnd); wStart = pp_sgpr(wStart); rho = pp_sgpr(rho); speed = pp_sgpr(speed);
    double endTime = pp_sgpr(fmin(p.horizon + 1e-12 + p.sst, wEnd));  // Edge.cpp:90
    bool infeasible = (srcT >= endTime);                              // :102-110
    bool throwsRef = colocated || (dub.type < 0);
    if (dub.type < 0) flags |= PPGPU_F_DUBINS_ERR;

    // ---- sweep state
    const double* tg = p.tgrid + (size_t)vi * p.ng;
#ifdef PP_DBG_EVENTS
    int dbgEvents = 0;
#endif
    int rdt = -1;                       // `auto ribbonsDoneTime = -1;` is an int (Edge.cpp:92)
    int nextEvent = 0;                  // toCoverDistance starts at 0: step 0 is an event
    int hitsAcc = 0;
    int steps = 0;
    double ix = srcX, iy = srcY, ih = srcH;   // `intermediate` pose
    double lastHeading = srcH;                // Edge.cpp:96
    double carryHeading = srcH;
    double tfinal = tg[0];
    bool dubErr = false;
    const double w = p.ribw;
    const double inc_d = p.inc_d;
    // bounds used by the obstacle culling: how far the vehicle / time advance over one 64-step chunk
    const double chunkTime = 64.0 * (p.inc_d / p.max_speed);
    const double chunkSpan = 64.0 * (p.inc_d / p.max_speed) * speed;

    if (!throwsRef) {
        for (int base = 0;; base += PP_WAVE) {
            const int k = base + lane;
            const double t = (k < p.ng) ? tg[k] : INFINITY;
            const double tFirst = pp_readlane(t, 0);
            if (!(tFirst < endTime)) { tfinal = tFirst; break; }      // `while (intermediate.time() < endTime)`
            const bool valid = t < endTime;

            // phase A: pose + static + dynamic obstacles for 64 steps
            double x = 0, y = 0, heading = 0;
            bool blk = false;
            int hits = 0;
            if (valid) {
                double dist = (t - wStart) * speed;                   // DubinsWrapper.cpp:36
                if (dist < 0 || dist > cv.length) dist = dist - 1e-5; // EDUBPARAM retry, :39-42
                if (dist < 0 || dist > cv.length) { dubErr = true; dist = fmin(fmax(dist, 0.0), cv.length); }
                double yaw;
#ifdef PP_ABL_NO_POSE
                x = srcX + dist * 1e-3; y = srcY; yaw = 1.0;
#else
                pp_curve_sample(cv, dist, x, y, yaw);
#endif
                heading = pp_heading_from_yaw(yaw);                   // :47
#ifndef PP_ABL_NO_GRID
                blk = pp_is_blocked(p.grid, x, y);                    // Edge.cpp:144
#endif
            }
#ifndef PP_ABL_NO_OBST
            if (p.n_obst > 0)                                         // :150-151
                hits = pp_obstacle_hits_chunk(p.obst, p.n_obst, x, y, t, valid, pp_readlane(x, 0), pp_readlane(y, 0), tFirst,
                                              chunkSpan, chunkTime);
#endif
            double prevHeading = __shfl_up(heading, 1, PP_WAVE);
            if (lane == 0) prevHeading = carryHeading;
            // Edge.cpp:159: cover only when coverage is allowed on this edge or the heading did not change since the last step
            const unsigned long long coverMask = cov ? ~0ull : __ballot(prevHeading == heading);

            const unsigned long long bm = __ballot(blk);
            const int fb = bm ? (__ffsll((long long)bm) - 1) : PP_WAVE;
            const int nvalid = __popcll(__ballot(valid));
            const int limit = fb < nvalid ? fb : nvalid;

            // phase B: coverage events among steps [0, limit)
            int lastEv = -1;
            bool runFailed = false, quietFailed = false;
#ifdef PP_ABL_NO_EVENTS
            nextEvent = 1 << 30;
#endif
            while (true) {
                const int j = __builtin_amdgcn_readfirstlane(nextEvent - base);
                if (j >= limit) break;
                const double tj = pp_readlane(t, j);
                if (!(tj < endTime)) break;
                const double xj = pp_readlane(x, j), yj = pp_readlane(y, j);
                double D;                                                                 // Edge.cpp:158-161
                int adv;
                nrib = pp_ribbons_event(rib, nrib, w, xj, yj, ((coverMask >> j) & 1ull) != 0ull, lds, D, adv);
                if (nrib > PP_WAVE) { nrib = PP_WAVE; flags |= PPGPU_F_RIBBON_OVF; }
#ifndef PP_NO_CORRIDOR_RUN
                if (adv >= 0 && j + 1 < limit && !runFailed) {
                    // this event only moved one piece's start: the following steps very likely do the same
                    double nsx, nsy;
                    const bool moveEnd = (adv & 0x100) != 0;
                    const int piece = adv & 0xff;
                    const int L = pp_corridor_run(rib, nrib, w, piece, moveEnd, x, y, (lane < limit) & (t < endTime), coverMask, j + 1, nsx, nsy);
                    runFailed = (L == 0);                  // do not keep paying for attempts that do not start
                    if (L > 0) {
                        if (lane == piece) {
                            if (moveEnd) { rib.ex = nsx; rib.ey = nsy; } else { rib.sx = nsx; rib.sy = nsy; }
                        }
                        lastEv = j + L;
                        nextEvent = base + j + L + 1;      // inside the corridor minDistanceFrom is 0: the next step is an event too
                        continue;
                    }
                }
#ifdef PP_QUIET_RUN   // opt-in: absorbs the non-mutating in-corridor events too, but costs ~18 VGPRs (one wave of occupancy)
                else if (adv == -2 && D == 0 && nrib > 0 && j + 1 < limit && !quietFailed) {
                    // inside a corridor, nothing changed: the following steps are very likely the same kind of event
                    const int L = pp_quiet_run(rib, nrib, w, x, y, (lane < limit) & (t < endTime), coverMask, j + 1);
                    quietFailed = (L == 0);
                    if (L > 0) {
                        lastEv = j + L;
                        nextEvent = base + j + L + 1;
                        continue;
                    }
                }
#endif
#endif
                if (nrib == 0) {                                                          // :162-170
                    if (cct == -1) cct = tj;
                    rdt = (int)tj;
                    endTime = fmin(endTime, cct + p.tmin);
                }
                lastEv = j;
#ifdef PP_DBG_EVENTS
                dbgEvents++;
#endif
                // steps until toCoverDistance <= increment again (:153-154): m subtractions
                int m = 0;
                if (D > inc_d) {
                    const double qd = D / inc_d;
                    if (qd > (double)(p.ng + 2)) {
                        m = p.ng + 1;                                  // beyond the grid: never again
                    } else {
                        const int m0 = (int)ceil(qd - 1.0);
                        const double r = fma(-(double)m0, inc_d, D);   // D - m0*inc, one rounding
                        const double margin = (double)m0 * D * 5e-16 + 1e-12;
                        if (m0 >= 1 && r > margin && r < inc_d - margin) {
                            m = m0;                                    // the running subtraction cannot differ
                        } else {
                            double tc = D;                             // too close to call: do it the long way
                            while (tc > inc_d && m <= p.ng) { tc -= inc_d; m++; }
                        }
                    }
                }
                nextEvent = base + j + m + 1;
            }

            const int cnt = __popcll(__ballot(valid && (t < endTime)));
            int nexec = cnt > lastEv + 1 ? cnt : lastEv + 1;
            nexec = nexec < limit ? nexec : limit;
            if (lane < nexec) hitsAcc += hits;

            if (fb < nvalid && nexec == fb) {           // reached the blocked step: `break` at :146
                infeasible = true;
                ix = pp_readlane(x, fb); iy = pp_readlane(y, fb); ih = pp_readlane(heading, fb);
                lastHeading = pp_readlane(prevHeading, fb);
                tfinal = pp_readlane(t, fb);
                steps = base + fb + 1;
                break;
            }
            if (nexec < PP_WAVE) {                      // loop condition failed inside this chunk
                ix = pp_readlane(x, nexec - 1); iy = pp_readlane(y, nexec - 1); ih = pp_readlane(heading, nexec - 1);
                lastHeading = ih;
                tfinal = pp_readlane(t, nexec);
                steps = base + nexec;
                break;
            }
            ix = pp_readlane(x, 63); iy = pp_readlane(y, 63); ih = pp_readlane(heading, 63);
            lastHeading = ih;
            carryHeading = ih;
            steps = base + PP_WAVE;
        }
    }
    if (__ballot(dubErr) != 0ull) flags |= PPGPU_F_DUBINS_ERR;

    // ---- phase C
    // end()->state().time() = endTime; wrapper.sample(end state)  (Edge.cpp:177-178)
    if (!throwsRef && !(wStart <= endTime && wEnd >= endTime)) throwsRef = true;  // DubinsWrapper::containsTime
    double endX = 0, endY = 0, endHeading = 0;
    if (!throwsRef) {
        double dist = (endTime - wStart) * speed;
        if (dist < 0 || dist > cv.length) dist = dist - 1e-5;
        if (dist < 0 || dist > cv.length) { flags |= PPGPU_F_DUBINS_ERR; dist = fmin(fmax(dist, 0.0), cv.length); }
        double yaw;
        pp_curve_sample(cv, dist, endX, endY, yaw);
        endHeading = pp_heading_from_yaw(yaw);
        // cover the last little bit (:182-191)
        if (cov || lastHeading == ih) {
            double Dunused;
            int advUnused;
            nrib = pp_ribbons_event(rib, nrib, w, ix, iy, true, lds, Dunused, advUnused);
            if (nrib > PP_WAVE) { nrib = PP_WAVE; flags |= PPGPU_F_RIBBON_OVF; }
        }
        if (nrib == 0) {
            if (cct == -1) cct = tfinal;
            rdt = (int)tfinal;
        }
    }
    const int hitsTotal = pp_wave_sum_i(hitsAcc);
    const double penalty = (double)hitsTotal * p.cpf;                             // :150-151 summed
    const double netTime = endTime - srcT;                                        // Edge::netTime
    double tc = fmax(netTime - ((nrib == 0) ? (endTime - (double)rdt) : 0), 0);  // :197
    if (startedDone) tc = 0;                                                      // :198
    const double trueCost = tc * p.tpf + penalty;                                 // :199
    const double g = srcG + trueCost;                                             // Vertex::setCurrentCost

    // h and f are filled in by pp_k_heuristic from the child ribbons written below
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
        const unsigned info = (unsigned)((dub.type < 0 ? 0 : dub.type) & 0xff) | ((unsigned)(nrib & 0xff) << 8) |
                              ((unsigned)(steps & 0xffff) << 16);
        double v;
        switch (lane) {
            case 0: v = __hiloint2double((int)info, (int)flags); break;   // {flags (low), info (high)}
            case 1: v = trueCost; break;
            case 2: v = penalty; break;
            case 3: v = approx; break;
            case 4: v = endX; break;
            case 5: v = endY; break;
            case 6: v = endHeading; break;
            case 7: v = speed; break;
            case 8: v = endTime; break;
            case 9: v = g; break;
            case 10: v = h; break;
            case 11: v = g + h; break;
            case 12: v = cct; break;
            case 13: v = dub.p0; break;
            case 14: v = dub.p1; break;
            default: v = dub.p2; break;
        }
        if (throwsRef && lane != 0) v = 0;
        if (lane < 16) reinterpret_cast<double*>(rec)[lane] = v;
    }
    if (!throwsRef) {
        if (nrib > p.stride && lane == 0) rec->flags = flags | PPGPU_F_RIBBON_OVF;   // after the record store above
        if (lane < nrib && lane < p.stride) {
            double* c = p.child + ((size_t)e * p.stride + lane) * 4;
            c[0] = rib.sx; c[1] = rib.sy; c[2] = rib.ex; c[3] = rib.ey;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Vertex::computeApproxToGo (Vertex.cpp:49-64) for every costed edge: h = heuristic(child pose,
// child ribbons) / maxSpeed, f = g + h, patched into the edge's record.  Its own kernel so that the
// sweep kernel's register budget is not set by the TSP enumeration.  One wavefront per edge.
#ifndef PP_H_MIN_WAVES
#define PP_H_MIN_WAVES 1
#endif
__global__ __launch_bounds__(PP_WPB * 64, PP_H_MIN_WAVES) void pp_k_heuristic(PPParams p) {
    __shared__ double lds_all[PP_WPB][PP_H_LDS];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = pp_lane();
    const long long e = (long long)blockIdx.x * PP_WPB + wave;
    if (e >= p.n_edges) return;
    double* pts = lds_all[wave];                 // x,y of the query point, then start/end of every child ribbon
    double* T = lds_all[wave] + PP_WAVE * 2;     // distance table of the TSP heuristics (<= 8 ribbons)
    double* KM = T + PP_H_PTS * (PP_H_PTS - 1);  // KM[p][i] = distance from point p to the nearer endpoint of ribbon i
    ppgpu_edge_result* rec = p.out + e;
    unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)rec->flags);
    if (flags & PPGPU_F_THROWS) return;
    int nrib = (int)((__builtin_amdgcn_readfirstlane((int)rec->info) >> 8) & 0xff);
    const double endX = rec->end_x, endY = rec->end_y, g = rec->g;
    double hdist = 0;
    if (nrib > 0 && nrib <= p.stride) {          // a truncated list (already flagged) carries no heuristic
        const bool tsp = p.heuristic != PPGPU_H_MAX_DISTANCE;
        if (tsp && nrib > PP_TSP_MAX) {
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
            if (!tsp) {
                hdist = pp_h_max_distance(pts, nrib, p.ribw);
            } else {
                const int npts = 2 * nrib + 1;
                const int ncol = npts - 1;
                for (int idx = lane; idx < npts * ncol; idx += PP_WAVE) {      // all distances, once
                    const int pp = idx / ncol, qq = 1 + (idx - pp * ncol);
                    T[pp * (PP_H_PTS - 1) + (qq - 1)] = pp_dist(pts[2 * pp], pts[2 * pp + 1], pts[2 * qq], pts[2 * qq + 1]);
                }
                pp_wave_lds_fence();
                for (int idx = lane; idx < npts * nrib; idx += PP_WAVE) {
                    const int pp = idx / nrib, ri = idx - pp * nrib;
                    KM[pp * PP_TSP_MAX + ri] = fmin(pp_h_T(T, pp, 1 + 2 * ri), pp_h_T(T, pp, 2 + 2 * ri));
                }
                pp_wave_lds_fence();
                if (p.heuristic == PPGPU_H_TSP_POINT_ALL) hdist = pp_h_tsp_point(T, KM, nrib, p.ribw, PP_TSP_MAX, false);
                else if (p.heuristic == PPGPU_H_TSP_POINT_K) hdist = pp_h_tsp_point(T, KM, nrib, p.ribw, p.tsp_k, true);
                else flags |= PPGPU_F_DUBINS_ERR;   // Dubins-TSP heuristics are rejected by ppgpu_set_config
            }
        }
    }
    const double h = hdist / p.max_speed * p.tpf;
    if (lane == 0) { rec->h = h; rec->f = g + h; rec->flags = flags; }
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

// k smallest (length, index) per (vertex, radius); one 256-thread workgroup each.  Round j finds
// the lexicographic successor of round j-1's winner, so no exclusion list is needed.
__global__ __launch_bounds__(256) void pp_k_select_nearest(const double* lengths, long long ns, int k, int* out_idx,
                                                           double* out_len) {
    __shared__ double sl[256];
    __shared__ long long si[256];
    const int vr = blockIdx.x;               // vertex * 2 + radius
    const int v = vr >> 1, r = vr & 1;
    const double* L = lengths + ((size_t)v * ns) * 2 + r;
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
        if (threadIdx.x == 0) { out_idx[(size_t)vr * k + j] = (int)prevI; out_len[(size_t)vr * k + j] = prevI >= 0 ? prevL : -1.0; }
        __syncthreads();
        if (prevI < 0) {                                                    // fewer than k candidates
            for (int jj = j + 1; jj < k; jj++) if (threadIdx.x == 0) { out_idx[(size_t)vr * k + jj] = -1; out_len[(size_t)vr * k + jj] = -1.0; }
            break;
        }
    }
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
