// pp_k_cover.h — the coverage state machine of Edge::computeTrueCost (Edge.cpp:153-171) and the rest of the function (:177-205):
// pp_k_approach_events (lane per edge), pp_k_cover_sweep (wave per edge), pp_k_cover_finish (lane per edge).  Included by pp_kernels.h.
#pragma once
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
__device__ __forceinline__ void pp_lane_ribbon_contains(double sx, double sy, double ex, double ey, double x, double y, double w, bool& inside, bool& strict,
                                                        double& px, double& py) {
    const double T = PP_RIBBON_TOL;
    const double dxr = ex - sx, dyr = ey - sy;
    const double sqL = dxr * dxr + dyr * dyr;
    const double dot = (x - sx) * dxr + (y - sy) * dyr;
    px = dxr * dot / sqL + sx;                               // Ribbon::getProjection (Ribbon.cpp:72-78)
    py = dyr * dot / sqL + sy;
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
// PPParams::track_far[e].y = (last event before the hand-over) + 1 in the low 20 bits, and, when the approach lane has already split the
// one ribbon the vehicle entered (round 4): PP_FAR_SPLIT, which piece the corridor run that follows moves (6 bits) and which of its ends
#define PP_FAR_LAST_MASK 0xfffff
#define PP_FAR_SPLIT (1 << 30)
#define PP_FAR_MOVE_END (1 << 29)
#define PP_FAR_PIECE_SHIFT 21
__device__ __forceinline__ void pp_lane_pose(const PPEdgeSetupBody* S, double t, double wStart, double speed, double length, double rho, double rho_inv,
                                             double qx, double qy, double hi0, double hi1, double p1, int word, double& x, double& y, double& uth, bool& err) {
    double dist = (t - wStart) * speed;                                     // DubinsWrapper.cpp:36
    if (dist < 0 || dist > length) dist = dist - 1e-5;                      // EDUBPARAM retry, :39-42
    if (dist < 0 || dist > length) { err = true; dist = fmin(fmax(dist, 0.0), length); }
    const double tprime = (rho_inv != 0.0) ? dist * rho_inv : dist / rho;
    double ux, uy;
    pp_setup_seg_pose(S, pp_seg_of(tprime, hi0, hi1), tprime, hi0, p1, word, ux, uy, uth);      // (hi0 = the record's p0)
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
    const double hi0 = S->p0, hi1 = S->hi1, segP1 = S->p1;
    const int dubWord = S->type;
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
    if (lastIdx >= 0) pp_lane_pose(S, tg[lastIdx], wStart, speed, length, rho, rho_inv, qx, qy, hi0, hi1, segP1, dubWord, ix, iy, uth, ignored);
    double endX, endY;
    pp_lane_pose(S, endTime, wStart, speed, length, rho, rho_inv, qx, qy, hi0, hi1, segP1, dubWord, endX, endY, uth, perr);
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
#ifndef PP_APPROACH_THREADS
#define PP_APPROACH_THREADS 256
#endif
#ifndef PP_LANE_NEAR_MAX
#define PP_LANE_NEAR_MAX 8
#endif
__global__ __launch_bounds__(PP_APPROACH_THREADS, PP_APPROACH_MIN_WAVES) void pp_k_approach_events(PPParams p) {
    __shared__ double s_rec[PP_APPROACH_THREADS * PP_REC_STRIDE];      // quiet edges' records, transposed through LDS (34 KB: four workgroups per CU still fit)
    const long long e0 = (long long)blockIdx.x * PP_APPROACH_THREADS;
    const long long e = e0 + threadIdx.x;
    const bool valid = e < p.n_edges;
    const PPEdgeSetupBody* S = p.setup + p.ws_base + (valid ? e : 0);
    int2 out; out.x = 0; out.y = 0;                 // {first event of the wave, (last event before it) + 1 | PP_FAR_* bits}
    int splitInfo = 0;
#ifdef PP_DBG_PHASES               // tools/cover_phases.py: the lane's cycles before / inside / after its event loop, and its events
    const long long apT0 = (long long)__builtin_readcyclecounter();
    long long apT1 = apT0, apT2 = apT0;
    int apEvents = 0;
#endif
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
            const double hi0 = S->p0, hi1 = S->hi1, segP1 = S->p1;
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
            // Round 3: an event within reach of ONE ribbon is no longer handed over at once.  The lane takes it exactly (does that
            // ribbon contain the point, strictly or not: the reference's own expressions) and goes on while it changes nothing — the
            // vehicle passes near a ribbon, or travels inside a corridor before it reaches the strict one, or may not cover while it
            // turns: what the wave used to start with (one event and one quiet run, two of a slow edge's nine operations).  At most
            // PP_LANE_NEAR_MAX such events per edge (a crawl along a corridor is the wave's, 64 steps at a time); within reach of
            // two ribbons at once the wave takes over as before.
            int nearBudget = PP_LANE_NEAR_MAX;
            const bool covEdge = (S->cbits & PPGPU_EDGE_COVERAGE) != 0;
            // a piece short enough to be erased makes every event a real one (Ribbon::covered is checked wherever the vehicle is)
#ifdef PP_DBG_PHASES
            apT1 = (long long)__builtin_readcyclecounter();
#endif
            while (!tiny) {
                if (k >= limit) break;
                const double t = tg[k];
                if (!(t < endTime0)) break;
#ifdef PP_DBG_PHASES
                apEvents++;
#endif
                double dist = (t - wStart) * speed;                                     // DubinsWrapper.cpp:36
                if (dist < 0 || dist > length) dist = dist - 1e-5;                      // EDUBPARAM retry, :39-42
                if (dist < 0 || dist > length) { handOver = true; break; }              // the wavefront's code flags the error
                const double tprime = (rho_inv != 0.0) ? dist * rho_inv : dist / rho;
                double ux, uy, uth;
                pp_setup_seg_pose(S, pp_seg_of(tprime, hi0, hi1), tprime, hi0, segP1, dubType, ux, uy, uth);
                const double x = ux * rho + qx, y = uy * rho + qy;
                bool inBox = false;
                int boxCount = 0, boxIdx = 0;
                double q = PP_DBL_MAX;
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
                    if (boxCount == 1 && nearBudget-- > 0) {
                        bool inside, strict;
                        double px, py;
                        const double bsx = rp[4 * boxIdx], bsy = rp[4 * boxIdx + 1], bex = rp[4 * boxIdx + 2], bey = rp[4 * boxIdx + 3];
                        pp_lane_ribbon_contains(bsx, bsy, bex, bey, x, y, w, inside, strict, px, py);
                        // Edge.cpp:159: cover() runs when coverage is allowed on this edge or the heading did not change since the last step
                        const bool coverOn = covEdge || (((p.track_eq[(size_t)(p.ws_base + e) * p.nch + (k >> 6)] >> (k & 63)) & 1ull) != 0ull);
                        if (!(strict && coverOn)) {
                            if (inside) D = 0;                                          // RibbonManager::minDistanceFrom: contained
                            lastEv = k;
                            k = k + pp_event_stride(D, p.inc_d, p.inv_inc_d, p.ng) + 1;
                            continue;
                        }
                        // Round 4: the event that SPLITS the one ribbon in reach is the lane's too, when both halves stay (the vehicle
                        // entered the strict corridor somewhere along the ribbon): RibbonManager::cover over the list in order changes
                        // this ribbon only — nothing else is in reach, nothing is short enough to be erased — into [start, projection]
                        // and [projection, end] (Ribbon::split, Ribbon.cpp:9-17; covered(strict), :23-25).  The new list goes into the
                        // edge's child-ribbon slot and the wave starts one step later, inside the corridor run that nearly always
                        // follows (the half the vehicle travels into, by the direction of travel: the run's own checks decide, a wrong
                        // guess costs it one attempt) — and in long-run mode from its first window.  The wave used to spend a window
                        // and its most expensive kind of event here, then a 64-step run before the first long one.
                        const double thr = minLength0 * minLength0 / (2.0 * 2.0);
                        const bool keepF = !(pp_sq_len(bsx, bsy, px, py) < thr), keepR = !(pp_sq_len(px, py, bex, bey) < thr);
                        if (keepF && keepR && p.lane_split && nrib + 1 <= p.stride && nrib + 1 <= PP_WAVE) {
                            double* c = p.child + (size_t)pp_edge_position(p, p.e_base + e) * p.stride * 4;
                            for (int i = 0; i < boxIdx; i++) { c[4 * i] = rp[4 * i]; c[4 * i + 1] = rp[4 * i + 1]; c[4 * i + 2] = rp[4 * i + 2]; c[4 * i + 3] = rp[4 * i + 3]; }
                            c[4 * boxIdx] = bsx; c[4 * boxIdx + 1] = bsy; c[4 * boxIdx + 2] = px; c[4 * boxIdx + 3] = py;
                            c[4 * boxIdx + 4] = px; c[4 * boxIdx + 5] = py; c[4 * boxIdx + 6] = bex; c[4 * boxIdx + 7] = bey;
                            for (int i = boxIdx + 1; i < nrib; i++) { c[4 * i + 4] = rp[4 * i]; c[4 * i + 5] = rp[4 * i + 1]; c[4 * i + 6] = rp[4 * i + 2]; c[4 * i + 7] = rp[4 * i + 3]; }
                            // which half does the vehicle travel into?  The direction of travel is the pose's yaw.
                            double sn, cs;
                            pp_sincos_bounded(uth, &sn, &cs);
                            const bool towardsEnd = (cs * (bex - bsx) + sn * (bey - bsy)) > 0.0;
                            splitInfo = PP_FAR_SPLIT | (towardsEnd ? 0 : PP_FAR_MOVE_END) | ((towardsEnd ? boxIdx + 1 : boxIdx) << PP_FAR_PIECE_SHIFT);
                            lastEv = k;
                            k = k + 1;
                            handOver = true; break;
                        }
                    }
                    handOver = true; break;                                             // a ribbon changes here (or two are in reach, or the budget is spent): the wavefront takes over
                }
                lastEv = k;
                k = k + pp_event_stride(D, p.inc_d, p.inv_inc_d, p.ng) + 1;
            }
            out.x = k; out.y = (lastEv + 1) | splitInfo;
#ifdef PP_DBG_PHASES
            apT2 = (long long)__builtin_readcyclecounter();
#endif
            if (!handOver && k >= limit && p.quiet_finish &&
                pp_finish_quiet_edge(p, S, V, p.ws_base + e, pp_edge_position(p, p.e_base + e), limit, lastEv, oneVertex ? rpU : rp, tg, s_rec + (size_t)threadIdx.x * PP_REC_STRIDE, oneVertex))
                out.x = PP_FAR_DONE;
        }
    }
    if (valid) p.track_far[p.ws_base + e] = out;
#ifdef PP_DBG_PHASES
    {
        const long long apT3 = (long long)__builtin_readcyclecounter();
        const double wPro = pp_wave_max((double)(apT1 - apT0)), wLoop = pp_wave_max((double)(apT2 - apT1)), wFin = pp_wave_max((double)(apT3 - apT2));
        const double wEv = pp_wave_max((double)apEvents);
        if (valid && out.x == PP_FAR_DONE) {
            double* st = s_rec + (size_t)threadIdx.x * PP_REC_STRIDE;
            st[12] = wPro; st[13] = wLoop * 4294967296.0 + (double)(apT2 - apT1); st[14] = -1.0 - wFin; st[15] = wEv * 1e6 + (double)apEvents;
        }
    }
#endif
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

    if (p.track_far && pp_const_i32(&p.track_far[e].x)[0] == PP_FAR_DONE) return;   // a quiet edge: pp_k_approach_events finished it
    // ---- phase 0 was done by pp_k_solve_edges: everything here is wave-uniform and arrives through scalar loads
    const PPEdgeSetup* S = p.setup + e;
    unsigned flags = 0;
    ppgpu_edge_result* rec = p.out + eg;
    const unsigned sflags = (unsigned)PP_SI32(sflags);
#if !defined(PP_DBG_COUNTS)
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

#ifdef PP_DBG_COUNTS
    int dbgWindows = 0, dbgCorr = 0, dbgQuiet = 0, dbgGeneric = 0, dbgCorrLen = 0, dbgQuietLen = 0, dbgFar = 0, dbgNoChange = 0, dbgInPlace = 0, dbgRestFar = 0;
#define PP_CNT(x) x
#else
#define PP_CNT(x)
#endif
#ifdef PP_DBG_PHASES               // (with PP_DBG_COUNTS) shader cycles per part of the event loop instead of the counts, tools/cover_phases.py
    long long phWin = 0, phCont = 0, phGen = 0, phRun = 0, phLoop = 0, phT = 0;
#define PP_PH_START() (phT = (long long)__builtin_readcyclecounter())
#define PP_PH_END(acc) (acc += (long long)__builtin_readcyclecounter() - phT)
#else
#define PP_PH_START()
#define PP_PH_END(acc)
#endif
#ifdef PP_DBG_TRACE
#define PP_TRACE(...) do { if (eg == (long long)(PP_DBG_TRACE) && lane == 0) printf(__VA_ARGS__); } while (0)
#else
#define PP_TRACE(...)
#endif
    int rdt = -1;                       // `auto ribbonsDoneTime = -1;` is an int (Edge.cpp:92)
    int nextEvent = 0;                  // toCoverDistance starts at 0: step 0 is an event
    int lastEv = -1;
    int handCont = 0, handPiece = 0, handCount = 0;   // (handCount: entries of the child slot the lane's split list occupies)
    bool handMoveEnd = false;
    if (p.track_far) {                  // the approach was walked by pp_k_approach_events: start where it handed over
        nextEvent = pp_const_i32(&p.track_far[e].x)[0];
        const int fy = pp_const_i32(&p.track_far[e].y)[0];
        lastEv = (fy & PP_FAR_LAST_MASK) - 1;
        if (fy & PP_FAR_SPLIT) {
            // the lane split the ribbon the vehicle entered: the list as it stands is in the edge's child slot, and the wave begins inside
            // the corridor run (long-run mode allowed at once)
            nrib = nrib + 1;
            rib = PPRibbon{0, 0, 0, 0};
            if (lane < nrib) {
                const double* rp = p.child + ((size_t)eg * p.stride + lane) * 4;
                rib.sx = rp[0]; rib.sy = rp[1]; rib.ex = rp[2]; rib.ey = rp[3];
            }
            handCont = 1 | 4; handPiece = (fy >> PP_FAR_PIECE_SHIFT) & 0x3f; handMoveEnd = (fy & PP_FAR_MOVE_END) != 0;
            handCount = nrib;
        }
    }
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
    int longStride = 1;
    {
        const double stepLen = PP_STEP_LEN();
        const double sd = PP_LONG_REACH / stepLen;
        longStride = sd >= (double)PP_LONG_MAX_STRIDE ? PP_LONG_MAX_STRIDE : (sd > 1.0 ? (int)sd : 1);
        // the half that vanishes between two samples must be shorter than the minimum length; the curve must not turn much between them
        if (!((double)longStride * stepLen + 1e-6 < 0.5 * w) || !(((double)longStride * stepLen + 1e-6) / PP_SF64(rho) < 0.5)) longStride = 1;
    }

    // ---- phase B: coverage events among steps [0, limit)
    if (!throwsRef) {
#ifdef PP_DBG_PHASES
        const long long phLoop0 = (long long)__builtin_readcyclecounter();
#endif
        bool ended = false;
        int cont = handCont, contPiece = handPiece;   // 1 / 2: the last window ended inside a corridor / quiet run (of piece contPiece); + 4: it was
                                            // one run from end to end (a long run is worth trying)
        bool contMoveEnd = handMoveEnd;
        while (!ended) {
            if (nextEvent >= limit) break;
            // a window of 64 steps of the track starting AT the next event, one step per lane (stretches without events are
            // never loaded)
            const int base = nextEvent;
            // ... or, after a window that was one run from end to end, 64 SAMPLES longStride steps apart (a long run)
            const int stride = ((cont & 4) != 0 && longStride > 1 && base + 2 * longStride < limit) ? longStride : 1;
            PP_TRACE("[wave] window at %d stride %d (limit %d, lastEv %d, nrib %d)\n", base, stride, limit, lastEv, nrib);
            PP_CNT(dbgWindows++);
            PP_PH_START();
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
            PP_PH_END(phWin);
            while (true) {
                const int j = nextEvent - base;
                if (j >= climit) break;
                const double tj = pp_readlane(t, j);
                if (!(tj < endTime)) { ended = true; break; }         // `while (intermediate.time() < endTime)`
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
                    PP_PH_START();
                    const double ell = (stride > 1) ? ((double)stride * PP_STEP_LEN() + 1e-6) : 0.0;
                    const double span = (stride > 1) ? 64.0 * ell : runSpan;
                    if (kind == 1) L = pp_corridor_run(rib, nrib, w, contPiece, contMoveEnd, q.x, q.y, stepOk, coverMask, 0, span, nsx, nsy, ell, ell / PP_SF64(rho));
                    else L = pp_quiet_run(rib, nrib, w, q.x, q.y, stepOk, coverAny, 0, span, ell, ell / PP_SF64(rho));
                    PP_TRACE("[wave]   continued run (kind %d, stride %d) from %d: L %d\n", kind, stride, base, L);
                    PP_PH_END(phCont);
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
                const double xj = pp_readlane(q.x, j), yj = pp_readlane(q.y, j);
                double D;                                             // Edge.cpp:158-161
                int adv;
                PP_CNT(dbgGeneric++);
                PP_PH_START();
                nrib = pp_ribbons_event(rib, nrib, w, xj, yj, ((coverMask >> j) & 1ull) != 0ull, lds, D, adv);
                PP_PH_END(phGen);
                PP_CNT(if (adv == -3) dbgFar++; else if (adv == -2) dbgNoChange++; else if (adv >= 0) dbgInPlace++);
                PP_TRACE("[wave]   event %d: adv %d D %.17g nrib %d cover %d x %.17g y %.17g\n", base + j, adv, D, nrib, (int)((coverMask >> j) & 1ull), xj, yj);
                if (nrib > PP_WAVE) { nrib = PP_WAVE; flags |= PPGPU_F_RIBBON_OVF | PPGPU_F_RIBBON_LOST; }

                bool guessed = false;
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
                if (adv >= 0 && j + 1 < climit && !runFailed) {
                    // this event only moved one piece's endpoint: the following steps very likely do the same
                    double nsx, nsy;
                    const bool moveEnd = (adv & 0x100) != 0;
                    const int piece = adv & 0xff;
                    PP_PH_START();
                    const int L = pp_corridor_run(rib, nrib, w, piece, moveEnd, q.x, q.y, (lane < climit) & (t < endTime), coverMask, j + 1, runSpan, nsx, nsy);
                    PP_PH_END(phRun);
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
                else if (adv == -2 && D == 0 && nrib > 0 && j + 1 < climit && !quietFailed) {
                    // inside a corridor, nothing changed: the following steps are very likely the same kind of event
                    PP_PH_START();
                    const int L = pp_quiet_run(rib, nrib, w, q.x, q.y, (lane < climit) & (t < endTime), coverMask, j + 1, runSpan);
                    PP_PH_END(phRun);
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
                if (nrib == 0) {                                      // :162-170
                    if (cct == -1) cct = tj;
                    rdt = (int)tj;
                    endTime = fmin(endTime, cct + p.tmin);
                }
                lastEv = base + j;
                // steps until toCoverDistance <= increment again (:153-154): m subtractions
                const int m = pp_event_stride(D, inc_d, p.inv_inc_d, p.ng);
                nextEvent = base + j + m + 1;
            }
        }
#ifdef PP_DBG_PHASES
        phLoop = (long long)__builtin_readcyclecounter() - phLoop0;
#endif
    }

    // ---- the rest is scalar work per edge: pp_k_cover_finish does it with one lane per edge, from what this wave knows now
    if (laneFinish && !throwsRef && (wStart <= endTime && wEnd >= endTime) && nrib <= PP_FINISH_MAX && nrib <= p.stride) {
        if (lane == 0) {
            PPCoverState* st = p.cover_state + e;
            st->cct = cct; st->endTime = endTime; st->lastEv = lastEv; st->rdt = rdt; st->flags = flags | (infeasible ? PPGPU_F_INFEASIBLE : 0u);
            st->nrib = nrib;                                          // (same lane, program order: after the -1 above)
        }
        if (lane < nrib || lane < handCount) {        // (entries of the lane's split list beyond what is left of it: back to zero)
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
            PPSeg cs = PPSeg{0, 0, 0, 0, 0, INFINITY, -INFINITY, 0, 0, 1};
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
#if defined(PP_DBG_PHASES)
            case 12: v = (double)phLoop; break;
            case 13: v = (double)phWin * 4294967296.0 + (double)phGen; break;
            case 14: v = (double)phCont; break;
            default: v = (double)phRun; break;
#elif defined(PP_DBG_COUNTS)
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
        if ((lane < nrib || lane < handCount) && lane < p.stride) {     // (lanes beyond nrib hold zeros: what is left of the lane's split list goes)
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
            if (lane == 0) { rec->h = hh; rec->f = g + hh; if (flags != flags0) rec->flags = flags | ((nrib > p.stride) ? PPGPU_F_RIBBON_OVF : 0u); }
        }
    }
#endif

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
    const double hi0 = S->p0, hi1 = S->hi1, segP1 = S->p1;
    const int dubWord = S->type;
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
    if (lastIdx >= 0) pp_lane_pose(S, tg[lastIdx], wStart, speed, length, rho, rho_inv, qx, qy, hi0, hi1, segP1, dubWord, ix, iy, uth, ignored);
    double endX, endY;
    pp_lane_pose(S, endTime, wStart, speed, length, rho, rho_inv, qx, qy, hi0, hi1, segP1, dubWord, endX, endY, uth, perr);
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
