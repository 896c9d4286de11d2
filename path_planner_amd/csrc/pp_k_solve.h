// pp_k_solve.h — phase 0 of an edge: pp_k_solve_edges (Vertex::connect + Edge::computeApproxCost, one lane per edge).  Included by pp_kernels.h.
#pragma once
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
    O->p0 = cv.p0; O->p1 = cv.p1; O->p2 = cv.p2; O->hi1 = cv.p0 + cv.p1;
    O->approx = S.approx; O->wStart = S.wStart; O->wEnd = S.wEnd; O->speed = S.speed;
    O->type = S.type; O->vi = S.vi; O->cbits = S.cbits; O->sflags = S.sflags;
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
    O->omask = omask;
}
