// pp_k_heuristic.h — Vertex::computeApproxToGo (Vertex.cpp:49-64): the wave-per-edge heuristic kernels, the workgroup-per-edge kernel for
// lists of 7-8 ribbons and the four-lanes-per-edge kernel for short lists.  Included by pp_kernels.h.
#pragma once
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
// TspPointRobotNoSplitKRibbons on child lists of 9..12 ribbons (rare: a vertex whose ribbons were split many times).  A whole
// WORKGROUP of sixteen wavefronts per such edge (round 4): every wave builds the same tables and takes every sixteenth pass of 64
// prefixes, the smallest of their minima is the minimum (exact, as in pp_k_heuristic_listed).  One wave per edge took a
// millisecond per edge — up to two million prefixes — and in a planner round trip, where such an edge comes alone, that
// millisecond was the round trip's: the deadline guard's largest under-predictions.
#define PP_BIG_GRID 1024
#define PP_BIG_WPB 16
__global__ __launch_bounds__(PP_BIG_WPB * 64) void pp_k_heuristic_big(PPParams p) {
    __shared__ double lds_all[PP_BIG_WPB][PPTsp<PP_TSP_MAX_BIG>::LDS];
    __shared__ double s_part[PP_BIG_WPB];
    if (pp_const_i32(p.need_big)[0] == 0) return;            // almost always: no child list beyond 8 ribbons in this launch
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = pp_lane();
    double* pts = lds_all[wave];
    const bool tsp = p.heuristic != PPGPU_H_MAX_DISTANCE;
    // a modest grid whose workgroups stride over the edges (ppgpu.hip: at most PP_BIG_GRID workgroups)
    for (long long e = (long long)blockIdx.x; e < p.n_edges; e += (long long)gridDim.x) {
        ppgpu_edge_result* rec = p.out + e;
        const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)rec->flags);
        if (flags & PPGPU_F_THROWS) continue;
        const int nrib = (int)((__builtin_amdgcn_readfirstlane((int)rec->info) >> 8) & 0xff);
        if (!(tsp && nrib <= p.stride && pp_tsp_big_ok(p.heuristic, p.tsp_k, nrib))) continue;      // (pp_heuristic_edge's own gate for this pass)
        const double endX = rec->end_x, endY = rec->end_y, g = rec->g;
        if (lane == 0) { pts[0] = endX; pts[1] = endY; }
        if (lane < nrib) {
            const double* c = p.child + ((size_t)e * p.stride + lane) * 4;
            pts[2 * (1 + 2 * lane)] = c[0]; pts[2 * (1 + 2 * lane) + 1] = c[1];
            pts[2 * (2 + 2 * lane)] = c[2]; pts[2 * (2 + 2 * lane) + 1] = c[3];
        }
        pp_wave_lds_fence();
        const double part = pp_h_point_from_pts<PP_TSP_MAX_BIG>(p.heuristic, p.tsp_k, p.ribw, pts, nrib, (unsigned)wave, (unsigned)PP_BIG_WPB);
        if (lane == 0) s_part[wave] = part;
        __syncthreads();
        if (threadIdx.x == 0) {
            double hdist = s_part[0];
            for (int w = 1; w < PP_BIG_WPB; w++) hdist = fmin(hdist, s_part[w]);
            const double h = hdist / p.max_speed * p.tpf;
            rec->h = h; rec->f = g + h;
        }
        __syncthreads();
    }
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
    __builtin_amdgcn_s_setprio(3);           // few, long waves sharing their SIMDs with pp_k_heuristic_lanes' many short ones: issue first
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
    if constexpr (REM >= PP_HL_SHARE_MIN_REM) gbest = pp_quad_min(gbest);
    return __ballot(!(sf + rb > gbest + PP_HL_MARGIN)) == 0ull;
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
                const double nrb = rb - c.CB[rid];
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
    for (int r = 0; r < N; r++) rbAll += c.CB[r];
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
        if constexpr (N > 1) v = pp_lane_tsp<N - 1>(c, nsf, pp_lane_tsp_drop(srt, cc), 2 * rid + 1 - dir, rbAll - c.CB[rid], gbest);
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
    if (have)
        for (int r = sub; r < n; r += PP_HL_SPLIT) {              // CB[r] = len[r] - 2w + the shortest way into ribbon r from another ribbon
            double m = (n > 1) ? PP_DBL_MAX : 0.0;
            for (int q = 0; q < 2 * n; q++)
                if ((q >> 1) != r) m = fmin(m, fmin(T[pp_tri(q, 2 * r)], T[pp_tri(q, 2 * r + 1)]));
            CB[r] = T[r * (2 * r + 1) + 2 * r] - 2 * p.ribw + m;
        }
    __syncthreads();
    if (have) {
        PPLaneTsp c;
        c.T = T; c.CB = CB; c.twoW = 2 * p.ribw; c.cnt = p.need_big + 14;
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
