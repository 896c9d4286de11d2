// pp_k_expand.h — SamplingBasedPlanner::expand on the device (SamplingBasedPlanner.cpp:82-149): Dubins lengths to every sample, the k nearest
// per radius in the reference's push order, the edge list of a round trip.  Included by pp_kernels.h.
#pragma once
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
// Round 4: Dubins lengths only for the samples that can matter.  The k winners of a (vertex, radius) are not longer than ANY k
// lengths one cares to compute, and a sample's Euclidean distance never exceeds its Dubins length — so a PROBE of the first
// PP_PROBE samples (the generator draws uniformly in the box, every prefix is spread like the whole) gives a bound Ub, and only the
// samples within Ub of the vertex — about a seventh of the set at the planner's sample counts — get their two lengths.  Round 3
// solved both Dubins problems for every (vertex, sample) pair: 97 us of every planner round trip for lengths 85 % of which nothing
// ever looked at.  Same winners, same push order: any upper bound of the k-th smallest length serves (pp_k_expand_bound then
// tightens it from the near samples' own block minima as before).
//   pp_k_expand_probe   thread per (vertex, probe sample): its lengths; per radius the minimum of every 32 consecutive probe samples
//   pp_k_expand_near    thread per (vertex, sample): Ub = the k-th smallest of the vertex's 32 group minima per radius (k different
//                       samples are not longer than it; +inf when fewer than k groups hold a valid sample, or k > 32: then every
//                       valid sample is "near" — the exhaustive path, as before); farther than the increment
//                       (SamplingBasedPlanner.cpp:111) and within the larger of the two bounds -> the vertex's near list (one
//                       atomic per workgroup; order arbitrary)
//   pp_k_near_lengths   thread per (vertex, near slot): both Dubins lengths, per 256 slots the smallest of each radius
#define PP_PROBE 1024
#define PP_PROBE_GROUPS (PP_PROBE / 32)      // minima of 32 probe samples each: the k-th smallest of them bounds the k-th smallest length
// thread per (vertex, probe sample): (PP_PROBE / 256, nv) workgroups
__global__ __launch_bounds__(256) void pp_k_expand_probe(const ppgpu_vertex* verts, const double* sx, const double* sy, const double* sh, long long ns,
                                                         double rho, double rho_cov, double inc_d, int two_radii, double* probe_min, int* near_count) {
    const int v = blockIdx.y;
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    const ppgpu_vertex* V = verts + v;
    const double ax = V->x, ay = V->y, ayaw = pp_yaw(V->heading);
    double b0 = INFINITY, b1 = INFINITY;
    if (s < ns) {
        const double bx = sx[s], by = sy[s], byaw = pp_yaw(sh[s]);
        if (sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by)) > inc_d) {   // State::distanceTo, SamplingBasedPlanner.cpp:111
            PPDubins d;
            pp_dubins_shortest(ax, ay, ayaw, bx, by, byaw, rho, d);
            b0 = pp_dubins_length(d, rho);
            if (two_radii) {
                pp_dubins_shortest(ax, ay, ayaw, bx, by, byaw, rho_cov, d);
                b1 = pp_dubins_length(d, rho_cov);
            }
        }
    }
    // the minimum of each half wavefront (32 consecutive probe samples)
    for (int o = 16; o > 0; o >>= 1) {
        b0 = fmin(b0, __shfl_xor(b0, o, PP_WAVE));
        b1 = fmin(b1, __shfl_xor(b1, o, PP_WAVE));
    }
    if ((threadIdx.x & 31) == 0) {
        const int g = (int)(blockIdx.x * 8 + (threadIdx.x >> 5));
        probe_min[((size_t)v * 2 + 0) * PP_PROBE_GROUPS + g] = b0;
        probe_min[((size_t)v * 2 + 1) * PP_PROBE_GROUPS + g] = two_radii ? b1 : b0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) near_count[v] = 0;
}
// the k-th smallest of a (vertex, radius)'s PP_PROBE_GROUPS group minima (ties ranked by group): +inf when fewer than k are finite.
// Every lane gets the value (lanes 0 .. PP_PROBE_GROUPS - 1 hold one group each).
__device__ __forceinline__ double pp_probe_bound(const double* probe_min, int v, int r, int k) {
    const int lane = pp_lane();
    const double x = (lane < PP_PROBE_GROUPS) ? probe_min[((size_t)v * 2 + r) * PP_PROBE_GROUPS + lane] : INFINITY;
    int rank = 0;
    for (int i = 0; i < PP_PROBE_GROUPS; i++) {
        const double y = pp_readlane(x, i);
        rank += ((y < x) | ((y == x) & (i < lane))) ? 1 : 0;
    }
    const unsigned long long hit = __ballot((x < INFINITY) & (rank == k - 1));
    return hit ? pp_readlane(x, __ffsll((long long)hit) - 1) : INFINITY;
}
#define PP_NEAR_PER 8            // samples per thread of pp_k_expand_near (a workgroup ranks the probe minima once for 2 048 samples)
__global__ __launch_bounds__(256) void pp_k_expand_near(const ppgpu_vertex* verts, const double* sx, const double* sy, long long ns, double inc_d,
                                                        const double* probe_min, int k, int* near_idx, int* near_count) {
    __shared__ int wcount[PP_NEAR_PER][4], wbase[PP_NEAR_PER][4];
    __shared__ double s_Ub;
    const int v = blockIdx.y, w = (int)(threadIdx.x >> 6), lane = pp_lane();
    if (w == 0) {
        const double u = (k <= PP_PROBE_GROUPS) ? fmax(pp_probe_bound(probe_min, v, 0, k), pp_probe_bound(probe_min, v, 1, k)) : INFINITY;
        if (lane == 0) s_Ub = u;
    }
    __syncthreads();
    const double Ub = s_Ub;
    const double vx = verts[v].x, vy = verts[v].y;
    const long long s0 = (long long)blockIdx.x * (256 * PP_NEAR_PER) + threadIdx.x;
    bool take[PP_NEAR_PER];
    unsigned long long m[PP_NEAR_PER];
#pragma unroll
    for (int j = 0; j < PP_NEAR_PER; j++) {
        const long long s = s0 + (long long)j * 256;
        take[j] = false;
        if (s < ns) {
            const double d = sqrt((vx - sx[s]) * (vx - sx[s]) + (vy - sy[s]) * (vy - sy[s]));
            take[j] = (d > inc_d) && !(d > Ub * (1.0 + 1e-9));
        }
        m[j] = __ballot(take[j]);
        if (lane == 0) wcount[j][w] = __popcll(m[j]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int j = 0; j < PP_NEAR_PER; j++) for (int i = 0; i < 4; i++) tot += wcount[j][i];
        int base = tot ? atomicAdd(&near_count[v], tot) : 0;
        for (int j = 0; j < PP_NEAR_PER; j++) for (int i = 0; i < 4; i++) { wbase[j][i] = base; base += wcount[j][i]; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PP_NEAR_PER; j++)
        if (take[j]) near_idx[(size_t)v * (size_t)ns + (size_t)(wbase[j][w] + __popcll(m[j] & ((1ull << lane) - 1ull)))] = (int)(s0 + (long long)j * 256);
}
#define PP_NEAR_GROUP 32
__global__ __launch_bounds__(256) void pp_k_near_lengths(const ppgpu_vertex* verts, const double* sx, const double* sy, const double* sh, long long ns,
                                                         const int* near_idx, const int* near_count, double rho, double rho_cov, int two_radii,
                                                         double* out, double* blockmin, int* blockcnt) {
    const int v = blockIdx.y;
    const int n = near_count[v];
    const long long slot = (long long)blockIdx.x * 256 + threadIdx.x;
    if ((long long)blockIdx.x * 256 >= (long long)n) return;            // (whole workgroups beyond the vertex's list)
    double l0 = -1, l1 = -1;
    if (slot < n) {
        const ppgpu_vertex* V = verts + v;
        const double ax = V->x, ay = V->y, ayaw = pp_yaw(V->heading);
        const int s = near_idx[(size_t)v * (size_t)ns + (size_t)slot];
        const double bx = sx[s], by = sy[s], byaw = pp_yaw(sh[s]);
        PPDubins d;
        pp_dubins_shortest(ax, ay, ayaw, bx, by, byaw, rho, d);
        l0 = pp_dubins_length(d, rho);
        pp_dubins_shortest(ax, ay, ayaw, bx, by, byaw, rho_cov, d);
        l1 = pp_dubins_length(d, rho_cov);
        double2 o; o.x = l0; o.y = l1;
        reinterpret_cast<double2*>(out)[(size_t)v * (size_t)ns + (size_t)slot] = o;
    }
    (void)two_radii;
    // the smallest length of every PP_NEAR_GROUP consecutive slots (a half wavefront): what pp_k_expand_bound ranks.  (Groups of 256,
    // as for the whole sample set in round 3, leave a near list of a thousand samples with fewer groups than k: no bound at all.)
    double a0 = l0 >= 0 ? l0 : INFINITY, a1 = l1 >= 0 ? l1 : INFINITY;
    for (int o = PP_NEAR_GROUP / 2; o > 0; o >>= 1) {
        a0 = fmin(a0, __shfl_xor(a0, o, PP_WAVE));
        a1 = fmin(a1, __shfl_xor(a1, o, PP_WAVE));
    }
    if ((threadIdx.x & (PP_NEAR_GROUP - 1)) == 0) {
        const size_t b = ((size_t)v * gridDim.x + blockIdx.x) * (256 / PP_NEAR_GROUP) + (threadIdx.x / PP_NEAR_GROUP);
        blockmin[2 * b] = a0;
        blockmin[2 * b + 1] = a1;
        const long long left = (long long)n - slot;
        blockcnt[b] = (int)(left < PP_NEAR_GROUP ? (left > 0 ? left : 0) : PP_NEAR_GROUP);
    }
}
#define PP_BOUND_CAP 512             // values the bound kernel ranks: block minima, merged into groups of consecutive blocks when there are more
// (nblk_row: blocks per vertex the arrays are laid out for; near_count: the vertex's own list length — only its blocks were written)
__global__ __launch_bounds__(256) void pp_k_expand_bound(const double* blockmin, const int* blockcnt, int nblk_row, const int* near_count, int k, double* bound, int* cand_count) {
    __shared__ double vals[PP_BOUND_CAP];
    __shared__ int valid;
    __shared__ double U;
    const int vr = blockIdx.x, v = vr >> 1, r = vr & 1;
    const int tid = (int)threadIdx.x;
    if (tid == 0) { valid = 0; U = INFINITY; cand_count[vr] = 0; }
    __syncthreads();
    const int nblk = (near_count[v] + PP_NEAR_GROUP - 1) / PP_NEAR_GROUP;
    const int per = (nblk + PP_BOUND_CAP - 1) / PP_BOUND_CAP;            // blocks per ranked value
    const int nval = per > 0 ? (nblk + per - 1) / per : 0;
    int c = 0;
    for (int j = tid; j < nval; j += 256) {
        double m = INFINITY;
        for (int b = j * per; b < (j + 1) * per && b < nblk; b++) {
            m = fmin(m, blockmin[2 * ((size_t)v * nblk_row + b) + r]);
            c += blockcnt[(size_t)v * nblk_row + b];
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
                                                              long long ns, const int* near_idx, const int* near_count, int two_radii, const double* bound,
                                                              double* g_key, int* g_val, double* g_len, long long g_cap, int* cand_count) {
    __shared__ int wcount[2][4], wbase[2][4];
    const int v = blockIdx.y, w = (int)(threadIdx.x >> 6), lane = pp_lane();
    const int n = near_count[v];
    if ((long long)blockIdx.x * 256 >= (long long)n) return;            // (whole workgroups beyond the vertex's near list)
    const long long slot = (long long)blockIdx.x * 256 + threadIdx.x;
    double l0 = -1, l1 = -1, d = 0;
    int s = 0;
    if (slot < n) {
        const double2 L = reinterpret_cast<const double2*>(lengths)[(size_t)v * (size_t)ns + (size_t)slot];
        l0 = L.x; l1 = L.y;
        s = near_idx[(size_t)v * (size_t)ns + (size_t)slot];
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
        const long long at_slot = wbase[r][w] + __popcll(m[r] & ((1ull << lane) - 1ull));
        if (at_slot < g_cap) {
            const size_t at = (size_t)(2 * v + r) * g_cap + at_slot;
            g_key[at] = d; g_val[at] = s; g_len[at] = r ? l1 : l0;
        }
    }
}
#define PP_ORD_INNER 1024            // candidates of the inner ring whose costs set the filter threshold
__global__ __launch_bounds__(256) void pp_k_expand_order(long long ns, int k, double max_speed, double tpf, int two_radii, const double* bound,
                                                         const double* g_key, const int* g_val, const double* g_len, long long g_cap,
                                                         const int* cand_count, const double* lengths, const int* near_idx, const int* near_count,
                                                         int* out_idx, unsigned* fallbacks) {
    __shared__ double cd[PP_ORD_CAP];        // 96 KB of the CU's 160 KB LDS: distance and list position (later sample index); the lengths
    __shared__ int ci[PP_ORD_CAP];           // stay in the list in memory and are fetched by position when the replay gets there
    __shared__ double inner[PP_ORD_INNER];
    __shared__ int nInner, nKept, fallbackVerdict;
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
    // the replay (wave 0; the other waves wait for its verdict: a list that has to fall back is selected by the whole workgroup)
    const int lane = tid & (PP_WAVE - 1);
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
    if (!fallback && tid < PP_WAVE) {
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
    if (tid == 0) fallbackVerdict = (fallback || unsafeFiltered) ? 1 : 0;
    __syncthreads();
    if (!fallbackVerdict) {
        if (tid < PP_WAVE && lane < k) out[lane] = (lane < hsize) ? h.idx : -1;      // the heap array, front to back
        return;
    }
    {
        // keep what a plain selection gives: the k cheapest of the list, ascending by (length, sample); only the push order is lost
        if (tid == 0) atomicAdd(fallbacks, 1u);
#ifdef PP_DBG_ORD
        if (tid == 0) printf("[ord] FALLBACK vr %d: k %d M %d cap %lld kept %d unsafeFiltered %d hsize %d | U %g dq %g inner %d threshold %g\n", vr, k, M, g_cap, Mk, (int)unsafeFiltered, hsize, U, dq, nInner, threshold);
#endif
        // More candidates within the bound than the list holds (M > g_cap: slots beyond it were dropped in the order the atomics
        // happened to arrive): the truncated list is not a set anyone can name, so the selection runs over the vertex's whole near
        // list instead (every sample within the probe bound, which is not below U: nothing that can win is missing from it)
        const bool fullRow = (long long)M > g_cap;
        const long long Mc = fullRow ? (long long)near_count[vr >> 1] : (long long)M;
        const double* row = lengths + (size_t)(vr >> 1) * (size_t)ns * 2 + r;
        const int* rowIdx = near_idx + (size_t)(vr >> 1) * (size_t)ns;
        // Round 4: ONE pass by the whole workgroup instead of k passes by one wavefront (a late-mission round trip over 2.5 million
        // samples whose lists all fell back took 200 ms and overran the planner's deadline by 145 ms).  U, the k-th smallest block
        // minimum of this row, bounds the k-th smallest length from above, and every sample not longer than U is in the list when
        // the list is whole (its distance is at most its length): the winners are among the entries with length <= U — usually a
        // few dozen — which go to LDS and are ranked there.
        __syncthreads();
        if (tid == 0) nKept = 0;
        __syncthreads();
        for (long long c = tid; c < Mc; c += 256) {
            const double l = fullRow ? row[2 * c] : gl[c];
            if (!(l >= 0) || l > U) continue;
            const int slot = atomicAdd(&nKept, 1);
            if (slot < PP_ORD_CAP) { cd[slot] = l; ci[slot] = fullRow ? rowIdx[c] : gv[c]; }
        }
        __syncthreads();
        const int Ms = nKept;
        if (Ms <= PP_ORD_CAP) {
            if (tid >= PP_WAVE) return;
            double prevL = -INFINITY; int prevI = -1;
            for (int j = 0; j < k; j++) {
                double bl = INFINITY; int bi = 0x7fffffff;
                for (int c = lane; c < Ms; c += PP_WAVE) {
                    const double l = cd[c];
                    const int i = ci[c];
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
        // more than PP_ORD_CAP entries within U (thousands of exactly equal lengths): k successive minimum scans of the source, by the
        // whole workgroup (cd / ci as the reduction's scratch)
        double prevL = -INFINITY; int prevI = -1;
        for (int j = 0; j < k; j++) {
            double bl = INFINITY; int bi = 0x7fffffff;
            for (long long c = tid; c < Mc; c += 256) {
                const double l = fullRow ? row[2 * c] : gl[c];
                const int i = fullRow ? rowIdx[c] : gv[c];
                if (!(l >= 0)) continue;
                const bool after = (l > prevL) || (l == prevL && i > prevI);
                if (after && (l < bl || (l == bl && i < bi))) { bl = l; bi = i; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const double l2 = __shfl_xor(bl, o, PP_WAVE); const int i2 = __shfl_xor(bi, o, PP_WAVE);
                if (l2 < bl || (l2 == bl && i2 < bi)) { bl = l2; bi = i2; }
            }
            __syncthreads();
            if (lane == 0) { cd[tid >> 6] = bl; ci[tid >> 6] = bi; }
            __syncthreads();
            for (int w = 0; w < 4; w++) {
                const double l2 = cd[w]; const int i2 = ci[w];
                if (w == 0 || l2 < bl || (l2 == bl && i2 < bi)) { bl = l2; bi = i2; }
            }
            if (tid == 0) out[j] = (bi == 0x7fffffff) ? -1 : bi;
            prevL = bl; prevI = bi;
            if (bi == 0x7fffffff) { for (int jj = j + 1 + tid; jj < k; jj += 256) out[jj] = -1; break; }
        }
    }
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
