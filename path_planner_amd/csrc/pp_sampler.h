// pp_sampler.h — StateGenerator on the device, bit-exact with libstdc++'s stream.
//
// Reference: path_planner/src/planner/utilities/StateGenerator.{h,cpp}.
//   * std::default_random_engine == minstd_rand0: x <- 16807 x mod (2^31 - 1)
//   * every uniform_real_distribution<double> draw = generate_canonical<double,53> = TWO engine
//     calls: ((e1 - 1) + (e2 - 1) * R) / (R * R), R = 2^31 - 2, evaluated in double
//   * g++ evaluates State(x(), y(), heading(), speed(), 0)'s arguments right to left, so a sample
//     consumes the draws speed, heading, y, x  (SURVEY.md section 8 a-1 probe)
//   * with ribbons a 5th draw decides (u < pi/50) whether the state is projected onto the nearest
//     ribbon, and only then a 6th draw decides the heading flip (StateGenerator.cpp:21-28)
//
// So the stream is measured in "pair slots" (one double draw each); sample i starts at slot q_i,
// q_{i+1} = q_i + 5 + proj(q_i), where proj(q) is a pure function of the slot (the draw at q + 4).
// The LCG admits O(log n) jump-ahead, so proj() is evaluated for every slot in parallel.  Which slots the
// chain visits is then a walk with ONE live state: d_q = how many slots lie between q and the next
// visited slot (0: q itself is visited), d_0 = 0 and
//     d_{q+1} = d_q > 0 ? d_q - 1 : 4 + proj[q].
// Each slot is a function on the six states {0..5}; functions compose associatively, so the walk is a
// prefix scan over function composition (18 bits per function: six 3-bit entries; a composition is six
// table look-ups).  Rounds 1-3 scanned 6x6 boolean transition matrices of the equivalent linear recurrence
// (vis[q+1] = vis[q-4] & !proj[q-4] | vis[q-5] & proj[q-5]: 36 bit-products per composition, most of the
// chain kernels' instructions); the visited bits are the same.  A second (integer) scan ranks the visited
// slots, giving every sample its slot; samples are then generated one per thread.
#pragma once
#include "pp_device.h"

#define PP_SCAN_TILE 2048   // elements per 256-thread workgroup in the scan kernels (8 per thread)

struct PPSamplerState {
    double b[6];               // minX, maxX, minY, maxY, minSpeed, maxSpeed
    unsigned seed;             // engine state before the first call
    int on_ribbons;            // m_SampleOnRibbons
    int n_ribbons;
    int initialised;
    const unsigned long long* d_pos;   // device: [0] pair slots consumed so far.  The position lives on the device so that a skip or an
                                       // add advances it with a launch (pp_k_sampler_advance) and the host never waits for it
};

// ------------------------------------------------------------------------------ minstd_rand0
#define PP_LCG_M 2147483647ull
__device__ __forceinline__ unsigned pp_mulmod(unsigned a, unsigned b) {
    unsigned long long p = (unsigned long long)a * (unsigned long long)b;   // < 2^62
    unsigned long long r = (p & PP_LCG_M) + (p >> 31);                      // < 2^32
    r = (r & PP_LCG_M) + (r >> 31);
    if (r >= PP_LCG_M) r -= PP_LCG_M;
    return (unsigned)r;
}
// engine state after `k` calls from state x: x * 16807^k, the power from its binary digits and a table of 16807^(2^i) mod m
struct PPLcgPowers {
    unsigned v[64];
    constexpr PPLcgPowers() : v() {
        unsigned long long a = 16807ull;
        for (int i = 0; i < 64; i++) { v[i] = (unsigned)a; a = (a * a) % PP_LCG_M; }
    }
};
__device__ const PPLcgPowers pp_lcg_powers = PPLcgPowers();
__device__ inline unsigned pp_lcg_jump(unsigned x, unsigned long long k) {
    unsigned acc = x;
    for (int i = 0; k; i++, k >>= 1)
        if (k & 1ull) acc = pp_mulmod(acc, pp_lcg_powers.v[i]);
    return acc;
}
// one generate_canonical<double,53> from state x (advanced by two calls)
__device__ __forceinline__ double pp_canonical(unsigned& x) {
    const double R = 2147483646.0;
    x = pp_mulmod(x, 16807u);
    double sum = (double)(x - 1u) * 1.0;
    double tmp = R;
    x = pp_mulmod(x, 16807u);
    sum += (double)(x - 1u) * tmp;
    tmp *= R;
    double ret = sum / tmp;
    if (ret >= 1.0) ret = 0.99999999999999988897769753748434595763683319091796875;   // nextafter(1, 0)
    return ret;
}
__device__ __forceinline__ double pp_uniform(unsigned& x, double a, double b) { return pp_canonical(x) * (b - a) + a; }
// `pp_uniform(x, 0, 2 pi) < pi / 50` (StateGenerator.cpp:22) for every stream slot: the division of pp_canonical is only taken when
// the quotient guessed from the reciprocal (a few ulp off at most) lies within 1e-9 of the threshold — one slot in a billion;
// everywhere else the guess decides, as the exact quotient would.
__device__ __forceinline__ bool pp_projection_draw(unsigned& x) {
    const double R = 2147483646.0;
    unsigned xs = x;
    x = pp_mulmod(x, 16807u);
    double sum = (double)(x - 1u) * 1.0;
    x = pp_mulmod(x, 16807u);
    sum += (double)(x - 1u) * R;
    const double guess = sum * (1.0 / (R * R)) * (PP_TWO_PI - 0) + 0;
    const double thr = PP_PI / 50;
    if (guess < thr * (1.0 - 1e-9)) return true;
    if (guess > thr * (1.0 + 1e-9)) return false;
    return pp_uniform(xs, 0, PP_TWO_PI) < thr;       // too close to call: the reference's own expression
}

// Step 1, proj bits: proj[qi] bit 0 = (5th draw of a sample starting at slot pos + qi) < pi/50   (StateGenerator.cpp:22)

// ------------------------------------------------------------------------------ 2. chain scan
// A function on the states {0..5}, entry d in bits [3d, 3d + 3).  pp_fn_compose(later, earlier) = later after earlier.
#define PP_FN_IDENTITY (0u | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12) | (5u << 15))
__device__ __forceinline__ unsigned pp_fn_apply(unsigned f, unsigned d) { return (f >> (3u * d)) & 7u; }
__device__ __forceinline__ unsigned pp_fn_compose(unsigned later, unsigned earlier) {
    unsigned r = 0;
#pragma unroll
    for (int d = 0; d < 6; d++) r |= ((later >> (3u * ((earlier >> (3 * d)) & 7u))) & 7u) << (3 * d);
    return r;
}
// one slot: a visited slot (state 0) sends the walk 5 + proj slots on, i.e. to state 4 + proj seen from the next slot; every
// other state counts down
__device__ __forceinline__ unsigned pp_fn_step(unsigned p) { return (4u + p) | (0u << 3) | (1u << 6) | (2u << 9) | (3u << 12) | (4u << 15); }
// the composition of the 256 thread aggregates of a workgroup, in thread order (later threads on the left); every thread gets it.
// sh = 4 words of LDS.
__device__ inline unsigned pp_fn_block_total(unsigned agg, unsigned* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {                        // lane i: the composition of lanes [i, i + 2o) after step o
        const unsigned hi = (unsigned)__shfl_down((int)agg, o, 64);
        if (lane + o < 64) agg = pp_fn_compose(hi, agg);
    }
    if (lane == 0) sh[wave] = agg;
    __syncthreads();
    const unsigned total = pp_fn_compose(sh[3], pp_fn_compose(sh[2], pp_fn_compose(sh[1], sh[0])));
    __syncthreads();
    return total;
}
// this thread's EXCLUSIVE prefix (identity for thread 0) of the thread aggregates, in thread order.  sh = 4 words of LDS.
__device__ inline unsigned pp_fn_block_exclusive(unsigned agg, unsigned* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = agg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {                        // lane i: the composition of lanes (i - 2o, i] after step o
        const unsigned lo = (unsigned)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc = pp_fn_compose(inc, lo);
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    unsigned before = PP_FN_IDENTITY;                          // the waves before this one
    for (int w = 0; w < wave; w++) before = pp_fn_compose(sh[w], before);
    __syncthreads();
    unsigned excl = (unsigned)__shfl_up((int)inc, 1, 64);
    if (lane == 0) excl = PP_FN_IDENTITY;
    return pp_fn_compose(excl, before);
}
// vis[q] = (state before slot q == 0); stored as bit 1 of proj[q]

// ------------------------------------------------------------------------------ 3. integer scans
__device__ inline unsigned pp_u32_block_scan(unsigned agg, unsigned* sh, unsigned& total) {
    const int t = threadIdx.x;
    sh[t] = agg;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        unsigned v = (t >= o) ? sh[t - o] : 0u;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    total = sh[255];
    unsigned excl = (t == 0) ? 0u : sh[t - 1];
    __syncthreads();
    return excl;
}


// ------------------------------------------------------------------------------ fused launches (round 3)
// The sampler's kernels are tiny (65 536 attempts = 393 k stream slots) and each launch waits for the one before it: twelve launches
// took ~80 us of which the work is a few.  The same steps in five launches: every workgroup works out the prefix of the workgroup
// aggregates before it for itself (a few hundred 8-byte values: cheaper than a launch that does it once), and neighbouring
// steps share a kernel where one only consumes what the other's own workgroup produced.  Same arithmetic, same bits.
//
// prefix of the tile functions blk[0 .. upto): composed in order (later on the left), by the whole workgroup
__device__ inline unsigned pp_fn_prefix_of(const unsigned* blk, int upto, unsigned* sh) {
    unsigned agg = PP_FN_IDENTITY;
    const int per = (upto + 255) / 256;                       // consecutive entries per thread
    const int b0 = (int)threadIdx.x * per;
    for (int i = 0; i < per; i++)
        if (b0 + i < upto) agg = pp_fn_compose(blk[b0 + i], agg);
    return pp_fn_block_total(agg, sh);
}
__device__ inline unsigned pp_u32_prefix_of(const unsigned* blk, int upto, unsigned* sh) {
    unsigned agg = 0;
    for (int i = (int)threadIdx.x; i < upto; i += 256) agg += blk[i];
    unsigned total;
    pp_u32_block_scan(agg, sh, total);
    return total;
}
// 1 + 2a: proj bits of a tile and the tile's function (the composition of its slots' steps)
__global__ __launch_bounds__(256) void pp_k_proj_reduce(unsigned seed, const unsigned long long* d_pos, long long nq, unsigned char* proj,
                                                        unsigned* blk, unsigned long long* zero16) {
    __shared__ unsigned sh[4];
    const unsigned long long pos = d_pos[0];
    if (blockIdx.x == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0ull;   // end slot / total of this call (written by later launches)
    const long long tile0 = (long long)blockIdx.x * PP_SCAN_TILE;
    // a thread's eight slots are consecutive: the draw that decides slot q ends where the one of slot q + 1 begins (two engine calls
    // each), so one jump-ahead per thread and then the engine's own steps
    const long long q0 = tile0 + (long long)threadIdx.x * 8;
    unsigned agg = PP_FN_IDENTITY;
    if (q0 < nq) {
        unsigned x = pp_lcg_jump(seed, 2ull * (pos + (unsigned long long)q0 + 4ull));
        unsigned long long bytes = 0ull;
        for (int i = 0; i < 8; i++) {
            if (q0 + i < nq) {
                const unsigned b = pp_projection_draw(x) ? 1u : 0u;
                bytes |= (unsigned long long)b << (8 * i);
                agg = pp_fn_compose(pp_fn_step(b), agg);
            }
        }
        if (q0 + 8 <= nq) *reinterpret_cast<unsigned long long*>(proj + q0) = bytes;      // (the buffer is 8-byte aligned, q0 a multiple of 8)
        else for (int i = 0; q0 + i < nq; i++) proj[q0 + i] = (unsigned char)((bytes >> (8 * i)) & 0xffu);
    }
    const unsigned total = pp_fn_block_total(agg, sh);
    if (threadIdx.x == 0) blk[blockIdx.x] = total;
}
// 2b + 3a: the tile's visited bits (the function of the tiles before it worked out here) and how many of its slots are visited
__global__ __launch_bounds__(256) void pp_k_chain_apply_count(unsigned char* proj, long long nq, const unsigned* blk, unsigned* cnt) {
    __shared__ unsigned sh[4];
    __shared__ unsigned shc[256];
    const unsigned before = pp_fn_prefix_of(blk, (int)blockIdx.x, sh);
    const long long q0 = (long long)blockIdx.x * PP_SCAN_TILE + (long long)threadIdx.x * 8;
    unsigned long long bytes = 0ull;
    if (q0 + 8 <= nq) bytes = *reinterpret_cast<const unsigned long long*>(proj + q0);
    else for (int i = 0; q0 + i < nq; i++) bytes |= (unsigned long long)proj[q0 + i] << (8 * i);
    unsigned agg = PP_FN_IDENTITY;
    for (int i = 0; i < 8; i++)
        if (q0 + i < nq) agg = pp_fn_compose(pp_fn_step((unsigned)(bytes >> (8 * i)) & 1u), agg);
    const unsigned excl = pp_fn_block_exclusive(agg, sh);
    // the walk starts on slot 0 (state 0); the state in front of this thread's first slot
    unsigned d = pp_fn_apply(excl, pp_fn_apply(before, 0u));
    unsigned c = 0;
    for (int i = 0; i < 8; i++) {
        if (q0 + i < nq) {
            const unsigned p = (unsigned)(bytes >> (8 * i)) & 1u;
            const unsigned vis = (d == 0u) ? 1u : 0u;
            bytes |= (unsigned long long)(vis << 1) << (8 * i);
            c += vis;
            d = d ? d - 1u : 4u + p;
        }
    }
    if (q0 + 8 <= nq) *reinterpret_cast<unsigned long long*>(proj + q0) = bytes;
    else for (int i = 0; q0 + i < nq; i++) proj[q0 + i] = (unsigned char)((bytes >> (8 * i)) & 0xffu);
    unsigned tot;
    pp_u32_block_scan(c, shc, tot);
    if (threadIdx.x == 0) cnt[blockIdx.x] = tot;
}
// 3b: slot of sample i for i < n, and the slot of sample n (= where the next call resumes); cnt = per-tile counts (not yet scanned).
// skip_pos != NULL (ppgpu_sampler_skip: nothing of this call reads the position any more): the thread that finds sample n's slot
// advances the stream position there and then — pos[0] += slot — and the last workgroup raises pos[1] if the chain never got that far.
__global__ __launch_bounds__(256) void pp_k_chain_positions_scan(const unsigned char* proj, long long nq, const unsigned* cnt,
                                                                 long long n, unsigned* qpos, unsigned long long* end_slot, unsigned long long* skip_pos) {
    __shared__ unsigned sh[256];
    const unsigned before = pp_u32_prefix_of(cnt, (int)blockIdx.x, sh);
    long long q0 = (long long)blockIdx.x * PP_SCAN_TILE + (long long)threadIdx.x * 8;
    unsigned c = 0;
    for (int i = 0; i < 8; i++) if (q0 + i < nq && (proj[q0 + i] & 2u)) c++;
    unsigned total;
    unsigned rank = pp_u32_block_scan(c, sh, total) + before;
    if (skip_pos && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && (long long)(before + total) <= n) skip_pos[1] = 1ull;
    for (int i = 0; i < 8; i++) {
        long long q = q0 + i;
        if (q < nq && (proj[q] & 2u)) {
            if ((long long)rank < n) { if (!skip_pos) qpos[rank] = (unsigned)q; }
            else if ((long long)rank == n) {
                *end_slot = (unsigned long long)q;   // relative; pp_k_sampler_advance (or the line below) adds it to the position
                if (skip_pos) skip_pos[0] += (unsigned long long)q;
            }
            rank++;
        }
    }
}

// ------------------------------------------------------------------------------ 4. generate
// RibbonManager::projectOntoNearestRibbon (RibbonManager.cpp:220-232) + Ribbon::getProjectionAsState
// (Ribbon.cpp:80-88): nearest by perpendicular distance to the infinite line, first minimum wins.
__device__ inline void pp_project_onto_nearest(const double* rb, int n, double& x, double& y, double& heading) {
    if (n == 0) return;
    double mn = PP_DBL_MAX;
    PPRibbon best = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        PPRibbon r = {rb[4 * i], rb[4 * i + 1], rb[4 * i + 2], rb[4 * i + 3]};
        double d = pp_ribbon_line_distance(r, x, y);
        if (d < mn) { mn = d; best = r; }
    }
    double px, py;
    pp_ribbon_projection(best, x, y, px, py);
    // State::setHeadingTowards(endX, endY) (State.cpp:51-57,64-67)
    double dx = best.ex - px;
    double dy = best.ey - py;
    double h = PP_PI_2 - atan2(dy, dx);
    if (h < 0) h += PP_TWO_PI;
    x = px; y = py; heading = h;
}


// ------------------------------------------------------------------------------ 5. map filter + compaction

// 4 + 5a: one candidate per thread, the map filter's verdict on it, and how many of the workgroup's 256 candidates are kept
__global__ __launch_bounds__(256) void pp_k_generate_keep(PPSamplerState s, const unsigned* qpos, const unsigned char* proj, const double* ribbons,
                                                          long long n, double* cand, PPGrid g, unsigned char* keep, unsigned* cnt) {
    __shared__ unsigned sc[4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    bool kept = false;
    if (i < n) {
        unsigned long long rel = qpos ? (unsigned long long)qpos[i] : 4ull * (unsigned long long)i;
        unsigned x = pp_lcg_jump(s.seed, 2ull * (s.d_pos[0] + rel));
        double speed = pp_uniform(x, s.b[4], s.b[5]);   // drawn first (right-to-left argument evaluation)
        (void)speed;                                    // expand() overwrites it (SamplingBasedPlanner.cpp:113)
        double heading = pp_uniform(x, 0, PP_TWO_PI);
        double yy = pp_uniform(x, s.b[2], s.b[3]);
        double xx = pp_uniform(x, s.b[0], s.b[1]);
        if (s.on_ribbons) {
            double u5 = pp_uniform(x, 0, PP_TWO_PI);
            if (u5 < PP_PI / 50) {                       // StateGenerator.cpp:22
                pp_project_onto_nearest(ribbons, s.n_ribbons, xx, yy, heading);
                double u6 = pp_uniform(x, 0, PP_TWO_PI);
                if (u6 < PP_PI) heading += PP_PI;        // :24-26, no wrap
            }
        }
        cand[i] = xx; cand[n + i] = yy; cand[2 * n + i] = heading;
        kept = !pp_is_blocked(g, xx, yy);               // SamplingBasedPlanner.cpp:161
        keep[i] = kept ? 1 : 0;
    }
    const unsigned long long m = __ballot(kept);
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
}
// 5b: order-preserving compaction; cnt = kept candidates per 256-candidate workgroup (not yet scanned)
__global__ __launch_bounds__(256) void pp_k_compact_scan(const unsigned char* keep, long long n, const unsigned* cnt, const double* cand,
                                                         double* sx, double* sy, double* sh, long long base, unsigned long long* total_out) {
    __shared__ unsigned shm[256];
    __shared__ unsigned sc[4];
    const unsigned before = pp_u32_prefix_of(cnt, (int)blockIdx.x, shm);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool kept = i < n && keep[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(kept);
    if (lane == 0) sc[wave] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned rank = before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) rank += sc[w];
    if (kept) { sx[base + rank] = cand[i]; sy[base + rank] = cand[n + i]; sh[base + rank] = cand[2 * n + i]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = (unsigned long long)(before + sc[0] + sc[1] + sc[2] + sc[3]);
}

// ------------------------------------------------------------------------------ 6. where the stream resumes
// The last launch of a skip or an add: pos[0] += the slots this call consumed (the chain scan's end slot, or 4 per sample without
// ribbons).  pos[1] = sticky error (the chain scan did not reach the end of its batch), pos[2] = samples the add kept, pos[3] = slots
// consumed by this call: what ppgpu_sampler_add reads back in one copy; ppgpu_sampler_skip reads nothing back.
__global__ void pp_k_sampler_advance(unsigned long long* pos, const unsigned long long* end_slot, const unsigned long long* total,
                                     unsigned long long fixed, int on_ribbons) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long rel = on_ribbons ? *end_slot : fixed;
    if (on_ribbons && rel == 0ull) pos[1] = 1ull; else pos[0] += rel;
    pos[2] = total ? *total : 0ull;
    pos[3] = rel;
}
