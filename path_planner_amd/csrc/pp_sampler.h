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
// The LCG admits O(log n) jump-ahead, so proj() is evaluated for every slot in parallel; which
// slots the chain actually visits is a linear recurrence over GF(2)-like booleans,
//     vis[q+1] = vis[q-4] & !proj[q-4]  |  vis[q-5] & proj[q-5],
// i.e. a prefix "product" of 6x6 boolean matrices — an associative scan.  A second (integer) scan
// ranks the visited slots, giving every sample its slot; samples are then generated one per thread.
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
// engine state after `k` calls from state x
__device__ inline unsigned pp_lcg_jump(unsigned x, unsigned long long k) {
    unsigned a = 16807u, acc = 1u;
    while (k) {
        if (k & 1ull) acc = pp_mulmod(acc, a);
        a = pp_mulmod(a, a);
        k >>= 1;
    }
    return pp_mulmod(x, acc);
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
// 6x6 boolean matrix, row i in bits [6i, 6i+6); C = A after B.
__device__ __forceinline__ unsigned long long pp_bm_compose(unsigned long long A, unsigned long long B) {
    unsigned long long C = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        unsigned ra = (unsigned)((A >> (6 * i)) & 63ull);
        unsigned rc = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) rc |= ((ra >> k) & 1u) ? (unsigned)((B >> (6 * k)) & 63ull) : 0u;
        C |= (unsigned long long)rc << (6 * i);
    }
    return C;
}
#define PP_BM_IDENTITY (1ull | (2ull << 6) | (4ull << 12) | (8ull << 18) | (16ull << 24) | (32ull << 30))
// transition of slot qi: V_{q+1} = M V_q with V_q = (vis[q], vis[q-1], ..., vis[q-5])
__device__ __forceinline__ unsigned long long pp_bm_step(const unsigned char* proj, long long qi) {
    unsigned p4 = qi >= 4 ? (proj[qi - 4] & 1u) : 0u;
    unsigned p5 = qi >= 5 ? (proj[qi - 5] & 1u) : 0u;
    unsigned row0 = ((p4 ^ 1u) << 4) | (p5 << 5);
    return (unsigned long long)row0 | (1ull << 6) | (2ull << 12) | (4ull << 18) | (8ull << 24) | (16ull << 30);
}
// inclusive scan over the 256 thread aggregates of a workgroup; returns this thread's EXCLUSIVE prefix
// (identity for thread 0) and the workgroup aggregate through `total`.  Later elements compose on the left.
__device__ inline unsigned long long pp_bm_block_scan(unsigned long long agg, unsigned long long* sh, unsigned long long& total) {
    const int t = threadIdx.x;
    sh[t] = agg;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        unsigned long long mine = sh[t];
        unsigned long long other = (t >= o) ? sh[t - o] : PP_BM_IDENTITY;
        __syncthreads();
        sh[t] = pp_bm_compose(mine, other);
        __syncthreads();
    }
    total = sh[255];
    unsigned long long excl = (t == 0) ? PP_BM_IDENTITY : sh[t - 1];
    __syncthreads();
    return excl;
}
// vis[q] = (P_q e_0)[0] = P_q[0][0]; stored as bit 1 of proj[q]

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
// prefix of the matrices blk[0 .. upto): composed in order (later on the left), by the whole workgroup
__device__ inline unsigned long long pp_bm_prefix_of(const unsigned long long* blk, int upto, unsigned long long* sh) {
    unsigned long long agg = PP_BM_IDENTITY;
    const int per = (upto + 255) / 256;                       // consecutive entries per thread
    const int b0 = (int)threadIdx.x * per;
    for (int i = 0; i < per; i++)
        if (b0 + i < upto) agg = pp_bm_compose(blk[b0 + i], agg);
    unsigned long long total;
    pp_bm_block_scan(agg, sh, total);
    return total;
}
__device__ inline unsigned pp_u32_prefix_of(const unsigned* blk, int upto, unsigned* sh) {
    unsigned agg = 0;
    for (int i = (int)threadIdx.x; i < upto; i += 256) agg += blk[i];
    unsigned total;
    pp_u32_block_scan(agg, sh, total);
    return total;
}
// 1 + 2a: proj bits of a tile (and of the five slots before it, which its first transitions read) and the tile's transition product
__global__ __launch_bounds__(256) void pp_k_proj_reduce(unsigned seed, const unsigned long long* d_pos, long long nq, unsigned char* proj,
                                                        unsigned long long* blk, unsigned long long* zero16) {
    __shared__ unsigned long long sh[256];
    const unsigned long long pos = d_pos[0];
    __shared__ unsigned char sp[PP_SCAN_TILE + 8];             // sp[5 + i] = proj of the tile's slot i; sp[0 .. 4] = the five slots before the tile
    if (blockIdx.x == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0ull;   // end slot / total of this call (written by later launches)
    const long long tile0 = (long long)blockIdx.x * PP_SCAN_TILE;
    {
        // a thread's eight slots are consecutive: the draw that decides slot q ends where the one of slot q + 1 begins (two engine calls
        // each), so one jump-ahead per thread and then the engine's own steps
        const int l0 = (int)threadIdx.x * 8;
        const long long q0 = tile0 + l0;
        unsigned x = (q0 < nq) ? pp_lcg_jump(seed, 2ull * (pos + (unsigned long long)q0 + 4ull)) : 1u;
        for (int i = 0; i < 8; i++) {
            unsigned char b = 0;
            if (q0 + i < nq) {
                b = pp_projection_draw(x) ? 1 : 0;
                proj[q0 + i] = b;
            }
            sp[5 + l0 + i] = b;
        }
        if (threadIdx.x < 5) {                                 // the five slots before the tile
            const long long q = tile0 - 5 + (long long)threadIdx.x;
            unsigned char b = 0;
            if (q >= 0 && q < nq) {
                unsigned xh = pp_lcg_jump(seed, 2ull * (pos + (unsigned long long)q + 4ull));
                b = pp_projection_draw(xh) ? 1 : 0;
            }
            sp[threadIdx.x] = b;
        }
    }
    __syncthreads();
    const int l0 = (int)threadIdx.x * 8;
    unsigned long long agg = PP_BM_IDENTITY;
    for (int i = 0; i < 8; i++) {
        const long long q = tile0 + l0 + i;
        if (q < nq) {
            const unsigned p4 = q >= 4 ? (unsigned)sp[5 + l0 + i - 4] : 0u, p5 = q >= 5 ? (unsigned)sp[5 + l0 + i - 5] : 0u;
            const unsigned row0 = ((p4 ^ 1u) << 4) | (p5 << 5);
            const unsigned long long m = (unsigned long long)row0 | (1ull << 6) | (2ull << 12) | (4ull << 18) | (8ull << 24) | (16ull << 30);
            agg = pp_bm_compose(m, agg);
        }
    }
    unsigned long long total;
    pp_bm_block_scan(agg, sh, total);
    if (threadIdx.x == 0) blk[blockIdx.x] = total;
}
// 2b + 3a: the tile's visited bits (prefix of the tiles before it worked out here) and how many of its slots are visited
__global__ __launch_bounds__(256) void pp_k_chain_apply_count(unsigned char* proj, long long nq, const unsigned long long* blk, unsigned* cnt) {
    __shared__ unsigned long long sh[256];
    __shared__ unsigned shc[256];
    const unsigned long long before = pp_bm_prefix_of(blk, (int)blockIdx.x, sh);
    long long q0 = (long long)blockIdx.x * PP_SCAN_TILE + (long long)threadIdx.x * 8;
    unsigned long long m[8];
    unsigned long long agg = PP_BM_IDENTITY;
    for (int i = 0; i < 8; i++) {
        long long q = q0 + i;
        m[i] = (q < nq) ? pp_bm_step(proj, q) : PP_BM_IDENTITY;
        agg = pp_bm_compose(m[i], agg);
    }
    unsigned long long total;
    unsigned long long pre = pp_bm_block_scan(agg, sh, total);
    pre = pp_bm_compose(pre, before);
    unsigned vis[8], c = 0;
    for (int i = 0; i < 8; i++) {
        vis[i] = (unsigned)(pre & 1ull);
        pre = pp_bm_compose(m[i], pre);
        if (q0 + i < nq) c += vis[i];
    }
    __syncthreads();   // every thread has read its proj[q-4], proj[q-5] neighbours (bit 0 only is read; bit 1 written)
    for (int i = 0; i < 8; i++) {
        long long q = q0 + i;
        if (q < nq) proj[q] = (unsigned char)((proj[q] & 1u) | (vis[i] << 1));
    }
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
