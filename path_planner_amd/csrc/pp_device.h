// pp_device.h — device-side building blocks for the gfx950 edge-costing kernels.
//
// Everything is IEEE double with -ffp-contract=off: the reference is built for baseline
// x86-64 (no FMA), so every product and sum below rounds separately and in the same order
// as the reference expression it cites.  Wavefront = 64 lanes, hard-coded.
#pragma once
#include "../../include/ppgpu.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PP_WAVE 64
#define PP_TWO_PI 6.283185307179586476925286766559
#define PP_PI 3.14159265358979323846
#define PP_PI_2 1.57079632679489661923
#define PP_DBL_MAX 1.7976931348623157e308

// ----------------------------------------------------------------------------- wave helpers
__device__ __forceinline__ int pp_lane() { return (int)(threadIdx.x & (PP_WAVE - 1)); }

// value of lane `src` (wave-uniform index) broadcast to all lanes, via v_readlane (no LDS traffic)
__device__ __forceinline__ double pp_readlane(double v, int src) {
    int s = __builtin_amdgcn_readfirstlane(src);
    int lo = __builtin_amdgcn_readlane(__double2loint(v), s);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), s);
    return __hiloint2double(hi, lo);
}
// A wave-uniform double moved to scalar registers: per-edge constants (curve bases, times, source pose) then
// occupy SGPRs (or, spilled, single lanes of a VGPR) instead of a full VGPR pair per lane — the sweep kernel is
// latency bound and its occupancy is set by VGPRs.
__device__ __forceinline__ double pp_sgpr(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int pp_readlane_i(int v, int src) {
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}
__device__ __forceinline__ double pp_wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, PP_WAVE));
    return v;
}
__device__ __forceinline__ double pp_wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, PP_WAVE));
    return v;
}
__device__ __forceinline__ double pp_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, PP_WAVE);
    return v;
}
__device__ __forceinline__ int pp_wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, PP_WAVE);
    return v;
}
// x / d for small operands (x < 2^21, d > 0) without the integer-division expansion: (x + 0.5) / d is at least 0.5 / d
// away from every integer while the float evaluation (x + 0.5 exact, rcp within 1 ulp, one rounded product) is off by
// less than 2e-7 of its value, i.e. by less than 0.5 / d whenever x < 2.5e6: truncation gives the exact quotient.
// (Largest use: 1 290 240 prefixes of an 8-ribbon TspPointRobotNoSplitAllRibbons tree.)
__device__ __forceinline__ unsigned pp_udiv_small(unsigned x, unsigned d) {
    return (unsigned)(((float)x + 0.5f) * __builtin_amdgcn_rcpf((float)d));
}
// orders this wave's LDS writes before its later LDS reads (single-wave producer/consumer)
__device__ __forceinline__ void pp_wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// ----------------------------------------------------------------------------- State helpers
// State::yaw(), ppc/include/path_planner_common/State.h:51-55
__device__ __forceinline__ double pp_yaw(double heading) {
    double h = PP_PI_2 - heading;
    if (h < 0) h += PP_TWO_PI;
    return h;
}
// State::setYaw(), State.h:62-65
__device__ __forceinline__ double pp_heading_from_yaw(double yaw) {
    double h = PP_PI_2 - yaw;
    if (h < 0) h += PP_TWO_PI;
    return h;
}

// State::headingTo / setHeadingTowards (ppc State.cpp:51-57,64-67)
__device__ __forceinline__ double pp_heading_to(double x, double y, double x1, double y1) {
    double h = PP_PI_2 - atan2(y1 - y, x1 - x);
    if (h < 0) h += PP_TWO_PI;
    return h;
}

// ----------------------------------------------------------------------------- sine / cosine
// sin and cos of the same angle for the bounded arguments of this kernel (|x| of a few pi): the classic
// reduce-by-pi/2-then-two-polynomials scheme (Cody & Waite reduction carried to ~118 bits in two steps, then the
// minimax polynomials for sin and cos on [-pi/4, pi/4] with a correction term for the low part of the reduced
// argument; error below 1 ulp).  It replaces the general-purpose device sincos, whose huge-argument path and extra
// selects cost about twice as many instructions; arguments outside the fast range take the library call.
// Like any libm pair, results can differ from the host libm's in the last bit: see DESIGN.md Appendix C.
template <bool TAB = false>
__device__ __forceinline__ void pp_sincos_bounded(double x, double* sn, double* cs);
__device__ __forceinline__ void pp_sincos(double x, double* sn, double* cs) {
    if (__ballot(!(fabs(x) < 1.0e5)) != 0ull) { sincos(x, sn, cs); return; }
    pp_sincos_bounded(x, sn, cs);
}
// The fast range only (|x| < 2^20 * pi/2 keeps the first reduction step exact).  The sweep kernels call this directly:
// their arguments are segment base heading +- arc, and pp_k_solve_edges refuses (PPGPU_F_DUBINS_ERR) any curve for which
// |start yaw| + arcs reaches 9e4 rad.  The library call's huge-argument reduction would otherwise sit in the sweep loop
// and set its register budget.
// TAB: the sixteen constants come from memory through scalar loads at the point of use instead of being literals.  As literals
// the compiler keeps them in registers across the cover sweep's whole per-edge loop — 32 registers the loop does not have: they
// were spilled to scratch in the kernel's prologue and reloaded at every window — (the pose sweep, which has the registers, keeps
// the literals).  Same values, same expressions.
__device__ const double pp_sincos_tab[16] = {
    6.36619772367581382433e-01, 1.57079632673412561417e+00, 6.07710050630396597660e-11, 2.02226624879595063154e-21,
    -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04, 2.75573137070700676789e-06,
    -2.50507602534068634195e-08, 1.58969099521155010221e-10,
    4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05, -2.75573143513906633035e-07,
    2.08757232129817482790e-09, -1.13596475577881948265e-11};
template <bool TAB>
__device__ __forceinline__ void pp_sincos_bounded(double x, double* sn, double* cs) {
    const double* tabp = pp_sincos_tab;
    if (TAB) asm volatile("" : "+s"(tabp));                        // keep the loads here, not hoisted out of the caller's loops
    const __attribute__((address_space(4))) double* K = (const __attribute__((address_space(4))) double*)(unsigned long long)tabp;
#define PP_SC(i, lit) (TAB ? K[i] : (lit))
    const double fn = rint(x * PP_SC(0, 6.36619772367581382433e-01));          // x * 2/pi, to nearest
    const int n = (int)fn;
    // pi/2 = pio2_1 + pio2_2 + pio2_2t (+ ...), the leading parts having 33 significant bits each
    double r = fma(-fn, PP_SC(1, 1.57079632673412561417e+00), x);             // exact either way: the product has <= 53 bits
    double wlo;
    {
        const double t = r;
        wlo = fn * PP_SC(2, 6.07710050630396597660e-11);                         // pio2_2
        r = t - wlo;
        wlo = fn * PP_SC(3, 2.02226624879595063154e-21) - ((t - r) - wlo);       // pio2_2t
    }
    const double y0 = r - wlo;
    const double y1 = (r - y0) - wlo;
    const double z = y0 * y0;
    // sin(y0 + y1)
    const double S1 = PP_SC(4, -1.66666666666666324348e-01), S2 = PP_SC(5, 8.33333333332248946124e-03), S3 = PP_SC(6, -1.98412698298579493134e-04),
                 S4 = PP_SC(7, 2.75573137070700676789e-06), S5 = PP_SC(8, -2.50507602534068634195e-08), S6 = PP_SC(9, 1.58969099521155010221e-10);
    const double v = z * y0;
    const double rs = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
    const double sinv = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    // cos(y0 + y1)
    const double C1 = PP_SC(10, 4.16666666666666019037e-02), C2 = PP_SC(11, -1.38888888888741095749e-03), C3 = PP_SC(12, 2.48015872894767294178e-05),
                 C4 = PP_SC(13, -2.75573143513906633035e-07), C5 = PP_SC(14, 2.08757232129817482790e-09), C6 = PP_SC(15, -1.13596475577881948265e-11);
#undef PP_SC
    const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    const double ax = fabs(y0);
    double qx = __hiloint2double(__double2hiint(ax) - 0x00200000, 0);  // |y0| / 4, low word cleared
    qx = (ax > 0.78125) ? 0.28125 : qx;
    qx = (ax < 0.3) ? 0.0 : qx;
    const double hz = 0.5 * z - qx;
    const double cosv = (1.0 - qx) - (hz - (z * rc - y0 * y1));
    // quadrant
    const double s2 = (n & 1) ? cosv : sinv;
    const double c2 = (n & 1) ? sinv : cosv;
    *sn = (n & 2) ? -s2 : s2;
    *cs = ((n + 1) & 2) ? -c2 : c2;
}

// ----------------------------------------------------------------------------- correctly rounded sin / cos / atan2 / acos (pp_cr.h)
#define PP_CR_FN __device__ __forceinline__
#define PP_CR_CONST static __device__ const
#include "pp_cr.h"
__device__ __forceinline__ void pp_cr_sincos(double x, double* sn, double* cs) {
    if (!(fabs(x) < PP_CR_MAX_ARG)) { sincos(x, sn, cs); return; }
    PPdd s, c;
    pp_cr_sincos_dd(x, &s, &c);
    *sn = s.h + s.l; *cs = c.h + c.l;
}
__device__ __forceinline__ double pp_cr_atan2(double y, double x) {
    const double t0 = atan2(y, x);
    // zeros, the axes and non-finite input: the library's value is exact (0, +-pi/2, +-pi are what the reference returns too)
    if (x == 0.0 || y == 0.0 || !(fabs(t0) > 1e-300) || !isfinite(x) || !isfinite(y)) return t0;
    PPdd s, c;
    pp_cr_sincos_dd(t0, &s, &c);
    return pp_cr_atan2_step(y, x, t0, s, c);
}
__device__ __forceinline__ double pp_cr_acos(double v) {
    const double t0 = acos(v);
    if (!(fabs(v) < 1.0) || t0 == 0.0) return t0;              // +-1 (exact 0 / pi) and out-of-range input (NaN)
    PPdd s, c;
    pp_cr_sincos_dd(t0, &s, &c);
    return pp_cr_acos_step(v, t0, s, c);
}

// ----------------------------------------------------------------------------- Dubins
// Third-party `dubins_curves` C library (absent from the reference tree).  Same published
// six-word formulation as path_planner_amd/csrc/dubins.c (the host library behind
// include/dubins.h); see that file for the derivation notes.
// mod2pi(t) = t - 2pi * floor(t / 2pi).  floor() of the quotient only depends on the division's rounding when the
// quotient is within rounding distance of an integer, so the quotient is first taken as t * (1/2pi) and the true
// division is evaluated only if some lane is within 1e-9 of an integer (same result, one division less).
__device__ __forceinline__ double pp_mod2pi(double t) {
    const double q = t * 0.15915494309189533576888376337251;
    double k = floor(q);
    const double fr = q - k;
    if (__ballot(!((fr > 1e-9) & (fr < 1.0 - 1e-9))) != 0ull) k = floor(t / PP_TWO_PI);
    return t - PP_TWO_PI * k;
}

struct PPDubins {
    double p0, p1, p2;  // DubinsPath::param
    int type;           // DubinsPathType, -1 = no path
};

__device__ __forceinline__ int pp_seg_type(int word, int i) {
    // 0 = L, 1 = S, 2 = R ; rows LSL, LSR, RSL, RSR, RLR, LRL, 2 bits per segment
    const unsigned tbl = (0u | (1u << 2) | (0u << 4)) | ((0u | (1u << 2) | (2u << 4)) << 6) |
                         ((2u | (1u << 2) | (0u << 4)) << 12) | ((2u | (1u << 2) | (2u << 4)) << 18) |
                         ((2u | (0u << 2) | (2u << 4)) << 24);
    if (word == 5) return (i == 1) ? 2 : 0;  // LRL
    return (int)((tbl >> (word * 6 + i * 2)) & 3u);
}

// dubins_shortest_path(): normalise, evaluate the six words in enum order, keep the first
// strictly-smallest t+p+q.  CR = true: every atan2 / acos / sin / cos is the correctly rounded one (pp_cr.h) — the edges the planner
// builds (pp_k_solve_edges); CR = false: the device library's — the lengths that only rank samples (pp_k_dubins_lengths, the
// push-order pipeline) and the Dubins-TSP heuristics' tables, where an ulp cannot change what comes out.
template <bool CR>
__device__ __forceinline__ double pp_dub_atan2(double y, double x) { if constexpr (CR) return pp_cr_atan2(y, x); else return atan2(y, x); }
template <bool CR>
__device__ __forceinline__ double pp_dub_acos(double v) { if constexpr (CR) return pp_cr_acos(v); else return acos(v); }
template <bool CR>
__device__ __forceinline__ void pp_dub_sincos(double x, double* s, double* c) { if constexpr (CR) pp_cr_sincos(x, s, c); else pp_sincos(x, s, c); }
template <bool CR = false>
__device__ inline void pp_dubins_shortest(double q0x, double q0y, double q0t, double q1x, double q1y, double q1t,
                                          double rho, PPDubins& out) {
    double dx = q1x - q0x;
    double dy = q1y - q0y;
    double D = sqrt(dx * dx + dy * dy);
    double d = D / rho;
    double theta = 0;
    if (d > 0) theta = pp_mod2pi(pp_dub_atan2<CR>(dy, dx));
    double alpha = pp_mod2pi(q0t - theta);
    double beta = pp_mod2pi(q1t - theta);
    double sa, ca, sb, cb;
    pp_dub_sincos<CR>(alpha, &sa, &ca);
    pp_dub_sincos<CR>(beta, &sb, &cb);
    double c_ab, s_ab_unused;
    pp_dub_sincos<CR>(alpha - beta, &s_ab_unused, &c_ab);
    double d_sq = d * d;

    double best = INFINITY;
    out.type = -1;
    out.p0 = out.p1 = out.p2 = 0;
    double t, p, q;
    {  // LSL
        double tmp0 = d + sa - sb;
        double p_sq = 2 + d_sq - (2 * c_ab) + (2 * d * (sa - sb));
        if (p_sq >= 0) {
            double tmp1 = pp_dub_atan2<CR>((cb - ca), tmp0);
            t = pp_mod2pi(tmp1 - alpha);
            p = sqrt(p_sq);
            q = pp_mod2pi(beta - tmp1);
            double cost = t + p + q;
            if (cost < best) { best = cost; out.p0 = t; out.p1 = p; out.p2 = q; out.type = 0; }
        }
    }
    {  // LSR
        double p_sq = -2 + (d_sq) + (2 * c_ab) + (2 * d * (sa + sb));
        if (p_sq >= 0) {
            p = sqrt(p_sq);
            double tmp0 = pp_dub_atan2<CR>((-ca - cb), (d + sa + sb)) - pp_dub_atan2<CR>(-2.0, p);
            t = pp_mod2pi(tmp0 - alpha);
            q = pp_mod2pi(tmp0 - pp_mod2pi(beta));
            double cost = t + p + q;
            if (cost < best) { best = cost; out.p0 = t; out.p1 = p; out.p2 = q; out.type = 1; }
        }
    }
    {  // RSL
        double p_sq = -2 + d_sq + (2 * c_ab) - (2 * d * (sa + sb));
        if (p_sq >= 0) {
            p = sqrt(p_sq);
            double tmp0 = pp_dub_atan2<CR>((ca + cb), (d - sa - sb)) - pp_dub_atan2<CR>(2.0, p);
            t = pp_mod2pi(alpha - tmp0);
            q = pp_mod2pi(beta - tmp0);
            double cost = t + p + q;
            if (cost < best) { best = cost; out.p0 = t; out.p1 = p; out.p2 = q; out.type = 2; }
        }
    }
    {  // RSR
        double tmp0 = d - sa + sb;
        double p_sq = 2 + d_sq - (2 * c_ab) + (2 * d * (sb - sa));
        if (p_sq >= 0) {
            double tmp1 = pp_dub_atan2<CR>((ca - cb), tmp0);
            t = pp_mod2pi(alpha - tmp1);
            p = sqrt(p_sq);
            q = pp_mod2pi(tmp1 - beta);
            double cost = t + p + q;
            if (cost < best) { best = cost; out.p0 = t; out.p1 = p; out.p2 = q; out.type = 3; }
        }
    }
    {  // RLR
        double tmp0 = (6. - d_sq + 2 * c_ab + 2 * d * (sa - sb)) / 8.;
        double phi = pp_dub_atan2<CR>(ca - cb, d - sa + sb);
        if (fabs(tmp0) <= 1) {
            p = pp_mod2pi((PP_TWO_PI) - pp_dub_acos<CR>(tmp0));
            t = pp_mod2pi(alpha - phi + pp_mod2pi(p / 2.));
            q = pp_mod2pi(alpha - beta - t + pp_mod2pi(p));
            double cost = t + p + q;
            if (cost < best) { best = cost; out.p0 = t; out.p1 = p; out.p2 = q; out.type = 4; }
        }
    }
    {  // LRL
        double tmp0 = (6. - d_sq + 2 * c_ab + 2 * d * (sb - sa)) / 8.;
        double phi = pp_dub_atan2<CR>(ca - cb, d + sa - sb);
        if (fabs(tmp0) <= 1) {
            p = pp_mod2pi(PP_TWO_PI - pp_dub_acos<CR>(tmp0));
            t = pp_mod2pi(-alpha - phi + p / 2.);
            q = pp_mod2pi(pp_mod2pi(beta) - alpha - t + pp_mod2pi(p));
            double cost = t + p + q;
            if (cost < best) { best = cost; out.p0 = t; out.p1 = p; out.p2 = q; out.type = 5; }
        }
    }
}

// dubins_path_length(): ((p0 + p1) + p2) * rho
__device__ __forceinline__ double pp_dubins_length(const PPDubins& d, double rho) {
    double length = 0.;
    length += d.p0;
    length += d.p1;
    length += d.p2;
    return length * rho;
}

// One Dubins segment advanced by t from (bx, by, bth) with sin/cos(bth) precomputed
// (dubins_segment()).  Writes the un-normalised pose.
template <bool CR = false>
__device__ __forceinline__ void pp_segment(int type, double t, double bx, double by, double bth, double sb, double cb,
                                           double& x, double& y, double& th) {
    if (type == 0) {  // L
        double s, c;
        pp_dub_sincos<CR>(bth + t, &s, &c);
        x = (+s - sb) + bx;
        y = (-c + cb) + by;
        th = t + bth;
    } else if (type == 2) {  // R
        double s, c;
        pp_dub_sincos<CR>(bth - t, &s, &c);
        x = (-s + sb) + bx;
        y = (+c - cb) + by;
        th = -t + bth;
    } else {  // S
        x = cb * t + bx;
        y = sb * t + by;
        th = 0.0 + bth;
    }
}

// Per-edge constants of the curve: the three segment bases of dubins_path_sample().
struct PPCurve {
    double qx, qy, qth;   // DubinsPath::qi
    double rho, length;
    double rho_inv;       // 1/rho when rho is a power of two (the division is then exact as a product), else 0
    double p0, p1, p2;
    int t0, t1, t2;       // segment types
    double b1x, b1y, b1th, b2x, b2y, b2th;     // end of segment 1 / 2 (unit radius, origin at qi)
    double s0, c0, s1, c1, s2, c2;             // sin/cos of the three base headings
};

template <bool CR = false>
__device__ inline void pp_curve_init(PPCurve& c, double qx, double qy, double qth, double rho, const PPDubins& d) {
    c.qx = qx; c.qy = qy; c.qth = qth; c.rho = rho;
    {
        int ex;
        c.rho_inv = (frexp(rho, &ex) == 0.5) ? (1.0 / rho) : 0.0;
    }
    c.p0 = d.p0; c.p1 = d.p1; c.p2 = d.p2;
    c.length = pp_dubins_length(d, rho);
    int w = d.type < 0 ? 0 : d.type;
    c.t0 = pp_seg_type(w, 0); c.t1 = pp_seg_type(w, 1); c.t2 = pp_seg_type(w, 2);
    pp_dub_sincos<CR>(qth, &c.s0, &c.c0);
    pp_segment<CR>(c.t0, d.p0, 0.0, 0.0, qth, c.s0, c.c0, c.b1x, c.b1y, c.b1th);
    pp_dub_sincos<CR>(c.b1th, &c.s1, &c.c1);
    pp_segment<CR>(c.t1, d.p1, c.b1x, c.b1y, c.b1th, c.s1, c.c1, c.b2x, c.b2y, c.b2th);
    pp_dub_sincos<CR>(c.b2th, &c.s2, &c.c2);
}
// one segment of dubins_path_sample(): advance by tt from base (bx, by, bth) whose sin/cos are (sb, cb)
template <bool TAB = false>
__device__ __forceinline__ void pp_curve_seg(int type, double tt, double bx, double by, double bth, double sb, double cb,
                                             double& ux, double& uy, double& uth) {
    if (type == 1) {            // straight: no transcendental
        ux = cb * tt + bx;
        uy = sb * tt + by;
        uth = 0.0 + bth;
    } else {
        const double arg = (type == 0) ? (bth + tt) : (bth - tt);
        double s, co;
        pp_sincos_bounded<TAB>(arg, &s, &co);
        if (type == 0) { ux = (+s - sb) + bx; uy = (-co + cb) + by; uth = tt + bth; }
        else           { ux = (-s + sb) + bx; uy = (+co - cb) + by; uth = -tt + bth; }
    }
}

// One segment of a solved curve as the sweep wants it: the interval of tprime (arc length / rho) it covers, the two
// subtractions dubins_path_sample() applies to tprime inside it (none / p0 / p0 then p1, in that order), and its base.
struct PPSeg {
    double bx, by, bth, sb, cb;   // base pose (unit radius, origin at qi), sin/cos of the base heading
    double lo, hi;                // tprime in [lo, hi) is sampled on this segment
    double o1, o2;                // tt = (tprime - o1) - o2
    int type;                     // 0 = L, 1 = S, 2 = R
};
// What the setup record keeps of a segment: its base.  Everything else of PPSeg follows from the record's p0, p1 and hi1 = p0 + p1
// (rounded once, by the solver) and from the Dubins word:
//     segment 0: tprime in (-inf, p0),  tt = (tprime - 0) - 0        segment 1: [p0, hi1), tt = (tprime - p0) - 0
//     segment 2: [hi1, +inf),           tt = (tprime - p0) - p1      (dubins_path_sample's own subtractions, in its order)
struct PPSegBase { double bx, by, bth, sb, cb; };
__device__ __forceinline__ void pp_curve_segments(const PPCurve& c, PPSegBase* sg) {
    sg[0] = PPSegBase{0.0, 0.0, c.qth, c.s0, c.c0};
    sg[1] = PPSegBase{c.b1x, c.b1y, c.b1th, c.s1, c.c1};
    sg[2] = PPSegBase{c.b2x, c.b2y, c.b2th, c.s2, c.c2};
}
__device__ __forceinline__ double pp_seg_o1(int i, double p0) { return i >= 1 ? p0 : 0.0; }
__device__ __forceinline__ double pp_seg_o2(int i, double p1) { return i == 2 ? p1 : 0.0; }
// the kinds of a word's segments (a record without a path keeps word 0's, as the solver's pp_curve_init does)
__device__ __forceinline__ int pp_word_seg_type(int word, int i) { return pp_seg_type(word < 0 ? 0 : word, i); }
// Wave-uniform reads of data written by an EARLIER kernel, through the constant address space: the compiler then always
// issues scalar loads (s_load) for them, which it otherwise only does where it can prove no store in this kernel aliases.
#define PP_AS4 __attribute__((address_space(4)))
__device__ __forceinline__ const PP_AS4 double* pp_const_f64(const void* p) { return (const PP_AS4 double*)(unsigned long long)p; }
__device__ __forceinline__ const PP_AS4 int* pp_const_i32(const void* p) { return (const PP_AS4 int*)(unsigned long long)p; }
__device__ __forceinline__ const PP_AS4 unsigned long long* pp_const_u64(const void* p) { return (const PP_AS4 unsigned long long*)(unsigned long long)p; }
// segment `cur` (wave-uniform) of a setup record as the sweeps keep it in scalar registers: five scalar loads for the base, the rest
// chosen among the record's p0 / p1 / hi1 / word, which the caller has in scalar registers already
__device__ __forceinline__ PPSeg pp_seg_load_uniform(const PPSegBase* g, int cur, double p0, double p1, double hi1, int word) {
    const PP_AS4 double* d = pp_const_f64(g);
    PPSeg s;
    s.bx = d[0]; s.by = d[1]; s.bth = d[2]; s.sb = d[3]; s.cb = d[4];
    s.lo = cur == 0 ? -INFINITY : (cur == 1 ? p0 : hi1);
    s.hi = cur == 0 ? p0 : (cur == 1 ? hi1 : INFINITY);
    s.o1 = pp_seg_o1(cur, p0); s.o2 = pp_seg_o2(cur, p1);
    s.type = pp_word_seg_type(word, cur);
    return s;
}
// which segment dubins_path_sample() picks for tprime: `tprime < p0`, else `tprime < p0 + p1`, else the third
__device__ __forceinline__ int pp_seg_of(double tprime, double hi0, double hi1) {
    return (tprime < hi0) ? 0 : ((tprime < hi1) ? 1 : 2);
}

// ----------------------------------------------------------------------------- occupancy grid
struct PPGrid {
    const uint32_t* bits;  // rows x words_per_row, bit (c & 31) of word c >> 5; NULL with rows == 0: base Map
    int rows, cols, wpr;
    double res, inv_res;
    // clearance[r * cols + c] = min(PP_CLEAR_CAP, chessboard distance in cells from cell (r, c) to the nearest cell that is blocked
    // or outside the grid); 0 for a blocked cell.  Built on the device when the grid is set (pp_k_grid_row_clear / pp_k_grid_clear);
    // lets the pose sweep skip whole chunks of steps that cannot touch a blocked cell.  NULL: no skipping.
    const unsigned char* clearance;
};
#define PP_CLEAR_CAP 64
// GridWorldMap::isBlocked (path_planner/src/common/map/GridWorldMap.cpp:84-93); Map::isBlocked (Map.cpp:4-6).
// The cell index is size_t(x / res): the quotient is first taken as x * (1/res) and the exact division is done only
// where that product is within 1e-9 of an integer (relative), i.e. where the division's rounding could matter.
__device__ __forceinline__ bool pp_is_blocked(const PPGrid& g, double x, double y) {
    if (g.rows == 0) return false;
    // size_t(x / res): the product x * (1 / res) is within a few ulp of the quotient, so if truncating it a little below and a
    // little above (4e-9 relative) gives the same cell, the quotient truncates to that cell too, and `x / res >= cols` is
    // `cell >= cols`; only when some lane sits that close to a cell boundary are the true divisions done (for all lanes).
    const double cx = x * g.inv_res, cy = y * g.inv_res;
    const unsigned cxl = (unsigned)(cx * (1.0 - 4e-9)), cxh = (unsigned)(cx * (1.0 + 4e-9));
    const unsigned cyl = (unsigned)(cy * (1.0 - 4e-9)), cyh = (unsigned)(cy * (1.0 + 4e-9));
    unsigned r = cyl, c = cxl;
    bool outside = (x < 0) | (cxl >= (unsigned)g.cols) | (y < 0) | (cyl >= (unsigned)g.rows);
    if (__ballot((cxl != cxh) | (cyl != cyh)) != 0ull) {
        const double qx = x / g.res, qy = y / g.res;                  // GridWorldMap.cpp:85-88, literally
        outside = (x < 0) | (qx >= (double)g.cols) | (y < 0) | (qy >= (double)g.rows);
        r = (unsigned)qy;
        c = (unsigned)qx;
    }
    if (outside) return true;
    uint32_t w = g.bits[(size_t)r * g.wpr + (c >> 5)];
    return (w >> (c & 31)) & 1u;
}
// (BASELINE's north_star asks for the occupancy grid "tiled into LDS".  Built and measured in round 3 — the wave copies the words under
// its 64 cells into an LDS tile, every lane reads its word there, commit c47d58f — parity green, pose sweep 496 -> 606 us: the whole
// 2048 x 2048 bit grid is 512 KiB and lives in every XCD's 4 MiB L2, so the direct lookup above stays.  DESIGN.md Appendix B.)

// ----------------------------------------------------------------------------- dynamic obstacles
// BinaryDynamicObstaclesManager::Obstacle with the per-call constants hoisted on the HOST with
// the host libm (ppgpu_set_obstacles): cosYaw/sinYaw = cos/sin(M_PI_2 - heading), halfL/halfW =
// (Length + 2) / 2, (Width + 2) / 2 for the strict test used by Edge::computeTrueCost (Edge.cpp:151);
// reach = sqrt(halfL^2 + halfW^2) rounded up: no point farther than that from the centre can hit.
struct PPObst { double X, Y, cosYaw, sinYaw, Speed, Time, halfL, halfW, reach, pad[3]; };
// GaussianDynamicObstaclesManager::Obstacle (.h:19-49): a, b, c, d = the inverse covariance (0,0), (0,1), (1,0), (1,1) and
// norm = 1 / 2pi / sqrt(det), both computed on the host in Eigen's 2x2 order; reach = distance from the mean beyond which
// the pdf is below 1e-13.  Same size and leading fields as PPObst so the culling code is shared.
struct PPGauss { double X, Y, cosYaw, sinYaw, Speed, Time, i00, i01, reach, i10, i11, norm; };

// One obstacle's contribution to BinaryDynamicObstaclesManager::collisionExists(x, y, t, strict=true)
// (.cpp:4-22): project to t, translate, rotate by +Yaw, strict box test.
__device__ __forceinline__ int pp_obstacle_hit(const PPObst& o, double x, double y, double t) {
    double dt = t - o.Time;
    double ddx = o.Speed * dt * o.cosYaw;
    double ddy = o.Speed * dt * o.sinYaw;
    double X = o.X + ddx;
    double Y = o.Y + ddy;
    double tx = x - X;
    double ty = y - Y;
    double rx = tx * o.cosYaw - ty * o.sinYaw;
    double ry = tx * o.sinYaw + ty * o.cosYaw;
    return (fabs(rx) < o.halfL && fabs(ry) < o.halfW) ? 1 : 0;
}
// GaussianDynamicObstaclesManager::Obstacle::project + pdf (.h:31-43)
__device__ __forceinline__ double pp_obstacle_pdf(const PPGauss& o, double x, double y, double t) {
    const double dt = t - o.Time;
    const double X = o.X + o.Speed * dt * o.cosYaw;
    const double Y = o.Y + o.Speed * dt * o.sinYaw;
    const double vx = x - X, vy = y - Y;
    const double r0 = vx * o.i00 + vy * o.i10, r1 = vx * o.i01 + vy * o.i11;
    const double quadform = r0 * vx + r1 * vy;
    return o.norm * exp(-0.5 * quadform);
}

// Hit count for this lane's step, for 64 consecutive steps held one per lane.
// Culling (exact): lane i first tests obstacle i against the whole chunk — every valid lane's pose lies
// within `span` of lane 0's pose (c0x, c0y at time t0) and every obstacle moves at most |Speed|*tspan
// during the chunk, so an obstacle farther than reach + span + |Speed|*tspan (+ slack) from lane 0's pose
// at t0 cannot hit any lane.  Only the survivors (usually none or one) get the exact test.
__device__ inline int pp_obstacle_hits_chunk(const PPObst* __restrict__ ob, int n, double x, double y, double t, bool valid,
                                             double c0x, double c0y, double t0, double span, double tspan) {
    int sum = 0;
    const int lane = pp_lane();
    for (int b = 0; b < n; b += PP_WAVE) {
        const int oi = b + lane;
        bool near = false;
        if (oi < n) {
            const PPObst o = ob[oi];
            double dt = t0 - o.Time;
            double X = o.X + o.Speed * dt * o.cosYaw;
            double Y = o.Y + o.Speed * dt * o.sinYaw;
            double R = o.reach + span + fabs(o.Speed) * tspan + 1e-3;
            double dx = c0x - X, dy = c0y - Y;
            near = !(dx * dx + dy * dy > R * R);
        }
        unsigned long long m = __ballot(near);
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            const PPObst& o = ob[b + j];       // wave-uniform address: scalar load
            if (valid) sum += pp_obstacle_hit(o, x, y, t);
        }
    }
    return sum;
}

// GaussianDynamicObstaclesManager::collisionExists (.cpp:3-13) for this lane's step: the sum of the pdfs of the obstacles
// that can matter anywhere in the chunk (same culling as above with reach = the 1e-13 radius), in obstacle order, floored
// at 1e-5.
__device__ inline double pp_obstacle_density_chunk(const PPGauss* __restrict__ ob, int n, double x, double y, double t, bool valid,
                                                   double c0x, double c0y, double t0, double span, double tspan) {
    double sum = 0;
    const int lane = pp_lane();
    for (int b = 0; b < n; b += PP_WAVE) {
        const int oi = b + lane;
        bool near = false;
        if (oi < n) {
            const PPGauss& o = ob[oi];
            const double dt = t0 - o.Time;
            const double X = o.X + o.Speed * dt * o.cosYaw;
            const double Y = o.Y + o.Speed * dt * o.sinYaw;
            const double R = o.reach + span + fabs(o.Speed) * tspan + 1e-3;
            const double dx = c0x - X, dy = c0y - Y;
            near = !(dx * dx + dy * dy > R * R);
        }
        unsigned long long m = __ballot(near);
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            const PPGauss& o = ob[b + j];      // wave-uniform address: scalar load
            sum += pp_obstacle_pdf(o, x, y, t);
        }
    }
    if (sum < 1e-5) sum = 0;                   // "questionable", .cpp:11
    return valid ? sum : 0.0;
}

// ----------------------------------------------------------------------------- ribbons (one per lane)
#define PP_RIBBON_TOL 1e-5  // Ribbon::c_Tolerance (Ribbon.h:129)

struct PPRibbon { double sx, sy, ex, ey; };

__device__ __forceinline__ double pp_sq_len(double sx, double sy, double ex, double ey) {  // Ribbon.h:133-135
    return (ex - sx) * (ex - sx) + (ey - sy) * (ey - sy);
}
__device__ __forceinline__ double pp_dist(double x1, double y1, double x2, double y2) {  // RibbonManager.h:285-287
    return sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
}
// Ribbon::getProjection (Ribbon.cpp:72-78)
__device__ __forceinline__ void pp_ribbon_projection(const PPRibbon& r, double x, double y, double& px, double& py) {
    double squaredL = pp_sq_len(r.sx, r.sy, r.ex, r.ey);
    double dot = (x - r.sx) * (r.ex - r.sx) + (y - r.sy) * (r.ey - r.sy);
    double projectedX = (r.ex - r.sx) * dot / squaredL;
    double projectedY = (r.ey - r.sy) * dot / squaredL;
    px = projectedX + r.sx;
    py = projectedY + r.sy;
}
// Ribbon::containsProjection (Ribbon.cpp:90-95)
__device__ __forceinline__ bool pp_ribbon_contains_projection(const PPRibbon& r, double px, double py) {
    const double T = PP_RIBBON_TOL;
    return !(((px - r.sx < -T && px - r.ex < -T) || (px - r.sx > T && px - r.ex > T)) ||
             ((py - r.sy < -T && py - r.ey < -T) || (py - r.sy > T && py - r.ey > T)));
}
// Ribbon::distance (Ribbon.h:118-121)
__device__ __forceinline__ double pp_ribbon_line_distance(const PPRibbon& r, double x, double y) {
    return (fabs((r.ey - r.sy) * x - (r.ex - r.sx) * y + r.ex * r.sy - r.ey * r.sx)) / sqrt(pp_sq_len(r.sx, r.sy, r.ex, r.ey));
}

// min over lanes [0, n) of a value that is PP_DBL_MAX elsewhere; n is wave-uniform and usually small
__device__ __forceinline__ double pp_min_first_n(double v, int n) {
    if (n <= 8) {
        double m = pp_readlane(v, 0);
        for (int i = 1; i < n; i++) m = fmin(m, pp_readlane(v, i));
        return m;
    }
    return pp_wave_min(v);
}

// |num| / sqrt(sqL) < lim, decided WITHOUT the square root and the division whenever the answer cannot depend on
// their rounding: with correctly rounded sqrt and divide the quotient is within (1 +- 2.1 * 2^-53) of the real
// value, so A = num^2 against C = lim^2 * sqL with a relative margin of 1e-11 is conclusive; only inside that
// margin (or for degenerate/NaN input) the reference's own expression (Ribbon.h:118-121) is evaluated.
__device__ __forceinline__ bool pp_line_distance_lt(double num, double sqL, double lim) {
    const double A = num * num;
    const double C = (lim * lim) * sqL;
    if (A < C * (1.0 - 1e-11)) return true;
    if (A > C * (1.0 + 1e-11)) return false;
    return (fabs(num) / sqrt(sqL)) < lim;
}

// One coverage event of Edge::computeTrueCost (Edge.cpp:158-161), lane i holding ribbon i (i < n):
//   D = RibbonManager::minDistanceFrom(x, y)                    (RibbonManager.cpp:142-152)
//   if (doCover) RibbonManager::cover(x, y, strict = true)      (RibbonManager.cpp:14-22, Ribbon::split
//                                                                Ribbon.cpp:9-17, Ribbon::covered :23-25)
// Both need the same projection and perpendicular distance per ribbon; they are computed once.
// cover() preserves list order: old ribbon i contributes [front part, if split and not covered] then
// [itself / the remainder, if not covered].  lds = this wave's 64 x 4-double scratch.
// Returns the new ribbon count (may exceed 64 -> the caller flags overflow).
__device__ inline int pp_ribbons_event(PPRibbon& r, int n, double w, double x, double y, bool doCover, double* lds, double& D,
                                       int& adv) {
    D = 0;
    adv = -1;
    if (n == 0) return 0;
    const int lane = pp_lane();
    const bool act = lane < n;
    {
        // Most events happen on the approach, far from every ribbon.  Ribbon::contains needs the projection inside the
        // segment (+-1e-5 per coordinate) and a perpendicular distance below w, so a point outside a ribbon's bounding box
        // grown by w (+1e-3 slack) cannot be contained in it: nothing contains, nothing splits, and unless some piece is
        // short enough to be erased as covered (Ribbon::covered, checked by cover() for every ribbon) the event only
        // measures the distance to the nearest endpoint.
        const double grow = w + 1e-3;
        const bool inBox = (x >= fmin(r.sx, r.ex) - grow) & (x <= fmax(r.sx, r.ex) + grow) & (y >= fmin(r.sy, r.ey) - grow) & (y <= fmax(r.sy, r.ey) + grow);
        const double minLength0 = 2 * w;
        const bool tiny = pp_sq_len(r.sx, r.sy, r.ex, r.ey) < minLength0 * minLength0 / (2.0 * 2.0);
        if (__ballot(act & (inBox | tiny)) == 0ull) {
            const double qS = (r.sx - x) * (r.sx - x) + (r.sy - y) * (r.sy - y);
            const double qE = (r.ex - x) * (r.ex - x) + (r.ey - y) * (r.ey - y);
            D = fmin(PP_DBL_MAX, sqrt(pp_min_first_n(act ? fmin(qE, qS) : PP_DBL_MAX, n)));
            adv = -3;                                                   // like -2 (nothing changed), told apart only by the debug counters
            return n;
        }
    }
    // Straight-line code on purpose (bitwise &,| instead of &&,||; inactive lanes hold a zero ribbon and are masked
    // out at the ballots): the event loop is the serial part of the kernel and every branch in it costs.
    // Ribbon::getProjection (Ribbon.cpp:72-78)
    const double dxr = r.ex - r.sx, dyr = r.ey - r.sy;
    const double sqL = dxr * dxr + dyr * dyr;
    const double dot = (x - r.sx) * dxr + (y - r.sy) * dyr;
    const double px = dxr * dot / sqL + r.sx;
    const double py = dyr * dot / sqL + r.sy;
    // Ribbon::containsProjection (Ribbon.cpp:90-95)
    const double T = PP_RIBBON_TOL;
    const double a1 = px - r.sx, a2 = px - r.ex, b1 = py - r.sy, b2 = py - r.ey;
    const bool outx = ((a1 < -T) & (a2 < -T)) | ((a1 > T) & (a2 > T));
    const bool outy = ((b1 < -T) & (b2 < -T)) | ((b1 > T) & (b2 > T));
    const bool cp = act & !(outx | outy);
    // Ribbon::distance (Ribbon.h:118-121) against w (contains, non-strict) and w/2 (strict, Ribbon.cpp:39-43):
    // decided on squares with a 1e-11 relative margin (see pp_line_distance_lt), exact expression only if ambiguous
    const double num = dyr * x - dxr * y + r.ex * r.sy - r.ey * r.sx;
    const double A = num * num;
    const double Cn = (w * w) * sqL;
    const double Cs = ((w / 2.0) * (w / 2.0)) * sqL;
    bool ltN = A < Cn * (1.0 - 1e-11), gtN = A > Cn * (1.0 + 1e-11);
    bool ltS = A < Cs * (1.0 - 1e-11), gtS = A > Cs * (1.0 + 1e-11);
    if (__ballot(cp & (!(ltN | gtN) | !(ltS | gtS))) != 0ull) {
        const double ld = fabs(num) / sqrt(sqL);
        ltN = ld < w;
        ltS = ld < (w / 2.0);
    }
    const bool ns = cp & ltN;
    const bool st = cp & ltS & doCover;
    // minDistanceFrom: 0 as soon as one ribbon contains the point (non-strict width), else nearest endpoint
    if (__ballot(ns) == 0ull) {
        // min over endpoints of sqrt(squared distance) (RibbonManager.h:285-287) = sqrt of the min: sqrt is monotone
        // and correctly rounded, so one square root after the reduction gives the same double
        const double qS = (r.sx - x) * (r.sx - x) + (r.sy - y) * (r.sy - y);
        const double qE = (r.ex - x) * (r.ex - x) + (r.ey - y) * (r.ey - y);
        const double m = act ? fmin(qE, qS) : PP_DBL_MAX;
        D = fmin(PP_DBL_MAX, sqrt(pp_min_first_n(m, n)));
    }
    if (!doCover) { adv = -2; return n; }                            // nothing can change without cover()
    const double minLength = 2 * w;                                  // Ribbon::minLength (Ribbon.cpp:52-58)
    const double thr = minLength * minLength / (2.0 * 2.0);          // covered(strict): c_StrictModifier^2
    const bool keepF = st & !(pp_sq_len(r.sx, r.sy, px, py) < thr);
    const bool keepR = act & (st ? !(pp_sq_len(px, py, r.ex, r.ey) < thr) : !(sqL < thr));
    const unsigned long long mS = __ballot(st), mF = __ballot(keepF), mR = __ballot(keepR);
    const unsigned long long actMask = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    if (mS == 0ull && mR == actMask) { adv = -2; return n; }          // nothing split, nothing erased
    if ((mF & mR) == 0ull && (mF | mR) == actMask) {
        // every split keeps exactly one of its halves and nothing else is erased: pieces change in place, order kept.
        // keepR: the front vanished, the start moves to the projection; keepF: the remainder vanished, the END moves.
        r.sx = (st & keepR) ? px : r.sx;
        r.sy = (st & keepR) ? py : r.sy;
        r.ex = (st & keepF) ? px : r.ex;
        r.ey = (st & keepF) ? py : r.ey;
        if ((mS & (mS - 1ull)) == 0ull) {                            // exactly one piece changed: a corridor run may follow
            adv = __ffsll((long long)mS) - 1;
            if (mF != 0ull) adv |= 0x100;                            // ... with its end (not its start) following the vehicle
        }
        return n;
    }
    // exactly one piece split, both halves kept, nothing erased: told to the caller (adv = -4 - index of the front half; the rest
    // follows it), which can then guess the corridor run that very likely starts at the next step
    if (mS != 0ull && (mS & (mS - 1ull)) == 0ull && mF == mS && mR == actMask && n < 64) adv = -4 - (__ffsll((long long)mS) - 1);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int posF = __popcll(mF & below) + __popcll(mR & below);
    const int posR = posF + (keepF ? 1 : 0);
    const int total = __popcll(mF) + __popcll(mR);
    if (keepF && posF < 64) { lds[posF * 4 + 0] = r.sx; lds[posF * 4 + 1] = r.sy; lds[posF * 4 + 2] = px; lds[posF * 4 + 3] = py; }
    if (keepR && posR < 64) {
        lds[posR * 4 + 0] = st ? px : r.sx; lds[posR * 4 + 1] = st ? py : r.sy;
        lds[posR * 4 + 2] = r.ex; lds[posR * 4 + 3] = r.ey;
    }
    pp_wave_lds_fence();
    if (lane < total && lane < 64) { r.sx = lds[lane * 4 + 0]; r.sy = lds[lane * 4 + 1]; r.ex = lds[lane * 4 + 2]; r.ey = lds[lane * 4 + 3]; }
    else if (lane >= total) { r.sx = 0; r.sy = 0; r.ex = 0; r.ey = 0; }
    pp_wave_lds_fence();
    return total;
}

// Corridor run.  While the vehicle travels inside one ribbon's strict corridor towards its end, EVERY step is a
// coverage event that does the same thing: split that one piece at the projection, drop the one-step-long front,
// keep the rest — i.e. the piece's start follows the vehicle (cover(): RibbonManager.cpp:14-22).  Sequentially that
// is one dependent chain of two divisions per step; here the following steps of the chunk (one per lane) are checked
// together and the longest prefix for which the outcome is beyond doubt is applied at once.
//
// "Beyond doubt": projections are taken on the piece's line as it stands at the start of the run (the sequential
// chain re-derives the line from the moved start each step; the two differ by rounding only, ~1e-14 m), and every
// decision the reference would take (containsProjection with its 1e-5 tolerance, strict distance, front covered,
// rest not covered, no other piece anywhere near) must hold with a guard of 1e-9 — a step that does not clear the
// guard ends the run and goes through the exact per-event code.  Flags therefore cannot differ; the moved start
// differs from the sequential value by rounding noise (<= 1e-12 m).
//
// lanes = steps of the current chunk; `first` is the first candidate step, `limit` the end of the executable range.
// moveEnd = false: the piece's START follows the vehicle (front part vanishes each step); true: its END does (the
// vehicle travels towards the piece's start and the remainder vanishes each step).
// Returns the run length L (0 = none) and the new position of the moving endpoint of piece `adv` (wave-uniform).
// Pieces no step of the window can reach, decided once with one piece per lane: every step of the window lies within `span`
// (64 steps of at most one increment each) of step `first`, so a piece whose midpoint is farther than its own reach + span
// from that step is `far` (in the sense of the per-step test in the runs below) for all of them.
__device__ __forceinline__ unsigned long long pp_pieces_in_reach(const PPRibbon& r, int n, double w, double x0, double y0, double span) {
    const int lane = pp_lane();
    const double mx = 0.5 * (r.sx + r.ex), my = 0.5 * (r.sy + r.ey);
    const double ql = pp_sq_len(r.sx, r.sy, r.ex, r.ey);
    const double reach = 0.5 * sqrt(ql) + w + 1e-3 + span;
    return __ballot((lane < n) & !(pp_sq_len(mx, my, x0, y0) > reach * reach));
}

// cover() erases every piece shorter than the strict minimum length wherever the vehicle is (Ribbon::covered, checked for each
// ribbon by RibbonManager::cover): while such a piece exists, a step at which cover() runs is never "the same thing again".
__device__ __forceinline__ bool pp_any_erasable_piece(const PPRibbon& r, int n, double w) {
    const double minLength = 2 * w;
    return __ballot((pp_lane() < n) & (pp_sq_len(r.sx, r.sy, r.ex, r.ey) < minLength * minLength / (2.0 * 2.0))) != 0ull;
}

//
// Long runs (round 3): ell > 0.  A slow edge crawling along a ribbon (0.5 m/s: one centimetre per collision-check step) stays inside
// one piece's corridor for hundreds of steps; taken 64 steps at a time every window costs a window of poses and a run check to
// learn "the same again".  When a run has just filled a whole window the caller therefore tries the next stretch with one SAMPLE
// every s steps (lane i = step base + i s) and ell = the arc length of s steps; every decision is made at the samples with a margin
// that covers the steps between them:
//   * every step between samples i-1 and i lies within ell of sample i, so a bound that holds at the sample with ell to spare —
//     inside the strict corridor, outside another piece's reach / extent / strict corridor — holds at those steps;
//   * the vehicle advances along the piece (the chord between consecutive samples makes an angle with the piece of less than 90
//     degrees minus the turn the curve can make between them; sinDt >= sin(ell / rho)), so projections move monotonically: each
//     step's projection lies between the previous step's (the moving endpoint) and the far end, which is what containsProjection
//     asks for; the half that must survive is shortest at the LAST step covered, where it is checked; the half that vanishes is at
//     most ell long (the caller checks ell is shorter than the minimum length);
//   * bit i of coverMask says cover() runs at EVERY step sample i vouches for.
// The run covers samples first .. first + L - 1; the endpoint moves to the projection of the last of them, the same expression the
// step-by-step run applies at its last step.  A sample that does not clear its margins ends the long run there and ordinary
// windows take over: flags cannot differ.  With ell = 0 (and sinDt = 0) every expression below is the step-by-step one.
__device__ inline int pp_corridor_run(const PPRibbon& r, int n, double w, int adv, bool moveEnd, double x, double y, bool stepOk,
                                      unsigned long long coverMask, int first, double span, double& newX, double& newY,
                                      double ell = 0.0, double sinDt = 0.0) {
    const int lane = pp_lane();
    if (pp_any_erasable_piece(r, n, w)) return 0;      // every step of a corridor run calls cover(): it would erase that piece first
    const double Sx = pp_readlane(r.sx, adv), Sy = pp_readlane(r.sy, adv), Ex = pp_readlane(r.ex, adv), Ey = pp_readlane(r.ey, adv);
    const double g = 1e-9;
    const double dxr = Ex - Sx, dyr = Ey - Sy;
    const double sqL = dxr * dxr + dyr * dyr;
    const double dot = (x - Sx) * dxr + (y - Sy) * dyr;
    const double px = dxr * dot / sqL + Sx;
    const double py = dyr * dot / sqL + Sy;
    const double num = dyr * x - dxr * y + Ex * Sy - Ey * Sx;
    // Long runs, round 4: what a sample vouches for is decided on the CHORD from the sample before it.  The steps between two samples
    // lie on a curve of curvature <= 1/rho and arc length <= ell, so they stay within sag = ell^2 / (8 rho) of the chord; a condition
    // whose region is convex (inside a corridor: a strip; a projection beyond one end of a piece: a half-plane, projections being
    // linear and contracting) holds at all of them when it holds at BOTH samples with sag to spare.  Rounds 3's margin was ell itself
    // (every step within ell of its sample): 5 cm where sag is 0.02 mm, so a long run could not start until the vehicle was 5 cm
    // inside the strict corridor and 5 cm past the piece the split left behind it — the first window after a split was never one.
    // The first sample of a window is the step after the last event and vouches for itself alone: it needs no partner (it is tested
    // with the same margin, being the next sample's partner).
    const double sag = (ell > 0.0) ? (ell * sinDt * 0.125) * (1.0 + 1e-9) + 1e-9 : 0.0;
    const double lim = w / 2.0 - sag;
    bool strictOk = (lim > 0.0) & ((num * num) < ((lim * lim) * sqL) * (1.0 - g));
    if (ell > 0.0) {                                              // wave-uniform
        const bool prevOk = __shfl_up((int)strictOk, 1, PP_WAVE) != 0;
        strictOk = strictOk & ((lane == first) | prevOk);
    }
    // the moving endpoint this step will see: the previous step's projection (the piece's own endpoint for the first)
    double qx = __shfl_up(px, 1, PP_WAVE), qy = __shfl_up(py, 1, PP_WAVE);
    if (lane == first) { qx = moveEnd ? Ex : Sx; qy = moveEnd ? Ey : Sy; }
    bool mono = true;
    if (ell > 0.0) {                                              // wave-uniform: long runs only
        const double cx = x - __shfl_up(x, 1, PP_WAVE), cy = y - __shfl_up(y, 1, PP_WAVE);
        const double along = (cx * dxr + cy * dyr) * (moveEnd ? -1.0 : 1.0);
        mono = (lane == first) | ((along > 0.0) & ((along * along) > ((cx * cx + cy * cy) * sqL) * (sinDt * sinDt) * (1.0 + 1e-6)));
    }
    const double csx = moveEnd ? Sx : qx, csy = moveEnd ? Sy : qy;   // the piece as this step sees it
    const double cex = moveEnd ? qx : Ex, cey = moveEnd ? qy : Ey;
    const double T = PP_RIBBON_TOL - g;
    const double a1 = px - csx, a2 = px - cex, b1 = py - csy, b2 = py - cey;
    const bool outx = ((a1 < -T) & (a2 < -T)) | ((a1 > T) & (a2 > T));
    const bool outy = ((b1 < -T) & (b2 < -T)) | ((b1 > T) & (b2 > T));
    const double thr = (2 * w) * (2 * w) / (2.0 * 2.0);
    const double frontSq = pp_sq_len(csx, csy, px, py), restSq = pp_sq_len(px, py, cex, cey);
    // start follows: front [start, proj] vanishes, rest kept; end follows: front kept, rest [proj, end] vanishes
    const bool halves = moveEnd ? ((frontSq > thr * (1.0 + g)) & (restSq < thr * (1.0 - g)))
                                : ((frontSq < thr * (1.0 - g)) & (restSq > thr * (1.0 + g)));
    bool ok = stepOk & (lane >= first) & strictOk & !(outx | outy) & halves & mono & (((coverMask >> lane) & 1ull) != 0ull);
    // no other piece may be touched by any step of the run: either it is out of reach (farther than half its length
    // + w from its midpoint: cheap), or — for the near ones, typically the sibling the first split left behind —
    // the reference's own test must fail with the guard: projection clearly outside the piece, or clearly outside
    // its strict corridor
    unsigned long long others = pp_pieces_in_reach(r, n, w, pp_readlane(x, first), pp_readlane(y, first), span) & ~(1ull << adv);
    while (others) {
        const int q = __ffsll((long long)others) - 1;
        others &= others - 1;
        const double sx = pp_readlane(r.sx, q), sy = pp_readlane(r.sy, q), ex = pp_readlane(r.ex, q), ey = pp_readlane(r.ey, q);
        const double mx = 0.5 * (sx + ex), my = 0.5 * (sy + ey);
        const double ql = pp_sq_len(sx, sy, ex, ey);
        const double reach = 0.5 * sqrt(ql) + w + 1e-3 + ell;
        const bool far = pp_sq_len(mx, my, x, y) > reach * reach;
        if (__ballot(ok & !far) != 0ull) {
            const double dq = ex - sx, eq = ey - sy;
            const double dt = (x - sx) * dq + (y - sy) * eq;
            const double ppx = dq * dt / ql + sx, ppy = eq * dt / ql + sy;
            const double T2 = PP_RIBBON_TOL + g + sag;
            const double c1 = ppx - sx, c2 = ppx - ex, d1 = ppy - sy, d2 = ppy - ey;
            // which of the four half-plane pairs of Ribbon::containsProjection puts the projection outside the piece
            const int outCode = (((c1 < -T2) & (c2 < -T2)) ? 1 : 0) | (((c1 > T2) & (c2 > T2)) ? 2 : 0) | (((d1 < -T2) & (d2 < -T2)) ? 4 : 0) | (((d1 > T2) & (d2 > T2)) ? 8 : 0);
            const double nq = eq * x - dq * y + ex * sy - ey * sx;
            const double lo = w / 2.0 + sag;
            const bool strictOut = (nq * nq) > ((lo * lo) * ql) * (1.0 + g);
            bool clear = (outCode != 0) | strictOut;
            if (ell > 0.0) {
                // ... on the same side as the sample before (a chord cannot leave a half-plane both its ends are in)
                const int prevCode = __shfl_up(outCode, 1, PP_WAVE);
                const bool prevStrictOut = __shfl_up((int)strictOut, 1, PP_WAVE) != 0;
                const bool prevPos = __shfl_up((int)(nq > 0.0), 1, PP_WAVE) != 0;
                const bool pair = ((outCode & prevCode) != 0) | (strictOut & prevStrictOut & ((nq > 0.0) == prevPos));
                clear = (lane == first) ? clear : pair;
            }
            ok = ok & (far | clear);
        }
    }
    const unsigned long long okMask = __ballot(ok);
    const unsigned long long bad = (~okMask) >> first;             // bit 0 = step `first`
    const int L = first >= 64 ? 0 : (bad ? (__ffsll((long long)bad) - 1) : (64 - first));
    if (L > 0) {
        newX = pp_readlane(px, first + L - 1);
        newY = pp_readlane(py, first + L - 1);
    }
    return L;
}

// Quiet run: consecutive steps that are all coverage events (the vehicle is inside some piece's NON-strict corridor,
// so minDistanceFrom is 0 and the next step is an event again) but change nothing — no piece is strictly contained,
// or cover() is not enabled at that step (Edge.cpp:159).  Same guarded, lanes-as-steps evaluation as the corridor run:
// a step joins the run only if "inside" is certain and "nothing splits" is certain.  Returns the run length.
// Long runs (ell > 0, see pp_corridor_run): a sample vouches for the steps between the previous sample and itself when it lies
// inside a piece with ell to spare and no piece could split anywhere within ell of it — or cover() is off at ALL of those steps
// (bit i of coverMask: cover() runs at SOME step sample i vouches for).
__device__ inline int pp_quiet_run(const PPRibbon& r, int n, double w, double x, double y, bool stepOk,
                                   unsigned long long coverMask, int first, double span, double ell = 0.0, double sinDt = 0.0) {
    const int lane = pp_lane();
    const double g = 1e-9;
    // long runs: decisions on the chord from the sample before, with the sagitta to spare at both of its ends (see pp_corridor_run)
    const double sagL = (ell > 0.0) ? (ell * sinDt * 0.125) * (1.0 + 1e-9) + 1e-9 : 0.0;
    const bool cand = stepOk & (lane >= first);
    bool inside = false, maySplit = false;
    unsigned long long pieces = pp_pieces_in_reach(r, n, w, pp_readlane(x, first), pp_readlane(y, first), span);
    while (pieces) {
        const int q = __ffsll((long long)pieces) - 1;
        pieces &= pieces - 1;
        const double sx = pp_readlane(r.sx, q), sy = pp_readlane(r.sy, q), ex = pp_readlane(r.ex, q), ey = pp_readlane(r.ey, q);
        const double mx = 0.5 * (sx + ex), my = 0.5 * (sy + ey);
        const double ql = pp_sq_len(sx, sy, ex, ey);
        const double reach = 0.5 * sqrt(ql) + w + 1e-3 + ell;
        const bool far = pp_sq_len(mx, my, x, y) > reach * reach;
        if (__ballot(cand & !far) != 0ull) {
            const double dq = ex - sx, eq = ey - sy;
            const double dt = (x - sx) * dq + (y - sy) * eq;
            const double ppx = dq * dt / ql + sx, ppy = eq * dt / ql + sy;
            const double c1 = ppx - sx, c2 = ppx - ex, d1 = ppy - sy, d2 = ppy - ey;
            const double Ti = PP_RIBBON_TOL - g, To = PP_RIBBON_TOL + g + sagL;
            bool cpIn = !((((c1 < -Ti) & (c2 < -Ti)) | ((c1 > Ti) & (c2 > Ti))) | (((d1 < -Ti) & (d2 < -Ti)) | ((d1 > Ti) & (d2 > Ti))));
            if (ell > 0.0) {
                // inside the piece's extent by the sagitta, measured along the piece (dt / |piece| = distance of the projection from
                // the start): every pose within it then projects inside the extent, where the reference's per-coordinate test passes
                const double marginLen = (sagL + 1e-6) * sqrt(ql);
                cpIn = (dt > marginLen) & ((ql - dt) > marginLen);
            }
            const int outCode = (((c1 < -To) & (c2 < -To)) ? 1 : 0) | (((c1 > To) & (c2 > To)) ? 2 : 0) | (((d1 < -To) & (d2 < -To)) ? 4 : 0) | (((d1 > To) & (d2 > To)) ? 8 : 0);
            const double nq = eq * x - dq * y + ex * sy - ey * sx;
            const double A = nq * nq;
            const double wi = w - sagL, wo = w / 2.0 + sagL;
            bool in = !far & cpIn & (wi > 0.0) & (A < ((wi * wi) * ql) * (1.0 - g));     // inside this piece's corridor (a rectangle: convex)
            const bool strictOut = A > ((wo * wo) * ql) * (1.0 + g);
            bool clear = far | (outCode != 0) | strictOut;                              // cannot be strictly inside this piece
            if (ell > 0.0) {
                // both this sample and the one before it, in the same piece / beyond the same face of it: then every step between them
                const bool prevIn = __shfl_up((int)in, 1, PP_WAVE) != 0;
                const int prevCode = __shfl_up(outCode, 1, PP_WAVE);
                const bool prevStrictOut = __shfl_up((int)strictOut, 1, PP_WAVE) != 0, prevPos = __shfl_up((int)(nq > 0.0), 1, PP_WAVE) != 0;
                // (far: farther from the piece than its reach plus ell, so every step within ell of this sample is out of its reach)
                const bool pairClear = far | ((outCode & prevCode) != 0) | (strictOut & prevStrictOut & ((nq > 0.0) == prevPos));
                if (lane != first) { in = in & prevIn; clear = pairClear; }
            }
            inside = inside | in;
            maySplit = maySplit | !clear;
        }
    }
    const bool coverOn = ((coverMask >> lane) & 1ull) != 0ull;
    const bool erasable = pp_any_erasable_piece(r, n, w);              // a cover() call at such a step would change the list
    const bool ok = cand & inside & !((maySplit | erasable) & coverOn);
    const unsigned long long bad = (~__ballot(ok)) >> first;
    return first >= 64 ? 0 : (bad ? (__ffsll((long long)bad) - 1) : (64 - first));
}

// ----------------------------------------------------------------------------- heuristics
// Brute-force TSP heuristics: table sizes for at most MAXN ribbons.  MAXN = 8 (remaining ribbons packed 4 bits each in 32
// bits) is the common kernel; MAXN = 12 (64 bits) serves the K variant on longer lists (see pp_tsp_big_ok).
#include <type_traits>
#define PP_TSP_MAX 8
#define PP_TSP_MAX_BIG 12
template <int MAXN>
struct PPTsp {
    static constexpr int MAX = MAXN;
    static constexpr int PTS = 2 * MAXN + 1;                                   // query point + both endpoints of every ribbon
    static constexpr int LDS = PP_WAVE * 2 + PTS * (PTS - 1) + PTS * MAXN;     // doubles of LDS per wave: points (MaxDistance uses up
                                                                              // to 129 of them: [0, 258)), distance table, KM table
    typedef typename std::conditional<(MAXN <= 8), unsigned, unsigned long long>::type Ord;
    static __device__ __forceinline__ Ord identity() { return (Ord)0xBA9876543210ull; }   // ribbon i at position i
};

// Every distance the heuristics need is between two of the points {query point, ribbon endpoints}; the TSP enumeration
// only ever stands on one of those points.  So all sqrt() are taken once, lane-parallel, into a table
//     T[p][q-1] = sqrt((xp - xq)^2 + (yp - yq)^2),   p in [0, 2n], q in [1, 2n]   (point 0 = query, 1+2i / 2+2i = start / end of ribbon i)
// (the same expression as RibbonManager::distance, RibbonManager.h:285-287, and Ribbon::length(): (a-b)^2 == (b-a)^2 exactly),
// and the enumeration itself is lookups, adds and compares.
template <int MAXN>
__device__ __forceinline__ double pp_h_T(const double* T, int p, int q) { return T[p * (PPTsp<MAXN>::PTS - 1) + (q - 1)]; }

// Number of lane-parallel prefixes pp_h_tsp_point uses for n ribbons and branching K (and how many levels they span)
__device__ __forceinline__ unsigned long long pp_tsp_prefixes(int n, int K, int& Ls) {
    unsigned long long NP = 1;
    Ls = 0;
    for (int l = 0; l < n; l++) {
        const int rem = n - l;
        if (NP >= 64ull && l >= n - 3) break;
        NP *= (unsigned long long)(2 * (rem < K ? rem : K));
        Ls = l + 1;
    }
    return NP;
}
// Lists of 9..12 ribbons are enumerated (by the MAXN = 12 kernel) only for the K variant of the point-robot heuristic and
// only while the prefix count stays below 2^21 (K = 2: up to 12 ribbons, K = 3: up to 10); everything else beyond 8 ribbons
// is reported as PPGPU_F_RIBBON_OVF.  (The All variants would need n! 2^n leaves: 1.9e8 at n = 9.)
__device__ __forceinline__ bool pp_tsp_big_ok(int heuristic, int K, int n) {
    if (heuristic != PPGPU_H_TSP_POINT_K || n <= PP_TSP_MAX || n > PP_TSP_MAX_BIG) return false;
    if (K <= 0) return true;
    int Ls;
    return pp_tsp_prefixes(n, K, Ls) < (1ull << 21);
}

// pp_k_heuristic_lanes (pp_kernels.h): the TSP heuristics with one lane per edge, for edges that leave their ribbons untouched
#define PP_H_DEFERRED (-1.0)        // what the cover sweep writes into h for such an edge (a heuristic is never negative)
#define PP_HL_MAX_LEAVES 4096
// how many leaves the enumeration of n ribbons has (branching 2 * min(K, remaining) per level)
__device__ __forceinline__ unsigned pp_lane_tsp_leaves(int n, int K) {
    unsigned long long L = 1;
    for (int rem = n; rem >= 1 && L <= (1ull << 20); rem--) L *= (unsigned long long)(2 * (rem < K ? rem : K));
    return L > (1ull << 20) ? (1u << 20) : (unsigned)L;
}
// may the cover sweep leave a child list of n ribbons to pp_k_heuristic_lanes?
// (Round 3 measured the same walk with tables for 8 ribbons on the lists of 7 and 8 — ~1 100 of config 3's 236 140 edges, a whole
// wave's work for ~170 us each: four lanes per edge took 404 us for them, the wave-uniform cut rarely fires on such deep trees and
// each lane walks a quarter of up to 32 768 leaves alone.  They stay with pp_k_heuristic_listed, a wave each, beside this kernel.)
#define PP_HL_MAX_N 6
static_assert(PP_HL_MAX_N == 6, "pp_k_heuristic_lanes dispatches on n = 1 .. 6");
__device__ __forceinline__ bool pp_lane_tsp_ok(int heuristic, int tsp_k, int n) {
    if (n < 1 || n > PP_HL_MAX_N) return false;
    if (heuristic == PPGPU_H_TSP_POINT_ALL) return pp_lane_tsp_leaves(n, PP_TSP_MAX) <= PP_HL_MAX_LEAVES;
    if (heuristic == PPGPU_H_TSP_POINT_K) return tsp_k > 0 && pp_lane_tsp_leaves(n, tsp_k) <= PP_HL_MAX_LEAVES;
    return false;
}

// RibbonManager::maxDistance (RibbonManager.cpp:234-248); pts = x,y of the query point then of every ribbon's start, end
__device__ inline double pp_h_max_distance(const double* pts, int n, double w) {
    const double x = pts[0], y = pts[1];
    double sumLength = 0, mn = PP_DBL_MAX, mx = 0;
    for (int i = 0; i < n; i++) {
        const double sx = pts[2 * (1 + 2 * i)], sy = pts[2 * (1 + 2 * i) + 1], ex = pts[2 * (2 + 2 * i)], ey = pts[2 * (2 + 2 * i) + 1];
        sumLength += sqrt(pp_sq_len(sx, sy, ex, ey)) - 2 * w;
        double dStart = pp_dist(sx, sy, x, y);
        double dEnd = pp_dist(ex, ey, x, y);
        mn = fmin(fmin(mn, dEnd), dStart);
        mx = fmax(fmax(mx, dEnd), dStart);
    }
    return fmax(sumLength + mn, mx);
}

// RibbonManager::tspPointRobotNoSplitAllRibbons (:53-67) and ...KRibbons (:69-94), wave-parallel.
//
// The reference is a depth-first enumeration: at each level it (K variant only) stable-sorts the
// remaining ribbons by DESCENDING nearest-endpoint distance from the current point
// (list::sort with comp = min1 > min2), branches on the first min(K, n) of them in both
// directions, and returns the min over leaves of the accumulated
//   soFar' = fmax(soFar + len - 2w + dist(point, entry endpoint), 0).
// fmin/fmax are exact, so any evaluation order gives the same bits.  Here the first Ls levels
// (the "prefix", chosen so that there are >= 64 prefixes when the tree is that large) are spread
// over the lanes — every lane walks ITS prefix with the same control flow, only the digits differ —
// and the remaining (at most three) levels are enumerated by wave-uniform loops, so that no lane ever
// waits for another lane's branch.  Each tree node is sorted once.
template <int MAXN>
struct PPTspNode { double sf; typename PPTsp<MAXN>::Ord ord; int pt; };   // accumulated distance, remaining ribbons (4 bits each), current point

// The order list::sort(comp = min1 > min2) leaves: element i goes to position
//   #{ j : key_j > key_i }  +  #{ j before i : key_j == key_i }          (stable, descending)
// computed as ranks, so nothing is swapped.  key_i = distance from point `pt` to the nearer endpoint of ribbon i,
// read from the table KM[pt][i] = fmin(T[pt][start_i], T[pt][end_i]) built next to T.
template <int MAXN, int REM>
__device__ __forceinline__ typename PPTsp<MAXN>::Ord pp_tsp_sort_n(const double* KM, typename PPTsp<MAXN>::Ord ord, int pt) {
    typedef typename PPTsp<MAXN>::Ord Ord;
    double key[REM];
#pragma unroll
    for (int i = 0; i < REM; i++) key[i] = KM[pt * MAXN + (int)((ord >> (4 * i)) & 0xfu)];
    Ord o = 0;
#pragma unroll
    for (int i = 0; i < REM; i++) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < REM; j++) {
            if (j != i) rank += ((key[j] > key[i]) | ((key[j] == key[i]) & (j < i))) ? 1 : 0;
        }
        o |= ((ord >> (4 * i)) & (Ord)0xfu) << (4 * rank);
    }
    return o;
}
// the same ranks with the keys re-read from the table: only for the rare lists of more than 8 remaining ribbons
template <int MAXN>
__device__ __noinline__ typename PPTsp<MAXN>::Ord pp_tsp_sort_loop(const double* KM, typename PPTsp<MAXN>::Ord ord, int rem, int pt) {
    typedef typename PPTsp<MAXN>::Ord Ord;
    Ord o = 0;
    for (int i = 0; i < rem; i++) {
        const double ki = KM[pt * MAXN + (int)((ord >> (4 * i)) & 0xfu)];
        int rank = 0;
        for (int j = 0; j < rem; j++) {
            const double kj = KM[pt * MAXN + (int)((ord >> (4 * j)) & 0xfu)];
            if (j != i) rank += ((kj > ki) | ((kj == ki) & (j < i))) ? 1 : 0;
        }
        o |= ((ord >> (4 * i)) & (Ord)0xfu) << (4 * rank);
    }
    return o;
}
// `rem` is wave-uniform: one branch picks the network of exactly that size (rem * (rem - 1) comparisons instead of 56)
template <int MAXN>
__device__ __forceinline__ typename PPTsp<MAXN>::Ord pp_tsp_sort(const double* KM, typename PPTsp<MAXN>::Ord ord, int rem, int pt) {
    switch (rem) {
        case 0: case 1: return ord;  // nothing to order
        case 2: return pp_tsp_sort_n<MAXN, 2>(KM, ord, pt);
        case 3: return pp_tsp_sort_n<MAXN, 3>(KM, ord, pt);
        case 4: return pp_tsp_sort_n<MAXN, 4>(KM, ord, pt);
        case 5: return pp_tsp_sort_n<MAXN, 5>(KM, ord, pt);
        case 6: return pp_tsp_sort_n<MAXN, 6>(KM, ord, pt);
        case 7: return pp_tsp_sort_n<MAXN, 7>(KM, ord, pt);
        case 8: return pp_tsp_sort_n<MAXN, 8>(KM, ord, pt);
        default: break;
    }
    if constexpr (MAXN > 8) return pp_tsp_sort_loop<MAXN>(KM, ord, rem, pt);
    return ord;
}

// take branch `digit` (ribbon position digit>>1 of `srt`, direction digit&1) from node `a`
// LEN == nullptr: T holds point-to-point distances and a ribbon's length is T[start][end]; otherwise T holds the Dubins
// distances between oriented endpoints (RibbonManager::dubinsDistance) and LEN[i] = Ribbon::length() of ribbon i.
template <int MAXN>
__device__ __forceinline__ PPTspNode<MAXN> pp_tsp_child(const double* T, const double* LEN, const PPTspNode<MAXN>& a,
                                                        typename PPTsp<MAXN>::Ord srt, int digit, double twoW) {
    typedef typename PPTsp<MAXN>::Ord Ord;
    const int c = digit >> 1, dir = digit & 1;
    const int rid = (int)((srt >> (4 * c)) & 0xfu);
    const int ps = 1 + 2 * rid, pe = 2 + 2 * rid;
    const double len = LEN ? LEN[rid] : pp_h_T<MAXN>(T, ps, pe);        // Ribbon::length()
    const double dd = pp_h_T<MAXN>(T, a.pt, dir == 0 ? ps : pe);        // distance(point, r.start()) / (point, r.end())
    PPTspNode<MAXN> b;
    b.sf = fmax(a.sf + len - twoW + dd, 0);
    b.pt = dir == 0 ? pe : ps;
    const Ord lowmask = (c == 0) ? (Ord)0 : ((((Ord)1) << (4 * c)) - (Ord)1);
    b.ord = (srt & lowmask) | ((srt >> 4) & ~lowmask);
    return b;
}

template <int MAXN>
// passFirst / passStride: this wave takes the passes (of 64 prefixes) passFirst, passFirst + passStride, ... — several waves can share one
// list, each with its own copy of the tables, and the minimum of their results is the result (pp_k_heuristic_listed).
__device__ inline double pp_h_tsp_point(const double* T, const double* KM, int n, double w, int K, bool sortK, const double* LEN = nullptr,
                                        unsigned passFirst = 0u, unsigned passStride = 1u) {
    if (n == 0) return 0;
    if (K <= 0) return PP_DBL_MAX;   // the reference's loop body never runs and it returns DBL_MAX
    const int lane = pp_lane();
    const double twoW = 2 * w;
    // prefix depth Ls: deep enough for >= 64 prefixes when the tree has them, and never leaving more than
    // two suffix levels (those are enumerated by wave-uniform loops below).  All of this is wave-uniform.
    int Ls = 0;
    const unsigned NP = (unsigned)pp_tsp_prefixes(n, K, Ls);     // < 2^21 by the callers' limits
    const int nsuf = n - Ls;         // 0 .. 3
    double best = PP_DBL_MAX;
    for (unsigned pbase = 64u * passFirst; pbase < NP; pbase += 64u * passStride) {
        const unsigned pid = pbase + (unsigned)lane;
        const bool act = pid < NP;
        unsigned rest = act ? pid : 0u;
        unsigned stride = NP;
        PPTspNode<MAXN> a;
        a.pt = 0; a.sf = 0; a.ord = PPTsp<MAXN>::identity();
        for (int l = 0; l < Ls; l++) {                      // this lane's prefix, level 0 = most significant digit
            const int rem = n - l;
            const unsigned b = (unsigned)(2 * (rem < K ? rem : K));
            stride = pp_udiv_small(stride, b);              // exact: NP is the product of the b's
            const unsigned dg = pp_udiv_small(rest, stride);
            rest -= dg * stride;
            const typename PPTsp<MAXN>::Ord srt = sortK ? pp_tsp_sort<MAXN>(KM, a.ord, rem, a.pt) : a.ord;
            a = pp_tsp_child<MAXN>(T, LEN, a, srt, (int)dg, twoW);
        }
        // the remaining (at most three) levels: wave-uniform loops, every lane below its own prefix node
        double v = PP_DBL_MAX;
        if (nsuf == 0) {
            v = a.sf;
        } else {
            const int remA = nsuf;
            const int bA = 2 * (remA < K ? remA : K);
            const typename PPTsp<MAXN>::Ord srtA = sortK ? pp_tsp_sort<MAXN>(KM, a.ord, remA, a.pt) : a.ord;
            for (int uA = 0; uA < bA; uA++) {
                const PPTspNode<MAXN> nb = pp_tsp_child<MAXN>(T, LEN, a, srtA, uA, twoW);
                if (nsuf == 1) { v = fmin(v, nb.sf); continue; }
                const int remB = remA - 1;
                const int bB = 2 * (remB < K ? remB : K);
                const typename PPTsp<MAXN>::Ord srtB = sortK ? pp_tsp_sort<MAXN>(KM, nb.ord, remB, nb.pt) : nb.ord;
                for (int uB = 0; uB < bB; uB++) {
                    const PPTspNode<MAXN> nc = pp_tsp_child<MAXN>(T, LEN, nb, srtB, uB, twoW);
                    if (nsuf == 2) { v = fmin(v, nc.sf); continue; }
                    const int remC = remB - 1;
                    const int bC = 2 * (remC < K ? remC : K);
                    const typename PPTsp<MAXN>::Ord srtC = sortK ? pp_tsp_sort<MAXN>(KM, nc.ord, remC, nc.pt) : nc.ord;
                    for (int uC = 0; uC < bC; uC++) v = fmin(v, pp_tsp_child<MAXN>(T, LEN, nc, srtC, uC, twoW).sf);
                }
            }
        }
        if (act) best = fmin(best, v);
    }
    return pp_wave_min(best);
}
