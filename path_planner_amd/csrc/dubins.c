/*
 * dubins.c — the C Dubins-curve library behind include/dubins.h (host side).
 *
 * Written from scratch: the reference links a third-party catkin package, `dubins_curves`
 * (path_planner/package.xml:31, no version pinned), that is not part of its tree.  What is
 * implemented is the published classification of the Dubins set (Shkel & Lumelsky 2001) in
 * its usual normalised form:
 *
 *   - translate/rotate so that q0 sits at the origin looking along the chord; with d the
 *     chord length in units of rho, alpha/beta the two headings relative to the chord,
 *     each of the six words (LSL LSR RSL RSR RLR LRL) has closed-form segment lengths
 *     (t, p, q), or no solution;
 *   - the shortest path is the feasible word with the smallest t + p + q, words tried in
 *     enum order and replaced only on a strictly smaller sum;
 *   - sampling walks the three segments with the unit-radius primitives
 *       L: (x + sin(th + s) - sin th, y - cos(th + s) + cos th, th + s)
 *       R: (x - sin(th - s) + sin th, y + cos(th - s) - cos th, th - s)
 *       S: (x + s cos th, y + s sin th, th)
 *     and scales by rho.
 *
 * The device code (pp_device.h: pp_dubins_shortest / pp_curve_sample) evaluates exactly
 * these expressions in the same order, so that host-built and device-built curves agree.
 * Build: plain C99, no FMA contraction (-ffp-contract=off).
 */
#include "../../include/dubins.h"

#include <math.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef enum { SEG_L = 0, SEG_S = 1, SEG_R = 2 } seg_t;

static const seg_t WORD_SEGMENTS[6][3] = {
    {SEG_L, SEG_S, SEG_L}, {SEG_L, SEG_S, SEG_R}, {SEG_R, SEG_S, SEG_L},
    {SEG_R, SEG_S, SEG_R}, {SEG_R, SEG_L, SEG_R}, {SEG_L, SEG_R, SEG_L}};

typedef struct {
    double alpha, beta, d, sa, sb, ca, cb, c_ab, d_sq;
} norm_t;

static double wrap_2pi(double theta) {
    const double two_pi = 2 * M_PI;
    return theta - two_pi * floor(theta / two_pi);
}

static int normalise(norm_t* n, const double q0[3], const double q1[3], double rho) {
    double dx, dy, D, d, theta, alpha, beta;
    if (rho <= 0.0) return EDUBBADRHO;
    dx = q1[0] - q0[0];
    dy = q1[1] - q0[1];
    D = sqrt(dx * dx + dy * dy);
    d = D / rho;
    theta = 0;
    if (d > 0) theta = wrap_2pi(atan2(dy, dx)); /* atan2(0,0) avoided for coincident points */
    alpha = wrap_2pi(q0[2] - theta);
    beta = wrap_2pi(q1[2] - theta);
    n->alpha = alpha;
    n->beta = beta;
    n->d = d;
    n->sa = sin(alpha);
    n->sb = sin(beta);
    n->ca = cos(alpha);
    n->cb = cos(beta);
    n->c_ab = cos(alpha - beta);
    n->d_sq = d * d;
    return EDUBOK;
}

static int solve_word(const norm_t* n, DubinsPathType type, double out[3]) {
    double tmp0, tmp1, p_sq, p, phi, t;
    switch (type) {
    case LSL:
        tmp0 = n->d + n->sa - n->sb;
        p_sq = 2 + n->d_sq - (2 * n->c_ab) + (2 * n->d * (n->sa - n->sb));
        if (p_sq < 0) return EDUBNOPATH;
        tmp1 = atan2((n->cb - n->ca), tmp0);
        out[0] = wrap_2pi(tmp1 - n->alpha);
        out[1] = sqrt(p_sq);
        out[2] = wrap_2pi(n->beta - tmp1);
        return EDUBOK;
    case RSR:
        tmp0 = n->d - n->sa + n->sb;
        p_sq = 2 + n->d_sq - (2 * n->c_ab) + (2 * n->d * (n->sb - n->sa));
        if (p_sq < 0) return EDUBNOPATH;
        tmp1 = atan2((n->ca - n->cb), tmp0);
        out[0] = wrap_2pi(n->alpha - tmp1);
        out[1] = sqrt(p_sq);
        out[2] = wrap_2pi(tmp1 - n->beta);
        return EDUBOK;
    case LSR:
        p_sq = -2 + (n->d_sq) + (2 * n->c_ab) + (2 * n->d * (n->sa + n->sb));
        if (p_sq < 0) return EDUBNOPATH;
        p = sqrt(p_sq);
        tmp0 = atan2((-n->ca - n->cb), (n->d + n->sa + n->sb)) - atan2(-2.0, p);
        out[0] = wrap_2pi(tmp0 - n->alpha);
        out[1] = p;
        out[2] = wrap_2pi(tmp0 - wrap_2pi(n->beta));
        return EDUBOK;
    case RSL:
        p_sq = -2 + n->d_sq + (2 * n->c_ab) - (2 * n->d * (n->sa + n->sb));
        if (p_sq < 0) return EDUBNOPATH;
        p = sqrt(p_sq);
        tmp0 = atan2((n->ca + n->cb), (n->d - n->sa - n->sb)) - atan2(2.0, p);
        out[0] = wrap_2pi(n->alpha - tmp0);
        out[1] = p;
        out[2] = wrap_2pi(n->beta - tmp0);
        return EDUBOK;
    case RLR:
        tmp0 = (6. - n->d_sq + 2 * n->c_ab + 2 * n->d * (n->sa - n->sb)) / 8.;
        phi = atan2(n->ca - n->cb, n->d - n->sa + n->sb);
        if (fabs(tmp0) > 1) return EDUBNOPATH;
        p = wrap_2pi((2 * M_PI) - acos(tmp0));
        t = wrap_2pi(n->alpha - phi + wrap_2pi(p / 2.));
        out[0] = t;
        out[1] = p;
        out[2] = wrap_2pi(n->alpha - n->beta - t + wrap_2pi(p));
        return EDUBOK;
    case LRL:
        tmp0 = (6. - n->d_sq + 2 * n->c_ab + 2 * n->d * (n->sb - n->sa)) / 8.;
        phi = atan2(n->ca - n->cb, n->d + n->sa - n->sb);
        if (fabs(tmp0) > 1) return EDUBNOPATH;
        p = wrap_2pi(2 * M_PI - acos(tmp0));
        t = wrap_2pi(-n->alpha - phi + p / 2.);
        out[0] = t;
        out[1] = p;
        out[2] = wrap_2pi(wrap_2pi(n->beta) - n->alpha - t + wrap_2pi(p));
        return EDUBOK;
    }
    return EDUBNOPATH;
}

int dubins_shortest_path(DubinsPath* path, double q0[3], double q1[3], double rho) {
    norm_t n;
    double params[3], cost, best_cost = INFINITY;
    int i, best_word = -1;
    int err = normalise(&n, q0, q1, rho);
    if (err != EDUBOK) return err;
    path->qi[0] = q0[0];
    path->qi[1] = q0[1];
    path->qi[2] = q0[2];
    path->rho = rho;
    for (i = 0; i < 6; i++) {
        if (solve_word(&n, (DubinsPathType)i, params) == EDUBOK) {
            cost = params[0] + params[1] + params[2];
            if (cost < best_cost) {
                best_word = i;
                best_cost = cost;
                path->param[0] = params[0];
                path->param[1] = params[1];
                path->param[2] = params[2];
                path->type = (DubinsPathType)i;
            }
        }
    }
    return best_word == -1 ? EDUBNOPATH : EDUBOK;
}

double dubins_path_length(const DubinsPath* path) {
    double length = 0.;
    length += path->param[0];
    length += path->param[1];
    length += path->param[2];
    length = length * path->rho;
    return length;
}

static void advance(double s, const double from[3], double to[3], seg_t type) {
    double st = sin(from[2]);
    double ct = cos(from[2]);
    if (type == SEG_L) {
        to[0] = +sin(from[2] + s) - st;
        to[1] = -cos(from[2] + s) + ct;
        to[2] = s;
    } else if (type == SEG_R) {
        to[0] = -sin(from[2] - s) + st;
        to[1] = +cos(from[2] - s) - ct;
        to[2] = -s;
    } else {
        to[0] = ct * s;
        to[1] = st * s;
        to[2] = 0.0;
    }
    to[0] += from[0];
    to[1] += from[1];
    to[2] += from[2];
}

int dubins_path_sample(const DubinsPath* path, double t, double q[3]) {
    double tprime = t / path->rho;
    double origin[3], end1[3], end2[3], p1, p2;
    const seg_t* seg = WORD_SEGMENTS[path->type];
    if (t < 0 || t > dubins_path_length(path)) return EDUBPARAM;
    origin[0] = 0.0;
    origin[1] = 0.0;
    origin[2] = path->qi[2];
    p1 = path->param[0];
    p2 = path->param[1];
    advance(p1, origin, end1, seg[0]);
    advance(p2, end1, end2, seg[1]);
    if (tprime < p1) advance(tprime, origin, q, seg[0]);
    else if (tprime < (p1 + p2)) advance(tprime - p1, end1, q, seg[1]);
    else advance(tprime - p1 - p2, end2, q, seg[2]);
    q[0] = q[0] * path->rho + path->qi[0];
    q[1] = q[1] * path->rho + path->qi[1];
    q[2] = wrap_2pi(q[2]);
    return EDUBOK;
}

/* The prefix [0, t].  The reference calls this from DubinsWrapper::updateStartTime
 * (DubinsWrapper.cpp:106-115) as if it returned the SUFFIX; that member is off the hot path
 * (one unit test, one commented-out call) and its semantics against the original library are
 * unpinned — see DESIGN.md. */
int dubins_extract_subpath(const DubinsPath* path, double t, DubinsPath* newpath) {
    double tprime = t / path->rho;
    if ((t < 0) || (t > dubins_path_length(path))) return EDUBPARAM;
    newpath->qi[0] = path->qi[0];
    newpath->qi[1] = path->qi[1];
    newpath->qi[2] = path->qi[2];
    newpath->rho = path->rho;
    newpath->type = path->type;
    newpath->param[0] = fmin(path->param[0], tprime);
    newpath->param[1] = fmin(path->param[1], tprime - newpath->param[0]);
    newpath->param[2] = fmin(path->param[2], tprime - newpath->param[0] - newpath->param[1]);
    return EDUBOK;
}
