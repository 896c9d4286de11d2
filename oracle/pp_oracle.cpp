// pp_oracle.cpp — TEST INFRASTRUCTURE ONLY (see pp_oracle.hpp header comment).
//
// Build: g++ -std=c++17 -O2 -ffp-contract=off (no -ffast-math, no -march=native): the
// reference is built for baseline x86-64 (pp/CMakeLists.txt:4-6 sets only -std=c++11
// -pthread), i.e. without fused multiply-add, so products and sums round separately.
#include "pp_oracle.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace ppo {

// ============================================================================ State
double State::yaw() const {  // ppc State.h:51-55
    double h = M_PI_2 - heading;
    if (h < 0) h += 2 * M_PI;
    return h;
}
void State::setYaw(double yaw1) {  // ppc State.h:62-65
    heading = M_PI_2 - yaw1;
    if (heading < 0) heading += 2 * M_PI;
}
void State::move(double distance) {  // ppc State.cpp:22-25
    x += std::cos(yaw()) * distance;
    y += std::sin(yaw()) * distance;
}
State State::push(double dt) const {  // ppc State.cpp:11-20
    State s;
    double displacement = dt * speed;
    s.x = x + std::sin(heading) * displacement;
    s.y = y + std::cos(heading) * displacement;
    s.heading = heading;
    s.speed = speed;
    s.time = time + dt;
    return s;
}
double State::headingTo(double x1, double y1) const {  // ppc State.cpp:51-57
    double dx = x1 - x;
    double dy = y1 - y;
    double h = M_PI_2 - std::atan2(dy, dx);
    if (h < 0) h += 2 * M_PI;
    return h;
}
void State::setHeadingTowards(double x1, double y1) {  // ppc State.cpp:64-67
    heading = headingTo(x1, y1);
    if (heading < 0) heading += 2 * M_PI;
}
double State::distanceTo(double x1, double y1) const {  // ppc State.cpp:91-93
    return std::sqrt((x - x1) * (x - x1) + (y - y1) * (y - y1));
}

// ============================================================================ Dubins
// Third-party `dubins_curves` (absent, unpinned; pp/package.xml:31).  Restated from the
// published algorithm: Shkel & Lumelsky, "Classification of the Dubins set" (2001), in
// the normalised (alpha, beta, d) form; words scanned LSL,LSR,RSL,RSR,RLR,LRL, strict <.
static inline double fmodr(double x, double y) { return x - y * std::floor(x / y); }
static inline double mod2pi(double theta) { return fmodr(theta, 2 * M_PI); }

namespace {
struct Inter { double alpha, beta, d, sa, sb, ca, cb, c_ab, d_sq; };
enum Seg { L_SEG = 0, S_SEG = 1, R_SEG = 2 };
const int DIRDATA[6][3] = {{L_SEG, S_SEG, L_SEG}, {L_SEG, S_SEG, R_SEG}, {R_SEG, S_SEG, L_SEG},
                           {R_SEG, S_SEG, R_SEG}, {R_SEG, L_SEG, R_SEG}, {L_SEG, R_SEG, L_SEG}};

int intermediate(Inter* in, const double q0[3], const double q1[3], double rho) {
    if (rho <= 0.0) return EDUBBADRHO;
    double dx = q1[0] - q0[0];
    double dy = q1[1] - q0[1];
    double D = std::sqrt(dx * dx + dy * dy);
    double d = D / rho;
    double theta = 0;
    if (d > 0) theta = mod2pi(std::atan2(dy, dx));
    double alpha = mod2pi(q0[2] - theta);
    double beta = mod2pi(q1[2] - theta);
    in->alpha = alpha; in->beta = beta; in->d = d;
    in->sa = std::sin(alpha); in->sb = std::sin(beta);
    in->ca = std::cos(alpha); in->cb = std::cos(beta);
    in->c_ab = std::cos(alpha - beta);
    in->d_sq = d * d;
    return EDUBOK;
}

int word(const Inter* in, int type, double out[3]) {
    switch (type) {
    case 0: {  // LSL
        double tmp0 = in->d + in->sa - in->sb;
        double p_sq = 2 + in->d_sq - (2 * in->c_ab) + (2 * in->d * (in->sa - in->sb));
        if (p_sq >= 0) {
            double tmp1 = std::atan2((in->cb - in->ca), tmp0);
            out[0] = mod2pi(tmp1 - in->alpha);
            out[1] = std::sqrt(p_sq);
            out[2] = mod2pi(in->beta - tmp1);
            return EDUBOK;
        }
        return EDUBNOPATH;
    }
    case 3: {  // RSR
        double tmp0 = in->d - in->sa + in->sb;
        double p_sq = 2 + in->d_sq - (2 * in->c_ab) + (2 * in->d * (in->sb - in->sa));
        if (p_sq >= 0) {
            double tmp1 = std::atan2((in->ca - in->cb), tmp0);
            out[0] = mod2pi(in->alpha - tmp1);
            out[1] = std::sqrt(p_sq);
            out[2] = mod2pi(tmp1 - in->beta);
            return EDUBOK;
        }
        return EDUBNOPATH;
    }
    case 1: {  // LSR
        double p_sq = -2 + (in->d_sq) + (2 * in->c_ab) + (2 * in->d * (in->sa + in->sb));
        if (p_sq >= 0) {
            double p = std::sqrt(p_sq);
            double tmp0 = std::atan2((-in->ca - in->cb), (in->d + in->sa + in->sb)) - std::atan2(-2.0, p);
            out[0] = mod2pi(tmp0 - in->alpha);
            out[1] = p;
            out[2] = mod2pi(tmp0 - mod2pi(in->beta));
            return EDUBOK;
        }
        return EDUBNOPATH;
    }
    case 2: {  // RSL
        double p_sq = -2 + in->d_sq + (2 * in->c_ab) - (2 * in->d * (in->sa + in->sb));
        if (p_sq >= 0) {
            double p = std::sqrt(p_sq);
            double tmp0 = std::atan2((in->ca + in->cb), (in->d - in->sa - in->sb)) - std::atan2(2.0, p);
            out[0] = mod2pi(in->alpha - tmp0);
            out[1] = p;
            out[2] = mod2pi(in->beta - tmp0);
            return EDUBOK;
        }
        return EDUBNOPATH;
    }
    case 4: {  // RLR
        double tmp0 = (6. - in->d_sq + 2 * in->c_ab + 2 * in->d * (in->sa - in->sb)) / 8.;
        double phi = std::atan2(in->ca - in->cb, in->d - in->sa + in->sb);
        if (std::fabs(tmp0) <= 1) {
            double p = mod2pi((2 * M_PI) - std::acos(tmp0));
            double t = mod2pi(in->alpha - phi + mod2pi(p / 2.));
            out[0] = t;
            out[1] = p;
            out[2] = mod2pi(in->alpha - in->beta - t + mod2pi(p));
            return EDUBOK;
        }
        return EDUBNOPATH;
    }
    case 5: {  // LRL
        double tmp0 = (6. - in->d_sq + 2 * in->c_ab + 2 * in->d * (in->sb - in->sa)) / 8.;
        double phi = std::atan2(in->ca - in->cb, in->d + in->sa - in->sb);
        if (std::fabs(tmp0) <= 1) {
            double p = mod2pi(2 * M_PI - std::acos(tmp0));
            double t = mod2pi(-in->alpha - phi + p / 2.);
            out[0] = t;
            out[1] = p;
            out[2] = mod2pi(mod2pi(in->beta) - in->alpha - t + mod2pi(p));
            return EDUBOK;
        }
        return EDUBNOPATH;
    }
    default: return EDUBNOPATH;
    }
}

void segment(double t, const double qi[3], double qt[3], int type) {
    double st = std::sin(qi[2]);
    double ct = std::cos(qi[2]);
    if (type == L_SEG) {
        qt[0] = +std::sin(qi[2] + t) - st;
        qt[1] = -std::cos(qi[2] + t) + ct;
        qt[2] = t;
    } else if (type == R_SEG) {
        qt[0] = -std::sin(qi[2] - t) + st;
        qt[1] = +std::cos(qi[2] - t) - ct;
        qt[2] = -t;
    } else {
        qt[0] = ct * t;
        qt[1] = st * t;
        qt[2] = 0.0;
    }
    qt[0] += qi[0];
    qt[1] += qi[1];
    qt[2] += qi[2];
}
}  // namespace

int dubins_word(int type, const double q0[3], const double q1[3], double rho, double out[3]) {
    Inter in;
    int e = intermediate(&in, q0, q1, rho);
    if (e != EDUBOK) return e;
    return word(&in, type, out);
}

int dubins_shortest_path(DubinsPath* path, const double q0[3], const double q1[3], double rho) {
    Inter in;
    int errcode = intermediate(&in, q0, q1, rho);
    if (errcode != EDUBOK) return errcode;
    path->qi[0] = q0[0]; path->qi[1] = q0[1]; path->qi[2] = q0[2];
    path->rho = rho;
    double best_cost = INFINITY;
    int best_word = -1;
    for (int i = 0; i < 6; i++) {
        double params[3];
        errcode = word(&in, i, params);
        if (errcode == EDUBOK) {
            double cost = params[0] + params[1] + params[2];
            if (cost < best_cost) {
                best_word = i;
                best_cost = cost;
                path->param[0] = params[0]; path->param[1] = params[1]; path->param[2] = params[2];
                path->type = i;
            }
        }
    }
    if (best_word == -1) return EDUBNOPATH;
    return EDUBOK;
}

double dubins_path_length(const DubinsPath* path) {
    double length = 0.;
    length += path->param[0];
    length += path->param[1];
    length += path->param[2];
    length = length * path->rho;
    return length;
}

int dubins_path_sample(const DubinsPath* path, double t, double q[3]) {
    double tprime = t / path->rho;
    if (t < 0 || t > dubins_path_length(path)) return EDUBPARAM;
    const int* types = DIRDATA[path->type];
    double qi[3] = {0.0, 0.0, path->qi[2]};
    double q1[3], q2[3];
    double p1 = path->param[0];
    double p2 = path->param[1];
    segment(p1, qi, q1, types[0]);
    segment(p2, q1, q2, types[1]);
    if (tprime < p1) {
        segment(tprime, qi, q, types[0]);
    } else if (tprime < (p1 + p2)) {
        segment(tprime - p1, q1, q, types[1]);
    } else {
        segment(tprime - p1 - p2, q2, q, types[2]);
    }
    q[0] = q[0] * path->rho + path->qi[0];
    q[1] = q[1] * path->rho + path->qi[1];
    q[2] = mod2pi(q[2]);
    return EDUBOK;
}

int dubins_extract_subpath(const DubinsPath* path, double t, DubinsPath* out) {
    double tprime = t / path->rho;
    if ((t < 0) || (t > dubins_path_length(path))) return EDUBPARAM;
    out->qi[0] = path->qi[0]; out->qi[1] = path->qi[1]; out->qi[2] = path->qi[2];
    out->rho = path->rho;
    out->type = path->type;
    out->param[0] = std::fmin(path->param[0], tprime);
    out->param[1] = std::fmin(path->param[1], tprime - out->param[0]);
    out->param[2] = std::fmin(path->param[2], tprime - out->param[0] - out->param[1]);
    return EDUBOK;
}

// ============================================================================ DubinsWrapper
void DubinsWrapper::set(const State& s1, const State& s2, double rho) {  // ppc DubinsWrapper.cpp:9-17
    double q1[3] = {s1.x, s1.y, s1.yaw()};
    double q2[3] = {s2.x, s2.y, s2.yaw()};
    dubins_shortest_path(&path, q1, q2, rho);
    speed = s1.speed;
    updatedStartTime = startTime = s1.time;
    setEndTime();
}
void DubinsWrapper::fill(const DubinsPath& p, double speed_, double startTime_) {  // :85-90
    path = p;
    speed = speed_;
    updatedStartTime = startTime = startTime_;
    setEndTime();
}
double DubinsWrapper::length() const {  // :19-22
    if (!isInitialized()) throw SampleError{"Cannot access unset Dubins wrapper"};
    return dubins_path_length(&path);
}
bool DubinsWrapper::containsTime(double t) const {  // :24-27
    if (!isInitialized()) throw SampleError{"Checking time constraints on uninitialized Dubins wrapper"};
    return updatedStartTime <= t && endTime >= t;
}
void DubinsWrapper::sample(State& s) const {  // :29-49
    if (!containsTime(s.time)) throw SampleError{"Invalid time in sample for Dubins path"};
    double distance = (s.time - startTime) * speed;
    double pose[3] = {s.x, s.y, s.heading};
    int err = dubins_path_sample(&path, distance, pose);
    if (err == EDUBPARAM) err = dubins_path_sample(&path, distance - 1e-5, pose);
    s.x = pose[0]; s.y = pose[1]; s.heading = pose[2];
    s.setYaw(s.heading);
    s.speed = speed;
}
void DubinsWrapper::setEndTime() { endTime = startTime + length() / speed; }  // :92-94
void DubinsWrapper::setSpeed(double s) { speed = s; setEndTime(); }          // :121-124
void DubinsWrapper::updateEndTime(double t) {                                 // :100-104
    if (endTime == -1) throw SampleError{"Cannot access unset Dubins wrapper"};
    if (t > endTime) throw SampleError{"Invalid end time for Dubins wrapper"};
    endTime = t;
}

// ============================================================================ Ribbon
static const double c_Tolerance = 1e-5;     // Ribbon.h:129
static const double c_StrictModifier = 2;   // Ribbon.h:131
double RibbonManager::RibbonWidth = 1.5;    // Ribbon.cpp:4

double Ribbon::squaredLength() const { return (ex - sx) * (ex - sx) + (ey - sy) * (ey - sy); }  // Ribbon.h:133-135
double Ribbon::length() const { return std::sqrt(squaredLength()); }                            // Ribbon.cpp:27-29
bool Ribbon::covered(bool strict, double w) const {                                            // Ribbon.cpp:23-25
    double minLength = 2 * w;
    return squaredLength() < minLength * minLength / (strict ? c_StrictModifier * c_StrictModifier : 1);
}
void Ribbon::projection(double x, double y, double& px, double& py) const {  // Ribbon.cpp:72-78
    double squaredL = squaredLength();
    double dot = (x - sx) * (ex - sx) + (y - sy) * (ey - sy);
    double projectedX = (ex - sx) * dot / squaredL;
    double projectedY = (ey - sy) * dot / squaredL;
    px = projectedX + sx;
    py = projectedY + sy;
}
bool Ribbon::containsProjection(double px, double py) const {  // Ribbon.cpp:90-95
    return !(((px - sx < -c_Tolerance && px - ex < -c_Tolerance) || (px - sx > c_Tolerance && px - ex > c_Tolerance)) ||
             ((py - sy < -c_Tolerance && py - ey < -c_Tolerance) || (py - sy > c_Tolerance && py - ey > c_Tolerance)));
}
double Ribbon::distance(double x, double y) const {  // Ribbon.h:118-121
    return (std::fabs((ey - sy) * x - (ex - sx) * y + ex * sy - ey * sx)) / std::sqrt(squaredLength());
}
bool Ribbon::contains(double x, double y, double px, double py, bool strict, double w) const {  // Ribbon.cpp:39-43
    if (!containsProjection(px, py)) return false;
    double d = distance(x, y);
    return d < (strict ? w / c_StrictModifier : w);
}
Ribbon Ribbon::split(double x, double y, bool strict, double w) {  // Ribbon.cpp:9-17
    double px, py;
    projection(x, y, px, py);
    if (!contains(x, y, px, py, strict, w)) return Ribbon{0, 0, 0, 0};
    Ribbon r{sx, sy, px, py};
    sx = px; sy = py;
    return r;
}
State Ribbon::startAsState() const {  // Ribbon.cpp:60-64
    State s(sx, sy, 0, 0, 0);
    s.setHeadingTowards(ex, ey);
    return s;
}
State Ribbon::endAsState() const {  // Ribbon.cpp:66-70
    State s(ex, ey, 0, 0, 0);
    s.setHeadingTowards(sx, sy);
    return s;
}

// ============================================================================ RibbonManager
static inline double dist2(double x1, double y1, double x2, double y2) {  // RibbonManager.h:285-287
    return std::sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
}

void RibbonManager::add(double x1, double y1, double x2, double y2) {  // RibbonManager.cpp:7-12,154-158
    Ribbon r{x1, y1, x2, y2};
    if (r.covered(false, RibbonWidth)) return;
    ribbons.push_back(r);
}

void RibbonManager::cover(double x, double y, bool strict) {  // RibbonManager.cpp:14-22
    size_t i = 0;
    while (i < ribbons.size()) {
        Ribbon r = ribbons[i].split(x, y, strict, RibbonWidth);
        if (!r.covered(strict, RibbonWidth)) {  // add(r, i, strict): insert before i
            ribbons.insert(ribbons.begin() + i, r);
            i++;
        }
        if (ribbons[i].covered(strict, RibbonWidth)) ribbons.erase(ribbons.begin() + i);
        else ++i;
    }
}

void RibbonManager::coverBetween(double x1, double y1, double x2, double y2, bool strict) {  // :391-403
    double theta = std::atan((y2 - y1) / (x2 - x1));
    double d = dist2(x1, y1, x2, y2);
    do {
        double d1 = dist2(x1, y1, x2, y2);
        if (d1 > d) break;
        else d = d1;
        cover(x1, y1, strict);
        x1 += minLength() * std::cos(theta) / 2;
        y1 += minLength() * std::sin(theta) / 2;
    } while (d > minLength());
    cover(x2, y2, strict);
}

double RibbonManager::minDistanceFrom(double x, double y) const {  // :142-152
    if (ribbons.empty()) return 0;
    double min = DBL_MAX;
    for (const auto& r : ribbons) {
        double px, py;
        r.projection(x, y, px, py);
        if (r.contains(x, y, px, py, false, RibbonWidth)) return 0;
        double dStart = dist2(r.sx, r.sy, x, y);
        double dEnd = dist2(r.ex, r.ey, x, y);
        min = std::fmin(std::fmin(min, dEnd), dStart);
    }
    return min;
}

double RibbonManager::maxDistance(double x, double y) const {  // :234-248
    double sumLength = 0, min = DBL_MAX, max = 0;
    for (const auto& r : ribbons) {
        sumLength += r.length() - 2 * RibbonWidth;
        double dStart = dist2(r.sx, r.sy, x, y);
        double dEnd = dist2(r.ex, r.ey, x, y);
        min = std::fmin(std::fmin(min, dEnd), dStart);
        max = std::fmax(std::fmax(max, dEnd), dStart);
    }
    return std::fmax(sumLength + min, max);
}

double RibbonManager::dubinsDistance(double x, double y, double h, const State& s) const {  // RibbonManager.h:210-216
    if (turningRadius == -1) throw std::logic_error("Cannot compute ribbon dubins distance with unset turning radius");
    DubinsPath p;
    double q1[] = {x, y, h}, q2[] = {s.x, s.y, s.yaw()};
    dubins_shortest_path(&p, q1, q2, turningRadius);
    return dubins_path_length(&p);
}

namespace {
const double W2 = 2;  // "2 * Ribbon::RibbonWidth"

// RibbonManager.cpp:53-67
double tspPointAll(std::vector<Ribbon> left, double soFar, double px, double py) {
    if (left.empty()) return soFar;
    double min = DBL_MAX;
    for (size_t i = 0; i < left.size(); i++) {
        const Ribbon r = left[i];
        left.erase(left.begin() + i);
        min = std::fmin(min, tspPointAll(left, std::fmax(soFar + r.length() - W2 * RibbonManager::RibbonWidth +
                                                         dist2(px, py, r.sx, r.sy), 0), r.ex, r.ey));
        min = std::fmin(min, tspPointAll(left, std::fmax(soFar + r.length() - W2 * RibbonManager::RibbonWidth +
                                                         dist2(px, py, r.ex, r.ey), 0), r.sx, r.sy));
        left.insert(left.begin() + i, r);
    }
    return min;
}

// RibbonManager.cpp:69-94.  list::sort(comp) is a stable sort with comp(r1,r2) = min1 > min2,
// i.e. DESCENDING nearest-endpoint distance: the first K branches are the K farthest ribbons.
double tspPointK(std::vector<Ribbon> left, double soFar, double px, double py, int K) {
    if (left.empty()) return soFar;
    double min = DBL_MAX;
    auto comp = [&](const Ribbon& r1, const Ribbon& r2) {
        double min1 = std::fmin(dist2(px, py, r1.sx, r1.sy), dist2(px, py, r1.ex, r1.ey));
        double min2 = std::fmin(dist2(px, py, r2.sx, r2.sy), dist2(px, py, r2.ex, r2.ey));
        return min1 > min2;
    };
    std::stable_sort(left.begin(), left.end(), comp);
    int i = 0;
    for (size_t it = 0; it < left.size(); it++) {
        if (i++ >= K) break;
        const Ribbon r = left[it];
        left.erase(left.begin() + it);
        min = std::fmin(min, tspPointK(left, std::fmax(soFar + r.length() - W2 * RibbonManager::RibbonWidth +
                                                       dist2(px, py, r.sx, r.sy), 0), r.ex, r.ey, K));
        min = std::fmin(min, tspPointK(left, std::fmax(soFar + r.length() - W2 * RibbonManager::RibbonWidth +
                                                       dist2(px, py, r.ex, r.ey), 0), r.sx, r.sy, K));
        left.insert(left.begin() + it, r);
    }
    return min;
}
}  // namespace

// RibbonManager.cpp:97-114 (All) and :116-140 (K).  In the K variant the comparator compares r1
// with r1 (:121-122) so the sort is the identity, and `i` is never incremented (:128), so with
// m_K >= 1 every ribbon is branched: it equals the All variant.  With m_K <= 0 it returns DBL_MAX.
static double tspDubins(const RibbonManager& rm, std::vector<Ribbon> left, double soFar, double x, double y, double yaw,
                        bool kVariant) {
    if (left.empty()) return soFar;
    double min = DBL_MAX;
    for (size_t it = 0; it < left.size(); it++) {
        if (kVariant && 0 >= rm.K) break;
        const Ribbon r = left[it];
        left.erase(left.begin() + it);
        State start = r.startAsState();
        State end = r.endAsState();
        min = std::fmin(min, tspDubins(rm, left, std::fmax(soFar + r.length() - 2 * RibbonManager::RibbonWidth +
                                                           rm.dubinsDistance(x, y, yaw, start), 0),
                                       end.x, end.y, end.yaw(), kVariant));
        min = std::fmin(min, tspDubins(rm, left, std::fmax(soFar + r.length() - 2 * RibbonManager::RibbonWidth +
                                                           rm.dubinsDistance(x, y, yaw, end), 0),
                                       start.x, start.y, start.yaw(), kVariant));
        left.insert(left.begin() + it, r);
    }
    return min;
}

double RibbonManager::approximateDistanceUntilDone(double x, double y, double yaw) const {  // :28-51
    if (done()) return 0;
    switch (heuristic) {
    case MaxDistance: return maxDistance(x, y);
    case TspPointRobotNoSplitAllRibbons: return tspPointAll(ribbons, 0, x, y);
    case TspDubinsNoSplitAllRibbons: return tspDubins(*this, ribbons, 0, x, y, yaw, false);
    case TspPointRobotNoSplitKRibbons: return tspPointK(ribbons, 0, x, y, K);
    case TspDubinsNoSplitKRibbons: return tspDubins(*this, ribbons, 0, x, y, yaw, true);
    default: return 0;
    }
}

State RibbonManager::getNearestEndpointAsState(const State& state) const {  // :160-195
    if (done()) throw std::logic_error("Attempting to get nearest endpoint when there are no ribbons");
    double min = DBL_MAX;
    State ret;
    for (const auto& r : ribbons) {
        State s = r.startAsState();
        s.move(minLength() / c_StrictModifier + 1e-5);
        double d = state.distanceTo(s);
        if (d < min) {
            if (d < minLength()) {
                ret = r.endAsState();
                ret.heading = s.heading;
                ret.move(-minLength() / c_StrictModifier + 1e-5);
            } else {
                ret = s;
            }
            min = d;
        }
        s = r.endAsState();
        s.move(minLength() / c_StrictModifier + 1e-5);
        d = state.distanceTo(s);
        if (d < min) {
            if (d < minLength()) {
                ret = r.startAsState();
                ret.heading = s.heading;
                ret.move(-minLength() / c_StrictModifier + 1e-5);
            } else {
                ret = s;
            }
            min = d;
        }
    }
    return ret;
}

void RibbonManager::projectOntoNearestRibbon(State& state) const {  // :220-232
    if (ribbons.empty()) return;
    double min = DBL_MAX;
    Ribbon ribbon{0, 0, 0, 0};
    for (const auto& r : ribbons) {
        double d = r.distance(state.x, state.y);
        if (d < min) {
            min = d;
            ribbon = r;
        }
    }
    // Ribbon::getProjectionAsState (Ribbon.cpp:80-88)
    double px, py;
    ribbon.projection(state.x, state.y, px, py);
    State s(px, py, 0, 0, 0);
    s.setHeadingTowards(ribbon.ex, ribbon.ey);
    state = s;
}

void RibbonManager::changeHeuristicIfTooManyRibbons() {  // :381-385, threshold RibbonManager.h:268
    if (ribbons.size() > 5) heuristic = MaxDistance;
}

std::vector<State> RibbonManager::findNearStatesOnRibbons(const State& start, double radius) const {  // :296-379
    std::vector<State> states;
    double h = start.yaw() + M_PI_2;
    double x1 = start.x + std::cos(h) * radius;
    double x2 = start.x - std::cos(h) * radius;
    double y1 = start.y + std::sin(h) * radius;
    double y2 = start.y - std::sin(h) * radius;
    for (const Ribbon& r : ribbons) {
        double spx, spy;
        r.projection(start.x, start.y, spx, spy);
        {
            double d;
            if (r.containsProjection(spx, spy)) d = start.distanceTo(spx, spy);
            else d = std::fmin(start.distanceTo(r.sx, r.sy), start.distanceTo(r.ex, r.ey));
            if (d > 2 * radius) continue;
        }
        double p1x, p1y, p2x, p2y;
        r.projection(x1, y1, p1x, p1y);
        r.projection(x2, y2, p2x, p2y);
        double projx = p2x, projy = p2y;
        double x = x2, y = y2;
        if (r.containsProjection(p1x, p1y)) {
            projx = p1x; projy = p1y;
            x = x1; y = y1;
        }
        State s1 = r.startAsState();
        State s2 = r.endAsState();
        State s;
        if (s1.distanceTo(start) < s2.distanceTo(start)) s = s1;
        else s = s2;
        double h2 = s.yaw() - M_PI_2;
        double dx1 = std::cos(h2) * radius / 2;
        double dy1 = std::sin(h2) * radius / 2;
        double x3 = projx + dx1;
        double y3 = projy + dy1;
        double a = dx1 * dx1 + dy1 * dy1;
        double rSquared = radius * radius;
        double b = std::sqrt(rSquared - a);
        double h3 = s.yaw();
        double x5 = x3 + b * std::cos(h3);
        double y5 = y3 + b * std::sin(h3);
        double x7 = x5 - x;
        double y7 = y5 - y;
        double h4 = std::atan(y7 / x7);
        double x8 = x5 + radius * std::cos(h4);
        double y8 = y5 + radius * std::sin(h4);
        double fx, fy;
        r.projection(x8, y8, fx, fy);
        double d = dist2(fx, fy, start.x, start.y);
        if (d > 1e-5 && d < 2 * radius) states.emplace_back(fx, fy, s.heading, 0, 0);
    }
    return states;
}

// ============================================================================ Map
bool GridMap::isBlocked(double x, double y) const {  // Map.cpp:4-6 / GridWorldMap.cpp:84-93
    if (rows == 0) return false;
    if (x < 0 || x / resolution >= (double)(size_t)cols) return true;
    if (y < 0 || y / resolution >= (double)(size_t)rows) return true;
    size_t r = (size_t)(y / resolution);
    size_t c = (size_t)(x / resolution);
    return cells[r * (size_t)cols + c] != 0;
}

void GridMap::setCells(const uint8_t* c, int rows_, int cols_, double res) {
    rows = rows_; cols = cols_; resolution = res;
    cells.assign(c, c + (size_t)rows * cols);
    extremes[0] = 0; extremes[1] = (double)(size_t)cols * resolution;   // GridWorldMap.cpp:29-30
    extremes[2] = 0; extremes[3] = (double)(size_t)rows * resolution;
}

bool GridMap::loadText(const std::string& text) {  // GridWorldMap.cpp:10-82
    std::istringstream infile(text);
    std::string line;
    std::vector<std::string> lines;
    if (!std::getline(infile, line)) return false;
    std::istringstream s(line);
    double res = 0;
    s >> res;
    int c = -1, r = 0;
    while (std::getline(infile, line)) {
        if (c == -1) c = (int)line.length();
        else if ((int)line.length() < c) c = (int)line.length();
        r++;
        lines.push_back(line);
    }
    if (r == 0 || c <= 0) return false;
    std::reverse(lines.begin(), lines.end());
    std::vector<uint8_t> cc((size_t)r * c, 0);
    for (int y = 0; y < r; y++)
        for (int x = 0; x < c; x++)
            if (lines[y][x] == '#') cc[(size_t)y * c + x] = 1;
    setCells(cc.data(), r, c, res);
    return true;
}

// ============================================================================ Obstacles
void Obstacles::update(double x, double y, double heading, double speed, double time, double width, double length) {
    list.push_back(BinaryObstacle{x, y, M_PI_2 - heading, speed, time, width, length});  // .h:17-19
}
void Obstacles::updateGaussian(double x, double y, double heading, double speed, double time, const double* cov4) {
    GaussianObstacle o{x, y, M_PI_2 - heading, speed, time, {{30, 10}, {10, 30}}};  // GaussianDynamicObstaclesManager.h:23-26
    if (cov4) { o.cov[0][0] = cov4[0]; o.cov[0][1] = cov4[1]; o.cov[1][0] = cov4[2]; o.cov[1][1] = cov4[3]; }  // :28-29
    gauss.push_back(o);
}
double Obstacles::collisionExists(double x, double y, double time, bool strict) const {  // .cpp:4-22
    if (model == 0) return 0;  // DynamicObstaclesManager.h:23
    if (model == 2) {          // GaussianDynamicObstaclesManager.cpp:3-13
        double sum = 0;
        for (GaussianObstacle o : gauss) {
            double dt = time - o.Time;  // Obstacle::project, .h:31-36
            double dx = o.Speed * dt * std::cos(o.Yaw);
            double dy = o.Speed * dt * std::sin(o.Yaw);
            o.X += dx; o.Y += dy;
            // Obstacle::pdf, .h:38-43
            double twoPi = 2 * M_PI;
            double det = o.cov[0][0] * o.cov[1][1] - o.cov[1][0] * o.cov[0][1];   // Eigen 2x2 determinant
            double invdet = 1.0 / det;                                            // Eigen compute_inverse, size 2
            double i00 = o.cov[1][1] * invdet, i10 = -o.cov[1][0] * invdet, i01 = -o.cov[0][1] * invdet, i11 = o.cov[0][0] * invdet;
            double vx = x - o.X, vy = y - o.Y;
            double r0 = vx * i00 + vy * i10, r1 = vx * i01 + vy * i11;            // (x - mean)^T * inverse
            double quadform = r0 * vx + r1 * vy;                                  // ... * (x - mean)
            double norm = 1.0 / twoPi / std::sqrt(det);
            sum += norm * std::exp(-0.5 * quadform);
        }
        if (sum < 1e-5) return 0;  // "questionable", .cpp:11
        return sum;
    }
    double sum = 0;
    for (BinaryObstacle o : list) {
        if (strict) {
            o.Width += 2;
            o.Length += 2;
        }
        double dt = time - o.Time;  // Obstacle::project, .h:20-25
        double dx = o.Speed * dt * std::cos(o.Yaw);
        double dy = o.Speed * dt * std::sin(o.Yaw);
        o.X += dx; o.Y += dy;
        double translatedX = x - o.X;
        double translatedY = y - o.Y;
        double rotatedX = translatedX * std::cos(o.Yaw) - translatedY * std::sin(o.Yaw);
        double rotatedY = translatedX * std::sin(o.Yaw) + translatedY * std::cos(o.Yaw);
        if (std::fabs(rotatedX) < o.Length / 2 && std::fabs(rotatedY) < o.Width / 2) sum++;
    }
    return sum;
}

// ============================================================================ Vertex / Edge
int Vertex::depth(const std::vector<Vertex>& arena) const {  // Vertex.cpp:87-90
    int d = 0;
    for (int p = parent; p >= 0; p = arena[p].parent) d++;
    return d;
}

Vertex makeRoot(const State& s, const RibbonManager& r) {  // Vertex.cpp:38-43
    Vertex v;
    v.state = s;
    v.currentCost = 0;
    v.ribbons = r;
    return v;
}

Vertex connectState(const Vertex& start, int startIndex, const State& next, double turningRadius, bool coverageAllowed) {
    Vertex v;  // Vertex.cpp:21-26,125-130
    v.state = next;
    v.parent = startIndex;
    v.ribbons = start.ribbons;
    v.turningRadius = turningRadius;
    v.coverageAllowed = coverageAllowed;
    return v;
}

Vertex connectWrapper(const Vertex& start, int startIndex, const DubinsWrapper& w, bool coverageAllowed) {
    Vertex v;  // Vertex.cpp:28-36 -> Edge::setEnd(wrapper) Edge.cpp:208-216
    v.wrapper = w;
    State s;
    s.time = w.endTime;
    w.sample(s);
    v.edgeApproxCost = (s.time - start.state.time) * 1.0;
    v.state = s;
    v.parent = startIndex;
    v.ribbons = start.ribbons;
    v.coverageAllowed = coverageAllowed;
    v.turningRadius = w.getRho();
    return v;
}

double computeApproxToGo(Vertex& v, const Config& cfg) {  // Vertex.cpp:49-64 (passes heading as "yaw")
    // Checker-only: mirror of the device's capacity rule (pp_device.h: pp_tsp_big_ok).  Beyond tspRibbonLimit (8) ribbons
    // only the K variant of the point-robot heuristic is enumerated, up to 12 ribbons and 2^21 lane-parallel prefixes.
    auto deviceEnumerates = [&](int n) {
        if (n <= cfg.tspRibbonLimit) return true;
        if (v.ribbons.heuristic != TspPointRobotNoSplitKRibbons || n > 12) return false;
        const int K = v.ribbons.K;
        if (K <= 0) return true;
        unsigned long long NP = 1;
        for (int l = 0; l < n; l++) {
            int rem = n - l;
            if (NP >= 64ull && l >= n - 3) break;
            NP *= (unsigned long long)(2 * (rem < K ? rem : K));
        }
        return NP < (1ull << 21);
    };
    if (cfg.tspRibbonLimit > 0 && v.ribbons.heuristic != MaxDistance && !deviceEnumerates((int)v.ribbons.ribbons.size())) {
        v.heuristicSkipped = true;
        v.approxToGo = 0;
        return 0;
    }
    if (cfg.skipHeuristicValue) {
        v.approxToGo = 0;
        return 0;
    }
    double max = v.ribbons.approximateDistanceUntilDone(v.state.x, v.state.y, v.state.heading);
    v.approxToGo = max / cfg.maxSpeed * cfg.timePenaltyFactor;
    return v.approxToGo;
}

double computeApproxCost(const Vertex& start, Vertex& end, double maxSpeed, double turningRadius) {  // Edge.cpp:11-20
    if (start.state.isCoLocated(end.state)) {
        end.edgeApproxCost = 0;
    } else {
        end.wrapper.set(start.state, end.state, turningRadius);
        end.edgeApproxCost = end.wrapper.length() / maxSpeed * 1.0;
    }
    return end.edgeApproxCost;
}

double computeTrueCost(const Vertex& start, Vertex& end, const Config& config) {  // Edge.cpp:68-206
    double speed = end.state.speed, turningRadius = config.turningRadius;
    if (end.coverageAllowed) turningRadius = config.coverageTurningRadius;
    if (end.edgeApproxCost == -1 || (end.wrapper.getRho() != turningRadius))
        computeApproxCost(start, end, speed, turningRadius);
    if (end.wrapper.speed != speed) end.wrapper.setSpeed(speed);
    if (end.edgeApproxCost < 0) throw std::runtime_error("Could not compute approximate cost");
    double collisionPenalty = 0;
    State intermediate(start.state);
    double endTime = std::fmin(config.timeHorizon + 1e-12 + config.startStateTime, end.wrapper.endTime);
    int ribbonsDoneTime = -1;  // `auto ribbonsDoneTime = -1;` => int (Edge.cpp:92)
    bool ribbonManagerStartedDone = end.ribbons.done();
    double toCoverDistance = 0;
    double lastHeading = start.state.heading;
    end.steps = 0;
    end.events = end.mutations = 0;
    for (int q = 0; q < 4; q++) end.mutKinds[q] = 0;

    if (intermediate.time >= endTime) end.infeasible = true;  // :102-110

    double timeIncrement = config.collisionCheckingIncrement / config.maxSpeed;  // :114
    double timeSinceStart = intermediate.time - config.startStateTime;
    double timeNudge = std::fmod(timeSinceStart, timeIncrement);
    intermediate.time += timeNudge;

    while (intermediate.time < endTime) {  // :125
        try {
            end.wrapper.sample(intermediate);
        } catch (SampleError&) {
            end.infeasible = true;
            break;
        }
        end.steps++;
        if (config.map->isBlocked(intermediate.x, intermediate.y)) {  // :144-147
            end.infeasible = true;
            break;
        }
        collisionPenalty += config.obstacles->collisionExists(intermediate.x, intermediate.y, intermediate.time, true) *
                            config.collisionPenaltyFactor;  // :150-151
        if (toCoverDistance > config.collisionCheckingIncrement) {
            toCoverDistance -= config.collisionCheckingIncrement;
        } else {
            toCoverDistance = end.ribbons.minDistanceFrom(intermediate.x, intermediate.y);
            end.events++;
            if (end.coverageAllowed || lastHeading == intermediate.heading) {
                auto before = end.ribbons.ribbons;
                end.ribbons.cover(intermediate.x, intermediate.y, true);
                bool same = before.size() == end.ribbons.ribbons.size();
                for (size_t q = 0; same && q < before.size(); q++)
                    same = before[q].sx == end.ribbons.ribbons[q].sx && before[q].sy == end.ribbons.ribbons[q].sy &&
                           before[q].ex == end.ribbons.ribbons[q].ex && before[q].ey == end.ribbons.ribbons[q].ey;
                if (!same) {
                    end.mutations++;
                    const auto& after = end.ribbons.ribbons;
                    if (before.size() != after.size()) end.mutKinds[2]++;
                    else {
                        int changed = 0, startOnly = 0;
                        for (size_t q = 0; q < before.size(); q++) {
                            bool diff = before[q].sx != after[q].sx || before[q].sy != after[q].sy || before[q].ex != after[q].ex || before[q].ey != after[q].ey;
                            if (diff) { changed++; if (before[q].ex == after[q].ex && before[q].ey == after[q].ey) startOnly++; }
                        }
                        if (changed == 1 && startOnly == 1) end.mutKinds[0]++;
                        else if (changed == 2) end.mutKinds[1]++;
                        else end.mutKinds[3]++;
                    }
                }
            }
            if (end.ribbons.done()) {
                if (end.ribbons.coverageCompletedTime == -1) end.ribbons.setCoverageCompletedTime(intermediate.time);
                ribbonsDoneTime = (int)intermediate.time;
                endTime = std::fmin(endTime, end.ribbons.coverageCompletedTime + config.timeMinimum);
            }
        }
        intermediate.time += timeIncrement;
        lastHeading = intermediate.heading;
    }
    // :177-179 (may throw out of computeTrueCost)
    end.state.time = endTime;
    end.wrapper.sample(end.state);
    end.wrapper.updateEndTime(end.state.time);

    if (end.coverageAllowed || lastHeading == intermediate.heading) {  // :182-184
        end.ribbons.cover(intermediate.x, intermediate.y, true);
    }
    if (end.ribbons.done()) {  // :185-191
        if (end.ribbons.coverageCompletedTime == -1) end.ribbons.setCoverageCompletedTime(intermediate.time);
        ribbonsDoneTime = (int)intermediate.time;
    }
    end.collisionPenalty = collisionPenalty;
    double netTime = end.state.time - start.state.time;
    double t = std::fmax(netTime - (end.ribbons.done() ? (endTime - ribbonsDoneTime) : 0), 0);  // :197
    if (ribbonManagerStartedDone) t = 0;
    end.edgeTrueCost = t * config.timePenaltyFactor + collisionPenalty;
    end.currentCost = start.currentCost + end.edgeTrueCost;  // Vertex::setCurrentCost, Vertex.cpp:102-104
    computeApproxToGo(end, config);
    return end.edgeTrueCost;
}

// ============================================================================ StateGenerator
StateGenerator::StateGenerator(double minX_, double maxX_, double minY_, double maxY_, double minSpeed_, double maxSpeed_,
                               unsigned long seed)
    : minX(minX_), maxX(maxX_), minY(minY_), maxY(maxY_), minSpeed(minSpeed_), maxSpeed(maxSpeed_) {
    // std::linear_congruential_engine<uint_fast32_t,16807,0,2147483647>::seed(s): c == 0 and s mod m == 0 -> 1
    const unsigned long m = 2147483647UL;
    unsigned long x = seed % m;
    engine = (uint32_t)(x == 0 ? 1 : x);
}
StateGenerator::StateGenerator(double minX_, double maxX_, double minY_, double maxY_, double minSpeed_, double maxSpeed_,
                               unsigned long seed, const RibbonManager& r)
    : StateGenerator(minX_, maxX_, minY_, maxY_, minSpeed_, maxSpeed_, seed) {
    ribbons = r;
    sampleOnRibbons = true;  // StateGenerator.cpp:33-38
}
uint32_t StateGenerator::next() {
    engine = (uint32_t)(((uint64_t)engine * 16807ULL) % 2147483647ULL);
    draws++;
    return engine;
}
double StateGenerator::canonical() {
    // std::generate_canonical<double,53>(minstd_rand0): range R = max-min+1 = 2147483646, log2 R -> 30,
    // k = max(1, (53 + 30 - 1) / 30) = 2 engine calls; sum and scale kept in double.
    const double R = 2147483646.0;
    double sum = 0, tmp = 1;
    for (int k = 0; k < 2; k++) {
        sum += (double)(next() - 1u) * tmp;
        tmp *= R;
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
    return ret;
}
double StateGenerator::uniform(double a, double b) { return canonical() * (b - a) + a; }

State StateGenerator::generate() {  // StateGenerator.cpp:15-31
    // g++ evaluates the constructor's arguments right to left: speed, heading, y, x (SURVEY.md 8 a-1 probe).
    double speed = uniform(minSpeed, maxSpeed);
    double heading = uniform(0, 2 * M_PI);
    double y = uniform(minY, maxY);
    double x = uniform(minX, maxX);
    State s(x, y, heading, speed, 0);
    if (sampleOnRibbons) {
        if (uniform(0, 2 * M_PI) < M_PI / 50) {
            ribbons.projectOntoNearestRibbon(s);
            if (uniform(0, 2 * M_PI) < M_PI) s.heading += M_PI;
        }
    }
    return s;
}

// ============================================================================ Planner
void AStarPlanner::pushVertexQueue(int vi) {  // SamplingBasedPlanner.cpp:7-19
    Vertex& v = arena[vi];
    if (v.parent >= 0 && v.infeasible) return;
    if (v.approxToGo == -1) throw std::runtime_error("Fetching unset approx to go (h)");
    if (best >= 0 && arena[best].f() < v.f()) return;
    if (best >= 0 && arena[best].f() == v.f() && goalCondition(v)) return;
    queue.push_back(vi);
    std::push_heap(queue.begin(), queue.end(), [&](int a, int b) { return arena[a].f() > arena[b].f(); });  // AStarPlanner.cpp:6-10
    stats.Generated++;
}

int AStarPlanner::popVertexQueue() {  // :21-27
    if (queue.empty()) return -1;
    std::pop_heap(queue.begin(), queue.end(), [&](int a, int b) { return arena[a].f() > arena[b].f(); });
    int r = queue.back();
    queue.pop_back();
    return r;
}

bool AStarPlanner::goalCondition(const Vertex& v) const {  // :42-50
    double coverageDoneTime = v.ribbons.coverageCompletedTime + cfg.timeMinimum;
    if (v.ribbons.coverageCompletedTime == -1 && v.ribbons.done())
        throw std::runtime_error("Unset coverage completed time but coverage is done");
    double nonCoverageDoneTime = startStateTime + cfg.timeHorizon;
    return v.state.time >= nonCoverageDoneTime || (v.ribbons.done() && v.state.time >= coverageDoneTime);
}

void AStarPlanner::expand(int source) {  // :52-151
    const double speeds[2] = {cfg.maxSpeed, cfg.maxSpeed == cfg.slowSpeed() ? -1 : cfg.slowSpeed()};
    const int nTurningRadii = 2;
    const double turningRadii[nTurningRadii] = {cfg.turningRadius,
                                                cfg.coverageTurningRadius == cfg.turningRadius ? -1 : cfg.coverageTurningRadius};
    auto costAndPush = [&](Vertex&& dest) {
        computeTrueCost(arena[source], dest, cfg);
        if (onEdge) onEdge(arena[source], dest);
        arena.push_back(std::move(dest));
        pushVertexQueue((int)arena.size() - 1);
    };
    if (!arena[source].ribbons.done()) {
        State s = arena[source].ribbons.getNearestEndpointAsState(arena[source].state);  // Vertex.cpp:92-95
        if (arena[source].state.distanceTo(s) > cfg.collisionCheckingIncrement) {
            for (double speed : speeds) {
                if (speed <= 0) continue;
                for (double turningRadius : turningRadii) {
                    if (turningRadius <= 0) continue;
                    bool coverageAllowed = turningRadius == cfg.coverageTurningRadius;
                    s.speed = speed;
                    costAndPush(connectState(arena[source], source, s, turningRadius, coverageAllowed));
                }
            }
        }
    }
    const State origin = arena[source].state;
    auto comp = [&](const State& s1, const State& s2) { return s1.distanceTo(origin) > s2.distanceTo(origin); };  // :36-40
    std::vector<Vertex> temps;  // candidates (Vertex::connect + computeApproxCost), referenced by index
    auto dubinsComp = [&](int a, int b) { return temps[a].edgeApproxCost < temps[b].edgeApproxCost; };  // :174-180
    std::make_heap(samples.begin(), samples.end(), comp);
    std::vector<int> bestSamplesHeaps[nTurningRadii];
    bool doneChecks[nTurningRadii] = {false, false};
    const size_t k = (size_t)cfg.branchingFactor;
    for (uint64_t i = 0; i < samples.size() && (!doneChecks[0] || !doneChecks[1]); i++) {
        State sample = samples.front();
        std::pop_heap(samples.begin(), samples.end() - i, comp);
        for (int j = 0; j < nTurningRadii; j++) {
            if (doneChecks[j]) continue;
            const double turningRadius = turningRadii[j];
            if (turningRadius <= 0) {
                doneChecks[j] = true;
                continue;
            }
            auto& bestSamples = bestSamplesHeaps[j];
            if (bestSamples.size() < k ||
                temps[bestSamples.front()].wrapper.length() > sample.distanceTo(origin)) {
                if (origin.distanceTo(sample) > cfg.collisionCheckingIncrement) {
                    sample.speed = cfg.maxSpeed;
                    bool coverageAllowed = turningRadius == cfg.coverageTurningRadius;
                    temps.push_back(connectState(arena[source], source, sample, turningRadius, coverageAllowed));
                    Vertex& t = temps.back();
                    computeApproxCost(arena[source], t, t.state.speed, t.turningRadius);  // Edge::computeApproxCost() :64-66
                    bestSamples.push_back((int)temps.size() - 1);
                    std::push_heap(bestSamples.begin(), bestSamples.end(), dubinsComp);
                    if (bestSamples.size() > k) {
                        std::pop_heap(bestSamples.begin(), bestSamples.end(), dubinsComp);
                        bestSamples.pop_back();
                    }
                }
            } else {
                doneChecks[j] = true;
            }
        }
    }
    for (auto& bestSamples : bestSamplesHeaps) {
        for (int ti : bestSamples) {
            DubinsWrapper wrapper = temps[ti].wrapper;
            for (double speed : speeds) {
                if (speed <= 0) continue;
                wrapper.setSpeed(speed);
                costAndPush(connectWrapper(arena[source], source, wrapper, temps[ti].coverageAllowed));
            }
        }
    }
    stats.Expanded++;
}

void AStarPlanner::addSamples(StateGenerator& g, int n) {  // :157-164
    attemptedSamples += n;
    for (int i = 0; i < n; i++) {
        const State s = g.generate();
        if (!cfg.map->isBlocked(s.x, s.y)) samples.push_back(s);
    }
}
void AStarPlanner::addSamples(StateGenerator& g) { addSamples(g, (int)samples.size()); }  // :166-168

int AStarPlanner::aStar(double endTime) {  // AStarPlanner.cpp:134-148
    int vertex = popVertexQueue();
    if (vertex < 0) throw std::out_of_range("Trying to pop an empty vertex queue");
    while (cfg.now() < endTime) {
        if (goalCondition(arena[vertex])) return vertex;
        expand(vertex);
        if (queue.empty()) return -1;
        vertex = popVertexQueue();
    }
    return -1;
}

void AStarPlanner::expandToCoverSpecificSamples(int root, const std::vector<State>& ss, bool coverageAllowed) {  // :150-162
    if (cfg.coverageTurningRadius > 0) {
        for (State s : ss) {
            for (double speed : {cfg.maxSpeed, cfg.slowSpeed()}) {
                s.speed = speed;
                Vertex d = connectState(arena[root], root, s, cfg.coverageTurningRadius, coverageAllowed);
                computeTrueCost(arena[root], d, cfg);
                if (onEdge) onEdge(arena[root], d);
                arena.push_back(std::move(d));
                pushVertexQueue((int)arena.size() - 1);
            }
        }
    }
}

std::vector<DubinsWrapper> AStarPlanner::tracePlan(int v) {  // Planner.cpp:12-32
    std::vector<DubinsWrapper> branch;
    if (v < 0) return branch;
    for (int cur = v; arena[cur].parent >= 0; cur = arena[cur].parent) {
        branch.push_back(arena[cur].wrapper);
        if (arena[cur].collisionPenalty > 0) stats.PlanCollisionPenalty += arena[cur].collisionPenalty;
    }
    std::reverse(branch.begin(), branch.end());
    return branch;
}

Stats AStarPlanner::plan(const RibbonManager& ribbons, const State& start, Config config,
                         const std::vector<DubinsWrapper>& previousPlan, double timeRemaining) {  // AStarPlanner.cpp:12-132
    cfg = std::move(config);
    double endTime = timeRemaining + cfg.now();
    cfg.startStateTime = start.time;
    ribbonManager = ribbons;
    ribbonManager.changeHeuristicIfTooManyRibbons();
    if (ribbonManager.done()) ribbonManager.setCoverageCompletedTime(start.time);
    stats = Stats();
    startStateTime = start.time;
    samples.clear();
    attemptedSamples = 0;
    arena.clear();
    queue.clear();
    double minSpeed = cfg.maxSpeed, maxSpeed = cfg.maxSpeed;
    double magnitude = cfg.maxSpeed * cfg.timeHorizon;
    const double* mapExtremes = cfg.map->extremes;
    double minX = std::fmax(start.x - magnitude, mapExtremes[0]);
    double maxX = std::fmin(start.x + magnitude, mapExtremes[1]);
    double minY = std::fmax(start.y - magnitude, mapExtremes[2]);
    double maxY = std::fmin(start.y + magnitude, mapExtremes[3]);
    unsigned long seed = (unsigned long)endTime;
    StateGenerator generator(minX, maxX, minY, maxY, minSpeed, maxSpeed, seed, ribbonManager);
    arena.push_back(makeRoot(start, ribbonManager));
    const int startV = 0;
    arena[startV].state.speed = cfg.maxSpeed;
    computeApproxToGo(arena[startV], cfg);
    best = -1;
    std::vector<State> brownPathSamples;
    if (cfg.useBrownPaths) brownPathSamples = ribbonManager.findNearStatesOnRibbons(start, cfg.coverageTurningRadius);

    int lastPlanEnd = startV;  // :46-59
    for (const auto& p : previousPlan) {
        if (p.endTime <= start.time) continue;
        if (p.endTime - p.updatedStartTime == 0) continue;
        Vertex d = connectWrapper(arena[lastPlanEnd], lastPlanEnd, p, p.getRho() == cfg.coverageTurningRadius);
        computeTrueCost(arena[lastPlanEnd], d, cfg);
        if (onEdge) onEdge(arena[lastPlanEnd], d);
        arena.push_back(std::move(d));
        lastPlanEnd = (int)arena.size() - 1;
        if (arena[lastPlanEnd].infeasible) {
            lastPlanEnd = startV;
            break;
        }
        if (goalCondition(arena[lastPlanEnd])) break;
    }
    while (cfg.now() < endTime) {  // :61
        queue.clear();
        if (best >= 0 && arena[best].f() <= arena[startV].f()) break;
        pushVertexQueue(startV);
        if (lastPlanEnd != startV) pushVertexQueue(lastPlanEnd);
        expandToCoverSpecificSamples(startV, brownPathSamples, true);
        if (samples.size() < (size_t)cfg.initialSamples) addSamples(generator, cfg.initialSamples);
        else addSamples(generator);
        int v = aStar(endTime);
        if (best < 0 || (v >= 0 && arena[v].f() + 0.0 < arena[best].f())) best = v;
        if (v >= 0 && stats.firstGoalIteration < 0) stats.firstGoalIteration = (long)stats.Iterations;
        stats.iterationBestF.push_back(best >= 0 ? arena[best].f() : std::numeric_limits<double>::quiet_NaN());
        stats.Iterations++;
    }
    stats.Samples = samples.size();
    if (best >= 0) {
        stats.PlanFValue = arena[best].f();
        stats.PlanDepth = arena[best].depth(arena);
        stats.PlanTimePenalty = (arena[best].state.time - startStateTime) * cfg.timePenaltyFactor;
        stats.PlanHValue = arena[best].approxToGo;
        stats.Plan = tracePlan(best);
    }
    return stats;
}

}  // namespace ppo
