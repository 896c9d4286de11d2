// Ribbon / RibbonManager — mirrors of path_planner/src/planner/utilities/{Ribbon,RibbonManager}.cpp (host, off the hot loop)
#include "path_planner_amd/RibbonManager.h"

#include <algorithm>
#include <cfloat>
#include <sstream>
#include <stdexcept>

extern "C" {
#include "../../../include/dubins.h"
}

namespace ppamd {

double Ribbon::RibbonWidth = 1.5;
static const double kTolerance = 1e-5;

Ribbon Ribbon::split(double x, double y, bool strict) {
    auto projected = getProjection(x, y);
    if (!contains(x, y, projected, strict)) return Ribbon::empty();
    Ribbon r(m_StartX, m_StartY, projected.first, projected.second);
    m_StartX = projected.first;
    m_StartY = projected.second;
    return r;
}

bool Ribbon::covered(bool strict) const {
    return squaredLength() < minLength() * minLength() / (strict ? strictModifier() * strictModifier() : 1);
}

double Ribbon::length() const { return std::sqrt(squaredLength()); }

bool Ribbon::contains(double x, double y, const std::pair<double, double>& projected, bool strict) const {
    if (!containsProjection(projected)) return false;
    double d = distance(x, y);
    return d < (strict ? RibbonWidth / strictModifier() : RibbonWidth);
}

std::string Ribbon::toString() const {
    std::stringstream stream;
    stream << "(" << m_StartX << ", " << m_StartY << ") -> (" << m_EndX << ", " << m_EndY << ") with length " << length();
    return stream.str();
}

State Ribbon::startAsState() const {
    State s(m_StartX, m_StartY, 0, 0, 0);
    s.setHeadingTowards(m_EndX, m_EndY);
    return s;
}

State Ribbon::endAsState() const {
    State s(m_EndX, m_EndY, 0, 0, 0);
    s.setHeadingTowards(m_StartX, m_StartY);
    return s;
}

std::pair<double, double> Ribbon::getProjection(double x, double y) const {
    double squaredL = squaredLength();
    double dot = (x - m_StartX) * (m_EndX - m_StartX) + (y - m_StartY) * (m_EndY - m_StartY);
    double projectedX = (m_EndX - m_StartX) * dot / squaredL;
    double projectedY = (m_EndY - m_StartY) * dot / squaredL;
    return std::make_pair(projectedX + m_StartX, projectedY + m_StartY);
}

State Ribbon::getProjectionAsState(double x, double y) const {
    auto p = getProjection(x, y);
    State s(p.first, p.second, 0, 0, 0);
    s.setHeadingTowards(m_EndX, m_EndY);
    return s;
}

bool Ribbon::containsProjection(const std::pair<double, double>& p) const {
    return !(((p.first - m_StartX < -kTolerance && p.first - m_EndX < -kTolerance) ||
              (p.first - m_StartX > kTolerance && p.first - m_EndX > kTolerance)) ||
             ((p.second - m_StartY < -kTolerance && p.second - m_EndY < -kTolerance) ||
              (p.second - m_StartY > kTolerance && p.second - m_EndY > kTolerance)));
}

double Ribbon::distance(double x, double y) const {
    return (std::fabs((m_EndY - m_StartY) * x - (m_EndX - m_StartX) * y + m_EndX * m_StartY - m_EndY * m_StartX)) /
           std::sqrt(squaredLength());
}

// ------------------------------------------------------------------------------------------------ RibbonManager
static double dist(double x1, double y1, double x2, double y2) { return std::sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2)); }
static double dist(std::pair<double, double> p, double x, double y) { return dist(p.first, p.second, x, y); }
static double dist(std::pair<double, double> a, std::pair<double, double> b) { return dist(a.first, a.second, b.first, b.second); }

void RibbonManager::add(double x1, double y1, double x2, double y2) {
    Ribbon r(x1, y1, x2, y2);
    if (r.covered(false)) return;
    m_Ribbons.push_back(r);
}

void RibbonManager::assign(const double* r4, int n, double cct) {
    m_Ribbons.clear();
    for (int i = 0; i < n; i++) m_Ribbons.emplace_back(r4[4 * i], r4[4 * i + 1], r4[4 * i + 2], r4[4 * i + 3]);
    m_CoverageCompletedTime = cct;
}

void RibbonManager::cover(double x, double y, bool strict) {
    auto i = m_Ribbons.begin();
    while (i != m_Ribbons.end()) {
        Ribbon r = i->split(x, y, strict);
        if (!r.covered(strict)) m_Ribbons.insert(i, r);
        if (i->covered(strict)) i = m_Ribbons.erase(i);
        else ++i;
    }
}

void RibbonManager::coverBetween(double x1, double y1, double x2, double y2, bool strict) {
    double theta = std::atan((y2 - y1) / (x2 - x1));
    double d = dist(x1, y1, x2, y2);
    do {
        double d1 = dist(x1, y1, x2, y2);
        if (d1 > d) break;
        else d = d1;
        cover(x1, y1, strict);
        x1 += Ribbon::minLength() * std::cos(theta) / 2;
        y1 += Ribbon::minLength() * std::sin(theta) / 2;
    } while (d > Ribbon::minLength());
    cover(x2, y2, strict);
}

double RibbonManager::minDistanceFrom(double x, double y) const {
    if (m_Ribbons.empty()) return 0;
    double mn = DBL_MAX;
    for (const auto& r : m_Ribbons) {
        if (r.contains(x, y, r.getProjection(x, y), false)) return 0;
        double dStart = dist(r.start(), x, y);
        double dEnd = dist(r.end(), x, y);
        mn = std::fmin(std::fmin(mn, dEnd), dStart);
    }
    return mn;
}

double RibbonManager::maxDistance(double x, double y) const {
    double sumLength = 0, mn = DBL_MAX, mx = 0;
    for (const auto& r : m_Ribbons) {
        sumLength += r.length() - 2 * Ribbon::RibbonWidth;
        double dStart = dist(r.start(), x, y);
        double dEnd = dist(r.end(), x, y);
        mn = std::fmin(std::fmin(mn, dEnd), dStart);
        mx = std::fmax(std::fmax(mx, dEnd), dStart);
    }
    return std::fmax(sumLength + mn, mx);
}

double RibbonManager::dubinsDistance(double x, double y, double h, const State& s) const {
    if (m_TurningRadius == -1) throw std::logic_error("Cannot compute ribbon dubins distance with unset turning radius");
    DubinsPath p;
    double q1[] = {x, y, h}, q2[] = {s.x(), s.y(), s.yaw()};
    dubins_shortest_path(&p, q1, q2, m_TurningRadius);
    return dubins_path_length(&p);
}

double RibbonManager::tspPoint(std::list<Ribbon> left, double soFar, std::pair<double, double> point, bool kVariant) const {
    if (left.empty()) return soFar;
    double mn = DBL_MAX;
    if (kVariant) {
        // list::sort with comp = (nearest endpoint of r1) > (nearest endpoint of r2): descending, stable
        left.sort([&](const Ribbon& r1, const Ribbon& r2) {
            double min1 = std::fmin(dist(point, r1.start()), dist(point, r1.end()));
            double min2 = std::fmin(dist(point, r2.start()), dist(point, r2.end()));
            return min1 > min2;
        });
    }
    int i = 0;
    for (auto it = left.begin(); it != left.end(); it++) {
        if (kVariant && i++ >= m_K) break;
        const Ribbon r = *it;
        it = left.erase(it);
        mn = std::fmin(mn, tspPoint(left, std::fmax(soFar + r.length() - 2 * Ribbon::RibbonWidth + dist(point, r.start()), 0), r.end(), kVariant));
        mn = std::fmin(mn, tspPoint(left, std::fmax(soFar + r.length() - 2 * Ribbon::RibbonWidth + dist(point, r.end()), 0), r.start(), kVariant));
        it = left.insert(it, r);
    }
    return mn;
}

double RibbonManager::tspDubins(std::list<Ribbon> left, double soFar, double x, double y, double yaw, bool kVariant) const {
    if (left.empty()) return soFar;
    double mn = DBL_MAX;
    for (auto it = left.begin(); it != left.end(); it++) {
        if (kVariant && 0 >= m_K) break;   // the reference never increments its counter (RibbonManager.cpp:128)
        const Ribbon r = *it;
        it = left.erase(it);
        State start = r.startAsState();
        State end = r.endAsState();
        mn = std::fmin(mn, tspDubins(left, std::fmax(soFar + r.length() - 2 * Ribbon::RibbonWidth + dubinsDistance(x, y, yaw, start), 0),
                                     end.x(), end.y(), end.yaw(), kVariant));
        mn = std::fmin(mn, tspDubins(left, std::fmax(soFar + r.length() - 2 * Ribbon::RibbonWidth + dubinsDistance(x, y, yaw, end), 0),
                                     start.x(), start.y(), start.yaw(), kVariant));
        it = left.insert(it, r);
    }
    return mn;
}

double RibbonManager::approximateDistanceUntilDone(double x, double y, double yaw) const {
    if (done()) return 0;
    switch (m_Heuristic) {
    case MaxDistance: return maxDistance(x, y);
    case TspPointRobotNoSplitAllRibbons: return tspPoint(m_Ribbons, 0, std::make_pair(x, y), false);
    case TspDubinsNoSplitAllRibbons: return tspDubins(m_Ribbons, 0, x, y, yaw, false);
    case TspPointRobotNoSplitKRibbons: return tspPoint(m_Ribbons, 0, std::make_pair(x, y), true);
    case TspDubinsNoSplitKRibbons: return tspDubins(m_Ribbons, 0, x, y, yaw, true);
    default: return 0;
    }
}

void RibbonManager::changeHeuristicIfTooManyRibbons() {
    if (m_Ribbons.size() > 5) m_Heuristic = MaxDistance;
}

State RibbonManager::getNearestEndpointAsState(const State& state) const {
    if (done()) throw std::logic_error("Attempting to get nearest endpoint when there are no ribbons");
    double mn = DBL_MAX;
    State ret;
    for (const auto& r : m_Ribbons) {
        State s = r.startAsState();
        s.move(Ribbon::minLength() / Ribbon::strictModifier() + 1e-5);
        double d = state.distanceTo(s);
        if (d < mn) {
            if (d < Ribbon::minLength()) {
                ret = r.endAsState();
                ret.heading() = s.heading();
                ret.move(-Ribbon::minLength() / Ribbon::strictModifier() + 1e-5);
            } else {
                ret = s;
            }
            mn = d;
        }
        s = r.endAsState();
        s.move(Ribbon::minLength() / Ribbon::strictModifier() + 1e-5);
        d = state.distanceTo(s);
        if (d < mn) {
            if (d < Ribbon::minLength()) {
                ret = r.startAsState();
                ret.heading() = s.heading();
                ret.move(-Ribbon::minLength() / Ribbon::strictModifier() + 1e-5);
            } else {
                ret = s;
            }
            mn = d;
        }
    }
    return ret;
}

std::string RibbonManager::dumpRibbons() const {
    std::stringstream stream;
    stream << "Ribbons: \n";
    if (m_Ribbons.empty()) stream << "None\n";
    else for (const auto& r : m_Ribbons) stream << r.toString() << "\n";
    return stream.str();
}

void RibbonManager::projectOntoNearestRibbon(State& state) const {
    if (m_Ribbons.empty()) return;
    double mn = DBL_MAX;
    Ribbon ribbon = Ribbon::empty();
    for (const auto& r : m_Ribbons) {
        double d = r.distance(state.x(), state.y());
        if (d < mn) { mn = d; ribbon = r; }
    }
    state = ribbon.getProjectionAsState(state.x(), state.y());
}

double RibbonManager::getTotalUncoveredLength() const {
    int sum = 0;   // the reference accumulates into an int (RibbonManager.cpp:415)
    for (const auto& r : m_Ribbons) sum += r.length();
    return sum;
}

std::vector<State> RibbonManager::findNearStatesOnRibbons(const State& start, double radius) const {
    std::vector<State> states;
    double h = start.yaw() + M_PI_2;
    double x1 = start.x() + std::cos(h) * radius;
    double x2 = start.x() - std::cos(h) * radius;
    double y1 = start.y() + std::sin(h) * radius;
    double y2 = start.y() - std::sin(h) * radius;
    for (const Ribbon& r : m_Ribbons) {
        auto startProj = r.getProjection(start.x(), start.y());
        {
            double d;
            if (r.containsProjection(startProj)) d = start.distanceTo(startProj.first, startProj.second);
            else d = std::fmin(start.distanceTo(r.start().first, r.start().second), start.distanceTo(r.end().first, r.end().second));
            if (d > 2 * radius) continue;
        }
        auto proj1 = r.getProjection(x1, y1);
        auto proj2 = r.getProjection(x2, y2);
        auto proj = proj2;
        double x = x2, y = y2;
        if (r.containsProjection(proj1)) { proj = proj1; x = x1; y = y1; }
        State s1 = r.startAsState();
        State s2 = r.endAsState();
        State s = (s1.distanceTo(start) < s2.distanceTo(start)) ? s1 : s2;
        double h2 = s.yaw() - M_PI_2;
        double dx1 = std::cos(h2) * radius / 2;
        double dy1 = std::sin(h2) * radius / 2;
        double x3 = proj.first + dx1;
        double y3 = proj.second + dy1;
        double a = dx1 * dx1 + dy1 * dy1;
        double b = std::sqrt(radius * radius - a);
        double h3 = s.yaw();
        double x5 = x3 + b * std::cos(h3);
        double y5 = y3 + b * std::sin(h3);
        double x7 = x5 - x;
        double y7 = y5 - y;
        double h4 = std::atan(y7 / x7);
        double x8 = x5 + radius * std::cos(h4);
        double y8 = y5 + radius * std::sin(h4);
        auto projFinal = r.getProjection(x8, y8);
        double d = dist(projFinal, start.x(), start.y());
        if (d > 1e-5 && d < 2 * radius) states.emplace_back(projFinal.first, projFinal.second, s.heading(), 0, 0);
    }
    return states;
}

}  // namespace ppamd
