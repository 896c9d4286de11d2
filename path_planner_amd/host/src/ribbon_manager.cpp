// Ribbon / RibbonManager on the host (interface, provenance and representation: include/path_planner_amd/RibbonManager.h).
//
// Every geometric predicate evaluates the reference's expression in the reference's operation order — the results are
// compared bit for bit with the reference's own objects (tests/golden/ribbon_ops.json) and with the CPU oracle
// (tests/test_host_cpu.py) — but nothing here is a list of ribbon objects: the manager edits one flat array, and the TSP
// heuristics are a depth-first search over index lists and a cost table.
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

namespace {
inline double sq(double v) { return v * v; }
inline double gap(double x1, double y1, double x2, double y2) { return std::sqrt(sq(x1 - x2) + sq(y1 - y2)); }   // RibbonManager.h:285-287
// is p outside [a, b] (either orientation) by more than the tolerance?
inline bool beyondBoth(double p, double a, double b) {
    const double da = p - a, db = p - b, tol = Ribbon::tolerance();
    return (da < -tol && db < -tol) || (da > tol && db > tol);
}
// squared length below which a piece counts as covered (Ribbon.cpp:23-25)
inline double coveredBelow(bool strict) {
    const double whole = Ribbon::minLength() * Ribbon::minLength();
    return strict ? whole / (Ribbon::strictModifier() * Ribbon::strictModifier()) : whole / 1;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ one ribbon
double Ribbon::length() const { return std::sqrt(squaredLength()); }

bool Ribbon::covered(bool strict) const { return squaredLength() < coveredBelow(strict); }

std::pair<double, double> Ribbon::getProjection(double x, double y) const {
    const double ux = ex - sx, uy = ey - sy;                        // the ribbon's direction, not normalised
    const double along = (x - sx) * ux + (y - sy) * uy;
    const double len2 = squaredLength();
    return {ux * along / len2 + sx, uy * along / len2 + sy};
}

bool Ribbon::containsProjection(const std::pair<double, double>& p) const {
    return !(beyondBoth(p.first, sx, ex) || beyondBoth(p.second, sy, ey));
}

double Ribbon::distance(double x, double y) const {
    const double cross = (ey - sy) * x - (ex - sx) * y + ex * sy - ey * sx;
    return std::fabs(cross) / std::sqrt(squaredLength());
}

bool Ribbon::contains(double x, double y, const std::pair<double, double>& projected, bool strict) const {
    const double halfWidth = strict ? RibbonWidth / strictModifier() : RibbonWidth;
    return containsProjection(projected) && distance(x, y) < halfWidth;
}

Ribbon Ribbon::split(double x, double y, bool strict) {
    const auto cut = getProjection(x, y);
    if (!contains(x, y, cut, strict)) return empty();
    const Ribbon before(sx, sy, cut.first, cut.second);
    sx = cut.first;
    sy = cut.second;
    return before;
}

State Ribbon::startAsState() const {
    State s(sx, sy, 0, 0, 0);
    s.setHeadingTowards(ex, ey);
    return s;
}

State Ribbon::endAsState() const {
    State s(ex, ey, 0, 0, 0);
    s.setHeadingTowards(sx, sy);
    return s;
}

State Ribbon::getProjectionAsState(double x, double y) const {
    const auto p = getProjection(x, y);
    State s(p.first, p.second, 0, 0, 0);
    s.setHeadingTowards(ex, ey);
    return s;
}

std::string Ribbon::toString() const {   // the format visualizer.py parses (Ribbon.cpp:45-50)
    std::ostringstream o;
    o << "(" << sx << ", " << sy << ") -> (" << ex << ", " << ey << ") with length " << length();
    return o.str();
}

// ------------------------------------------------------------------------------------------------ the list
void RibbonManager::add(double x1, double y1, double x2, double y2) {
    const Ribbon r(x1, y1, x2, y2);
    if (!r.covered(false)) m_Ribbons.push_back(r);
}

void RibbonManager::assign(const double* r4, int n, double cct) {
    m_Ribbons.clear();
    m_Ribbons.reserve((size_t)(n > 0 ? n : 0));
    for (int i = 0; i < n; i++) m_Ribbons.emplace_back(r4[4 * i], r4[4 * i + 1], r4[4 * i + 2], r4[4 * i + 3]);
    m_CoverageCompletedTime = cct;
}

void RibbonManager::cover(double x, double y, bool strict) {
    // One pass that collects the surviving pieces in list order: a ribbon the point lies in is cut at the projection; of the
    // two parts — and of every untouched ribbon, whose "part before the cut" is the empty ribbon — only those not yet
    // `covered` stay.
    const size_t n = m_Ribbons.size();
    std::vector<Ribbon> kept;
    kept.reserve(n + 1);
    for (size_t i = 0; i < n; i++) {
        Ribbon rest = m_Ribbons[i];
        const Ribbon before = rest.split(x, y, strict);
        if (!before.covered(strict)) kept.push_back(before);
        if (!rest.covered(strict)) kept.push_back(rest);
    }
    m_Ribbons.swap(kept);
}

void RibbonManager::coverBetween(double x1, double y1, double x2, double y2, bool strict) {
    // cover() at points half a minimum length apart from (x1, y1) towards (x2, y2), as long as the remaining distance shrinks,
    // then at (x2, y2) itself
    const double bearing = std::atan((y2 - y1) / (x2 - x1));
    double px = x1, py = y1;
    double remaining = gap(px, py, x2, y2);
    for (;;) {
        const double now = gap(px, py, x2, y2);
        if (now > remaining) break;
        remaining = now;
        cover(px, py, strict);
        px += Ribbon::minLength() * std::cos(bearing) / 2;
        py += Ribbon::minLength() * std::sin(bearing) / 2;
        if (!(remaining > Ribbon::minLength())) break;
    }
    cover(x2, y2, strict);
}

double RibbonManager::minDistanceFrom(double x, double y) const {
    if (m_Ribbons.empty()) return 0;
    double nearest = DBL_MAX;
    for (const Ribbon& r : m_Ribbons) {
        if (r.contains(x, y, r.getProjection(x, y), false)) return 0;
        const double toStart = gap(r.sx, r.sy, x, y), toEnd = gap(r.ex, r.ey, x, y);
        nearest = std::fmin(std::fmin(nearest, toEnd), toStart);
    }
    return nearest;
}

void RibbonManager::changeHeuristicIfTooManyRibbons() {
    if (m_Ribbons.size() > 5) m_Heuristic = MaxDistance;   // c_RibbonCountDangerThreshold (RibbonManager.h:268)
}

std::string RibbonManager::dumpRibbons() const {
    std::string out = "Ribbons: \n";
    if (m_Ribbons.empty()) return out + "None\n";
    for (const Ribbon& r : m_Ribbons) out += r.toString() + "\n";
    return out;
}

void RibbonManager::projectOntoNearestRibbon(State& state) const {
    // nearest by distance to the infinite line, first of equals; an empty list leaves the state alone
    const Ribbon* pick = nullptr;
    double nearest = DBL_MAX;
    for (const Ribbon& r : m_Ribbons) {
        const double d = r.distance(state.x(), state.y());
        if (d < nearest) { nearest = d; pick = &r; }
    }
    if (m_Ribbons.empty()) return;
    state = (pick ? *pick : Ribbon::empty()).getProjectionAsState(state.x(), state.y());
}

double RibbonManager::getTotalUncoveredLength() const {
    int metres = 0;   // the reference's accumulator is an int (`auto sum = 0`, RibbonManager.cpp:415): truncated after every ribbon
    for (const Ribbon& r : m_Ribbons) metres = (int)(metres + r.length());
    return metres;
}

State RibbonManager::getNearestEndpointAsState(const State& state) const {
    if (done()) throw std::logic_error("RibbonManager::getNearestEndpointAsState: there are no ribbons");
    // Each ribbon offers two entry points: just inside either end (half a minimum length plus 1e-5 in), heading along the
    // ribbon.  The nearest entry wins; but if the vehicle is already within a minimum length of it, it is sent to the far end
    // of that ribbon instead (pulled the same amount back inside, keeping the entry's heading).
    const double inset = Ribbon::minLength() / Ribbon::strictModifier() + 1e-5;
    const double pullBack = -Ribbon::minLength() / Ribbon::strictModifier() + 1e-5;
    double nearest = DBL_MAX;
    State chosen;
    auto offer = [&](State entry, State farEnd) {
        entry.move(inset);
        const double d = state.distanceTo(entry);
        if (!(d < nearest)) return;
        nearest = d;
        if (d < Ribbon::minLength()) {
            farEnd.heading() = entry.heading();
            farEnd.move(pullBack);
            chosen = farEnd;
        } else {
            chosen = entry;
        }
    };
    for (const Ribbon& r : m_Ribbons) {
        offer(r.startAsState(), r.endAsState());
        offer(r.endAsState(), r.startAsState());
    }
    return chosen;
}

// Brown-path seeds (AStarPlanner.cpp:40-43): for every ribbon near the start, the point on it that a vehicle reaches by
// swinging out on one turning circle and back in on another.  The construction (RibbonManager.cpp:296-379), step by step:
//   side points     one radius to the left and right of the start
//   foot            the projection of a side point onto the ribbon (the left one if it falls inside the ribbon, else the right)
//   entry heading   that of the ribbon end nearer to the start
//   offset point    the foot moved half a radius sideways (entry yaw - 90 deg) ...
//   circle point    ... then forward along the entry yaw until it is one radius from the foot's line (sqrt(r^2 - offset^2))
//   seed            one radius further along the line from the side point through the circle point, projected back onto the
//                   ribbon; kept if it lies between 1e-5 and two radii from the start
std::vector<State> RibbonManager::findNearStatesOnRibbons(const State& start, double radius) const {
    std::vector<State> seeds;
    const double across = start.yaw() + M_PI_2;
    const double leftX = start.x() + std::cos(across) * radius, rightX = start.x() - std::cos(across) * radius;
    const double leftY = start.y() + std::sin(across) * radius, rightY = start.y() - std::sin(across) * radius;
    for (const Ribbon& r : m_Ribbons) {
        // only ribbons within two radii of the start
        const auto under = r.getProjection(start.x(), start.y());
        const double away = r.containsProjection(under)
                                ? start.distanceTo(under.first, under.second)
                                : std::fmin(start.distanceTo(r.sx, r.sy), start.distanceTo(r.ex, r.ey));
        if (away > 2 * radius) continue;

        const auto footLeft = r.getProjection(leftX, leftY);
        const bool useLeft = r.containsProjection(footLeft);
        const auto foot = useLeft ? footLeft : r.getProjection(rightX, rightY);
        const double sideX = useLeft ? leftX : rightX, sideY = useLeft ? leftY : rightY;

        const State atStart = r.startAsState(), atEnd = r.endAsState();
        const State& entry = (atStart.distanceTo(start) < atEnd.distanceTo(start)) ? atStart : atEnd;

        const double sideways = entry.yaw() - M_PI_2;
        const double offX = std::cos(sideways) * radius / 2, offY = std::sin(sideways) * radius / 2;
        const double forward = std::sqrt(radius * radius - (offX * offX + offY * offY));
        const double ahead = entry.yaw();
        const double circleX = (foot.first + offX) + forward * std::cos(ahead);
        const double circleY = (foot.second + offY) + forward * std::sin(ahead);

        const double bearing = std::atan((circleY - sideY) / (circleX - sideX));
        const auto seed = r.getProjection(circleX + radius * std::cos(bearing), circleY + radius * std::sin(bearing));

        const double reach = gap(seed.first, seed.second, start.x(), start.y());
        if (reach > 1e-5 && reach < 2 * radius) seeds.emplace_back(seed.first, seed.second, entry.heading(), 0, 0);
    }
    return seeds;
}

// ------------------------------------------------------------------------------------------------ heuristics
double RibbonManager::maxDistance(double x, double y) const {
    // max(sum of (length - 2 w) + nearest endpoint, farthest endpoint) (RibbonManager.cpp:234-248)
    double lengths = 0, nearest = DBL_MAX, farthest = 0;
    for (const Ribbon& r : m_Ribbons) {
        lengths += r.length() - 2 * Ribbon::RibbonWidth;
        const double toStart = gap(r.sx, r.sy, x, y), toEnd = gap(r.ex, r.ey, x, y);
        nearest = std::fmin(std::fmin(nearest, toEnd), toStart);
        farthest = std::fmax(std::fmax(farthest, toEnd), toStart);
    }
    return std::fmax(lengths + nearest, farthest);
}

namespace {
// The four "TSP, no split" heuristics (RibbonManager.cpp:53-140) are one search: visit the ribbons in some order, each from one
// end to the other; a visit costs  max(soFar + length - 2 w + approach, 0); minimise the total.  What differs is the approach
// cost (straight line, or a Dubins curve between oriented endpoints) and which ribbons may be visited next (all of them, or —
// K variants — the first K of the remaining list after a stable sort that puts the ribbon whose nearer endpoint is FARTHEST
// first; the Dubins K variant's comparator compares a ribbon with itself, so its sort is the identity, and its counter never
// advances, so it branches on every ribbon unless K <= 0: both quirks kept).
//
// Places: 0 = where the vehicle is, 1 + 2 i = start of ribbon i, 2 + 2 i = its end.  approach[p][q - 1] = cost of going from
// place p to endpoint q, filled once.  A subtree is abandoned when even the most favourable continuation (every remaining
// ribbon adding just its `length - 2 w`, every approach free, the nearest remaining endpoint reached first) cannot get below
// the best total found so far; the comparison carries a margin far above the rounding of the sums it bounds, so the minimum
// returned is the exhaustive one.
struct Tour {
    int n = 0;
    double twoW = 0;
    int K = 0;
    bool limited = false, sorted = false, counting = false;   // K variant / re-sorted at every level / the K counter advances
    std::vector<double> approach;      // (2 n + 1) x (2 n)
    std::vector<double> length;        // per ribbon
    std::vector<double> net;           // per ribbon: length - 2 w, what a visit adds at least
    double best = DBL_MAX;
    // The enumeration is exponential in the list length (the reference's too: 4^n leaves for K = 2).  Lists the planner meets
    // are pieces of at most five ribbons and finish in microseconds to milliseconds; should a list ever need more than this
    // many nodes the search stops and returns the best complete tour found so far, an over-estimate of the exhaustive minimum
    // (the reference's recursion would not return within any planning budget there).
    long nodes = 0;
    static constexpr long kNodeBudget = 50000000;

    double leg(int from, int to) const { return approach[(size_t)from * (2 * n) + (to - 1)]; }

    void visit(const int* left, int m, double soFar, int at) {
        if (m == 0) { best = std::fmin(best, soFar); return; }
        if (++nodes > kNodeBudget && best < DBL_MAX) return;
        if (best < DBL_MAX) {
            double floor = soFar, firstLeg = DBL_MAX;
            for (int i = 0; i < m; i++) {
                floor += net[left[i]];
                firstLeg = std::fmin(firstLeg, std::fmin(leg(at, 1 + 2 * left[i]), leg(at, 2 + 2 * left[i])));
            }
            if (floor + firstLeg > best + (1e-9 + 1e-12 * (std::fabs(best) + std::fabs(floor) + firstLeg))) return;
        }
        int order[64], rest[64];
        std::copy(left, left + m, order);
        if (sorted) {
            // stable, descending by the distance from `at` to the ribbon's nearer endpoint (insertion sort: m is small)
            double key[64];
            for (int i = 0; i < m; i++) key[i] = std::fmin(leg(at, 1 + 2 * order[i]), leg(at, 2 + 2 * order[i]));
            for (int i = 1; i < m; i++) {
                const int r = order[i];
                const double k = key[i];
                int j = i;
                while (j > 0 && k > key[j - 1]) { order[j] = order[j - 1]; key[j] = key[j - 1]; j--; }
                order[j] = r; key[j] = k;
            }
        }
        const int branches = !limited ? m : (counting ? std::min(m, std::max(K, 0)) : (K > 0 ? m : 0));
        for (int b = 0; b < branches; b++) {
            const int r = order[b];
            int c = 0;
            for (int i = 0; i < m; i++) if (i != b) rest[c++] = order[i];
            const double through = soFar + length[r] - twoW;
            visit(rest, m - 1, std::fmax(through + leg(at, 1 + 2 * r), 0), 2 + 2 * r);   // in at the start, out at the end
            visit(rest, m - 1, std::fmax(through + leg(at, 2 + 2 * r), 0), 1 + 2 * r);   // in at the end, out at the start
        }
    }
};
}  // namespace

double RibbonManager::tour(double x, double y, double yaw, bool dubins, bool kVariant) const {
    const int n = (int)m_Ribbons.size();
    if (n > 64) throw std::length_error("RibbonManager: more than 64 ribbons in a TSP heuristic");
    if (dubins && m_TurningRadius == -1) throw std::logic_error("RibbonManager: the Dubins heuristics need a turning radius");
    Tour t;
    t.n = n;
    t.twoW = 2 * Ribbon::RibbonWidth;
    t.K = m_K;
    t.limited = kVariant;
    t.sorted = t.counting = kVariant && !dubins;
    t.length.resize((size_t)n);
    t.net.resize((size_t)n);
    // the places: position and, for the Dubins costs, yaw (a ribbon endpoint looks along the ribbon towards its other end;
    // the vehicle's own "yaw" is whatever the caller passed — the planner passes a heading there, Vertex.cpp:51)
    std::vector<double> px((size_t)2 * n + 1), py((size_t)2 * n + 1), pyaw((size_t)2 * n + 1);
    px[0] = x; py[0] = y; pyaw[0] = yaw;
    for (int i = 0; i < n; i++) {
        const Ribbon& r = m_Ribbons[(size_t)i];
        t.length[(size_t)i] = r.length();
        t.net[(size_t)i] = t.length[(size_t)i] - t.twoW;
        px[(size_t)1 + 2 * i] = r.sx; py[(size_t)1 + 2 * i] = r.sy;
        px[(size_t)2 + 2 * i] = r.ex; py[(size_t)2 + 2 * i] = r.ey;
        if (dubins) {
            pyaw[(size_t)1 + 2 * i] = r.startAsState().yaw();
            pyaw[(size_t)2 + 2 * i] = r.endAsState().yaw();
        }
    }
    t.approach.assign((size_t)(2 * n + 1) * (2 * n), 0.0);
    for (int p = 0; p <= 2 * n; p++)
        for (int q = 1; q <= 2 * n; q++) {
            double cost;
            if (dubins) {
                DubinsPath path;
                double from[3] = {px[(size_t)p], py[(size_t)p], pyaw[(size_t)p]}, to[3] = {px[(size_t)q], py[(size_t)q], pyaw[(size_t)q]};
                dubins_shortest_path(&path, from, to, m_TurningRadius);
                cost = dubins_path_length(&path);
            } else {
                cost = gap(px[(size_t)p], py[(size_t)p], px[(size_t)q], py[(size_t)q]);
            }
            t.approach[(size_t)p * (2 * n) + (q - 1)] = cost;
        }
    int all[64];
    for (int i = 0; i < n; i++) all[i] = i;
    t.visit(all, n, 0.0, 0);
    return t.best;
}

double RibbonManager::approximateDistanceUntilDone(double x, double y, double yaw) const {
    if (done()) return 0;
    switch (m_Heuristic) {
    case MaxDistance: return maxDistance(x, y);
    case TspPointRobotNoSplitAllRibbons: return tour(x, y, yaw, false, false);
    case TspPointRobotNoSplitKRibbons: return tour(x, y, yaw, false, true);
    case TspDubinsNoSplitAllRibbons: return tour(x, y, yaw, true, false);
    case TspDubinsNoSplitKRibbons: return tour(x, y, yaw, true, true);
    }
    return 0;
}

}  // namespace ppamd
