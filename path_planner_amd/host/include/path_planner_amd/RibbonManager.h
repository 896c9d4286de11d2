// Ribbon / RibbonManager — the survey lines still to be covered, on the host.
//
// Public interface = the reference's (path_planner/src/planner/utilities/{Ribbon,RibbonManager}.h): same member functions and
// the same results — tests/golden/ribbon_ops.json holds the reference Ribbon object's own outputs and tests/test_golden.py
// requires these classes to reproduce them bit for bit; tests/test_host_cpu.py compares the manager with the CPU oracle.
// On the GPU path the host needs them only OFF the hot loop: the root's heuristic, the nearest ribbon endpoint of a vertex
// being expanded, Brown-path seeds, Executive bookkeeping, and h of the rare child whose ribbon list is longer than the
// device's enumeration limit.  The per-step cover()/minDistanceFrom() work and h of every costed edge run on the device.
//
// Representation (this build's): a Ribbon is four doubles {startX, startY, endX, endY} and nothing else, and a manager keeps
// its ribbons in ONE contiguous array in list order — the row layout of the device's ribbon pool (ppgpu_set_vertices) and of a
// costed edge's child ribbons, so lists go to and come from the device with a memcpy (rows() / assign()).  The TSP heuristics
// run over a table of point-to-endpoint costs with index lists instead of copying lists of ribbons at every level of the
// enumeration, and prune subtrees that provably cannot lower the minimum (which is what makes lists longer than the device
// handles affordable); the value returned is the reference's, to the bit.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "State.h"

namespace ppamd {

class Ribbon {
public:
    static double RibbonWidth;   // HALF width of the swath; process-global like the reference's (Ribbon.h:16, default 1.5)

    Ribbon(double startX, double startY, double endX, double endY) : sx(startX), sy(startY), ex(endX), ey(endY) {}
    static Ribbon empty() { return Ribbon(0, 0, 0, 0); }

    // cut at the projection of (x, y) if the point lies in the swath: returns the part before the cut, keeps the part after
    // it; a point outside returns empty() and changes nothing (Ribbon.cpp:9-17)
    Ribbon split(double x, double y, bool strict);
    bool covered(bool strict) const;                       // shorter than minLength (strict: half of it) (Ribbon.cpp:23-25)
    double length() const;
    std::pair<double, double> start() const { return {sx, sy}; }
    std::pair<double, double> end() const { return {ex, ey}; }
    State startAsState() const;                            // at the start, heading towards the end; speed 0, time 0
    State endAsState() const;                              // at the end, heading towards the start
    bool contains(double x, double y, const std::pair<double, double>& projected, bool strict) const;
    bool containsProjection(const std::pair<double, double>& projected) const;
    std::string toString() const;
    static double minLength() { return 2 * RibbonWidth; }
    std::pair<double, double> getProjection(double x, double y) const;
    State getProjectionAsState(double x, double y) const;
    double distance(double x, double y) const;             // to the infinite line through the ribbon (Ribbon.h:118-121)
    static constexpr double strictModifier() { return 2; }
    static constexpr double tolerance() { return 1e-5; }   // Ribbon::c_Tolerance (Ribbon.h:129)

    const double* row() const { return &sx; }              // {startX, startY, endX, endY}

private:
    double sx, sy, ex, ey;
    double squaredLength() const { return (ex - sx) * (ex - sx) + (ey - sy) * (ey - sy); }
    friend class RibbonManager;
};
static_assert(sizeof(Ribbon) == 4 * sizeof(double), "a Ribbon is one row of the device's ribbon pool");

class RibbonManager {
public:
    enum Heuristic { MaxDistance, TspPointRobotNoSplitAllRibbons, TspPointRobotNoSplitKRibbons, TspDubinsNoSplitAllRibbons,
                     TspDubinsNoSplitKRibbons };
    RibbonManager() : RibbonManager(MaxDistance) {}
    explicit RibbonManager(Heuristic h) : m_Heuristic(h) {}
    RibbonManager(Heuristic h, double turningRadius) : m_Heuristic(h), m_TurningRadius(turningRadius) {}
    RibbonManager(Heuristic h, double turningRadius, int k) : m_Heuristic(h), m_TurningRadius(turningRadius), m_K(k) {}

    void add(double x1, double y1, double x2, double y2);                        // appended unless already shorter than minLength
    void cover(double x, double y, bool strict);                                 // RibbonManager.cpp:14-22
    void coverBetween(double x1, double y1, double x2, double y2, bool strict);  // cover() along a straight line (:250-264)
    bool done() const { return m_Ribbons.empty(); }
    double approximateDistanceUntilDone(double x, double y, double yaw) const;   // the configured heuristic (:28-51)
    void changeHeuristicIfTooManyRibbons();                                      // more than 5 ribbons: MaxDistance (:381-385)
    double minDistanceFrom(double x, double y) const;                            // 0 inside a swath, else nearest endpoint (:142-152)
    State getNearestEndpointAsState(const State& state) const;                   // where expand() aims first (:160-195)
    std::string dumpRibbons() const;
    void projectOntoNearestRibbon(State& state) const;                           // the sampler's 1 % projection (:220-232)
    const std::vector<Ribbon>& get() const { return m_Ribbons; }
    std::vector<State> findNearStatesOnRibbons(const State& start, double radius) const;   // Brown-path seeds (:296-379)
    void setHeuristic(Heuristic h) { m_Heuristic = h; }
    Heuristic heuristic() const { return m_Heuristic; }
    int k() const { return m_K; }
    double turningRadius() const { return m_TurningRadius; }
    static void setRibbonWidth(double w) { Ribbon::RibbonWidth = w; }
    double coverageCompletedTime() const { return m_CoverageCompletedTime; }
    void setCoverageCompletedTime(double t) { if (m_CoverageCompletedTime == -1) m_CoverageCompletedTime = t; }
    double getTotalUncoveredLength() const;

    // ---- the device's view: the list as rows of four doubles, in list order
    int count() const { return (int)m_Ribbons.size(); }
    const double* rows() const { return m_Ribbons.empty() ? nullptr : m_Ribbons.front().row(); }
    // a costed edge came back from the device: the child's ribbon list and completion time
    void assign(const double* ribbons4, int n, double coverageCompletedTime);

private:
    Heuristic m_Heuristic;
    double m_TurningRadius = -1;
    int m_K = 0;
    double m_CoverageCompletedTime = -1;
    std::vector<Ribbon> m_Ribbons;
    double maxDistance(double x, double y) const;
    double tour(double x, double y, double yaw, bool dubins, bool kVariant) const;
};

}  // namespace ppamd
