// Ribbon / RibbonManager — host mirrors of /root/reference/path_planner/src/planner/utilities/{Ribbon,RibbonManager}.{h,cpp}.
// On the GPU path these are needed only OFF the hot loop: the root's heuristic, the nearest ribbon endpoint of the vertex
// being expanded, coverBetween for the executive, Brown-path seeds.  The per-step cover/minDistance work and the heuristic
// of every costed edge run on the device (pp_k_cost_edges / pp_k_heuristic).
#pragma once
#include <list>
#include <string>
#include <utility>
#include <vector>

#include "State.h"

namespace ppamd {

class Ribbon {
public:
    static double RibbonWidth;   // half width; process-global like the reference's (Ribbon.h:16)
    Ribbon(double startX, double startY, double endX, double endY) : m_StartX(startX), m_StartY(startY), m_EndX(endX), m_EndY(endY) {}
    Ribbon split(double x, double y, bool strict);
    bool covered(bool strict) const;
    static Ribbon empty() { return Ribbon(0, 0, 0, 0); }
    double length() const;
    std::pair<double, double> start() const { return {m_StartX, m_StartY}; }
    std::pair<double, double> end() const { return {m_EndX, m_EndY}; }
    State startAsState() const;
    State endAsState() const;
    bool contains(double x, double y, const std::pair<double, double>& projected, bool strict) const;
    bool containsProjection(const std::pair<double, double>& projected) const;
    std::string toString() const;
    static double minLength() { return 2 * RibbonWidth; }
    std::pair<double, double> getProjection(double x, double y) const;
    State getProjectionAsState(double x, double y) const;
    double distance(double x, double y) const;
    static constexpr double strictModifier() { return 2; }

private:
    double m_StartX, m_StartY, m_EndX, m_EndY;
    double squaredLength() const { return (m_EndX - m_StartX) * (m_EndX - m_StartX) + (m_EndY - m_StartY) * (m_EndY - m_StartY); }
};

class RibbonManager {
public:
    enum Heuristic { MaxDistance, TspPointRobotNoSplitAllRibbons, TspPointRobotNoSplitKRibbons, TspDubinsNoSplitAllRibbons,
                     TspDubinsNoSplitKRibbons };
    RibbonManager() : RibbonManager(MaxDistance) {}
    explicit RibbonManager(Heuristic h) : m_Heuristic(h) {}
    RibbonManager(Heuristic h, double turningRadius) : m_Heuristic(h), m_TurningRadius(turningRadius) {}
    RibbonManager(Heuristic h, double turningRadius, int k) : m_Heuristic(h), m_TurningRadius(turningRadius), m_K(k) {}

    void add(double x1, double y1, double x2, double y2);
    void cover(double x, double y, bool strict);
    void coverBetween(double x1, double y1, double x2, double y2, bool strict);
    bool done() const { return m_Ribbons.empty(); }
    double approximateDistanceUntilDone(double x, double y, double yaw) const;
    void changeHeuristicIfTooManyRibbons();
    double minDistanceFrom(double x, double y) const;
    State getNearestEndpointAsState(const State& state) const;
    std::string dumpRibbons() const;
    void projectOntoNearestRibbon(State& state) const;
    const std::list<Ribbon>& get() const { return m_Ribbons; }
    std::vector<State> findNearStatesOnRibbons(const State& start, double radius) const;
    void setHeuristic(Heuristic h) { m_Heuristic = h; }
    Heuristic heuristic() const { return m_Heuristic; }
    int k() const { return m_K; }
    double turningRadius() const { return m_TurningRadius; }
    static void setRibbonWidth(double w) { Ribbon::RibbonWidth = w; }
    double coverageCompletedTime() const { return m_CoverageCompletedTime; }
    void setCoverageCompletedTime(double t) { if (m_CoverageCompletedTime == -1) m_CoverageCompletedTime = t; }
    double getTotalUncoveredLength() const;
    // used when a costed edge comes back from the device: the child's ribbon list and completion time
    void assign(const double* ribbons4, int n, double coverageCompletedTime);

private:
    Heuristic m_Heuristic;
    double m_TurningRadius = -1;
    int m_K = 0;
    double m_CoverageCompletedTime = -1;
    std::list<Ribbon> m_Ribbons;
    double dubinsDistance(double x, double y, double h, const State& s) const;
    double maxDistance(double x, double y) const;
    double tspPoint(std::list<Ribbon> left, double soFar, std::pair<double, double> point, bool kVariant) const;
    double tspDubins(std::list<Ribbon> left, double soFar, double x, double y, double yaw, bool kVariant) const;
};

}  // namespace ppamd
