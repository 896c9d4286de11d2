// State.h — the vehicle state that crosses the Planner::plan() seam.
//
// Public interface = the reference's State (path_planner_common/include/path_planner_common/State.h:13-213): the same
// accessors and the same arithmetic results (tests/golden/state_ops.json holds the reference object's own outputs and
// tests/test_golden.py requires these functions to reproduce them bit for bit).  Storage and bodies are this build's: five
// doubles in one array — x, y, heading, speed, time — whose first three slots are what dubins_path_sample() writes into
// (DubinsWrapper::sample hands pose() to it), so a State is also exactly one row of the device's sample/vertex records.
// Units: metres, radians east of north (heading) or north of east (yaw), m/s, seconds.
#pragma once
#include <cmath>
#include <string>
#include <utility>

namespace ppamd {

class State {
public:
    enum Slot { X = 0, Y = 1, Heading = 2, Speed = 3, Time = 4 };

    State() = default;                                            // time -1 marks "no state" (State.h:201-202)
    State(double x_, double y_, double heading_, double speed_, double t) : m_V{x_, y_, heading_, speed_, t} {}

    double x() const { return m_V[X]; }
    double& x() { return m_V[X]; }
    double y() const { return m_V[Y]; }
    double& y() { return m_V[Y]; }
    double heading() const { return m_V[Heading]; }
    double& heading() { return m_V[Heading]; }
    double speed() const { return m_V[Speed]; }
    double& speed() { return m_V[Speed]; }
    double time() const { return m_V[Time]; }
    double& time() { return m_V[Time]; }
    double* pose() { return m_V; }                                // {x, y, heading, speed}
    const double* pose() const { return m_V; }

    // heading <-> yaw: both are pi/2 - angle, brought back into [0, 2 pi) by one addition (State.h:51-55,62-65)
    static double flipAngle(double a) {
        const double r = M_PI_2 - a;
        return r < 0 ? r + 2 * M_PI : r;
    }
    double yaw() const { return flipAngle(heading()); }
    void setYaw(double yaw1) { heading() = flipAngle(yaw1); }

    State push(double timeInterval) const;                 // dead reckoning along the heading (State.cpp:11-20)
    void move(double distance);                            // along the yaw, time unchanged (State.cpp:22-25)
    std::string toString() const;                          // "x y heading[deg] speed time" (State.cpp:27-33)
    std::string toStringRad() const;                       // "x y heading[rad] speed time" (State.cpp:35-41)
    double headingTo(double x1, double y1) const;          // State.cpp:51-57
    double headingTo(const State& o) const { return headingTo(o.x(), o.y()); }
    double headingTo(const std::pair<double, double> p) const { return headingTo(p.first, p.second); }
    void setHeadingTowards(double x1, double y1) { heading() = headingTo(x1, y1); }   // State.cpp:64-67
    void setHeadingTowards(const State& o) { setHeadingTowards(o.x(), o.y()); }
    double timeUntil(const State& o) const { return o.time() - time(); }
    bool isCoLocated(const State& r) const { return x() == r.x() && y() == r.y() && heading() == r.heading(); }
    bool operator==(const State& r) const { return isCoLocated(r) && speed() == r.speed() && time() == r.time(); }
    double distanceTo(double x1, double y1) const {
        const double dx = x() - x1, dy = y() - y1;
        return std::sqrt(dx * dx + dy * dy);
    }
    double distanceTo(const State& o) const { return distanceTo(o.x(), o.y()); }
    State interpolate(const State& other, double desiredTime) const;   // State.cpp:95-113
    double headingDifference(double otherHeading) const;               // signed, in [-pi, pi) (State.cpp:119-121)
    double headingDifference(const State& o) const { return headingDifference(o.heading()); }

private:
    double m_V[5] = {0, 0, 0, 0, -1};
};

}  // namespace ppamd
