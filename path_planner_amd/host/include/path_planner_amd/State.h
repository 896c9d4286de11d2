// State.h — host mirror of the reference's State
// (/root/reference/path_planner_common/include/path_planner_common/State.h:13-213, src/state/State.cpp).
// Same member names and meanings; units: metres, radians east of north, m/s, seconds.
#pragma once
#include <cmath>
#include <string>
#include <utility>

namespace ppamd {

class State {
public:
    double x() const { return m_Pose[0]; }
    double& x() { return m_Pose[0]; }
    double y() const { return m_Pose[1]; }
    double& y() { return m_Pose[1]; }
    double heading() const { return m_Pose[2]; }
    double& heading() { return m_Pose[2]; }
    double speed() const { return m_Pose[3]; }
    double& speed() { return m_Pose[3]; }
    double time() const { return m_Time; }
    double& time() { return m_Time; }
    double* pose() { return m_Pose; }
    const double* pose() const { return m_Pose; }

    // heading north of east (State.h:51-55)
    double yaw() const {
        double h = M_PI_2 - heading();
        if (h < 0) h += 2 * M_PI;
        return h;
    }
    // State.h:62-65 (declared double in the reference but returns nothing; void here)
    void setYaw(double yaw1) {
        heading() = M_PI_2 - yaw1;
        if (heading() < 0) heading() += 2 * M_PI;
    }

    State() = default;
    State(double x_, double y_, double heading_, double speed_, double t) {
        m_Pose[0] = x_; m_Pose[1] = y_; m_Pose[2] = heading_; m_Pose[3] = speed_; m_Time = t;
    }

    State push(double timeInterval) const;                 // State.cpp:11-20
    void move(double distance);                            // State.cpp:22-25
    std::string toString() const;                          // State.cpp:27-33
    std::string toStringRad() const;                       // State.cpp:35-41
    double headingTo(double x1, double y1) const;          // State.cpp:51-57
    double headingTo(const State& o) const { return headingTo(o.x(), o.y()); }
    double headingTo(const std::pair<double, double> p) const { return headingTo(p.first, p.second); }
    void setHeadingTowards(double x1, double y1);          // State.cpp:64-67
    void setHeadingTowards(const State& o) { setHeadingTowards(o.x(), o.y()); }
    double timeUntil(const State& o) const { return o.time() - time(); }
    bool operator==(const State& r) const {
        return x() == r.x() && y() == r.y() && heading() == r.heading() && speed() == r.speed() && time() == r.time();
    }
    bool isCoLocated(const State& r) const { return x() == r.x() && y() == r.y() && heading() == r.heading(); }
    double distanceTo(double x1, double y1) const { return std::sqrt((x() - x1) * (x() - x1) + (y() - y1) * (y() - y1)); }
    double distanceTo(const State& o) const { return distanceTo(o.x(), o.y()); }
    State interpolate(const State& other, double desiredTime) const;   // State.cpp:95-113
    double headingDifference(double otherHeading) const;               // State.cpp:119-121
    double headingDifference(const State& o) const { return headingDifference(o.heading()); }

private:
    double m_Pose[4] = {0, 0, 0, 0};
    double m_Time = -1;
};

}  // namespace ppamd
