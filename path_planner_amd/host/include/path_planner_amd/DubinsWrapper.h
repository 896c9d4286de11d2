// DubinsWrapper / DubinsPlan — a timed Dubins curve and a sequence of them: what Planner::Stats::Plan carries across the
// plan() seam and what the node serialises field by field (NodeBase.h:201-220).
//
// Public interface = the reference's (path_planner_common/include/path_planner_common/{DubinsWrapper,DubinsPlan}.h): same
// member functions, same meaning, exceptions of the same types at the same places.  The representation is this build's: a
// curve (DubinsPath of include/dubins.h) plus a time window
//     t0     when arc length 0 is passed (what sample() measures distance from)
//     from   first time that may be sampled (moves forward with updateStartTime)
//     until  last time that may be sampled (moves backward with updateEndTime)
// and a plan is a flat vector of those, ordered in time.
#pragma once
#include <stdexcept>
#include <vector>

#include "State.h"

extern "C" {
#include "../../../../include/dubins.h"
}

namespace ppamd {

class DubinsWrapper {
public:
    DubinsWrapper() = default;
    DubinsWrapper(const State& s1, const State& s2, double rho) { set(s1, s2, rho); }

    void set(const State& s1, const State& s2, double rho);               // solve s1 -> s2; speed and start time from s1
    void fill(const DubinsPath& path, double speed, double startTime);    // adopt a curve solved elsewhere (the device, a message)
    double length() const;
    bool containsTime(double time) const;
    void sample(State& s) const;                                          // pose at s.time(); sets x, y, heading AND speed
    std::vector<State> getSamples(double timeInterval, double offset) const;
    double getRho() const { return m_Curve.rho; }
    double getSpeed() const { return m_Speed; }
    void setSpeed(double speed) { m_Speed = speed; m_Until = arrival(); }
    double getStartTime() const { return m_From; }
    double getEndTime() const { return m_Until; }
    double getNetTime() const { return m_Until - m_From; }
    void updateEndTime(double endTime);
    void updateStartTime(double startTime);
    const DubinsPath& unwrap() const { return m_Curve; }
    // the start time of the underlying curve: what sample() measures distance from
    double curveStartTime() const { return m_T0; }

private:
    DubinsPath m_Curve{};
    double m_Speed = 0;
    double m_T0 = -1, m_From = -1, m_Until = -1;
    bool solved() const { return m_T0 >= 0; }                             // a default-constructed wrapper holds no curve
    double arrival() const { return m_T0 + length() / m_Speed; }
    void requireSolved(const char* what) const { if (!solved()) throw std::runtime_error(what); }
};

class DubinsPlan {
public:
    DubinsPlan() = default;
    DubinsPlan(const State& s1, const State& s2, double rho) { m_Legs.emplace_back(s1, s2, rho); }
    void append(const DubinsPlan& plan) { m_Legs.insert(m_Legs.end(), plan.m_Legs.begin(), plan.m_Legs.end()); }
    void append(const DubinsWrapper& p) { m_Legs.push_back(p); }
    void sample(State& s) const;                     // the first leg whose window holds s.time(); throws outside the plan
    bool empty() const { return m_Legs.empty(); }
    double totalTime() const { return empty() ? 0 : m_Legs.back().getEndTime() - m_Legs.front().getStartTime(); }
    double getStartTime() const { return first().getStartTime(); }
    double getEndTime() const { return last().getEndTime(); }
    bool containsTime(double time) const;
    void changeIntoSuffix(double startTime);         // drop the legs that end before startTime
    std::vector<State> getHalfSecondSamples() const;
    const std::vector<DubinsWrapper>& get() const { return m_Legs; }
    static constexpr double planTimeDensity() { return 0.5; }
    bool dangerous() const { return m_Dangerous; }
    void setDangerous(bool d) { m_Dangerous = d; }

private:
    std::vector<DubinsWrapper> m_Legs;
    bool m_Dangerous = false;
    const DubinsWrapper& first() const { if (empty()) throw std::runtime_error("the plan has no legs"); return m_Legs.front(); }
    const DubinsWrapper& last() const { if (empty()) throw std::runtime_error("the plan has no legs"); return m_Legs.back(); }
};

}  // namespace ppamd
