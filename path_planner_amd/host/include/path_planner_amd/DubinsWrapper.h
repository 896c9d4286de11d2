// DubinsWrapper.h / DubinsPlan — host mirrors of
// /root/reference/path_planner_common/include/path_planner_common/{DubinsWrapper,DubinsPlan}.h and
// src/dubinsPlan/{DubinsWrapper,DubinsPlan}.cpp, over the C library of include/dubins.h.
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
    void set(const State& s1, const State& s2, double rho);               // DubinsWrapper.cpp:9-17
    void fill(const DubinsPath& path, double speed, double startTime);    // :85-90
    double length() const;                                                // :19-22
    bool containsTime(double time) const;                                 // :24-27
    void sample(State& s) const;                                          // :29-49
    std::vector<State> getSamples(double timeInterval, double offset) const;   // :55-67
    double getRho() const { return m_DubinsPath.rho; }
    double getSpeed() const { return m_Speed; }
    void setSpeed(double speed) { m_Speed = speed; setEndTime(); }        // :121-124
    double getStartTime() const { return m_UpdatedStartTime; }
    double getEndTime() const { return m_EndTime; }
    double getNetTime() const { return getEndTime() - getStartTime(); }
    void updateEndTime(double endTime);                                   // :100-104
    void updateStartTime(double startTime);                               // :106-115
    const DubinsPath& unwrap() const { return m_DubinsPath; }
    // the start time of the underlying curve (m_StartTime): what sample() measures distance from
    double curveStartTime() const { return m_StartTime; }

private:
    DubinsPath m_DubinsPath{};
    double m_Speed{};
    double m_StartTime = -1, m_EndTime = -1, m_UpdatedStartTime = -1;
    bool isInitialized() const { return m_StartTime >= 0; }
    void setEndTime() { m_EndTime = m_StartTime + length() / m_Speed; }
};

class DubinsPlan {
public:
    DubinsPlan() = default;
    DubinsPlan(const State& s1, const State& s2, double rho) { m_DubinsPaths.emplace_back(s1, s2, rho); }
    void append(const DubinsPlan& plan) { for (const auto& s : plan.m_DubinsPaths) append(s); }
    void append(const DubinsWrapper& p) { m_DubinsPaths.push_back(p); }
    void sample(State& s) const;                     // DubinsPlan.cpp:11-19
    bool empty() const { return m_DubinsPaths.empty(); }
    double totalTime() const;
    double getStartTime() const;
    double getEndTime() const;
    bool containsTime(double time) const;
    void changeIntoSuffix(double startTime);         // DubinsPlan.cpp:66-77
    std::vector<State> getHalfSecondSamples() const; // :29-40
    const std::vector<DubinsWrapper>& get() const { return m_DubinsPaths; }
    static constexpr double planTimeDensity() { return 0.5; }
    bool dangerous() const { return m_Dangerous; }
    void setDangerous(bool d) { m_Dangerous = d; }

private:
    std::vector<DubinsWrapper> m_DubinsPaths;
    bool m_Dangerous = false;
};

}  // namespace ppamd
