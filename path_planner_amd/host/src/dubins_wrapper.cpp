// DubinsWrapper / DubinsPlan — timed Dubins curves over the C library of include/dubins.h (interface and provenance:
// include/path_planner_amd/DubinsWrapper.h).  Behaviour follows path_planner_common/src/dubinsPlan/{DubinsWrapper,DubinsPlan}.cpp
// — which calls throw (std::runtime_error), the EDUBPARAM retry 1e-5 short of the requested arc length (:39-42), speed written
// into the sampled state (:48) — the code is this build's.
#include "path_planner_amd/DubinsWrapper.h"

#include <algorithm>
#include <cstdio>
#include <iostream>

namespace ppamd {

// ------------------------------------------------------------------------------------------------ one curve
void DubinsWrapper::set(const State& s1, const State& s2, double rho) {
    double from[3] = {s1.x(), s1.y(), s1.yaw()};
    double to[3] = {s2.x(), s2.y(), s2.yaw()};
    dubins_shortest_path(&m_Curve, from, to, rho);
    m_Speed = s1.speed();
    m_T0 = m_From = s1.time();
    m_Until = arrival();
}

void DubinsWrapper::fill(const DubinsPath& path, double speed, double startTime) {
    m_Curve = path;
    m_Speed = speed;
    m_T0 = m_From = startTime;
    m_Until = arrival();
}

double DubinsWrapper::length() const {
    requireSolved("DubinsWrapper::length: no curve has been set");
    return dubins_path_length(&m_Curve);
}

bool DubinsWrapper::containsTime(double time) const {
    requireSolved("DubinsWrapper::containsTime: no curve has been set");
    return time >= m_From && time <= m_Until;   // closed window; NaN is in no window
}

void DubinsWrapper::sample(State& s) const {
    if (!containsTime(s.time())) {
        char msg[192];
        std::snprintf(msg, sizeof(msg), "DubinsWrapper::sample: time %f is outside the curve's window [%f, %f]", s.time(), m_From, m_Until);
        throw std::runtime_error(msg);
    }
    const double arc = (s.time() - m_T0) * m_Speed;
    // the library writes (x, y, yaw) into the first three pose slots; a parameter just past either end of the curve — the last
    // step of an edge, by rounding — is retried a hair earlier, as the reference does
    int rc = dubins_path_sample(&m_Curve, arc, s.pose());
    if (rc == EDUBPARAM) rc = dubins_path_sample(&m_Curve, arc - 1e-5, s.pose());
    if (rc != EDUBOK) std::cerr << "DubinsWrapper::sample: the Dubins library reported error " << rc << std::endl;
    s.setYaw(s.heading());
    s.speed() = m_Speed;
}

std::vector<State> DubinsWrapper::getSamples(double timeInterval, double) const {
    std::vector<State> out;
    State probe;
    for (double t = m_From; t < m_Until; t += timeInterval) {   // a running sum, like every time grid of the planner
        probe.time() = t;
        sample(probe);
        out.push_back(probe);
    }
    return out;
}

void DubinsWrapper::updateEndTime(double endTime) {
    if (m_Until == -1) throw std::runtime_error("DubinsWrapper::updateEndTime: no curve has been set");
    if (endTime > m_Until) throw std::runtime_error("DubinsWrapper::updateEndTime: a curve can only be cut shorter");
    m_Until = endTime;
}

void DubinsWrapper::updateStartTime(double startTime) {
    requireSolved("DubinsWrapper::updateStartTime: no curve has been set");
    if (startTime < m_T0) throw std::runtime_error("DubinsWrapper::updateStartTime: a curve can only start later");
    // The reference re-bases the curve here through dubins_extract_subpath (DubinsWrapper.cpp:106-115); that call's semantics
    // are unpinned and nothing on the planning path reaches it.
    const double skipped = (startTime - m_T0) * m_Speed;
    m_From = m_T0 = startTime;
    const DubinsPath whole = m_Curve;
    dubins_extract_subpath(&whole, skipped, &m_Curve);
}

// ------------------------------------------------------------------------------------------------ a sequence of curves
void DubinsPlan::sample(State& s) const {
    const auto leg = std::find_if(m_Legs.begin(), m_Legs.end(), [&](const DubinsWrapper& w) { return w.containsTime(s.time()); });
    if (leg == m_Legs.end()) throw std::runtime_error("DubinsPlan::sample: the requested time is outside the plan");
    leg->sample(s);
}

bool DubinsPlan::containsTime(double time) const {
    return std::any_of(m_Legs.begin(), m_Legs.end(), [&](const DubinsWrapper& w) { return w.containsTime(time); });
}

void DubinsPlan::changeIntoSuffix(double startTime) {
    if (empty()) throw std::runtime_error("DubinsPlan::changeIntoSuffix: the plan has no legs");
    // legs are in time order: the ones to drop form a prefix
    const auto keep = std::find_if(m_Legs.begin(), m_Legs.end(), [&](const DubinsWrapper& w) { return !(w.getEndTime() < startTime); });
    m_Legs.erase(m_Legs.begin(), keep);
}

std::vector<State> DubinsPlan::getHalfSecondSamples() const {
    std::vector<State> out;
    if (empty()) return out;
    const double end = getEndTime();
    State probe;
    for (double t = getStartTime(); t < end; t += planTimeDensity()) {
        probe.time() = t;
        sample(probe);
        out.push_back(probe);
    }
    return out;
}

}  // namespace ppamd
