// DubinsWrapper / DubinsPlan — mirrors of path_planner_common/src/dubinsPlan/{DubinsWrapper,DubinsPlan}.cpp
#include "path_planner_amd/DubinsWrapper.h"

#include <iostream>
#include <sstream>

namespace ppamd {

void DubinsWrapper::set(const State& s1, const State& s2, double rho) {
    double q1[3] = {s1.x(), s1.y(), s1.yaw()};
    double q2[3] = {s2.x(), s2.y(), s2.yaw()};
    dubins_shortest_path(&m_DubinsPath, q1, q2, rho);
    m_Speed = s1.speed();
    m_UpdatedStartTime = m_StartTime = s1.time();
    setEndTime();
}

void DubinsWrapper::fill(const DubinsPath& path, double speed, double startTime) {
    m_DubinsPath = path;
    m_Speed = speed;
    m_UpdatedStartTime = m_StartTime = startTime;
    setEndTime();
}

double DubinsWrapper::length() const {
    if (!isInitialized()) throw std::runtime_error("Cannot access unset Dubins wrapper");
    return dubins_path_length(&m_DubinsPath);
}

bool DubinsWrapper::containsTime(double time) const {
    if (!isInitialized()) throw std::runtime_error("Checking time constraints on uninitialized Dubins wrapper");
    return m_UpdatedStartTime <= time && m_EndTime >= time;
}

void DubinsWrapper::sample(State& s) const {
    if (!containsTime(s.time())) {
        std::stringstream stream;
        stream << "Invalid time " << std::to_string(s.time()) << " in sample for Dubins path which spans from "
               << std::to_string(getStartTime()) << " to " << std::to_string(getEndTime());
        throw std::runtime_error(stream.str());
    }
    double distance = (s.time() - m_StartTime) * m_Speed;
    int err = dubins_path_sample(&m_DubinsPath, distance, s.pose());   // heading slot receives yaw
    if (err == EDUBPARAM) err = dubins_path_sample(&m_DubinsPath, distance - 1e-5, s.pose());
    if (err != EDUBOK) std::cerr << "Encountered error in dubins library" << std::endl;
    s.setYaw(s.heading());
    s.speed() = m_Speed;
}

std::vector<State> DubinsWrapper::getSamples(double timeInterval, double) const {
    std::vector<State> result;
    State intermediate;
    intermediate.speed() = m_Speed;
    for (double s = m_UpdatedStartTime; s < m_EndTime; s += timeInterval) {
        intermediate.time() = s;
        sample(intermediate);
        result.push_back(intermediate);
    }
    return result;
}

void DubinsWrapper::updateEndTime(double endTime) {
    if (m_EndTime == -1) throw std::runtime_error("Cannot access unset Dubins wrapper");
    if (endTime > m_EndTime) throw std::runtime_error("Invalid end time for Dubins wrapper");
    m_EndTime = endTime;
}

void DubinsWrapper::updateStartTime(double startTime) {
    if (!isInitialized()) throw std::runtime_error("Cannot access unset Dubins wrapper");
    if (startTime < m_StartTime) throw std::runtime_error("Invalid start time for Dubins wrapper");
    m_UpdatedStartTime = startTime;
    double d = (m_UpdatedStartTime - m_StartTime) * m_Speed;
    m_StartTime = startTime;
    DubinsPath copy = m_DubinsPath;
    dubins_extract_subpath(&copy, d, &m_DubinsPath);
}

void DubinsPlan::sample(State& s) const {
    for (const auto& p : m_DubinsPaths) {
        if (p.containsTime(s.time())) {
            p.sample(s);
            return;
        }
    }
    throw std::runtime_error("Requested time outside plan bounds");
}

double DubinsPlan::totalTime() const {
    if (empty()) return 0;
    return m_DubinsPaths.back().getEndTime() - m_DubinsPaths.front().getStartTime();
}

double DubinsPlan::getStartTime() const {
    if (m_DubinsPaths.empty()) throw std::runtime_error("Cannot access empty plan");
    return m_DubinsPaths.front().getStartTime();
}

double DubinsPlan::getEndTime() const {
    if (m_DubinsPaths.empty()) throw std::runtime_error("Cannot access empty plan");
    return m_DubinsPaths.back().getEndTime();
}

bool DubinsPlan::containsTime(double time) const {
    for (const auto& p : m_DubinsPaths) if (p.containsTime(time)) return true;
    return false;
}

void DubinsPlan::changeIntoSuffix(double startTime) {
    if (m_DubinsPaths.empty()) throw std::runtime_error("Cannot access empty plan");
    while (!m_DubinsPaths.empty() && m_DubinsPaths.front().getEndTime() < startTime) m_DubinsPaths.erase(m_DubinsPaths.begin());
}

std::vector<State> DubinsPlan::getHalfSecondSamples() const {
    std::vector<State> result;
    if (empty()) return result;
    State s;
    for (double time = getStartTime(); time < getEndTime(); time += planTimeDensity()) {
        s.time() = time;
        sample(s);
        result.push_back(s);
    }
    return result;
}

}  // namespace ppamd
