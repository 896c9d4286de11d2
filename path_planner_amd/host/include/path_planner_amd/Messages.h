// Wire formats of the planner's ROS interface without ROS: the five messages of path_planner_common/msg
// (DubinsPath.msg, Plan.msg, StateMsg.msg, Stats.msg, TaskLevelStats.msg) as plain structs with the same field names and
// order, the conversions the node performs (NodeBase.h:200-219 convertToPlanMsg; TrajectoryDisplayerHelper's state
// conversions), and ROS 1 serialisation (little-endian fields in declaration order, arrays prefixed by a uint32 count, bool as
// one byte) so that a bridge can hand the bytes to a ROS publisher unchanged.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "Planner.h"

namespace ppamd {
namespace msg {

struct DubinsPath {   // DubinsPath.msg
    double initial_x = 0, initial_y = 0, initial_yaw = 0;
    double length0 = 0, length1 = 0, length2 = 0;
    double rho = 0;
    int32_t type = 0;     // LSL = 0, LSR = 1, RSL = 2, RSR = 3, RLR = 4, LRL = 5
    double speed = 0, start_time = 0;
};
struct Plan {         // Plan.msg
    std::vector<DubinsPath> paths;
    double endtime = 0;   // paths may have been truncated
};
struct StateMsg { double x = 0, y = 0, heading = 0, speed = 0, time = 0; };   // StateMsg.msg
struct Stats {        // Stats.msg
    int64_t samples = 0, generated = 0, expanded = 0, iterations = 0;
    double plan_f_value = 0, plan_collision_penalty = 0, plan_time_penalty = 0, plan_h_value = 0;
    int64_t plan_depth = 0;
    double collision_penalty = 0;
    int64_t cpu_time = 0;
    bool last_plan_achievable = false;
};
struct TaskLevelStats { double time = 0, collision_penalty = 0, score = 0, uncovered_length = 0; };   // TaskLevelStats.msg

// ---- conversions (NodeBase.h:200-219, path_planner_node.cpp publishStats / publishTaskLevelStats)
inline Plan convertToPlanMsg(const DubinsPlan& plan) {
    Plan planMsg;
    for (const auto& d : plan.get()) {
        DubinsPath path;
        const ::DubinsPath& p = d.unwrap();
        path.initial_x = p.qi[0]; path.initial_y = p.qi[1]; path.initial_yaw = p.qi[2];
        path.length0 = p.param[0]; path.length1 = p.param[1]; path.length2 = p.param[2];
        path.type = (int32_t)p.type;
        path.rho = d.getRho();
        path.speed = d.getSpeed();
        path.start_time = d.getStartTime();
        planMsg.paths.push_back(path);
    }
    planMsg.endtime = plan.empty() ? 0.0 : plan.getEndTime();
    return planMsg;
}
// what the controller does with a received Plan (path_planner_common's DubinsPlan(msg) constructor: fill + truncate the last)
inline DubinsPlan convertFromPlanMsg(const Plan& planMsg) {
    DubinsPlan plan;
    for (const auto& m : planMsg.paths) {
        ::DubinsPath p;
        p.qi[0] = m.initial_x; p.qi[1] = m.initial_y; p.qi[2] = m.initial_yaw;
        p.param[0] = m.length0; p.param[1] = m.length1; p.param[2] = m.length2;
        p.rho = m.rho;
        p.type = (DubinsPathType)m.type;
        DubinsWrapper w;
        w.fill(p, m.speed, m.start_time);
        plan.append(w);
    }
    if (!planMsg.paths.empty()) {
        std::vector<DubinsWrapper> v = plan.get();
        if (v.back().getEndTime() > planMsg.endtime) {
            v.back().updateEndTime(planMsg.endtime);
            DubinsPlan q;
            for (const auto& w : v) q.append(w);
            q.setDangerous(plan.dangerous());
            return q;
        }
    }
    return plan;
}
inline StateMsg convertToStateMsg(const State& s) { return StateMsg{s.x(), s.y(), s.heading(), s.speed(), s.time()}; }
inline State convertToStateFromMsg(const StateMsg& m) { return State(m.x, m.y, m.heading, m.speed, m.time); }
inline Stats convertToStatsMsg(const Planner::Stats& stats, double collisionPenalty, unsigned long cpuTime, bool lastPlanAchievable) {
    Stats m;
    m.samples = (int64_t)stats.Samples; m.generated = (int64_t)stats.Generated; m.expanded = (int64_t)stats.Expanded;
    m.iterations = (int64_t)stats.Iterations;
    m.plan_f_value = stats.PlanFValue; m.plan_collision_penalty = stats.PlanCollisionPenalty; m.plan_time_penalty = stats.PlanTimePenalty;
    m.plan_h_value = stats.PlanHValue; m.plan_depth = (int64_t)stats.PlanDepth;
    m.collision_penalty = collisionPenalty; m.cpu_time = (int64_t)cpuTime; m.last_plan_achievable = lastPlanAchievable;
    return m;
}

// ---- ROS 1 serialisation
class Writer {
public:
    std::vector<uint8_t> bytes;
    template <typename T> void put(const T& v) { const uint8_t* p = reinterpret_cast<const uint8_t*>(&v); bytes.insert(bytes.end(), p, p + sizeof(T)); }
};
class Reader {
public:
    Reader(const uint8_t* p, size_t n) : m_P(p), m_N(n) {}
    template <typename T> T get() {
        if (m_Off + sizeof(T) > m_N) throw std::runtime_error("message truncated");
        T v;
        std::memcpy(&v, m_P + m_Off, sizeof(T));
        m_Off += sizeof(T);
        return v;
    }
    size_t offset() const { return m_Off; }
private:
    const uint8_t* m_P; size_t m_N, m_Off = 0;
};
inline void serialize(Writer& w, const DubinsPath& m) {
    w.put(m.initial_x); w.put(m.initial_y); w.put(m.initial_yaw); w.put(m.length0); w.put(m.length1); w.put(m.length2); w.put(m.rho);
    w.put(m.type); w.put(m.speed); w.put(m.start_time);
}
inline void deserialize(Reader& r, DubinsPath& m) {
    m.initial_x = r.get<double>(); m.initial_y = r.get<double>(); m.initial_yaw = r.get<double>(); m.length0 = r.get<double>();
    m.length1 = r.get<double>(); m.length2 = r.get<double>(); m.rho = r.get<double>(); m.type = r.get<int32_t>(); m.speed = r.get<double>();
    m.start_time = r.get<double>();
}
inline std::vector<uint8_t> serialize(const Plan& m) {
    Writer w;
    w.put((uint32_t)m.paths.size());
    for (const auto& p : m.paths) serialize(w, p);
    w.put(m.endtime);
    return w.bytes;
}
inline Plan deserializePlan(const uint8_t* p, size_t n) {
    Reader r(p, n);
    Plan m;
    uint32_t count = r.get<uint32_t>();
    if ((size_t)count * 76 > n) throw std::runtime_error("message truncated");
    m.paths.resize(count);
    for (auto& q : m.paths) deserialize(r, q);
    m.endtime = r.get<double>();
    return m;
}
inline std::vector<uint8_t> serialize(const StateMsg& m) { Writer w; w.put(m.x); w.put(m.y); w.put(m.heading); w.put(m.speed); w.put(m.time); return w.bytes; }
inline StateMsg deserializeState(const uint8_t* p, size_t n) {
    Reader r(p, n); StateMsg m;
    m.x = r.get<double>(); m.y = r.get<double>(); m.heading = r.get<double>(); m.speed = r.get<double>(); m.time = r.get<double>();
    return m;
}
inline std::vector<uint8_t> serialize(const Stats& m) {
    Writer w;
    w.put(m.samples); w.put(m.generated); w.put(m.expanded); w.put(m.iterations); w.put(m.plan_f_value); w.put(m.plan_collision_penalty);
    w.put(m.plan_time_penalty); w.put(m.plan_h_value); w.put(m.plan_depth); w.put(m.collision_penalty); w.put(m.cpu_time);
    w.put((uint8_t)(m.last_plan_achievable ? 1 : 0));
    return w.bytes;
}
inline Stats deserializeStats(const uint8_t* p, size_t n) {
    Reader r(p, n); Stats m;
    m.samples = r.get<int64_t>(); m.generated = r.get<int64_t>(); m.expanded = r.get<int64_t>(); m.iterations = r.get<int64_t>();
    m.plan_f_value = r.get<double>(); m.plan_collision_penalty = r.get<double>(); m.plan_time_penalty = r.get<double>(); m.plan_h_value = r.get<double>();
    m.plan_depth = r.get<int64_t>(); m.collision_penalty = r.get<double>(); m.cpu_time = r.get<int64_t>(); m.last_plan_achievable = r.get<uint8_t>() != 0;
    return m;
}
inline std::vector<uint8_t> serialize(const TaskLevelStats& m) { Writer w; w.put(m.time); w.put(m.collision_penalty); w.put(m.score); w.put(m.uncovered_length); return w.bytes; }

}  // namespace msg
}  // namespace ppamd
