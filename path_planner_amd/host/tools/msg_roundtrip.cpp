// msg_roundtrip — checks the ROS-free wire formats (Messages.h): plan -> Plan message -> ROS 1 bytes -> Plan message -> plan,
// the byte layout against hand-counted sizes, and truncated input.  No GPU.  Prints "ok" and exits 0, or the failure.
#include <cmath>
#include <cstdio>

#include "path_planner_amd/Messages.h"

using namespace ppamd;

#define CHECK(c) do { if (!(c)) { std::printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
    DubinsPlan plan;
    State a(0, 0, 0, 2.5, 10), b(40, 25, 1.1, 2.5, 0), c(10, 60, 4.0, 0.5, 0);
    DubinsWrapper w1(a, b, 8);
    State mid; mid.time() = w1.getEndTime(); w1.sample(mid);
    DubinsWrapper w2(mid, c, 16);
    w2.updateEndTime(w2.getStartTime() + 0.6 * (w2.getEndTime() - w2.getStartTime()));   // a truncated last segment, as the planner returns
    plan.append(w1); plan.append(w2);
    msg::Plan m = msg::convertToPlanMsg(plan);
    CHECK(m.paths.size() == 2 && m.endtime == plan.getEndTime());
    CHECK(m.paths[0].initial_x == 0 && m.paths[0].rho == 8 && m.paths[1].rho == 16 && m.paths[1].speed == w2.getSpeed() && m.paths[1].start_time == w2.getStartTime());
    std::vector<uint8_t> bytes = msg::serialize(m);
    CHECK(bytes.size() == 4 + 2 * (9 * 8 + 4) + 8);          // uint32 count, 2 x (9 float64 + int32), float64 endtime
    uint32_t count; std::memcpy(&count, bytes.data(), 4);
    CHECK(count == 2);
    msg::Plan m2 = msg::deserializePlan(bytes.data(), bytes.size());
    CHECK(m2.paths.size() == 2 && std::memcmp(&m2.paths[1], &m.paths[1], sizeof(msg::DubinsPath)) == 0 && m2.endtime == m.endtime);
    DubinsPlan back = msg::convertFromPlanMsg(m2);
    CHECK(back.get().size() == 2 && back.getEndTime() == plan.getEndTime() && back.getStartTime() == plan.getStartTime());
    for (double f : {0.0, 0.3, 0.77, 1.0}) {
        State s1, s2;
        s1.time() = s2.time() = plan.getStartTime() + f * (plan.getEndTime() - plan.getStartTime());
        plan.sample(s1); back.sample(s2);
        CHECK(s1.x() == s2.x() && s1.y() == s2.y() && s1.heading() == s2.heading() && s1.speed() == s2.speed());
    }
    bool threw = false;
    try { msg::deserializePlan(bytes.data(), bytes.size() - 3); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
    Planner::Stats st; st.Samples = 7; st.Generated = 9; st.Expanded = 3; st.Iterations = 2; st.PlanFValue = 1.5; st.PlanDepth = 4;
    msg::Stats sm = msg::convertToStatsMsg(st, 600.0, 12, true);
    std::vector<uint8_t> sb = msg::serialize(sm);
    CHECK(sb.size() == 11 * 8 + 1);
    msg::Stats sm2 = msg::deserializeStats(sb.data(), sb.size());
    CHECK(sm2.samples == 7 && sm2.plan_depth == 4 && sm2.collision_penalty == 600.0 && sm2.cpu_time == 12 && sm2.last_plan_achievable);
    msg::StateMsg s = msg::convertToStateMsg(State(1, 2, 3, 4, 5));
    std::vector<uint8_t> stb = msg::serialize(s);
    CHECK(stb.size() == 40);
    State rs = msg::convertToStateFromMsg(msg::deserializeState(stb.data(), stb.size()));
    CHECK(rs.x() == 1 && rs.y() == 2 && rs.heading() == 3 && rs.speed() == 4 && rs.time() == 5);
    CHECK(msg::serialize(msg::TaskLevelStats{1, 2, 3, 4}).size() == 32);
    std::printf("ok\n");
    return 0;
}
