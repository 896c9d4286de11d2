// mission_trace — a SCRIPTED mission through ppamd::Executive, deterministic from end to end, printing what the planning loop
// decided in every cycle.  tests/test_gpu_mission.py compares the trace line by line with oracle/mission_oracle, the oracle's
// restatement of Executive::planLoop (executive.cpp:43-305) around the oracle's planner, run on the same script.
//
// Everything the ROS node, the controller and the wall clock do in the reference is done here by one TrajectoryPublisher whose
// callbacks run on the planning thread, in the order the loop makes them:
//   getTime()          a counting clock, t0 + calls * dt (every call advances it; the planner polls it through PlannerConfig::now)
//   displayRibbons()   top of cycle c
//   (cycle observer)   right before plan(): arms the scripted clock fault of that cycle (the clock then throws inside plan())
//   publishStats()     after plan(): the odometry callback (updateCovered with the scripted pose of this cycle, if any), the
//                      contact callbacks (every obstacle reported again), scripted reconfiguration, cancellation after the last cycle
//   publishPlan()      the controller: the vehicle is on the plan now (updateCovered with the plan's pose at the present time) and
//                      will be at the plan's pose one planning period from now — unless the script moves it this cycle
//
// usage: mission_trace scenario.txt     (vocabulary of mission_sim plus:  clock t0 dt | max_cycles n |
//          at <cycle> teleport x y heading speed | at <cycle> clock_fault | at <cycle> horizon h)
#include <array>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "path_planner_amd/Executive.h"

using namespace ppamd;

namespace {
struct Script {
    std::map<unsigned long, std::array<double, 4>> teleport;
    std::map<unsigned long, bool> clockFault;
    std::map<unsigned long, double> horizon;
    unsigned long maxCycles = 30;
};

class ScriptedWorld : public TrajectoryPublisher {
public:
    ScriptedWorld(double t0, double dt, double lookahead, const Script& script, const std::vector<std::array<double, 7>>& obst, double cfg[11], int flags[3])
        : m_T0(t0), m_Dt(dt), m_Lookahead(lookahead), m_Script(script), m_Obst(obst) {
        for (int i = 0; i < 11; i++) m_Cfg[i] = cfg[i];
        for (int i = 0; i < 3; i++) m_Flags[i] = flags[i];
    }
    Executive* exec = nullptr;
    bool done = false;

    double peek() const { return m_T0 + (double)m_Calls * m_Dt; }
    double getTime() const override {
        if (m_FaultAt >= 0 && m_Calls == m_FaultAt) {
            m_FaultAt = -1;
            m_Calls++;
            throw std::runtime_error("scripted clock fault");
        }
        return m_T0 + (double)(m_Calls++) * m_Dt;
    }
    void displayRibbons(const RibbonManager&) override { m_Cycle++; }   // cycle numbers count the tops of the loop that got this far
    // right before plan() (the cycle observer): the scripted clock fault of this cycle hits the planner's first poll of the clock
    void beforePlan() { if (m_Script.clockFault.count((unsigned long)m_Cycle)) m_FaultAt = m_Calls; }
    void publishStats(const Planner::Stats& st, double collisionPenalty, unsigned long, bool lastPlanAchievable) override {
        std::printf("{\"k\": \"stats\", \"cycle\": %ld, \"samples\": %lu, \"iterations\": %lu, \"expanded\": %lu, \"generated\": %lu, \"plan_legs\": %zu, "
                    "\"plan_f\": %.17g, \"plan_depth\": %lu, \"collision_penalty\": %.17g, \"last_plan_achievable\": %d}\n",
                    m_Cycle, st.Samples, st.Iterations, st.Expanded, st.Generated, st.Plan.get().size(), st.PlanFValue, st.PlanDepth, collisionPenalty,
                    lastPlanAchievable ? 1 : 0);
        const unsigned long c = (unsigned long)m_Cycle;
        auto tp = m_Script.teleport.find(c);
        if (tp != m_Script.teleport.end()) {                         // the odometry callback: the vehicle is somewhere else
            const auto& p = tp->second;
            exec->updateCovered(p[0], p[1], p[3], p[2], peek());
            m_Moved = true; m_MovedTo = p;
        }
        uint32_t mmsi = 1;                                           // the contact callbacks
        for (const auto& o : m_Obst) exec->updateDynamicObstacle(mmsi++, State(o[0], o[1], o[2], o[3], o[4]), o[5], o[6]);
        auto hz = m_Script.horizon.find(c);
        if (hz != m_Script.horizon.end())                            // dynamic reconfiguration: a new time horizon, everything else as it was
            exec->setConfiguration(m_Cfg[0], m_Cfg[1], m_Cfg[2], m_Cfg[3], m_Cfg[4], (int)m_Cfg[5], (int)m_Cfg[6], hz->second, m_Cfg[8], m_Cfg[9], (int)m_Cfg[10],
                                   m_Flags[0] != 0, m_Flags[1] != 0, m_Flags[2] != 0, false);
        if (c + 1 >= m_Script.maxCycles) exec->cancelPlanner();
    }
    State publishPlan(const DubinsPlan& plan) override {
        const double tNow = peek();
        State now;
        now.time() = tNow;
        if (!m_Moved && plan.containsTime(tNow)) {
            plan.sample(now);
            exec->updateCovered(now.x(), now.y(), now.speed(), now.heading(), tNow);
        }
        State next;
        next.time() = getTime() + m_Lookahead;
        if (m_Moved) {                                               // not where the plan says: the controller answers with where the vehicle is
            next = State(m_MovedTo[0], m_MovedTo[1], m_MovedTo[2], m_MovedTo[3], next.time());
            m_Moved = false;
        } else if (plan.containsTime(next.time())) {
            plan.sample(next);
        } else {
            next = State();
        }
        std::printf("{\"k\": \"publish\", \"cycle\": %ld, \"next\": [%.17g, %.17g, %.17g, %.17g, %.17g]}\n", m_Cycle, next.x(), next.y(), next.heading(),
                    next.speed(), next.time());
        return next;
    }
    void publishTaskLevelStats(double wall, double cumCollision, double cumG, double uncovered) override {
        std::printf("{\"k\": \"task\", \"wall\": %.17g, \"collision\": %.17g, \"g\": %.17g, \"uncovered\": %.17g}\n", wall, cumCollision, cumG, uncovered);
    }
    void allDone() override {
        done = true;
        std::printf("{\"k\": \"all_done\", \"cycle\": %ld}\n", m_Cycle);
    }

private:
    double m_T0, m_Dt, m_Lookahead;
    Script m_Script;
    std::vector<std::array<double, 7>> m_Obst;
    double m_Cfg[11];
    int m_Flags[3];
    mutable long m_Calls = 0, m_FaultAt = -1;
    long m_Cycle = -1;
    bool m_Moved = false;
    std::array<double, 4> m_MovedTo{};
};
}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s scenario.txt\n", argv[0]); return 2; }
    std::ifstream in(argv[1]);
    if (!in) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    State start(0, 0, 0, 0, 1);
    std::vector<std::array<double, 4>> ribs;
    std::vector<std::array<double, 7>> obst;
    std::string mapFile;
    double cfg[11] = {8, 16, 2.5, 0.5, 1.5, 9, 1, 30, 5, 0.05, 100};
    int flags[3] = {0, 0, 0};
    double planningTime = 0.1, t0 = 1000, dt = 1e-3;
    int speculation = 16;
    Script script;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream s(line);
        std::string k;
        if (!(s >> k)) continue;
        if (k == "start") { double x, y, h, v, t; s >> x >> y >> h >> v >> t; start = State(x, y, h, v, t); }
        else if (k == "ribbon") { std::array<double, 4> r; s >> r[0] >> r[1] >> r[2] >> r[3]; ribs.push_back(r); }
        else if (k == "obstacle") { std::array<double, 7> o; for (auto& v : o) s >> v; obst.push_back(o); }
        else if (k == "map_file") s >> mapFile;
        else if (k == "config") { for (auto& v : cfg) s >> v; for (auto& f : flags) s >> f; }
        else if (k == "planning_time") s >> planningTime;
        else if (k == "speculation") s >> speculation;
        else if (k == "clock") s >> t0 >> dt;
        else if (k == "max_cycles") s >> script.maxCycles;
        else if (k == "at") {
            unsigned long c; std::string what;
            s >> c >> what;
            if (what == "teleport") { std::array<double, 4> p; s >> p[0] >> p[1] >> p[2] >> p[3]; script.teleport[c] = p; }
            else if (what == "clock_fault") script.clockFault[c] = true;
            else if (what == "horizon") { double h; s >> h; script.horizon[c] = h; }
        }
    }
    ScriptedWorld world(t0, dt, planningTime, script, obst, cfg, flags);
    Executive exec(&world);
    world.exec = &exec;
    exec.setPlanningTimeSeconds(planningTime);
    exec.setSpeculation(speculation);
    exec.setDeadlineGuard(false);            // the scripted clock does not advance while the device works
    exec.setConfiguration(cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], (int)cfg[5], (int)cfg[6], cfg[7], cfg[8], cfg[9], (int)cfg[10], flags[0] != 0, flags[1] != 0,
                          flags[2] != 0, false);
    if (!mapFile.empty()) exec.refreshMap(mapFile, 0, 0);
    for (auto& r : ribs) exec.addRibbon(r[0], r[1], r[2], r[3]);
    exec.setCycleObserver([&world](const Executive::CycleRecord& r) {
        world.beforePlan();
        std::printf("{\"k\": \"cycle\", \"cycle\": %lu, \"from\": [%.17g, %.17g, %.17g, %.17g, %.17g], \"previous_plan_legs\": %zu, \"time_horizon\": %.17g, "
                    "\"time_remaining\": %.17g, \"ribbons\": %zu, \"uncovered\": %.17g, \"empty_in_a_row\": %d, \"last_plan_achievable\": %d}\n",
                    r.cycle, r.from.x(), r.from.y(), r.from.heading(), r.from.speed(), r.from.time(), r.previousPlanLegs, r.timeHorizon, r.timeRemaining,
                    r.ribbons, r.uncoveredLength, r.emptyInARow, r.lastPlanAchievable ? 1 : 0);
    });
    exec.updateCovered(start.x(), start.y(), start.speed(), start.heading(), start.time());
    exec.startPlanner();
    const bool ended = exec.waitUntilInactive(600.0);
    std::printf("{\"k\": \"end\", \"ended\": %s, \"finished\": %s, \"cycles\": %lu, \"empty_plans\": %lu}\n", ended ? "true" : "false", world.done ? "true" : "false",
                exec.cycles(), exec.emptyPlans());
    return ended ? 0 : 1;
}
