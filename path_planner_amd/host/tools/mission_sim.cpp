// mission_sim — a whole coverage mission through ppamd::Executive (the reference's 10 Hz planning harness, ROS-free) with a
// simulated vehicle that follows the published plan exactly: it stands in for the ROS node + controller of the reference
// (path_planner_node.cpp, the controller's update_reference_trajectory service).  Prints one JSON line of task-level results.
//
// usage: mission_sim scenario.txt      lines (same vocabulary as plan_cli where it overlaps):
//   start x y heading speed time | ribbon x1 y1 x2 y2 | obstacle x y heading speed time width length | map_file path |
//   config turningRadius coverageTurningRadius maxSpeed slowSpeed lineWidth k heuristic timeHorizon timeMinimum increment initialSamples
//          useBrownPaths useGaussian ignoreObstacles | planning_time s | max_seconds s | speculation n
#include <atomic>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>

#include "path_planner_amd/Executive.h"

using namespace ppamd;

class SimulatedVehicle : public TrajectoryPublisher {
public:
    explicit SimulatedVehicle(const State& start) : m_T0(std::chrono::steady_clock::now()), m_Start(start), m_Current(start) {}
    double getTime() const override { return m_Start.time() + std::chrono::duration<double>(std::chrono::steady_clock::now() - m_T0).count(); }
    // the controller's answer: where the vehicle will be one planning period from now if it follows this plan
    State publishPlan(const DubinsPlan& plan) override {
        std::lock_guard<std::mutex> lock(m_Mutex);
        m_Plan = plan;
        State s;
        s.time() = getTime() + m_Lookahead;
        if (plan.containsTime(s.time())) plan.sample(s);
        else s = State();
        return s;
    }
    void publishStats(const Planner::Stats& stats, double collisionPenalty, unsigned long, bool) override {
        plans++; iterations += stats.Iterations; expanded += stats.Expanded; samples += stats.Samples; edges += stats.EdgesCosted;
        collision += collisionPenalty;
    }
    void publishTaskLevelStats(double wall, double cumCollision, double cumG, double uncovered) override {
        taskWall = wall; taskCollision = cumCollision; taskG = cumG; taskUncovered = uncovered;
    }
    void allDone() override { done = true; }
    // pose of the vehicle now: on the latest plan if it covers the present, otherwise dead reckoning from the last pose
    State pose() {
        std::lock_guard<std::mutex> lock(m_Mutex);
        State s;
        s.time() = getTime();
        if (!m_Plan.empty() && m_Plan.containsTime(s.time())) { m_Plan.sample(s); m_Current = s; }
        else { m_Current = m_Current.push(s.time() - m_Current.time()); }
        return m_Current;
    }
    void setLookahead(double s) { m_Lookahead = s; }
    std::atomic<bool> done{false};
    unsigned long plans = 0, iterations = 0, expanded = 0, samples = 0, edges = 0;
    double collision = 0, taskWall = 0, taskCollision = 0, taskG = 0, taskUncovered = -1;

private:
    std::chrono::steady_clock::time_point m_T0;
    State m_Start, m_Current;
    DubinsPlan m_Plan;
    std::mutex m_Mutex;
    double m_Lookahead = 0.1;
};

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
    double planningTime = 0.1, maxSeconds = 60;
    int speculation = 16;
    std::vector<int> devices;
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
        else if (k == "max_seconds") s >> maxSeconds;
        else if (k == "speculation") s >> speculation;
        else if (k == "devices") { int d; while (s >> d) devices.push_back(d); }
    }
    SimulatedVehicle vehicle(start);
    vehicle.setLookahead(planningTime);
    Executive exec(&vehicle);
    exec.setPlanningTimeSeconds(planningTime);
    exec.setSpeculation(speculation);
    if (!devices.empty()) exec.setDevices(devices);
    exec.setConfiguration(cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], (int)cfg[5], (int)cfg[6], cfg[7], cfg[8], cfg[9], (int)cfg[10], flags[0] != 0, flags[1] != 0,
                          flags[2] != 0, false);
    if (!mapFile.empty()) exec.refreshMap(mapFile, 0, 0);
    for (auto& r : ribs) exec.addRibbon(r[0], r[1], r[2], r[3]);
    // the node's contact callback: every contact is reported again and again while it is tracked (a planning run starts by
    // forgetting all contacts, executive.cpp:46-50)
    auto reportContacts = [&] {
        uint32_t mmsi = 1;
        for (auto& o : obst) exec.updateDynamicObstacle(mmsi++, State(o[0], o[1], o[2], o[3], o[4]), o[5], o[6]);
    };
    reportContacts();
    exec.updateCovered(start.x(), start.y(), start.speed(), start.heading(), start.time());
    exec.startPlanner();
    // the node's odometry callback: 20 Hz position updates feeding Executive::updateCovered (path_planner_node.cpp)
    const auto w0 = std::chrono::steady_clock::now();
    bool timedOut = false;
    while (!vehicle.done) {
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        State p = vehicle.pose();
        exec.updateCovered(p.x(), p.y(), p.speed(), p.heading(), p.time());
        reportContacts();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > maxSeconds) { timedOut = true; break; }
    }
    exec.cancelPlanner();
    exec.waitUntilInactive(5.0);
    std::printf("{\"finished\": %s, \"timed_out\": %s, \"cycles\": %lu, \"empty_plans\": %lu, \"plans_published\": %lu, \"mean_iterations\": %.2f, "
                "\"mean_expanded\": %.1f, \"mean_samples\": %.1f, \"edges_costed\": %lu, \"task_wall_s\": %.3f, \"task_collision_penalty\": %.6g, "
                "\"task_cumulative_g\": %.6g, \"uncovered_length\": %.6g}\n",
                vehicle.done ? "true" : "false", timedOut ? "true" : "false", exec.cycles(), exec.emptyPlans(), vehicle.plans,
                vehicle.plans ? (double)vehicle.iterations / vehicle.plans : 0.0, vehicle.plans ? (double)vehicle.expanded / vehicle.plans : 0.0,
                vehicle.plans ? (double)vehicle.samples / vehicle.plans : 0.0, vehicle.edges, vehicle.taskWall, vehicle.taskCollision, vehicle.taskG,
                vehicle.taskUncovered);
    return 0;
}
