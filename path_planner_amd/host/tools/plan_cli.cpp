// plan_cli — drives GpuAStarPlanner::plan() from a small text scenario and prints the Stats as JSON.
// Used by tests/test_gpu_host_planner.py to compare the C++ host path with the CPU oracle's AStarPlanner::plan
// under the same injected clock (PlannerConfig::setNowFunction).
//
// Scenario lines:  cfg <key> <value> | start x y heading speed time | ribbon x1 y1 x2 y2 | heuristic H K radius |
//                  ribbon_width w | obstacle x y heading speed time width length | gaussian x y heading speed time [c00 c01 c10 c11] | map_file path | clock t0 dt |
//                  time_remaining T | prev qi0 qi1 qi2 p0 p1 p2 rho type speed start end | repeat n |
//                  sharded_batch attempts seed   (instead of plan(): one iteration's batch, sample-sharded over `devices`: ShardedIteration)
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "path_planner_amd/Planner.h"

using namespace ppamd;

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: plan_cli scenario.txt\n"); return 2; }
    std::ifstream in(argv[1]);
    if (!in) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    PlannerConfig config(&std::cerr);
    State start;
    int H = 0, K = 0;
    double hr = 8;
    std::vector<std::array<double, 4>> ribs;
    auto obst = std::make_shared<BinaryDynamicObstaclesManager>();
    auto gauss = std::make_shared<GaussianDynamicObstaclesManager>();
    bool haveObst = false, haveGauss = false;
    Map::SharedPtr map = std::make_shared<Map>();
    double t0 = 1000, dt = 1e-3, timeRemaining = 0.05;
    DubinsPlan prev;
    int repeat = 1;
    bool realClock = false;
    int replans = 0;
    double replanStep = 0.1;
    uint32_t mmsi = 1;
    std::vector<int> devices;
    long long shardedAttempts = 0;
    unsigned long shardedSeed = 7;
    int failShard = -1;
    std::string cycleLogPath;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream s(line);
        std::string k;
        if (!(s >> k)) continue;
        if (k == "cfg") {
            std::string name; double v; s >> name >> v;
            if (name == "max_speed") config.setMaxSpeed(v);
            else if (name == "slow_speed") config.setSlowSpeed(v);
            else if (name == "turning_radius") config.setTurningRadius(v);
            else if (name == "coverage_turning_radius") config.setCoverageTurningRadius(v);
            else if (name == "time_horizon") config.setTimeHorizon(v);
            else if (name == "time_minimum") config.setTimeMinimum(v);
            else if (name == "collision_checking_increment") config.setCollisionCheckingIncrement(v);
            else if (name == "branching_factor") config.setBranchingFactor((int)v);
            else if (name == "initial_samples") config.setInitialSamples((int)v);
            else if (name == "use_brown_paths") config.setUseBrownPaths(v != 0);
            else if (name == "speculation") config.setSpeculation((int)v);
            else { std::fprintf(stderr, "unknown cfg %s\n", name.c_str()); return 2; }
        } else if (k == "start") {
            double x, y, h, v, t; s >> x >> y >> h >> v >> t; start = State(x, y, h, v, t);
        } else if (k == "ribbon") {
            std::array<double, 4> r; s >> r[0] >> r[1] >> r[2] >> r[3]; ribs.push_back(r);
        } else if (k == "heuristic") { s >> H >> K >> hr;
        } else if (k == "ribbon_width") { double w; s >> w; RibbonManager::setRibbonWidth(w);
        } else if (k == "obstacle") {
            double x, y, h, v, t, w, l; s >> x >> y >> h >> v >> t >> w >> l; obst->update(mmsi++, x, y, h, v, t, w, l); haveObst = true;
        } else if (k == "gaussian") {      // x y heading speed time [c00 c01 c10 c11]
            double x, y, h, v, t, c[4]; s >> x >> y >> h >> v >> t;
            if (s >> c[0] >> c[1] >> c[2] >> c[3]) gauss->update(mmsi++, x, y, h, v, t, c); else gauss->update(mmsi++, x, y, h, v, t);
            haveGauss = true;
        } else if (k == "map_file") { std::string p; s >> p; map = std::make_shared<GridWorldMap>(p);
        } else if (k == "visualization_file") {   // the search dump visualizer.py reads (executive.cpp:443-449)
            std::string p; s >> p;
            config.setVisualizations(true);
            config.setVisualizer(std::make_shared<Visualizer>(p));
        } else if (k == "clock") { s >> t0 >> dt;
        } else if (k == "time_remaining") { s >> timeRemaining;
        } else if (k == "devices") {          // device ids of the planner; an id that repeats gets its own second context on that device
            int d; while (s >> d) devices.push_back(d);
        } else if (k == "sharded_batch") { s >> shardedAttempts >> shardedSeed;
        } else if (k == "fail_shard") { s >> failShard;      // tests: this shard of a sharded_batch throws before its work
        } else if (k == "cycle_log") { s >> cycleLogPath;    // replan: one JSON line per cycle (Stats::Budget and what the cycle reached)
        } else if (k == "repeat") { s >> repeat;
        } else if (k == "replan") { s >> replans >> replanStep;   // N consecutive cycles, start moved replanStep seconds along the plan
        } else if (k == "real_clock") { int v; s >> v; realClock = v != 0;   // now() = t0 + wall seconds since plan() began
        } else if (k == "prev") {
            DubinsPath p; double speed, st, en; int type;
            s >> p.qi[0] >> p.qi[1] >> p.qi[2] >> p.param[0] >> p.param[1] >> p.param[2] >> p.rho >> type >> speed >> st >> en;
            p.type = (DubinsPathType)type;
            DubinsWrapper w; w.fill(p, speed, st);
            if (w.getEndTime() > en) w.updateEndTime(en);
            prev.append(w);
        }
    }
    RibbonManager rm((RibbonManager::Heuristic)H, hr, K);
    for (auto& r : ribs) rm.add(r[0], r[1], r[2], r[3]);
    config.setMap(map);
    if (haveObst) config.setObstaclesManager(obst);
    if (haveGauss) config.setObstaclesManager(gauss);
    std::vector<std::shared_ptr<GpuContext>> contexts;
    try {
        if (!devices.empty()) contexts = GpuContext::shared(devices);     // an id that repeats: a further context (stream) on that device
        if (contexts.empty()) contexts.push_back(GpuContext::shared(0));
    } catch (const std::exception& e) {
        std::printf("{\"exception\": \"%s\"}\n", e.what());
        return 1;
    }
    if (shardedAttempts > 0) {
        try {
            ShardedIteration it(contexts);
            it.failShardForTest = failShard;
            ShardedIteration::Result r;
            std::vector<double> wall;
            for (int rep = 0; rep < repeat; rep++) {
                const auto w0 = std::chrono::steady_clock::now();
                r = it.run(rm, start, config, shardedSeed, shardedAttempts);
                wall.push_back(1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count());
            }
            std::sort(wall.begin(), wall.end());
            std::printf("{\"sharded_batch\": %lld, \"shards\": %zu, \"rccl_ranks\": %d, \"agreed\": %s, \"best_f\": %.17g, \"best_f_bits\": %llu, \"best_edge\": %llu, "
                        "\"best_shard\": %d, \"edges\": %lld, \"wall_ms_median\": %.4f, \"kept\": [",
                        shardedAttempts, contexts.size(), r.rcclRanks, r.agreed ? "true" : "false", r.f, (unsigned long long)r.fBits, (unsigned long long)r.edge, r.shard,
                        (long long)r.edges, wall[wall.size() / 2]);
            for (size_t i = 0; i < r.kept.size(); i++) std::printf("%s%lld", i ? ", " : "", (long long)r.kept[i]);
            std::printf("]}\n");
            return 0;
        } catch (const std::exception& e) {
            std::printf("{\"exception\": \"%s\"}\n", e.what());
            return 1;
        }
    }
    try {
        Planner::Stats st;
        std::vector<double> wall;
        if (replans > 0) {
            // the 10 Hz loop of Executive::planLoop (executive.cpp:85-305) without ROS: plan, move the start replanStep seconds
            // along the returned plan, hand the plan back as previousPlan, repeat.  Real clock, fixed budget per cycle.
            double tNow = t0;
            State cur = start;
            unsigned long iters = 0, expanded = 0, failures = 0, samples = 0, deadlineStops = 0, edges = 0, trips = 0, gridUploads = 0;
            unsigned long failuresExplained = 0;       // failed plans whose start state was in collision (an obstacle's box or a blocked cell)
            unsigned long nodeRegrowths = 0, deviceGrowths = 0, late = 0;
            long firstGoalSum = 0, firstGoalCount = 0, firstGoalMax = -1;
            std::vector<long> firstGoals;
            std::vector<int> failedCycles;
            double insidePlanMs = 0;                   // plan() wall time over the cycles after the first
            Planner::Stats::BudgetTrace worst;         // the budget trace of the slowest cycle after the first
            int worstCycle = -1;
            double worstMs = 0;
            int failureCount = 0, horizonHalvings = 0;  // Executive::planLoop's back-off (executive.cpp:263-277)
            FILE* cycleLog = cycleLogPath.empty() ? nullptr : std::fopen(cycleLogPath.c_str(), "w");
            for (int cyc = 0; cyc < replans; cyc++) {
                const auto w0 = std::chrono::steady_clock::now();
                config.setNowFunction([&]() { return tNow + std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count(); });
                GpuAStarPlanner planner(contexts);
                st = planner.plan(rm, cur, config, prev, timeRemaining);
                wall.push_back(1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count());
                iters += st.Iterations; expanded += st.Expanded; samples += st.Samples; deadlineStops += st.DeadlineStops; edges += st.EdgesCosted;
                trips += st.Budget.RoundTrips; gridUploads += st.Budget.GridUploaded ? 1 : 0;
                nodeRegrowths += st.Budget.NodeRegrowths; deviceGrowths += st.Budget.DeviceGrowths;
                if (st.FirstGoalIteration >= 0) { firstGoalSum += st.FirstGoalIteration; firstGoalCount++; firstGoalMax = std::max(firstGoalMax, st.FirstGoalIteration); }
                firstGoals.push_back(st.FirstGoalIteration);
                const bool startHit = config.obstaclesManager().collisionExists(cur, true) > 0 || (config.map() && config.map()->isBlocked(cur.x(), cur.y()));
                if (cyc > 0) {
                    insidePlanMs += wall.back();
                    if (wall.back() >= 1e3 * timeRemaining) late++;
                    if (wall.back() > worstMs) { worstMs = wall.back(); worst = st.Budget; worstCycle = cyc; }
                }
                const Planner::Stats::BudgetTrace& b = st.Budget;
                if (cycleLog)
                    std::fprintf(cycleLog, "{\"cycle\": %d, \"wall_ms\": %.3f, \"iterations\": %lu, \"first_goal_iteration\": %ld, \"samples\": %lu, \"expanded\": %lu, "
                                 "\"edges\": %lu, \"round_trips\": %lu, \"deadline_stops\": %lu, \"plan_empty\": %s, \"start_in_collision\": %s, \"prologue_ms\": %.3f, "
                                 "\"loop_end_ms\": %.3f, \"total_ms\": %.3f, \"last_op\": %d, \"last_op_start_ms\": %.3f, \"last_op_predicted_ms\": %.3f, "
                                 "\"last_op_actual_ms\": %.3f, \"margin_ms\": %.3f, \"max_trip_ms\": %.3f, \"worst_under_prediction_ms\": %.3f, \"node_regrowths\": %lu, "
                                 "\"node_regrowth_ms\": %.3f, \"device_growths\": %lu, \"device_growth_ms\": %.3f, \"grid_uploaded\": %s, \"pick_ms\": %.3f, \"max_pick_ms\": %.3f, "
                                 "\"order_fallbacks\": %lu, \"max_wake_ms\": %.3f, \"stride_retries\": %lu}\n",
                                 cyc, wall.back(), (unsigned long)st.Iterations, st.FirstGoalIteration, (unsigned long)st.Samples, (unsigned long)st.Expanded,
                                 (unsigned long)st.EdgesCosted, b.RoundTrips, (unsigned long)st.DeadlineStops, st.Plan.empty() ? "true" : "false", startHit ? "true" : "false",
                                 b.PrologueMs, b.LoopEndMs, b.TotalMs, b.LastOpKind, b.LastOpStartMs, b.LastOpPredictedMs, b.LastOpActualMs, b.MarginMs, b.MaxTripMs,
                                 b.WorstUnderPredictionMs, b.NodeRegrowths, b.NodeRegrowthMs, b.DeviceGrowths, b.DeviceGrowthMs, b.GridUploaded ? "true" : "false", b.PickMs, b.MaxPickMs,
                                 (unsigned long)st.OrderFallbacks, b.MaxWakeMs, b.StrideRetries);
                if (cyc > 0 && wall.back() >= 1e3 * timeRemaining)      // late: what was the cycle doing when the budget ran out?
                    std::fprintf(stderr, "[replan] cycle %d LATE: %.3f ms of %.1f | loop left at %.3f, last op kind %d started %.3f predicted %.3f took %.3f (margin %.3f) | "
                                 "max trip %.3f, worst under-prediction %.3f | node regrowths %lu (%.3f ms), device growths %lu (%.3f ms), prologue %.3f | picks %.3f ms (longest %.3f), "
                                 "%lu order fallbacks, longest thread wake %.3f | %lu iterations, %lu samples, %lu expanded\n",
                                 cyc, wall.back(), 1e3 * timeRemaining, b.LoopEndMs, b.LastOpKind, b.LastOpStartMs, b.LastOpPredictedMs, b.LastOpActualMs, b.MarginMs, b.MaxTripMs,
                                 b.WorstUnderPredictionMs, b.NodeRegrowths, b.NodeRegrowthMs, b.DeviceGrowths, b.DeviceGrowthMs, b.PrologueMs, b.PickMs, b.MaxPickMs,
                                 (unsigned long)st.OrderFallbacks, b.MaxWakeMs, (unsigned long)st.Iterations, (unsigned long)st.Samples, (unsigned long)st.Expanded);
                tNow += replanStep;
                if (st.Plan.empty()) {
                    failures++; failedCycles.push_back(cyc);
                    if (startHit) failuresExplained++;
                    // executive.cpp:263-277: the third empty plan in a row halves the time horizon (never below timeMinimum, never restored)
                    failureCount++;
                    if (failureCount > 2) {
                        config.setTimeHorizon(config.timeHorizon() / 2);
                        if (config.timeHorizon() < config.timeMinimum()) config.setTimeHorizon(config.timeMinimum());
                        else { failureCount = 0; horizonHalvings++; }
                    }
                    cur.time() = tNow; prev = DubinsPlan(); continue;
                }
                failureCount = 0;                       // executive.cpp:220
                prev = st.Plan;
                State nxt; nxt.time() = tNow;
                if (prev.containsTime(tNow)) { prev.sample(nxt); cur = nxt; } else { cur.time() = tNow; }
                cur.speed() = config.maxSpeed();
                rm.cover(cur.x(), cur.y(), false);      // Executive::updateCovered: the vehicle covers as it moves
            }
            if (cycleLog) std::fclose(cycleLog);
            // the first cycle of a process allocates the device buffers (they persist in the contexts): reported on its own,
            // the percentiles are over the cycles after it
            const double firstCycle = wall.front();
            if (wall.size() > 1) wall.erase(wall.begin());
            std::sort(wall.begin(), wall.end());
            const double p50 = wall[wall.size() / 2], p99 = wall[std::min(wall.size() - 1, (size_t)(0.99 * wall.size()))];
            std::vector<long> fg = firstGoals;
            std::sort(fg.begin(), fg.end());
            std::printf("{\"replans\": %d, \"budget_ms\": %.3f, \"first_cycle_ms\": %.3f, \"wall_ms_p50\": %.3f, \"wall_ms_p99\": %.3f, \"wall_ms_max\": %.3f, "
                        "\"late_cycles\": %lu, \"mean_iterations\": %.2f, \"mean_expanded\": %.1f, \"mean_samples\": %.1f, \"mean_edges\": %.1f, \"mean_round_trips\": %.1f, "
                        "\"first_goal_iteration_median\": %ld, \"first_goal_iteration_max\": %ld, \"first_goal_iteration_mean\": %.3f, \"cycles_with_a_goal\": %ld, "
                        "\"failed_plans\": %lu, \"failed_plans_with_start_in_collision\": %lu, \"horizon_halvings\": %d, \"final_time_horizon\": %.3f, \"deadline_stops\": %lu, \"grid_uploads\": %lu, "
                        "\"node_regrowths\": %lu, \"device_growths\": %lu, \"expansions_per_s_inside_plan\": %.1f, \"edges_per_s_inside_plan\": %.1f, "
                        "\"worst_cycle\": {\"cycle\": %d, \"wall_ms\": %.3f, \"loop_end_ms\": %.3f, \"last_op\": %d, \"last_op_start_ms\": %.3f, \"last_op_predicted_ms\": %.3f, "
                        "\"last_op_actual_ms\": %.3f, \"margin_ms\": %.3f, \"max_trip_ms\": %.3f, \"node_regrowths\": %lu, \"node_regrowth_ms\": %.3f, \"device_growths\": %lu, "
                        "\"device_growth_ms\": %.3f, \"prologue_ms\": %.3f}, \"failed_cycles\": [",
                        replans, 1e3 * timeRemaining, firstCycle, p50, p99, wall.back(), late, (double)iters / replans, (double)expanded / replans,
                        (double)samples / replans, (double)edges / replans, (double)trips / replans, fg[fg.size() / 2], firstGoalMax,
                        firstGoalCount ? (double)firstGoalSum / firstGoalCount : -1.0, firstGoalCount, failures, failuresExplained, horizonHalvings, config.timeHorizon(), deadlineStops, gridUploads,
                        nodeRegrowths, deviceGrowths, insidePlanMs > 0 ? 1e3 * (double)expanded / insidePlanMs : 0.0, insidePlanMs > 0 ? 1e3 * (double)edges / insidePlanMs : 0.0,
                        worstCycle, worstMs, worst.LoopEndMs, worst.LastOpKind, worst.LastOpStartMs, worst.LastOpPredictedMs, worst.LastOpActualMs, worst.MarginMs,
                        worst.MaxTripMs, worst.NodeRegrowths, worst.NodeRegrowthMs, worst.DeviceGrowths, worst.DeviceGrowthMs, worst.PrologueMs);
            for (size_t i = 0; i < failedCycles.size(); i++) std::printf("%s%d", i ? ", " : "", failedCycles[i]);
            std::printf("], \"devices\": %zu}\n", contexts.size());
            return 0;
        }
        for (int rep = 0; rep < repeat; rep++) {
            long calls = 0;
            const auto w0 = std::chrono::steady_clock::now();
            if (realClock)
                config.setNowFunction([&]() { return t0 + std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count(); });
            else
                config.setNowFunction([&]() { return t0 + (double)(calls++) * dt; });
            config.setDeadlineGuard(realClock);   // a counting clock does not advance while the device works
            GpuAStarPlanner planner(contexts);   // a fresh planner every cycle, like Executive::planLoop (executive.cpp:85-90)
            st = planner.plan(rm, start, config, prev, timeRemaining);
            wall.push_back(1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count());
        }
        std::sort(wall.begin(), wall.end());
        std::fprintf(stderr, "plan() wall ms: min %.3f median %.3f max %.3f over %d calls\n", wall.front(), wall[wall.size() / 2], wall.back(), repeat);
        std::printf("{\"samples\": %lu, \"generated\": %lu, \"expanded\": %lu, \"iterations\": %lu, \"plan_f\": %.17g, "
                    "\"plan_collision_penalty\": %.17g, \"plan_time_penalty\": %.17g, \"plan_h\": %.17g, \"plan_depth\": %lu, "
                    "\"first_goal_iteration\": %ld, \"edges_costed\": %lu, \"host_heuristics\": %lu, \"order_fallbacks\": %lu, \"wall_ms_median\": %.4f, "
                    "\"wall_ms_max\": %.4f, \"plan\": [",
                    st.Samples, st.Generated, st.Expanded, st.Iterations, st.PlanFValue, st.PlanCollisionPenalty, st.PlanTimePenalty,
                    st.PlanHValue, st.PlanDepth, st.FirstGoalIteration, st.EdgesCosted, st.HostHeuristics, st.OrderFallbacks, wall[wall.size() / 2], wall.back());
        bool first = true;
        for (const auto& w : st.Plan.get()) {
            const DubinsPath& p = w.unwrap();
            std::printf("%s[%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %d, %.17g, %.17g, %.17g]", first ? "" : ", ", p.qi[0], p.qi[1],
                        p.qi[2], p.param[0], p.param[1], p.param[2], p.rho, (int)p.type, w.getSpeed(), w.getStartTime(), w.getEndTime());
            first = false;
        }
        std::printf("]}\n");
    } catch (const std::exception& e) {
        std::printf("{\"exception\": \"%s\"}\n", e.what());
        return 1;
    }
    return 0;
}
