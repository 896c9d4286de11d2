// ppamd::Executive — the 10 Hz planning harness around GpuAStarPlanner, ROS-free (interface: Executive.h).
//
// Behaviour follows the reference's Executive (pp/src/executive/executive.cpp; the cited lines say which rule each step
// implements); the code is organised as a small state machine of one planning cycle:
//
//     admit()  ->  [ cycle: gather() -> solve() -> pace() -> handOver() | backOff() ] *  ->  summarise()
//
// Not on this path: PotentialFieldPlanner, GeoTiffMap, the visualisation stream, radius shrinking
// (compiled out in the reference: c_RadiusShrinkEnabled = false, executive.h:171).
#include "path_planner_amd/Executive.h"

#include <chrono>
#include <fstream>
#include <iostream>
#include <thread>

namespace ppamd {

namespace {
const double kPenaltyPerCollision = 600.0;   // Edge::collisionPenaltyFactor(), Edge.h:151
const double kPenaltyPerSecond = 1.0;        // Edge::timePenaltyFactor(), Edge.h:152
RibbonManager freshRibbons(double turningRadius) {   // what clearRibbons() and the constructor start from (executive.cpp:404-407)
    return RibbonManager(RibbonManager::TspPointRobotNoSplitKRibbons, turningRadius, 2);
}
}  // namespace

Executive::Executive(TrajectoryPublisher* trajectoryPublisher) : m_TrajectoryPublisher(trajectoryPublisher), m_PlannerConfig(&std::cerr) {
    m_PlannerConfig.setNowFunction([this] { return m_TrajectoryPublisher->getTime(); });   // executive.cpp:18
    m_BinaryDynamicObstaclesManager = std::make_shared<BinaryDynamicObstaclesManager>();
    m_GaussianDynamicObstaclesManager = std::make_shared<GaussianDynamicObstaclesManager>();
    m_RibbonManager = freshRibbons(m_PlannerConfig.turningRadius());
}

Executive::~Executive() {
    terminate();
    waitUntilInactive(2.0);   // executive.cpp:21-24
}

// ------------------------------------------------------------------------------------------------ callbacks of the node
void Executive::updateCovered(double x, double y, double speed, double heading, double t) {
    // cover only while the heading is steady (executive.cpp:36)
    const bool steady = (m_LastHeading - heading) / m_LastUpdateTime <= c_CoverageHeadingRateMax;
    if (steady) {
        std::lock_guard<std::mutex> guard(m_RibbonManagerMutex);
        m_RibbonManager.cover(x, y, false);
    }
    m_LastHeading = heading;
    m_LastUpdateTime = t;
    m_LastState = State(x, y, heading, speed, t);
}

void Executive::addRibbon(double x1, double y1, double x2, double y2) {
    std::lock_guard<std::mutex> guard(m_RibbonManagerMutex);
    m_RibbonManager.add(x1, y1, x2, y2);
}

void Executive::clearRibbons() {
    std::lock_guard<std::mutex> guard(m_RibbonManagerMutex);
    m_RibbonManager = freshRibbons(m_PlannerConfig.turningRadius());
}

void Executive::updateDynamicObstacle(uint32_t mmsi, State o, double width, double length) {   // both models are kept current (:313-319)
    std::lock_guard<std::mutex> guard(m_ObstaclesMutex);
    m_BinaryDynamicObstaclesManager->update(mmsi, o.x(), o.y(), o.heading(), o.speed(), o.time(), width, length);
    m_GaussianDynamicObstaclesManager->update(mmsi, o.x(), o.y(), o.heading(), o.speed(), o.time());
}

void Executive::refreshMap(const std::string& path, double, double) {
    // The reference loads in a detached thread (:321-382); here the load is synchronous and only GridWorldMap text files (or ""
    // = no map) are accepted.  The planning loop picks the new map up at the top of its next cycle, as there.
    std::lock_guard<std::mutex> guard(m_MapMutex);
    std::ostream& log = *m_PlannerConfig.output();
    auto useEmpty = [&](const char* why) {
        m_NewMap = std::make_shared<Map>();
        m_CurrentMapPath.clear();
        m_TrajectoryPublisher->displayMap("");
        log << why << std::endl;
    };
    if (path.empty()) return useEmpty("Map cleared. Using empty map now.");
    if (!std::ifstream(path).good()) return useEmpty(("Cannot find map file: " + path + " - using empty map for now.").c_str());
    try {
        if (path.find(".map") == std::string::npos) throw std::runtime_error("not a GridWorldMap file");
        m_NewMap = std::make_shared<GridWorldMap>(path);
        m_CurrentMapPath = path;
        m_TrajectoryPublisher->displayMap(path);
        log << "Loaded map file: " << path << std::endl;
    } catch (const std::exception& e) {
        m_NewMap = nullptr;   // keep the map in use
        m_CurrentMapPath.clear();
        log << "Map at " << path << " was not loaded (" << e.what() << "); the map in use stays." << std::endl;
    }
}

void Executive::setConfiguration(double turningRadius, double coverageTurningRadius, double maxSpeed, double slowSpeed, double lineWidth, int k,
                                 int heuristic, double timeHorizon, double timeMinimum, double collisionCheckingIncrement, int initialSamples,
                                 bool useBrownPaths, bool useGaussianDynamicObstacles, bool ignoreDynamicObstacles, bool usePotentialField) {
    PlannerConfig& c = m_PlannerConfig;
    c.setTurningRadius(turningRadius); c.setCoverageTurningRadius(coverageTurningRadius);
    c.setMaxSpeed(maxSpeed); c.setSlowSpeed(slowSpeed);
    c.setBranchingFactor(k);
    c.setTimeHorizon(timeHorizon); c.setTimeMinimum(timeMinimum);
    c.setCollisionCheckingIncrement(collisionCheckingIncrement);
    c.setInitialSamples(initialSamples);
    c.setUseBrownPaths(useBrownPaths);
    RibbonManager::setRibbonWidth(lineWidth);
    // numbering of path_planner.cfg (executive.cpp:419-427)
    static const RibbonManager::Heuristic byCfgIndex[5] = {RibbonManager::TspPointRobotNoSplitAllRibbons, RibbonManager::TspPointRobotNoSplitKRibbons,
                                                           RibbonManager::MaxDistance, RibbonManager::TspDubinsNoSplitAllRibbons,
                                                           RibbonManager::TspDubinsNoSplitKRibbons};
    if (heuristic >= 0 && heuristic < 5) m_RibbonManager.setHeuristic(byCfgIndex[heuristic]);
    else *c.output() << "Unknown heuristic. Ignoring." << std::endl;
    m_UseGaussianDynamicObstacles = useGaussianDynamicObstacles;
    m_IgnoreDynamicObstacles = ignoreDynamicObstacles;
    if (usePotentialField) *c.output() << "PotentialFieldPlanner is outside this build: the A* planner is used." << std::endl;
}

void Executive::startPlanner() {
    if (!m_PlannerConfig.map()) m_PlannerConfig.setMap(std::make_shared<Map>());
    std::unique_lock<std::mutex> guard(m_PlannerStateMutex);
    // a loop that is being cancelled finishes by itself; only an idle executive gets a new thread (:442-451)
    if (m_PlannerState != PlannerState::Running) m_PlanningFuture = std::async(std::launch::async, &Executive::planLoop, this);
}

void Executive::cancelPlanner() {
    std::unique_lock<std::mutex> guard(m_PlannerStateMutex);
    if (m_PlannerState == PlannerState::Running) m_PlannerState = PlannerState::Cancelled;
}

void Executive::terminate() { cancelPlanner(); }

bool Executive::waitUntilInactive(double seconds) {
    return !m_PlanningFuture.valid() || m_PlanningFuture.wait_for(std::chrono::duration<double>(seconds)) == std::future_status::ready;
}

// ------------------------------------------------------------------------------------------------ the planning thread
namespace {
struct Cycle {                  // what survives from one planning cycle to the next (executive.cpp:69-78)
    State from;                 // state to plan from; time -1 = "ask dead reckoning"
    Planner::Stats last;        // holds the plan that may be reused
    bool lastPlanAchievable = false;
    int emptyInARow = 0;
};
}  // namespace

void Executive::planLoop() {
    TrajectoryPublisher& out = *m_TrajectoryPublisher;
    const double missionStart = out.getTime();
    double collisionsSeen = 0;   // sum over cycles of collisionExists at the vehicle's reported state (:158-167)

    // every run starts without contacts: the node reports the live ones again within a second (executive.cpp:46-50)
    {
        std::lock_guard<std::mutex> guard(m_ObstaclesMutex);
        m_BinaryDynamicObstaclesManager = std::make_shared<BinaryDynamicObstaclesManager>();
        m_GaussianDynamicObstaclesManager = std::make_shared<GaussianDynamicObstaclesManager>();
    }
    if (m_Contexts.empty()) m_Contexts = GpuContext::shared(std::vector<int>{0});

    auto cancelled = [this] {
        std::unique_lock<std::mutex> guard(m_PlannerStateMutex);
        return m_PlannerState == PlannerState::Cancelled;
    };

    try {
        // admit(): a previous loop may still be winding down with the cancel flag up; give it two seconds (:56-68)
        {
            std::unique_lock<std::mutex> guard(m_PlannerStateMutex);
            m_CancelCV.wait_for(guard, std::chrono::seconds(2), [this] { return m_PlannerState != PlannerState::Cancelled; });
            if (m_PlannerState == PlannerState::Cancelled) {
                std::cerr << "Planner initialization timed out: the cancel flag of an earlier run is still set." << std::endl;
                return;
            }
            m_PlannerState = PlannerState::Running;
        }

        Cycle cyc;
        for (;;) {
            const double cycleStart = out.getTime();
            if (cancelled()) break;

            // ---- gather(): ribbons, start state, map, previous plan, obstacle model
            RibbonManager ribbons;
            {
                std::lock_guard<std::mutex> guard(m_RibbonManagerMutex);
                if (m_RibbonManager.done()) {                                     // nothing left to cover: the mission is over (:97-104)
                    std::cerr << "Finished covering ribbons" << std::endl;
                    out.allDone();
                    break;
                }
                out.displayRibbons(m_RibbonManager);
                ribbons = m_RibbonManager;                                        // private copy for this cycle (:181-185)
            }
            if (cyc.from.time() == -1)                                            // no answer from the controller: dead reckoning (:114-118)
                cyc.from = m_LastState.push(out.getTime() + m_PlanningTimeSeconds - m_LastState.time());
            {
                std::unique_lock<std::mutex> guard(m_MapMutex, std::defer_lock);   // never wait for a map that is still loading (:121-142)
                if (guard.try_lock()) {
                    if (m_NewMap) m_PlannerConfig.setMap(m_NewMap);
                    m_NewMap = nullptr;
                    if (m_PlannerConfig.map()->isBlocked(cyc.from.x(), cyc.from.y())) {
                        *m_PlannerConfig.output() << "We've run aground, according to the most recent map! Ending task now" << std::endl;
                        out.allDone();
                        break;
                    }
                }
            }
            if (!c_ReusePlanEnabled) cyc.last.Plan = DubinsPlan();
            if (!cyc.last.Plan.empty()) cyc.last.Plan.changeIntoSuffix(cyc.from.time());   // what is left of the last plan (:146)
            // this cycle's snapshot of the contacts: the callbacks keep updating the live managers while the planner reads its copy
            DynamicObstaclesManager::SharedPtr contacts;
            {
                std::lock_guard<std::mutex> guard(m_ObstaclesMutex);
                if (m_UseGaussianDynamicObstacles) contacts = std::make_shared<GaussianDynamicObstaclesManager>(*m_GaussianDynamicObstaclesManager);
                else contacts = std::make_shared<BinaryDynamicObstaclesManager>(*m_BinaryDynamicObstaclesManager);
            }
            const double hitNow = contacts->DynamicObstaclesManager::collisionExists(m_LastState, false);   // base class on purpose (:160-165)
            collisionsSeen += hitNow;
            m_Cycles++;

            // ---- solve(): one plan() call with whatever is left of this cycle's time budget
            try {
                if (m_IgnoreDynamicObstacles) m_PlannerConfig.setObstaclesManager(std::make_shared<DynamicObstaclesManager>());
                else m_PlannerConfig.setObstaclesManager(contacts);
                ribbons.coverBetween(m_LastState.x(), m_LastState.y(), cyc.from.x(), cyc.from.y(), false);   // up to where we plan from (:186)
                GpuAStarPlanner planner(m_Contexts);                              // stateless: a new one every cycle (:85-90); the device contexts persist
                const double budget = cycleStart + m_PlanningTimeSeconds - out.getTime();
                if (m_CycleObserver) {
                    CycleRecord rec;
                    rec.cycle = m_Cycles - 1; rec.from = cyc.from; rec.previousPlanLegs = cyc.last.Plan.get().size();
                    rec.timeHorizon = m_PlannerConfig.timeHorizon(); rec.timeRemaining = budget;
                    rec.ribbons = (size_t)ribbons.count(); rec.uncoveredLength = ribbons.getTotalUncoveredLength();
                    rec.emptyInARow = cyc.emptyInARow; rec.lastPlanAchievable = cyc.lastPlanAchievable;
                    m_CycleObserver(rec);
                }
                cyc.last = planner.plan(ribbons, cyc.from, m_PlannerConfig, cyc.last.Plan, budget);
            } catch (const std::exception& e) {                                   // logged, plan dropped, loop continues (:191-195)
                std::cerr << "Exception thrown while planning: " << e.what() << " - proceeding without a plan." << std::endl;
                cyc.last.Plan = DubinsPlan();
            } catch (...) {                                                       // anything else stops the executive (:196-200)
                std::cerr << "Unknown exception thrown while planning; pausing" << std::endl;
                cancelPlanner();
                throw;
            }
            out.publishStats(cyc.last, hitNow * kPenaltyPerCollision, 0, cyc.lastPlanAchievable);

            // ---- pace(): the loop runs at one cycle per planning period (:205-211)
            const int spareMs = (int)((m_PlanningTimeSeconds - (out.getTime() - cycleStart)) * 1000);
            if (spareMs >= 0) std::this_thread::sleep_for(std::chrono::milliseconds(spareMs));
            out.displayTrajectory(cyc.last.Plan.getHalfSecondSamples(), true, cyc.last.Plan.dangerous());

            if (cyc.last.Plan.empty()) {
                // ---- backOff(): three empty plans in a row halve the horizon, never below the minimum (:270-287)
                std::cerr << "Planner returned empty trajectory." << std::endl;
                m_EmptyPlans++;
                cyc.from = State();
                if (++cyc.emptyInARow > 2) {
                    const double halved = m_PlannerConfig.timeHorizon() / 2;
                    if (halved < m_PlannerConfig.timeMinimum()) {
                        m_PlannerConfig.setTimeHorizon(m_PlannerConfig.timeMinimum());
                    } else {
                        m_PlannerConfig.setTimeHorizon(halved);
                        std::cerr << "Failed " << cyc.emptyInARow << " times in a row. Reducing time horizon to " << halved << std::endl;
                        cyc.emptyInARow = 0;
                    }
                }
                continue;
            }

            // ---- handOver(): the controller takes the plan and says where the next one starts (:217-268)
            cyc.emptyInARow = 0;
            try {
                cyc.from = out.publishPlan(cyc.last.Plan);
            } catch (const std::exception& e) {
                std::cerr << "Exception thrown while updating controller's reference trajectory: " << e.what() << " - pausing." << std::endl;
                cancelPlanner();
            } catch (...) {
                cancelPlanner();
                throw;
            }
            if (!cyc.last.Plan.containsTime(cyc.from.time()) && cancelled()) break;   // a cancelled controller may answer nonsense (:234-241)
            State onPlan(cyc.from);
            cyc.last.Plan.sample(onPlan);
            cyc.lastPlanAchievable = cyc.from.isCoLocated(onPlan);                 // the controller expects to be on the plan: reuse it (:242-262)
            if (!cyc.lastPlanAchievable) cyc.last.Plan = DubinsPlan();
        }
    } catch (const std::exception& e) {
        std::cerr << "Exception thrown in plan loop: " << e.what() << " - pausing." << std::endl;
        cancelPlanner();
    } catch (...) {
        std::cerr << "Unknown exception thrown in plan loop" << std::endl;
    }

    // ---- summarise(): task-level figures (:293-304)
    const double wall = out.getTime() - missionStart;
    const double collisionScore = collisionsSeen * kPenaltyPerCollision;
    double uncovered;
    {
        std::lock_guard<std::mutex> guard(m_RibbonManagerMutex);
        uncovered = m_RibbonManager.getTotalUncoveredLength();
    }
    out.publishTaskLevelStats(wall, collisionScore, wall * kPenaltyPerSecond + collisionScore, uncovered);
    std::unique_lock<std::mutex> guard(m_PlannerStateMutex);
    m_PlannerState = PlannerState::Inactive;
    m_CancelCV.notify_all();
}

}  // namespace ppamd
