// Executive — see Executive.h.  The control flow of planLoop() follows executive.cpp:43-305 line by line; what differs is
// named where it differs (no PotentialFieldPlanner, no GeoTiffMap, no visualisation stream, GpuAStarPlanner as the planner).
#include "path_planner_amd/Executive.h"

#include <chrono>
#include <fstream>
#include <iostream>
#include <thread>

namespace ppamd {

static const double kCollisionPenaltyFactorExec = 600.0;   // Edge::collisionPenaltyFactor(), Edge.h:151
static const double kTimePenaltyFactorExec = 1.0;          // Edge::timePenaltyFactor(), Edge.h:152

Executive::Executive(TrajectoryPublisher* trajectoryPublisher) : m_PlannerConfig(&std::cerr) {   // executive.cpp:14-19
    m_TrajectoryPublisher = trajectoryPublisher;
    m_PlannerConfig.setNowFunction([&] { return m_TrajectoryPublisher->getTime(); });
    m_BinaryDynamicObstaclesManager = std::make_shared<BinaryDynamicObstaclesManager>();
    m_GaussianDynamicObstaclesManager = std::make_shared<GaussianDynamicObstaclesManager>();
    m_RibbonManager = RibbonManager(RibbonManager::TspPointRobotNoSplitKRibbons, m_PlannerConfig.turningRadius(), 2);
}

Executive::~Executive() {   // :21-24
    terminate();
    if (m_PlanningFuture.valid()) m_PlanningFuture.wait_for(std::chrono::seconds(2));
}

void Executive::updateCovered(double x, double y, double speed, double heading, double t) {   // :34-41
    if ((m_LastHeading - heading) / m_LastUpdateTime <= c_CoverageHeadingRateMax) {
        std::lock_guard<std::mutex> lock(m_RibbonManagerMutex);
        m_RibbonManager.cover(x, y, false);
    }
    m_LastUpdateTime = t; m_LastHeading = heading;
    m_LastState = State(x, y, heading, speed, t);
}

void Executive::addRibbon(double x1, double y1, double x2, double y2) {   // :384-387
    std::lock_guard<std::mutex> lock(m_RibbonManagerMutex);
    m_RibbonManager.add(x1, y1, x2, y2);
}

void Executive::clearRibbons() {   // :404-407
    std::lock_guard<std::mutex> lock(m_RibbonManagerMutex);
    m_RibbonManager = RibbonManager(RibbonManager::TspPointRobotNoSplitKRibbons, m_PlannerConfig.turningRadius(), 2);
}

void Executive::updateDynamicObstacle(uint32_t mmsi, State obstacle, double width, double length) {   // :313-319
    m_BinaryDynamicObstaclesManager->update(mmsi, obstacle.x(), obstacle.y(), obstacle.heading(), obstacle.speed(), obstacle.time(), width, length);
    m_GaussianDynamicObstaclesManager->update(mmsi, obstacle.x(), obstacle.y(), obstacle.heading(), obstacle.speed(), obstacle.time());
}

void Executive::refreshMap(const std::string& pathToMapFile, double, double) {   // :321-382; synchronous here, GridWorldMap files only
    std::lock_guard<std::mutex> lock(m_MapMutex);
    if (pathToMapFile.empty()) {
        m_NewMap = std::make_shared<Map>();
        m_CurrentMapPath = pathToMapFile;
        *m_PlannerConfig.output() << "Map cleared. Using empty map now." << std::endl;
        m_TrajectoryPublisher->displayMap(pathToMapFile);
        return;
    }
    try {
        if (!std::ifstream(pathToMapFile).good()) {
            *m_PlannerConfig.output() << "Cannot find map file: " << pathToMapFile << "\\nUsing empty map  for now." << std::endl;
            m_NewMap = std::make_shared<Map>();
            m_CurrentMapPath = "";
            m_TrajectoryPublisher->displayMap("");
            return;
        }
        if (pathToMapFile.find(".map") == std::string::npos)
            throw std::runtime_error("GeoTIFF maps are outside this build (SURVEY 2 row 14): rasterise to a GridWorldMap file");
        m_NewMap = std::make_shared<GridWorldMap>(pathToMapFile);
        m_TrajectoryPublisher->displayMap(pathToMapFile);
        m_CurrentMapPath = pathToMapFile;
        *m_PlannerConfig.output() << "Loaded map file: " << pathToMapFile << std::endl;
    } catch (...) {
        *m_PlannerConfig.output() << "Encountered an error loading map at path " << pathToMapFile << ".\\nMap was not updated." << std::endl;
        m_NewMap = nullptr;
        m_CurrentMapPath = "";
    }
}

void Executive::setConfiguration(double turningRadius, double coverageTurningRadius, double maxSpeed, double slowSpeed, double lineWidth, int k,
                                 int heuristic, double timeHorizon, double timeMinimum, double collisionCheckingIncrement, int initialSamples,
                                 bool useBrownPaths, bool useGaussianDynamicObstacles, bool ignoreDynamicObstacles, bool usePotentialField) {   // :409-440
    m_PlannerConfig.setTurningRadius(turningRadius);
    m_PlannerConfig.setCoverageTurningRadius(coverageTurningRadius);
    m_PlannerConfig.setMaxSpeed(maxSpeed);
    m_PlannerConfig.setSlowSpeed(slowSpeed);
    RibbonManager::setRibbonWidth(lineWidth);
    m_PlannerConfig.setBranchingFactor(k);
    switch (heuristic) {   // path_planner.cfg numbering
        case 0: m_RibbonManager.setHeuristic(RibbonManager::TspPointRobotNoSplitAllRibbons); break;
        case 1: m_RibbonManager.setHeuristic(RibbonManager::TspPointRobotNoSplitKRibbons); break;
        case 2: m_RibbonManager.setHeuristic(RibbonManager::MaxDistance); break;
        case 3: m_RibbonManager.setHeuristic(RibbonManager::TspDubinsNoSplitAllRibbons); break;
        case 4: m_RibbonManager.setHeuristic(RibbonManager::TspDubinsNoSplitKRibbons); break;
        default: *m_PlannerConfig.output() << "Unknown heuristic. Ignoring." << std::endl; break;
    }
    m_PlannerConfig.setTimeHorizon(timeHorizon);
    m_PlannerConfig.setTimeMinimum(timeMinimum);
    m_PlannerConfig.setCollisionCheckingIncrement(collisionCheckingIncrement);
    m_PlannerConfig.setInitialSamples(initialSamples);
    m_PlannerConfig.setUseBrownPaths(useBrownPaths);
    m_UseGaussianDynamicObstacles = useGaussianDynamicObstacles;
    m_IgnoreDynamicObstacles = ignoreDynamicObstacles;
    if (usePotentialField) *m_PlannerConfig.output() << "PotentialFieldPlanner is outside this build: using the A* planner." << std::endl;
}

void Executive::startPlanner() {   // :442-451
    if (!m_PlannerConfig.map()) m_PlannerConfig.setMap(std::make_shared<Map>());
    std::unique_lock<std::mutex> lock(m_PlannerStateMutex);
    if (m_PlannerState != PlannerState::Running) m_PlanningFuture = std::async(std::launch::async, &Executive::planLoop, this);
}

void Executive::cancelPlanner() {   // :453-459
    std::unique_lock<std::mutex> lock(m_PlannerStateMutex);
    if (m_PlannerState == PlannerState::Running) m_PlannerState = PlannerState::Cancelled;
}

void Executive::terminate() { cancelPlanner(); }   // :308-311

bool Executive::waitUntilInactive(double seconds) {
    if (!m_PlanningFuture.valid()) return true;
    return m_PlanningFuture.wait_for(std::chrono::duration<double>(seconds)) == std::future_status::ready;
}

void Executive::planLoop() {   // executive.cpp:43-305
    double trialStartTime = m_TrajectoryPublisher->getTime(), cumulativeCollisionPenalty = 0;
    try {
        {
            std::unique_lock<std::mutex> lock(m_PlannerStateMutex);
            m_CancelCV.wait_for(lock, std::chrono::seconds(2), [=] { return m_PlannerState != PlannerState::Cancelled; });
            if (m_PlannerState == PlannerState::Cancelled) {
                std::cerr << "Planner initialization timed out. Cancel flag is still set." << std::endl;
                return;
            }
            m_PlannerState = PlannerState::Running;
        }
        State startState;
        Planner::Stats stats;               // declared here so that the plan persists between loops
        bool lastPlanAchievable = false;
        int failureCount = 0;               // how many times in a row no plan was found
        while (true) {
            double startTime = m_TrajectoryPublisher->getTime();
            std::unique_ptr<Planner> planner(new GpuAStarPlanner);   // planner is stateless: a new instance each time (:85-90)
            {
                std::unique_lock<std::mutex> lock(m_PlannerStateMutex);
                if (m_PlannerState == PlannerState::Cancelled) break;
            }
            {
                std::lock_guard<std::mutex> lock1(m_RibbonManagerMutex);
                if (m_RibbonManager.done()) {
                    std::cerr << "Finished covering ribbons" << std::endl;
                    m_TrajectoryPublisher->allDone();
                    break;
                }
            }
            {
                std::lock_guard<std::mutex> lock1(m_RibbonManagerMutex);
                m_TrajectoryPublisher->displayRibbons(m_RibbonManager);
            }
            // if the state estimator returned an error naively do it ourselves (:114-118)
            if (startState.time() == -1)
                startState = m_LastState.push(m_TrajectoryPublisher->getTime() + m_PlanningTimeSeconds - m_LastState.time());
            {
                std::unique_lock<std::mutex> lock1(m_MapMutex, std::defer_lock);
                if (lock1.try_lock()) {
                    if (m_NewMap) m_PlannerConfig.setMap(m_NewMap);
                    m_NewMap = nullptr;
                    if (m_PlannerConfig.map()->isBlocked(startState.x(), startState.y())) {
                        *m_PlannerConfig.output() << "We've run aground, according to the most recent map!\\nEnding task now" << std::endl;
                        m_TrajectoryPublisher->allDone();
                        break;
                    }
                }
            }
            if (!c_ReusePlanEnabled) stats.Plan = DubinsPlan();
            if (!stats.Plan.empty()) stats.Plan.changeIntoSuffix(startState.time());   // update the last plan
            // check for collision penalty (:158-167)
            double collisionPenalty = 0;
            if (m_UseGaussianDynamicObstacles)
                collisionPenalty = m_GaussianDynamicObstaclesManager->DynamicObstaclesManager::collisionExists(m_LastState, false);
            else
                collisionPenalty = m_BinaryDynamicObstaclesManager->DynamicObstaclesManager::collisionExists(m_LastState, false);
            cumulativeCollisionPenalty += collisionPenalty;
            m_Cycles++;
            try {
                if (m_IgnoreDynamicObstacles) m_PlannerConfig.setObstaclesManager(std::make_shared<DynamicObstaclesManager>());
                else if (m_UseGaussianDynamicObstacles) m_PlannerConfig.setObstaclesManager(m_GaussianDynamicObstaclesManager);
                else m_PlannerConfig.setObstaclesManager(m_BinaryDynamicObstaclesManager);
                RibbonManager ribbonManagerCopy;
                {
                    std::lock_guard<std::mutex> lock(m_RibbonManagerMutex);
                    ribbonManagerCopy = m_RibbonManager;
                }
                // cover up to the state that we're planning from (:186)
                ribbonManagerCopy.coverBetween(m_LastState.x(), m_LastState.y(), startState.x(), startState.y(), false);
                stats = planner->plan(ribbonManagerCopy, startState, m_PlannerConfig, stats.Plan,
                                      startTime + m_PlanningTimeSeconds - m_TrajectoryPublisher->getTime());
            } catch (const std::exception& e) {
                std::cerr << "Exception thrown while planning:\\n" << e.what() << "\\nIgnoring that and just trying to proceed." << std::endl;
                stats.Plan = DubinsPlan();
            } catch (...) {
                std::cerr << "Unknown exception thrown while planning; pausing" << std::endl;
                cancelPlanner();
                throw;
            }
            m_TrajectoryPublisher->publishStats(stats, collisionPenalty * kCollisionPenaltyFactorExec, 0, lastPlanAchievable);
            // calculate remaining time (to sleep) (:205-211)
            double endTime = m_TrajectoryPublisher->getTime();
            int sleepTime = ((int)((m_PlanningTimeSeconds - (endTime - startTime)) * 1000));
            if (sleepTime >= 0) std::this_thread::sleep_for(std::chrono::milliseconds(sleepTime));
            m_TrajectoryPublisher->displayTrajectory(stats.Plan.getHalfSecondSamples(), true, stats.Plan.dangerous());
            if (!stats.Plan.empty()) {
                failureCount = 0;
                try {
                    startState = m_TrajectoryPublisher->publishPlan(stats.Plan);   // send trajectory to controller
                } catch (const std::exception& e) {
                    std::cerr << "Exception thrown while updating controller's reference trajectory:\\n" << e.what() << "\\nPausing." << std::endl;
                    cancelPlanner();
                } catch (...) {
                    cancelPlanner();
                    throw;
                }
                if (!stats.Plan.containsTime(startState.time())) {
                    std::unique_lock<std::mutex> lock2(m_PlannerStateMutex);
                    if (m_PlannerState == PlannerState::Cancelled) break;
                }
                State expectedStartState(startState);
                stats.Plan.sample(expectedStartState);
                if (!startState.isCoLocated(expectedStartState)) {
                    stats.Plan = DubinsPlan();   // reset plan because controller says we can't make it
                    lastPlanAchievable = false;
                } else {
                    lastPlanAchievable = true;   // expected start state is along plan: pass it to the planner as previous plan
                }
            } else {
                std::cerr << "Planner returned empty trajectory." << std::endl;
                m_EmptyPlans++;
                startState = State();
                failureCount++;
                if (failureCount > 2) {   // :276-287
                    m_PlannerConfig.setTimeHorizon(m_PlannerConfig.timeHorizon() / 2);
                    if (m_PlannerConfig.timeHorizon() < m_PlannerConfig.timeMinimum()) {
                        m_PlannerConfig.setTimeHorizon(m_PlannerConfig.timeMinimum());   // prevent from getting too small
                    } else {
                        std::cerr << "Failed " << failureCount << " times in a row. Reducing time horizon to " << m_PlannerConfig.timeHorizon() << std::endl;
                        failureCount = 0;
                    }
                }
            }
        }
    } catch (const std::exception& e) {
        std::cerr << "Exception thrown in plan loop:\\n" << e.what() << "\\nPausing." << std::endl;
        cancelPlanner();
    } catch (...) {
        std::cerr << "Unknown exception thrown in plan loop" << std::endl;
    }
    // task-level stats reporting (:293-304)
    auto trialEndTime = m_TrajectoryPublisher->getTime();
    auto wallClockTime = trialEndTime - trialStartTime;
    cumulativeCollisionPenalty *= kCollisionPenaltyFactorExec;
    auto timePenalty = wallClockTime * kTimePenaltyFactorExec;
    double uncoveredLength;
    {
        std::lock_guard<std::mutex> lock(m_RibbonManagerMutex);
        uncoveredLength = m_RibbonManager.getTotalUncoveredLength();
    }
    m_TrajectoryPublisher->publishTaskLevelStats(wallClockTime, cumulativeCollisionPenalty, timePenalty + cumulativeCollisionPenalty, uncoveredLength);
    std::unique_lock<std::mutex> lock2(m_PlannerStateMutex);
    m_PlannerState = PlannerState::Inactive;
    m_CancelCV.notify_all();
}

}  // namespace ppamd
