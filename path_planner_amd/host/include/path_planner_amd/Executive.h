// Executive / TrajectoryPublisher — ROS-free mirrors of
//   /root/reference/path_planner/src/executive/executive.{h,cpp}  (planLoop :43-305, callbacks :29-41,308-440)
//   /root/reference/path_planner/src/trajectory_publisher.h       (the interface the ROS node implements)
// so that the 10 Hz planning harness — plan reuse, covering up to the start state, failure back-off, cancellation —
// runs unchanged around GpuAStarPlanner.  PotentialFieldPlanner and GeoTiffMap are not on this path.
#pragma once
#include <condition_variable>
#include <functional>
#include <future>
#include <mutex>
#include <string>

#include "Planner.h"

namespace ppamd {

class TrajectoryPublisher {   // trajectory_publisher.h:12-60
public:
    virtual ~TrajectoryPublisher() = default;
    virtual State publishPlan(const DubinsPlan& plan) = 0;           // returns the state to plan from next
    virtual void displayTrajectory(std::vector<State> trajectory, bool plannerTrajectory, bool dangerous) {}
    virtual void displayDynamicObstacle(double x, double y, double yaw, double width, double length, uint32_t id) {}
    virtual void publishStats(const Planner::Stats& stats, double collisionPenalty, unsigned long cpuTime, bool lastPlanAchievable) {}
    virtual void publishTaskLevelStats(double wallClockTime, double cumulativeCollisionPenalty, double cumulativeGValue, double uncoveredLength) {}
    virtual void displayMap(std::string path) {}
    virtual void displayRibbons(const RibbonManager& ribbonManager) {}
    virtual void allDone() = 0;
    virtual double getTime() const = 0;
};

class Executive {
public:
    explicit Executive(TrajectoryPublisher* trajectoryPublisher);
    ~Executive();
    void updateCovered(double x, double y, double speed, double heading, double t);   // executive.cpp:34-41
    void addRibbon(double x1, double y1, double x2, double y2);
    void clearRibbons();
    void updateDynamicObstacle(uint32_t mmsi, State obstacle, double width, double length);
    void refreshMap(const std::string& pathToMapFile, double latitude, double longitude);   // GridWorldMap files, or "" = no map
    void setConfiguration(double turningRadius, double coverageTurningRadius, double maxSpeed, double slowSpeed, double lineWidth, int k,
                          int heuristic, double timeHorizon, double timeMinimum, double collisionCheckingIncrement, int initialSamples,
                          bool useBrownPaths, bool useGaussianDynamicObstacles, bool ignoreDynamicObstacles, bool usePotentialField);
    void startPlanner();
    void cancelPlanner();
    void terminate();
    void setPlannerVisualization(bool visualize, const std::string& visualizationFilePath) {   // executive.cpp:443-449
        m_PlannerConfig.setVisualizations(visualize);
        if (visualize) m_PlannerConfig.setVisualizer(std::make_shared<Visualizer>(visualizationFilePath));
    }
    // this build's knobs
    void setPlanningTimeSeconds(double s) { m_PlanningTimeSeconds = s; }   // reference: c_PlanningTimeSeconds = 0.85 (executive.h:183)
    void setSpeculation(int n) { m_PlannerConfig.setSpeculation(n); }
    // PlannerConfig::setDeadlineGuard: off for a clock that does not advance while the device works (a scripted, counting clock)
    void setDeadlineGuard(bool on) { m_PlannerConfig.setDeadlineGuard(on); }
    // What the loop decided for a cycle, handed to an observer right before that cycle's plan() call (called on the planning
    // thread): the state it plans from (executive.cpp:114-118,217-268), how much of the last plan it hands back (:144-146), the
    // horizon after any back-off (:270-287), the time budget (:189-190), the ribbons left.  tests/test_gpu_mission.py compares these
    // with the oracle's restatement of the loop, cycle by cycle.
    struct CycleRecord {
        unsigned long cycle = 0;
        State from;
        size_t previousPlanLegs = 0;
        double timeHorizon = 0, timeRemaining = 0;
        size_t ribbons = 0;
        double uncoveredLength = 0;
        int emptyInARow = 0;
        bool lastPlanAchievable = false;
    };
    void setCycleObserver(std::function<void(const CycleRecord&)> f) { m_CycleObserver = std::move(f); }
    // the devices every cycle's planner works on (default: device 0).  The contexts are process-level (GpuContext::shared) and
    // are held here for the life of the executive, so no cycle pays for device allocations an earlier cycle already made.
    void setDevices(const std::vector<int>& devices) { m_Contexts = GpuContext::shared(devices); }
    bool waitUntilInactive(double seconds);
    unsigned long cycles() const { return m_Cycles; }
    unsigned long emptyPlans() const { return m_EmptyPlans; }

    enum class PlannerState { Running, Cancelled, Inactive };

private:
    void planLoop();
    static constexpr bool c_ReusePlanEnabled = true;               // executive.h:175
    static constexpr double c_CoverageHeadingRateMax = 0.1;        // executive.h (coverage only while the heading is steady)

    TrajectoryPublisher* m_TrajectoryPublisher;
    PlannerConfig m_PlannerConfig;
    RibbonManager m_RibbonManager;
    std::mutex m_RibbonManagerMutex, m_MapMutex, m_PlannerStateMutex, m_ObstaclesMutex;
    std::vector<std::shared_ptr<GpuContext>> m_Contexts;
    std::condition_variable m_CancelCV;
    PlannerState m_PlannerState = PlannerState::Inactive;
    std::future<void> m_PlanningFuture;
    Map::SharedPtr m_NewMap;
    std::string m_CurrentMapPath;
    std::shared_ptr<BinaryDynamicObstaclesManager> m_BinaryDynamicObstaclesManager;
    std::shared_ptr<GaussianDynamicObstaclesManager> m_GaussianDynamicObstaclesManager;
    bool m_UseGaussianDynamicObstacles = false, m_IgnoreDynamicObstacles = false;
    State m_LastState;
    double m_LastHeading = 0, m_LastUpdateTime = 1;
    double m_PlanningTimeSeconds = 0.85;
    unsigned long m_Cycles = 0, m_EmptyPlans = 0;
    std::function<void(const CycleRecord&)> m_CycleObserver;
};

}  // namespace ppamd
