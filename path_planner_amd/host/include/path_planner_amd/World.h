// Map / GridWorldMap / DynamicObstaclesManager / Binary- and GaussianDynamicObstaclesManager / PlannerConfig — host mirrors of
//   /root/reference/path_planner/src/common/map/{Map,GridWorldMap}.{h,cpp}
//   /root/reference/path_planner/src/common/dynamic_obstacles/{DynamicObstaclesManager,BinaryDynamicObstaclesManager,
//       GaussianDynamicObstaclesManager}.{h,cpp}
//   /root/reference/path_planner/src/planner/PlannerConfig.h
// The planner uploads snapshots of them to the device at the start of every plan() call.
#pragma once
#include <cfloat>
#include <cstdint>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "State.h"

namespace ppamd {

class Map {
public:
    typedef std::shared_ptr<Map> SharedPtr;
    virtual ~Map() = default;
    virtual bool isBlocked(double x, double y) const { return false; }        // Map.cpp:4-6
    virtual const double* extremes() const { return m_Extremes; }             // Map.cpp:8-10
    virtual double resolution() const { return 0; }                           // Map.cpp:12-14
    // Occupancy as a byte grid for the device (row 0 = y in [0,res)).  The base map has none (rows = 0).
    // Any subclass that only implements isBlocked/extremes/resolution is rasterised cell centre by cell centre.
    virtual void rasterize(std::vector<uint8_t>& cells, int& rows, int& cols, double& res) const;
    // Non-zero: the map's cells never change while this value is the same (an immutable map returns one number for its whole
    // life, a map that is edited in place a new one after every edit); the planner then rasterises and uploads it once and
    // reuses the device's copy in later cycles (Executive hands every cycle's planner the same shared_ptr<Map> until the map
    // loader replaces it, executive.cpp:321-369).  0 (the default): unknown, upload at every plan().
    virtual unsigned long version() const { return 0; }

protected:
    static unsigned long nextVersion();       // process-unique, never 0

private:
    double m_Extremes[4] = {-DBL_MAX, DBL_MAX, -DBL_MAX, DBL_MAX};
};

// GridWorldMap: the text occupancy grid of the reference's tests and simulator (GridWorldMap.cpp:10-93).  File format: first
// line the cell size in metres, then one text line per row, '#' = blocked, the LAST line being y = 0; the grid is as wide as
// the shortest line; everything outside [0, cols * res) x [0, rows * res) is blocked.
// Stored the way the device wants it: one bit per cell, 64 cells per word, row 0 = y in [0, res).
class GridWorldMap : public Map {
public:
    explicit GridWorldMap(const std::string& path);
    static std::shared_ptr<GridWorldMap> fromText(const std::string& text);
    bool isBlocked(double x, double y) const override;
    const double* extremes() const override { return m_Box; }
    double resolution() const override { return m_CellSize; }
    void rasterize(std::vector<uint8_t>& cells, int& rows, int& cols, double& res) const override;
    unsigned long version() const override { return m_Version; }      // immutable after construction
    int rows() const { return m_Rows; }
    int cols() const { return m_Cols; }

private:
    GridWorldMap() = default;
    void parse(std::istream& in);
    bool cell(size_t row, size_t col) const { return (m_Bits[row * m_WordsPerRow + (col >> 6)] >> (col & 63)) & 1u; }
    std::vector<uint64_t> m_Bits;
    int m_Rows = 0, m_Cols = 0;
    size_t m_WordsPerRow = 0;
    double m_CellSize = 0;
    double m_Box[4] = {0, 0, 0, 0};      // minX, maxX, minY, maxY
    unsigned long m_Version = nextVersion();
};

class DynamicObstaclesManager {
public:
    typedef std::shared_ptr<DynamicObstaclesManager> SharedPtr;
    virtual ~DynamicObstaclesManager() = default;
    virtual double collisionExists(double x, double y, double time, bool strict) const { return 0; }
    double collisionExists(const State& s, bool strict) const { return collisionExists(s.x(), s.y(), s.time(), strict); }
    // device snapshot: model id (PPGPU_OBST_*) and rows of {x, y, heading, speed, time, width, length} (binary) or
    // {x, y, heading, speed, time, c00, c01, c10, c11} (Gaussian)
    virtual int deviceModel() const { return 0; }
    virtual void deviceRows(std::vector<double>& rows) const { rows.clear(); }
};

// Obstacle tracks keyed by MMSI, kept as a dense array in the order the contacts were first reported (the row order the device
// receives; the reference keeps an unordered_map, whose iteration order is unspecified — for the binary model the result is a
// count and cannot depend on it) with a side index from MMSI to slot.  forget() moves the last track into the freed slot.
template <typename Track>
class TrackTable {
public:
    Track* find(uint32_t mmsi) {
        const auto it = m_Slot.find(mmsi);
        return it == m_Slot.end() ? nullptr : &m_Tracks[it->second];
    }
    void put(uint32_t mmsi, const Track& t) {
        if (Track* have = find(mmsi)) { *have = t; return; }
        m_Slot[mmsi] = m_Tracks.size();
        m_Tracks.push_back(t);
        m_Mmsi.push_back(mmsi);
    }
    void drop(uint32_t mmsi) {
        const auto it = m_Slot.find(mmsi);
        if (it == m_Slot.end()) return;
        const size_t slot = it->second, last = m_Tracks.size() - 1;
        if (slot != last) { m_Tracks[slot] = m_Tracks[last]; m_Mmsi[slot] = m_Mmsi[last]; m_Slot[m_Mmsi[slot]] = slot; }
        m_Tracks.pop_back(); m_Mmsi.pop_back();
        m_Slot.erase(it);
    }
    const std::vector<Track>& tracks() const { return m_Tracks; }
    void mute(uint32_t mmsi) { m_Muted.insert(mmsi); }
    void unmute(uint32_t mmsi) { m_Muted.erase(mmsi); }
    bool muted(uint32_t mmsi) const { return m_Muted.count(mmsi) != 0; }

private:
    std::vector<Track> m_Tracks;
    std::vector<uint32_t> m_Mmsi;
    std::unordered_map<uint32_t, size_t> m_Slot;
    std::unordered_set<uint32_t> m_Muted;
};

// BinaryDynamicObstaclesManager (.h:14-47, .cpp:4-35): each contact is a box of Width x Length (each 2 m larger for the strict
// test) moving at constant speed along its heading; collisionExists counts the boxes that hold the query point at the query
// time.  cos/sin of the contact's yaw are taken once, when the contact is reported (the same libm calls the reference makes
// per query).
class BinaryDynamicObstaclesManager : public DynamicObstaclesManager {
public:
    typedef std::shared_ptr<BinaryDynamicObstaclesManager> SharedPtr;
    struct Track { double x, y, heading, speed, time, width, length, cosYaw, sinYaw; };
    void update(uint32_t mmsi, double x, double y, double heading, double speed, double time, double width, double length);
    void forget(uint32_t mmsi) { m_Table.drop(mmsi); }
    void addIgnore(uint32_t mmsi) { m_Table.mute(mmsi); }
    void removeIgnore(uint32_t mmsi) { m_Table.unmute(mmsi); }
    double collisionExists(double x, double y, double time, bool strict) const override;
    size_t size() const { return m_Table.tracks().size(); }
    int deviceModel() const override { return 1; }
    void deviceRows(std::vector<double>& rows7) const override;

private:
    TrackTable<Track> m_Table;
};

// GaussianDynamicObstaclesManager (.h:19-49, .cpp:3-47), without Eigen: a contact is a bivariate normal that moves like a binary
// contact's box; the inverse covariance and the normalisation are computed when the contact is reported, in the order Eigen's
// fixed-size 2x2 code evaluates them; collisionExists is the sum of the densities in report order, 0 below 1e-5.
class GaussianDynamicObstaclesManager : public DynamicObstaclesManager {
public:
    typedef std::shared_ptr<GaussianDynamicObstaclesManager> SharedPtr;
    struct Track { double x, y, heading, speed, time, cosYaw, sinYaw, cov[4], inv[4], norm; };
    void update(uint32_t mmsi, double x, double y, double heading, double speed, double time);   // covariance [[30,10],[10,30]]
    void update(uint32_t mmsi, double x, double y, double heading, double speed, double time, const double covariance[4]);
    void forget(uint32_t mmsi) { m_Table.drop(mmsi); }
    void addIgnore(uint32_t mmsi) { m_Table.mute(mmsi); }
    void removeIgnore(uint32_t mmsi) { m_Table.unmute(mmsi); }
    double collisionExists(double x, double y, double time, bool strict) const override;
    size_t size() const { return m_Table.tracks().size(); }
    int deviceModel() const override { return 2; }
    void deviceRows(std::vector<double>& rows9) const override;

private:
    TrackTable<Track> m_Table;
};

// Visualizer.h: owns the file the search is dumped to (the text format visualizer.py:485-547 reads)
class Visualizer {
public:
    typedef std::shared_ptr<Visualizer> SharedPtr;
    explicit Visualizer(const std::string& path) { m_File.open(path, std::ios::trunc | std::ios::out); }
    std::ostream& stream() { return m_File; }

private:
    std::ofstream m_File;
};

class PlannerConfig {
public:
    explicit PlannerConfig(std::ostream* output) : m_Output(output) {}
    int branchingFactor() const { return m_BranchingFactor; }
    void setBranchingFactor(int b) { m_BranchingFactor = b; }
    double maxSpeed() const { return m_MaxSpeed; }
    void setMaxSpeed(double s) { m_MaxSpeed = s; }
    double turningRadius() const { return m_TurningRadius; }
    void setTurningRadius(double r) { m_TurningRadius = r; }
    double coverageTurningRadius() const { return m_CoverageTurningRadius; }
    void setCoverageTurningRadius(double r) { m_CoverageTurningRadius = r; }
    const Map::SharedPtr& map() const { return m_Map; }
    void setMap(const Map::SharedPtr& m) { m_Map = m; }
    const DynamicObstaclesManager& obstaclesManager() const { return *m_ObstaclesManager; }
    void setObstaclesManager(DynamicObstaclesManager::SharedPtr m) { m_ObstaclesManager = std::move(m); }
    std::ostream* output() const { return m_Output; }
    void setNowFunction(const std::function<double()>& f) { m_NowFunction = f; }
    double now() const { return m_NowFunction(); }
    double startStateTime() const { return m_StartStateTime; }
    void setStartStateTime(double t) { m_StartStateTime = t; }
    double timeHorizon() const { return m_TimeHorizon; }
    void setTimeHorizon(double t) { m_TimeHorizon = t; }
    bool useBrownPaths() const { return m_UseBrownPaths; }
    void setUseBrownPaths(bool b) { m_UseBrownPaths = b; }
    int initialSamples() const { return m_InitialSamples; }
    void setInitialSamples(int n) { m_InitialSamples = n; }
    double collisionCheckingIncrement() const { return m_CollisionCheckingIncrement; }
    void setCollisionCheckingIncrement(double d) { m_CollisionCheckingIncrement = d; }
    double timeMinimum() const { return m_TimeMinimum; }
    void setTimeMinimum(double t) { m_TimeMinimum = t; }
    double slowSpeed() const { return m_SlowSpeed <= 0 ? m_MaxSpeed : m_SlowSpeed; }
    void setSlowSpeed(double s) { m_SlowSpeed = s; }
    // How many open vertices one device round trip expands (this planner only; 1 = one vertex at a time, exactly the
    // reference's call pattern).  Results do not depend on it, only when the arithmetic happens.
    // "Guaranteed to return before timeRemaining has elapsed" (Planner.h:42): the reference only polls now() between expansions
    // (AStarPlanner.cpp:61,136), which is enough when an expansion takes microseconds.  Here one device round trip costs 0.3-3 ms
    // and a doubling of the sample set more, so with the guard on (the default) the planner does not START a round trip or a sample
    // doubling that its own measurements of the previous ones say cannot end before the deadline.  It uses the value of the poll
    // the reference makes anyway — no extra now() call — and assumes the clock advances while the device works.  A clock that
    // does not (the counting clock of the oracle comparisons: t0 + calls * dt) must switch it off.
    bool deadlineGuard() const { return m_DeadlineGuard; }
    void setDeadlineGuard(bool on) { m_DeadlineGuard = on; }
    int speculation() const { return m_Speculation; }
    void setSpeculation(int n) { m_Speculation = n < 1 ? 1 : n; }
    // PlannerConfig.h:60-80,116-118: the search dump.  The device keeps no per-step poses, so the planner rebuilds the
    // "Trajectory:" samples of an edge on the host from the child's curve when (and only when) this is on.
    bool visualizations() const { return m_Visualizations && (m_Visualizer || m_VisualizationStream); }
    void setVisualizations(bool v) { m_Visualizations = v; }
    void setVisualizer(Visualizer::SharedPtr v) { m_Visualizer = std::move(v); }
    void setVisualizationStream(std::ostream* s) { m_VisualizationStream = s; }   // any stream instead of a file
    std::ostream& visualizationStream() const { return m_Visualizer ? m_Visualizer->stream() : *m_VisualizationStream; }

private:
    int m_BranchingFactor = 9;
    int m_Speculation = 64;      // measured on config 5 (round 4, 100 ms cycles): one context 8 -> 1 870 expansions per cycle, 16 -> 3 050, 32 -> 5 150;
                                 // two contexts on the one GPU (two round trips in flight) 32 -> 6 060, 64 -> 7 570, 128 -> 7 480
    double m_MaxSpeed = 2.5, m_SlowSpeed = 0.5, m_TurningRadius = 8, m_CoverageTurningRadius = 16;
    double m_TimeHorizon = 30, m_TimeMinimum = 5;
    double m_CollisionCheckingIncrement = 0.05;
    int m_InitialSamples = 100;
    bool m_UseBrownPaths = false;
    bool m_DeadlineGuard = true;
    bool m_Visualizations = false;
    Visualizer::SharedPtr m_Visualizer;
    std::ostream* m_VisualizationStream = nullptr;
    Map::SharedPtr m_Map;
    DynamicObstaclesManager::SharedPtr m_ObstaclesManager = std::make_shared<DynamicObstaclesManager>();
    std::ostream* m_Output;
    std::function<double()> m_NowFunction;
    double m_StartStateTime = 0;
};

}  // namespace ppamd
