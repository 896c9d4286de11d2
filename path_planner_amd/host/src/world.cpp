// Map / GridWorldMap / BinaryDynamicObstaclesManager — host mirrors, plus the snapshots the device consumes
#include "path_planner_amd/World.h"

#include <algorithm>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <tuple>

namespace ppamd {

void Map::rasterize(std::vector<uint8_t>& cells, int& rows, int& cols, double& res) const {
    const double* e = extremes();
    res = resolution();
    if (!(res > 0) || e[1] >= DBL_MAX || e[3] >= DBL_MAX || e[0] != 0 || e[2] != 0) {
        cells.clear(); rows = cols = 0; res = 0;   // unbounded / resolution-less map: nothing is ever blocked on the device either
        return;
    }
    cols = (int)std::ceil(e[1] / res);
    rows = (int)std::ceil(e[3] / res);
    cells.assign((size_t)rows * cols, 0);
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++)
            cells[(size_t)r * cols + c] = isBlocked((c + 0.5) * res, (r + 0.5) * res) ? 1 : 0;
}

GridWorldMap::GridWorldMap(const std::string& path) {
    std::ifstream infile(path);
    if (!infile) throw std::runtime_error("GridWorldMap: cannot open " + path);
    load(infile);
}

std::shared_ptr<GridWorldMap> GridWorldMap::fromText(const std::string& text) {
    std::shared_ptr<GridWorldMap> m(new GridWorldMap());
    std::istringstream in(text);
    m->load(in);
    return m;
}

void GridWorldMap::load(std::istream& infile) {   // GridWorldMap.cpp:10-82
    std::string line;
    std::vector<std::string> lines;
    std::getline(infile, line);
    std::istringstream s(line);
    s >> m_Resolution;
    int cols = -1, rows = 0;
    while (std::getline(infile, line)) {
        if (cols == -1) cols = (int)line.length();
        else if ((int)line.length() < cols) cols = (int)line.length();
        rows++;
        lines.push_back(line);
    }
    if (rows == 0 || cols <= 0) throw std::runtime_error("GridWorldMap: empty map");
    std::reverse(lines.begin(), lines.end());
    m_Blocked = std::vector<std::vector<bool>>(rows, std::vector<bool>(cols, false));
    m_Extremes[0] = 0; m_Extremes[1] = m_Blocked.front().size() * m_Resolution;
    m_Extremes[2] = 0; m_Extremes[3] = m_Blocked.size() * m_Resolution;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++)
            if (lines[y][x] == '#') m_Blocked[y][x] = true;
}

bool GridWorldMap::isBlocked(double x, double y) const {
    if (x < 0 || x / m_Resolution >= m_Blocked.front().size()) return true;
    if (y < 0 || y / m_Resolution >= m_Blocked.size()) return true;
    return m_Blocked.at(y / m_Resolution).at(x / m_Resolution);
}

void GridWorldMap::rasterize(std::vector<uint8_t>& cells, int& rows, int& cols, double& res) const {
    rows = (int)m_Blocked.size();
    cols = (int)m_Blocked.front().size();
    res = m_Resolution;
    cells.assign((size_t)rows * cols, 0);
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++)
            if (m_Blocked[r][c]) cells[(size_t)r * cols + c] = 1;
}

void BinaryDynamicObstaclesManager::update(uint32_t mmsi, double x, double y, double heading, double speed, double time,
                                           double width, double length) {
    if (m_Ignored.find(mmsi) != m_Ignored.end()) return;
    auto result = m_Obstacles.emplace(std::piecewise_construct, std::forward_as_tuple(mmsi),
                                      std::forward_as_tuple(x, y, heading, speed, time, width, length));
    if (!result.second) result.first->second = Obstacle(x, y, heading, speed, time, width, length);
}

double BinaryDynamicObstaclesManager::collisionExists(double x, double y, double time, bool strict) const {
    double sum = 0;
    for (auto o : m_Obstacles) {
        auto& obstacle = o.second;
        if (strict) { obstacle.Width += 2; obstacle.Length += 2; }
        obstacle.project(time);
        double translatedX = x - obstacle.X;
        double translatedY = y - obstacle.Y;
        double rotatedX = translatedX * std::cos(obstacle.Yaw) - translatedY * std::sin(obstacle.Yaw);
        double rotatedY = translatedX * std::sin(obstacle.Yaw) + translatedY * std::cos(obstacle.Yaw);
        if (std::fabs(rotatedX) < obstacle.Length / 2 && std::fabs(rotatedY) < obstacle.Width / 2) sum++;
    }
    return sum;
}

void BinaryDynamicObstaclesManager::deviceRows(std::vector<double>& rows7) const {
    rows7.clear();
    for (const auto& kv : m_Obstacles) {
        const Obstacle& o = kv.second;
        rows7.insert(rows7.end(), {o.X, o.Y, o.Heading, o.Speed, o.Time, o.Width, o.Length});
    }
}

}  // namespace ppamd
