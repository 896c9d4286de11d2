// State — out-of-line members (interface and provenance: include/path_planner_amd/State.h).
// Every function evaluates the reference's formula in the reference's operation order, because the results are compared
// bit for bit with the reference object's (tests/golden/state_ops.json); how the code is laid out is this build's.
#include "path_planner_amd/State.h"

#include <cstdio>

namespace ppamd {

namespace {
std::string fiveNumbers(double a, double b, double c, double d, double e) {
    char buf[5 * 330];                               // std::to_string(double) prints "%f": at most 318 characters each
    std::snprintf(buf, sizeof(buf), "%f %f %f %f %f", a, b, c, d, e);
    return buf;
}
}  // namespace

State State::push(double timeInterval) const {
    // the vehicle keeps heading and speed; heading is measured from north, so x grows with its sine
    const double run = timeInterval * speed();
    return State(x() + std::sin(heading()) * run, y() + std::cos(heading()) * run, heading(), speed(), time() + timeInterval);
}

void State::move(double distance) {
    const double a = yaw();
    m_V[X] += std::cos(a) * distance;
    m_V[Y] += std::sin(a) * distance;
}

std::string State::toString() const { return fiveNumbers(x(), y(), heading() * 180 / M_PI, speed(), time()); }

std::string State::toStringRad() const { return fiveNumbers(x(), y(), heading(), speed(), time()); }

double State::headingTo(double x1, double y1) const { return flipAngle(std::atan2(y1 - y(), x1 - x())); }

double State::headingDifference(double otherHeading) const {
    const double turn = 2 * M_PI;
    const double raw = std::fmod(otherHeading - heading(), turn);
    return std::fmod(raw + 3 * M_PI, turn) - M_PI;
}

State State::interpolate(const State& other, double desiredTime) const {
    // linear in every component, the heading along the shorter way round; rates first, then rate * elapsed
    const double span = other.time() - time();
    const double elapsed = desiredTime - time();
    const double rate[4] = {(other.x() - x()) / span, (other.y() - y()) / span, headingDifference(other) / span,
                            (other.speed() - speed()) / span};
    State s(*this);
    s.x() += rate[X] * elapsed;
    s.y() += rate[Y] * elapsed;
    s.heading() = heading() + (rate[Heading] * elapsed);
    if (s.heading() >= 2 * M_PI) s.heading() -= 2 * M_PI;
    s.speed() += rate[Speed] * elapsed;
    s.time() = desiredTime;
    return s;
}

}  // namespace ppamd
