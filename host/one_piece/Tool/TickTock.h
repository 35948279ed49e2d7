// Tool/TickTock.h -- the wall-clock harness the drivers time their stages with (reference: src/Tool/TickTock.h:10-76;
// SURVEY 8f N3 "timing harness"): tool::Duration (TICK / TOCK / Elapsed in milliseconds) and tool::Timer, a set of named
// durations (TICK(name) / TOCK(name) / Elapsed / Log / LogAll / Reset).  Header-only like the reference's.
// NOTE for GPU timing: CubeHandler::IntegrateImage only enqueues; bracket it with Synchronize() (or any reader) when the
// elapsed time should include the kernels.
#pragma once
#include <chrono>
#include <iostream>
#include <map>
#include <string>

#include "Tool/ConsoleColor.h"

namespace one_piece {
namespace tool {

class Duration {
  public:
    void TICK() { start_time = std::chrono::steady_clock::now(); }
    void TOCK() { end_time = std::chrono::steady_clock::now(); }
    // milliseconds between the last TICK and the last TOCK
    float Elapsed() {
        const float ms = std::chrono::duration<float, std::milli>(end_time - start_time).count();
        if (ms < 0) std::cout << YELLOW << "[TICKTOCK]::[WARNING]::Elapsed time is less than 0." << RESET << std::endl;
        return ms;
    }
    std::chrono::steady_clock::time_point start_time;
    std::chrono::steady_clock::time_point end_time;
};

class Timer {
  public:
    void TICK(const std::string& name) { durations[name].TICK(); }
    void TOCK(const std::string& name) {
        std::map<std::string, Duration>::iterator it = durations.find(name);
        if (it == durations.end()) {
            std::cout << YELLOW << "[TICKTOCK]::[WARNING]::TOCK without TICK!" << RESET << std::endl;
            return;
        }
        it->second.TOCK();
    }
    float Elapsed(const std::string& name) { return durations[name].Elapsed(); }
    void Log(const std::string& name) { std::cout << name << "::" << durations[name].Elapsed() << "ms" << std::endl; }
    void LogAll() {
        for (std::map<std::string, Duration>::iterator it = durations.begin(); it != durations.end(); ++it) Log(it->first);
    }
    void Reset() { durations.clear(); }
    std::map<std::string, Duration> durations;
};

} // namespace tool
} // namespace one_piece
