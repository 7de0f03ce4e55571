// xw_device.h -- small device helpers shared by the XWorld2D kernels.
#pragma once
#include "xwb_common.h"

namespace xwb {

enum : int { STAGE_IDLE = 0, STAGE_NAV = 1, STAGE_TERMINAL = 2 };
enum : int { EV_NONE = 0, EV_CORRECT = 1, EV_WRONG = 2, EV_TIMEUP = 3 };

__device__ __forceinline__ int pack_task(int target, int stage, int event) {
    return (target & 0xffff) | (stage << 16) | (event << 24);
}

__device__ __forceinline__ int done_code(const XwParams &p, int num_steps, int event) {
    // AgentSpecificSimulator::game_over = GameSimulator::game_over | XWorldSimulator::game_over
    int code = (p.max_steps > 0 && num_steps >= p.max_steps) ? MAX_STEP : ALIVE;
    if (p.task_mode == 0) {       // lang_acquisition, xworld_simulator.cpp:166-177
        if (event == EV_CORRECT) code |= SUCCESS;
        else if (event == EV_WRONG) code |= DEAD;
        else if (event == EV_TIMEUP) code |= MAX_STEP;
    }
    return code;
}


}  // namespace xwb
