// xw_device.h -- small device helpers shared by the XWorld2D kernels.
#pragma once
#include "xwb_common.h"

namespace xwb {

enum : int { STAGE_IDLE = 0, STAGE_NAV = 1, STAGE_TERMINAL = 2 };
enum : int { EV_NONE = 0, EV_CORRECT = 1, EV_WRONG = 2, EV_TIMEUP = 3 };

// tasks of the XWorld3DNav group (confs/navigation2d.json order) and the direction words of NavTargetDirection
enum : int { TASK_TARGET = 0, TASK_NEAR = 1, TASK_BETWEEN = 2, TASK_DIRECTION = 3, TASK_AVOID = 4,
              // the 2-D-native group "XWorldNav" of confs/walls.json (games/xworld/tasks/XWorldNav*.py, rule D14b)
              TASK2D_TARGET = 5, TASK2D_NEAR = 6, TASK2D_COLOR = 7, TASK2D_BETWEEN = 8 };
enum : int { DIR_FRONT = 1, DIR_BEHIND = 2, DIR_LEFT = 3, DIR_RIGHT = 4 };

// task_state word: target (name id for NavTarget, middle cell for NavTargetBetween, target cell for the
// 2-D-native tasks, else -1) | stage | event | task
__device__ __forceinline__ int pack_task(int target, int stage, int event, int kind) {
    return (target & 0xffff) | (stage << 16) | (event << 20) | (kind << 24);
}
__device__ __forceinline__ int task_target(int ts) { return (int)(int16_t)(ts & 0xffff); }
__device__ __forceinline__ int task_stage(int ts) { return (ts >> 16) & 0xf; }
__device__ __forceinline__ int task_event(int ts) { return (ts >> 20) & 0xf; }
__device__ __forceinline__ int task_kind(int ts) { return (ts >> 24) & 0xf; }

// cell code: bits 0..14 = palette icon + 1 (0 = empty), bit 15 = the goal belongs to the teacher's target set
constexpr uint32_t CELL_ICON_MASK = 0x7fffu, CELL_TARGET_BIT = 0x8000u;

// The idle stage of a 2-D-native task (XWorldNavTarget.py:22-33, XWorldNavColorTarget.py:8-20; Near / Between never
// find a target in this snapshot): Task::reset, then a uniformly chosen reachable [coloured] goal.
// `draw(n)` supplies the decisions (reset stream at reset time, stream 2 / block = num_steps at step time).
template <typename Draw>
__device__ __forceinline__ void idle_2d(int kind, uint32_t cand, const uint8_t *goal_cells, Draw draw,
                                        int &target, int &stage, int &tsteps) {
    tsteps = 0;
    target = -1;
    stage = STAGE_IDLE;
    if (kind != TASK2D_TARGET && kind != TASK2D_COLOR) return;
    const uint32_t m = kind == TASK2D_TARGET ? (cand & 0xffffu) : (cand >> 16);
    const int nc = __popc(m);
    if (nc == 0) return;
    int k = (int)draw((uint32_t)nc);                       // random.choice(targets)
    uint32_t mm = m;
    while (k-- > 0) mm &= mm - 1;                          // drop the k lowest set bits
    target = goal_cells[__ffs(mm) - 1];
    stage = STAGE_NAV;
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
