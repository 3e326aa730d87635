// controller.cpp — the reference application's camera controller and per-frame seeding
// policy as a headless state machine behind the C ABI.
//
// Restates, on the reference's own variables and constants:
//   * the state and its initial values            src/main.cpp:63-96, 250-273
//   * one main-loop iteration's time step          src/main.cpp:283-285
//   * keyboard handling (processInput)             src/main.cpp:402-496
//   * mouse capture / look / scroll                src/main.cpp:499-550
//   * updateCamera                                 src/main.cpp:554-565
//   * camera block + push constants of a frame     src/main.cpp:1331-1340, 1424-1442, 1466-1497
//   * the FPS-window frame counter that seeds RNG  src/main.cpp:2156-2174
// Window-system input is replaced by explicit calls: a key mask and a clock value per
// iteration, cursor positions, scroll offsets.  Arithmetic is binary32 in the order the
// reference writes it (its mouse offsets are formed in double and narrowed, as there).
//
// Deviations, on purpose:
//   * `lastFrame` is uninitialised in the reference (main.cpp:250), so its first delta time
//     is indeterminate; here it starts at 0.
//   * Until the first updateCamera() the reference uses the POINT (2.75,2.75,-5) as the view
//     direction (main.cpp:81, 260); the controller starts from the state after that first
//     updateCamera() instead (yaw 180, pitch 0, roll 0).
#include "../../../include/rt_abi.h"
#include "rtm.h"

#include <cmath>
#include <new>

namespace {

// main.cpp:63-96
const rtm::vec4 kWorldFront = {0.0f, 0.0f, -1.0f, 0.0f};
const rtm::vec4 kWorldUp = {0.0f, 1.0f, 0.0f, 0.0f};
const float kMoveSpeed = 1.0f, kRollSpeed = 80.0f, kShiftMult = 2.5f, kFovIncrease = 1.0f, kSensitivity = 0.1f;
const float kFovInitial = 50.0f;
const rtm::vec3 kPosInitial = {2.75f, 2.75f, -6.0f};
const rtm::vec3 kFrontPointInitial = {2.75f, 2.75f, -5.0f};
const float kRollInitial = 0.0f;
const bool kAccumulationInitial = true;

float degrees(float rad) { return rad * 57.295779513082320876798154814105f; }  // glm::degrees

}  // namespace

struct RtController {
    // main.cpp:250-273
    float last_frame = 0.0f;
    int frame_count = 0;
    bool mouse_captured = true, first_mouse = true;
    float fov = kFovInitial;
    rtm::vec3 pos = kPosInitial, front = kFrontPointInitial, up = {0.0f, 1.0f, 0.0f};
    float yaw = 0.0f, pitch = 0.0f, roll = kRollInitial;
    float yaw_initial = 0.0f, pitch_initial = 0.0f;
    bool reset_accumulation = true, accumulation_on = kAccumulationInitial;
    bool plane_mode = false;  // main.cpp:90
    // function-local statics of the reference
    float last_x = 0.0f, last_y = 0.0f;  // mouse_callback, main.cpp:510-511
    bool x_bounce = false;               // processInput, main.cpp:475
    bool fps_started = false;            // showFPS, main.cpp:2157
    double fps_last_time = 0.0;
    bool should_close = false;

    // main.cpp:554-565
    void update_camera() {
        reset_accumulation = true;
        const rtm::mat4 rotation = rtm::yawPitchRoll(-rtm::radians(yaw), rtm::radians(pitch), -rtm::radians(roll));
        front = rtm::normalize(rtm::xyz(rtm::mul(rotation, kWorldFront)));
        up = rtm::normalize(rtm::xyz(rtm::mul(rotation, kWorldUp)));
    }
};

extern "C" {

int rt_controller_create(RtController** out) {
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    RtController* c = new (std::nothrow) RtController();
    if (!c) return RT_ERR_TOO_LARGE;
    // main.cpp:83-85
    const rtm::vec3 f = rtm::normalize(kFrontPointInitial - kPosInitial);
    c->yaw_initial = degrees(std::atan2(f.x, -f.z));
    c->pitch_initial = degrees(std::asin(f.y));
    c->yaw = c->yaw_initial;
    c->pitch = c->pitch_initial;
    c->update_camera();
    *out = c;
    return RT_OK;
}

int rt_controller_destroy(RtController* c) {
    delete c;
    return RT_OK;
}

int rt_controller_set_plane_mode(RtController* c, int on) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    c->plane_mode = on != 0;
    return RT_OK;
}

int rt_controller_get(const RtController* c, RtControllerState* out) {
    if (!c || !out) return RT_ERR_INVALID_ARGUMENT;
    out->pos[0] = c->pos.x; out->pos[1] = c->pos.y; out->pos[2] = c->pos.z;
    out->front[0] = c->front.x; out->front[1] = c->front.y; out->front[2] = c->front.z;
    out->up[0] = c->up.x; out->up[1] = c->up.y; out->up[2] = c->up.z;
    out->yaw = c->yaw; out->pitch = c->pitch; out->roll = c->roll; out->fov = c->fov;
    out->last_frame = c->last_frame;
    out->frame_count = c->frame_count;
    out->mouse_captured = c->mouse_captured;
    out->reset_accumulation = c->reset_accumulation;
    out->accumulation_on = c->accumulation_on;
    out->should_close = c->should_close;
    return RT_OK;
}

// main.cpp:283-288: the clock step of one main-loop iteration, then processInput (402-496).
int rt_controller_begin_frame(RtController* c, double now_seconds, uint32_t keys, float* delta_time_out) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    const float current = (float)now_seconds;
    const float dt = current - c->last_frame;
    c->last_frame = current;
    if (delta_time_out) *delta_time_out = dt;

    if (!c->mouse_captured) return RT_OK;
    float move = kMoveSpeed * dt;
    float rolld = kRollSpeed * dt;
    bool rotate = false;
    if (keys & RT_KEY_LEFT_SHIFT) {
        move *= kShiftMult;
        rolld *= kShiftMult;
    }
    if (c->plane_mode) {
        c->pos = c->pos + c->front * move;
    } else {
        if (keys & RT_KEY_W) { c->pos = c->pos + rtm::normalize(c->front) * move; c->reset_accumulation = true; }
        if (keys & RT_KEY_S) { c->pos = c->pos - rtm::normalize(c->front) * move; c->reset_accumulation = true; }
        if (keys & RT_KEY_A) { c->pos = c->pos - rtm::normalize(rtm::cross(c->front, c->up)) * move; c->reset_accumulation = true; }
        if (keys & RT_KEY_D) { c->pos = c->pos + rtm::normalize(rtm::cross(c->front, c->up)) * move; c->reset_accumulation = true; }
        if (keys & RT_KEY_SPACE) { c->pos = c->pos + c->up * move; c->reset_accumulation = true; }
        if (keys & RT_KEY_LEFT_CONTROL) { c->pos = c->pos - c->up * move; c->reset_accumulation = true; }
    }
    if (keys & RT_KEY_K) c->should_close = true;
    if (keys & RT_KEY_Q) { c->roll -= rolld; rotate = true; }
    if (keys & RT_KEY_E) { c->roll += rolld; rotate = true; }
    if (keys & RT_KEY_F) { c->roll = kRollInitial; rotate = true; }
    if (keys & RT_KEY_R) {
        c->roll = kRollInitial;
        c->pitch = c->pitch_initial;
        c->yaw = c->yaw_initial;
        c->pos = kPosInitial;
        c->fov = kFovInitial;
        rotate = true;
        c->accumulation_on = kAccumulationInitial;
    }
    if ((keys & RT_KEY_X) && !c->x_bounce) {
        c->accumulation_on = !c->accumulation_on;
        c->x_bounce = true;
    }
    if (!(keys & RT_KEY_X) && c->x_bounce) c->x_bounce = false;
    if (rotate) {
        if (c->roll > 360.0) c->roll -= 360.0;
        if (c->roll < 0.0) c->roll += 360.0;
        c->update_camera();
    }
    if (keys & RT_KEY_RELEASE_MOUSE) c->mouse_captured = false;
    return RT_OK;
}

// main.cpp:499-505
int rt_controller_mouse_button(RtController* c, int left_pressed) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (left_pressed && !c->mouse_captured) {
        c->mouse_captured = true;
        c->first_mouse = true;
    }
    return RT_OK;
}

// main.cpp:508-543
int rt_controller_mouse_move(RtController* c, double xpos, double ypos) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->mouse_captured) return RT_OK;
    if (c->first_mouse) {
        c->last_x = (float)xpos;
        c->last_y = (float)ypos;
        c->first_mouse = false;
    }
    float xoffset = (float)(xpos - (double)c->last_x);
    float yoffset = (float)((double)c->last_y - ypos);
    c->last_x = (float)xpos;
    c->last_y = (float)ypos;
    const float sensitivity = kSensitivity * c->fov / kFovInitial;
    xoffset *= sensitivity;
    yoffset *= sensitivity;
    const float roll_rad = -rtm::radians(c->roll);
    c->yaw += xoffset * std::cos(roll_rad) - yoffset * std::sin(roll_rad);
    c->pitch += xoffset * std::sin(roll_rad) + yoffset * std::cos(roll_rad);
    if (c->pitch > 89.0f) c->pitch = 89.0f;
    if (c->pitch < -89.0f) c->pitch = -89.0f;
    if (c->yaw > 360.0) c->yaw -= 360.0;
    if (c->yaw < 0.0) c->yaw += 360.0;
    c->update_camera();
    return RT_OK;
}

// main.cpp:545-550
int rt_controller_scroll(RtController* c, double yoffset) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    c->fov -= static_cast<float>(yoffset) * kFovIncrease;
    if (c->fov < 1.0) c->fov = 1.0;
    if (c->fov > 160.0) c->fov = 160.0;
    c->reset_accumulation = true;
    return RT_OK;
}

// drawFrame's per-frame inputs (main.cpp:1331-1340): updatePushConstantsPre (1424-1438),
// updateUniformBuffer (1466-1497), then updatePushConstantsPost (1440-1442).
int rt_controller_frame_inputs(RtController* c, const RtScene* s, int32_t width, int32_t height, RtCamera* cam,
                               RtPushConstants* pc) {
    if (!c || !s || !cam || !pc) return RT_ERR_INVALID_ARGUMENT;
    int rc = rt_push_constants(pc, s, c->last_frame, (uint32_t)c->frame_count, c->accumulation_on ? c->reset_accumulation : 1);
    if (rc != RT_OK) return rc;
    const float p[3] = {c->pos.x, c->pos.y, c->pos.z}, f[3] = {c->front.x, c->front.y, c->front.z}, u[3] = {c->up.x, c->up.y, c->up.z};
    rc = rt_camera_look(cam, p, f, u, c->fov, width, height);
    if (rc != RT_OK) return rc;
    c->reset_accumulation = false;
    return RT_OK;
}

// showFPS (main.cpp:2156-2174): the counter that becomes pushConstants.frameCount restarts
// whenever a second has passed since the window began.
int rt_controller_end_frame(RtController* c, double now_seconds) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->fps_started) {
        c->fps_started = true;
        c->fps_last_time = now_seconds;
    }
    const double delta = now_seconds - c->fps_last_time;
    c->frame_count++;
    if (delta >= 1.0) {
        c->frame_count = 0;
        c->fps_last_time = now_seconds;
    }
    return RT_OK;
}

}  // extern "C"
