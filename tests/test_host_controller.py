"""The camera controller against values worked out by hand from the reference's own code
(src/main.cpp:63-96 constants, 250-273 state, 283-288 clock step, 402-496 processInput,
499-550 mouse, 554-565 updateCamera, 1424-1442 push constants, 2156-2174 showFPS).  No GPU."""
import ctypes as C
import math

import numpy as np
import pytest

from raytracer_amd import capi
from raytracer_amd.renderer import Controller, Scene, look_camera

F = np.float32


def vec(a):
    return np.array(list(a), dtype=np.float32)


def glm_yaw_pitch_roll(yaw, pitch, roll):
    """glm::yawPitchRoll (gtx/euler_angles), column-major, binary32."""
    ch, sh = F(math.cos(yaw)), F(math.sin(yaw))
    cp, sp = F(math.cos(pitch)), F(math.sin(pitch))
    cb, sb = F(math.cos(roll)), F(math.sin(roll))
    m = np.zeros((4, 4), np.float32)  # m[col][row]
    m[0] = [ch * cb + sh * sp * sb, sb * cp, -sh * cb + ch * sp * sb, 0]
    m[1] = [-ch * sb + sh * sp * cb, cb * cp, sb * sh + ch * sp * cb, 0]
    m[2] = [sh * cp, -sp, ch * cp, 0]
    m[3] = [0, 0, 0, 1]
    return m


def expected_axes(yaw_deg, pitch_deg, roll_deg):
    """updateCamera (main.cpp:554-565): front = R * (0,0,-1), up = R * (0,1,0), normalised."""
    r = glm_yaw_pitch_roll(-math.radians(yaw_deg), math.radians(pitch_deg), -math.radians(roll_deg))
    front, up = -r[2][:3], r[1][:3]
    return front / np.linalg.norm(front), up / np.linalg.norm(up)


def test_initial_state_is_the_pose_after_the_first_update_camera():
    st = Controller().state
    assert list(st.pos) == [2.75, 2.75, -6.0]                      # main.cpp:80
    assert (st.yaw, st.pitch, st.roll, st.fov) == (180.0, 0.0, 0.0, 50.0)  # main.cpp:83-86, 75
    np.testing.assert_allclose(vec(st.front), [0, 0, 1], atol=1e-6)
    np.testing.assert_allclose(vec(st.up), [0, 1, 0], atol=1e-6)
    assert st.mouse_captured == 1 and st.reset_accumulation == 1 and st.accumulation_on == 1 and st.frame_count == 0


def test_movement_keys_scale_with_delta_time_and_shift():
    c = Controller()
    assert c.begin_frame(0.5, capi.RT_KEY_W) == 0.5               # deltaTime = now - lastFrame (0 at start)
    np.testing.assert_allclose(vec(c.state.pos), [2.75, 2.75, -5.5], atol=1e-6)  # moveSpeed 1.0 along front
    c.begin_frame(1.0, capi.RT_KEY_S | capi.RT_KEY_LEFT_SHIFT)      # 0.5 s * 2.5
    np.testing.assert_allclose(vec(c.state.pos), [2.75, 2.75, -6.75], atol=1e-6)
    c.begin_frame(1.5, capi.RT_KEY_D)                               # cross(front, up) = (-1, 0, 0)
    np.testing.assert_allclose(vec(c.state.pos), [2.25, 2.75, -6.75], atol=1e-6)
    c.begin_frame(2.0, capi.RT_KEY_A | capi.RT_KEY_SPACE)
    np.testing.assert_allclose(vec(c.state.pos), [2.75, 3.25, -6.75], atol=1e-6)
    c.begin_frame(2.25, capi.RT_KEY_LEFT_CONTROL)
    np.testing.assert_allclose(vec(c.state.pos), [2.75, 3.0, -6.75], atol=1e-6)
    assert c.state.reset_accumulation == 1


def test_movement_sets_the_reset_flag_and_a_frame_clears_it():
    s = Scene(capi.RT_SCENE_EMPTY)
    c = Controller()
    _, pc = c.frame_inputs(s, 64, 36)
    assert pc.reset_frame_accumulation == 1 and c.state.reset_accumulation == 0   # first frame, then Post clears
    c.begin_frame(0.1, 0)
    _, pc = c.frame_inputs(s, 64, 36)
    assert pc.reset_frame_accumulation == 0
    c.begin_frame(0.2, capi.RT_KEY_W)
    _, pc = c.frame_inputs(s, 64, 36)
    assert pc.reset_frame_accumulation == 1
    _, pc = c.frame_inputs(s, 64, 36)
    assert pc.reset_frame_accumulation == 0


def test_mouse_look_sensitivity_clamp_and_wrap():
    c = Controller()
    c.mouse_move(400.0, 300.0)                  # first event only latches the position
    assert (c.state.yaw, c.state.pitch) == (180.0, 0.0)
    c.mouse_move(410.0, 295.0)                  # 10 px right, 5 px up at sensitivity 0.1 * 50/50
    st = c.state
    assert st.yaw == pytest.approx(181.0, abs=1e-4) and st.pitch == pytest.approx(0.5, abs=1e-4)
    f, u = expected_axes(st.yaw, st.pitch, st.roll)
    np.testing.assert_allclose(vec(st.front), f, atol=2e-6)
    np.testing.assert_allclose(vec(st.up), u, atol=2e-6)
    c.mouse_move(410.0, -2000.0)                # pitch clamps at 89
    assert c.state.pitch == 89.0
    c.mouse_move(410.0 + 1795.0, -2000.0)       # yaw 181 + 179.5 = 360.5 -> wraps
    assert c.state.yaw == pytest.approx(0.5, abs=1e-3)
    c.mouse_move(410.0 + 1795.0 - 20.0, -2000.0)  # yaw 0.5 - 2 < 0 -> + 360
    assert c.state.yaw == pytest.approx(358.5, abs=1e-3)


def test_mouse_look_is_rotated_by_the_roll():
    c = Controller()
    c.begin_frame(0.5, capi.RT_KEY_E)           # roll += 80 * 0.5 = 40 degrees
    assert c.state.roll == pytest.approx(40.0, abs=1e-5)
    c.mouse_move(0.0, 0.0)
    c.mouse_move(10.0, 0.0)                     # xoffset 1 degree, rollRad = -radians(40)
    st = c.state
    assert st.yaw == pytest.approx(180.0 + math.cos(math.radians(-40.0)), abs=1e-4)
    assert st.pitch == pytest.approx(math.sin(math.radians(-40.0)), abs=1e-4)
    f, u = expected_axes(st.yaw, st.pitch, st.roll)
    np.testing.assert_allclose(vec(st.front), f, atol=2e-6)
    np.testing.assert_allclose(vec(st.up), u, atol=2e-6)


def test_roll_keys_wrap_and_reset_keys():
    c = Controller()
    c.begin_frame(0.1, capi.RT_KEY_Q)           # roll = -8 -> + 360
    assert c.state.roll == pytest.approx(352.0, abs=1e-4)
    c.begin_frame(0.3, capi.RT_KEY_E | capi.RT_KEY_LEFT_SHIFT)  # + 80 * 0.2 * 2.5 = 40 -> 392 -> wraps to 32
    assert c.state.roll == pytest.approx(32.0, abs=1e-3)
    c.begin_frame(0.4, capi.RT_KEY_F)
    assert c.state.roll == 0.0
    c.scroll(-30.0)
    c.mouse_move(0, 0); c.mouse_move(100, 50)
    c.begin_frame(0.5, capi.RT_KEY_W | capi.RT_KEY_X)
    assert c.state.accumulation_on == 0 and c.state.fov == 80.0
    c.begin_frame(0.6, capi.RT_KEY_R)
    st = c.state
    assert list(st.pos) == [2.75, 2.75, -6.0] and (st.yaw, st.pitch, st.roll, st.fov) == (180.0, 0.0, 0.0, 50.0)
    assert st.accumulation_on == 1 and st.reset_accumulation == 1
    np.testing.assert_allclose(vec(st.front), [0, 0, 1], atol=1e-6)


def test_scroll_changes_fov_within_limits():
    c = Controller()
    c.scroll(3.0)
    assert c.state.fov == 47.0 and c.state.reset_accumulation == 1
    c.scroll(100.0)
    assert c.state.fov == 1.0
    c.scroll(-1000.0)
    assert c.state.fov == 160.0


def test_accumulation_toggle_is_debounced_and_forces_reset_while_off():
    s = Scene(capi.RT_SCENE_EMPTY)
    c = Controller()
    c.frame_inputs(s, 32, 32)
    c.begin_frame(0.1, capi.RT_KEY_X)
    c.begin_frame(0.2, capi.RT_KEY_X)           # still held: one toggle only
    assert c.state.accumulation_on == 0
    for _ in range(2):                          # every frame resets while accumulation is off (main.cpp:1430-1434)
        assert c.frame_inputs(s, 32, 32)[1].reset_frame_accumulation == 1
    c.begin_frame(0.3, 0)
    c.begin_frame(0.4, capi.RT_KEY_X)
    assert c.state.accumulation_on == 1
    assert c.frame_inputs(s, 32, 32)[1].reset_frame_accumulation == 0


def test_released_mouse_ignores_input_until_a_click_recaptures_it():
    c = Controller()
    c.mouse_move(0, 0)
    c.begin_frame(0.1, capi.RT_KEY_RELEASE_MOUSE)
    assert c.state.mouse_captured == 0
    c.begin_frame(0.6, capi.RT_KEY_W)
    c.mouse_move(500, 500)
    st = c.state
    assert list(st.pos) == [2.75, 2.75, -6.0] and st.yaw == 180.0
    c.mouse_button(True)
    c.mouse_move(900, 900)                      # first event after the capture: no jump
    assert c.state.yaw == 180.0 and c.state.mouse_captured == 1
    c.mouse_move(910, 900)
    assert c.state.yaw == pytest.approx(181.0, abs=1e-4)


def test_frame_inputs_carry_time_frame_count_and_the_camera_block():
    s = Scene(capi.RT_SCENE_CORNELL_STAR)
    c = Controller()
    c.begin_frame(0.25, 0)
    c.scroll(-10.0)
    cam, pc = c.frame_inputs(s, 1920, 1080)
    assert pc.time == 0.25 and pc.frameCount == 0
    ref = s.push_constants(0.25, 0, True)
    assert bytes(pc) == bytes(ref)
    st = c.state
    assert bytes(cam) == bytes(look_camera(list(st.pos), list(st.front), list(st.up), 60.0, 1920, 1080))
    # main.cpp:1494: tan(glm::radians(fov) / 2.0) in double on the binary32 radians, then narrowed
    rad = np.float32(60.0) * np.float32(0.01745329251994329576923690768489)
    assert cam.tanHalfFOV == np.float32(math.tan(float(rad) / 2.0))


def test_frame_counter_restarts_every_second_like_the_fps_window():
    c = Controller()
    counts = []
    for t in (0.0, 0.4, 0.8, 1.2, 1.6, 2.0, 2.3):
        c.end_frame(t)
        counts.append(c.state.frame_count)
    # first call opens the window at t=0; 1.2 - 0 >= 1 restarts it at 1.2; 2.3 - 1.2 >= 1 again
    assert counts == [1, 2, 3, 0, 1, 2, 0]


def test_plane_mode_flies_forward_without_keys():
    c = Controller()
    c.set_plane_mode(True)
    c.begin_frame(2.0, capi.RT_KEY_S)           # movement keys are not read in plane mode
    np.testing.assert_allclose(vec(c.state.pos), [2.75, 2.75, -4.0], atol=1e-6)


def test_close_key_and_null_arguments():
    c = Controller()
    c.begin_frame(0.1, capi.RT_KEY_K)
    assert c.state.should_close == 1
    lib = capi.load()
    assert lib.rt_controller_create(None) == capi.RT_ERR_INVALID_ARGUMENT
    assert lib.rt_controller_get(None, None) == capi.RT_ERR_INVALID_ARGUMENT
    assert lib.rt_controller_scroll(None, 1.0) == capi.RT_ERR_INVALID_ARGUMENT
