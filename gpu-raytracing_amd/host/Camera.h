// Camera.h -- mirrors the reference's Camera.cuh / Input.cuh (Camera.cu:8-91).
#pragma once
#include "Common.h"

struct InputState {  // Input.cuh:4-14
    bool key_pressed_w = false, key_pressed_a = false, key_pressed_s = false, key_pressed_d = false;
    bool key_pressed_q = false, key_pressed_e = false, key_pressed_space = false, mouse_down = false;
    int prev_x = 0, prev_y = 0;
};

void UpdateCamera(Camera& camera);                                  // Camera.cu:8-29: yaw/pitch -> u, v, w
void UpdateCameraPosition(Camera& camera, InputState input);        // Camera.cu:31-45
void UpdateCameraLookDelta(Camera& camera, float dx, float dy);     // Camera.cu:47-51
void UpdateCameraZoom(Camera& camera, int dir);                     // Camera.cu:53-60
void InitialiseCamera(Camera& camera, AABB scene_aabb);             // Camera.cu:62-91
