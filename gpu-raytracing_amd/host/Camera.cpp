#include "Camera.h"

#include <algorithm>
#include <cmath>

static const float kPi = 3.14159265358979323846f;

void UpdateCamera(Camera& camera)
{
    // pitch is kept strictly inside (-pi/2, pi/2) so that w never becomes parallel to the up vector
    const float limit = kPi / 2;
    if (camera.pitch > limit) camera.pitch = limit - 0.0001f;
    else if (camera.pitch < -limit) camera.pitch = -limit + 0.0001f;
    const float pitch = camera.pitch, yaw = camera.yaw;
    camera.w = normalize(make_vec3(-sinf(yaw) * cosf(pitch), -sinf(pitch), cosf(yaw) * cosf(pitch)));
    camera.u = normalize(cross(camera.w, make_vec3(0, 1, 0)));
    camera.v = normalize(cross(camera.w, camera.u));
}

void UpdateCameraPosition(Camera& camera, InputState input)
{
    const float step = camera.scale * 0.25f;
    if (input.key_pressed_w) camera.position = camera.position + camera.w * step;
    if (input.key_pressed_s) camera.position = camera.position - camera.w * step;
    if (input.key_pressed_a) camera.position = camera.position - camera.u * step;
    if (input.key_pressed_d) camera.position = camera.position + camera.u * step;
    if (input.key_pressed_q || input.key_pressed_space) camera.position = camera.position - camera.v * step;
    if (input.key_pressed_e) camera.position = camera.position + camera.v * step;
}

void UpdateCameraLookDelta(Camera& camera, float dx, float dy)
{
    camera.yaw += dx * 0.01f;
    camera.pitch += dy * 0.01f;
}

void UpdateCameraZoom(Camera& camera, int dir)
{
    camera.position = dir > 0 ? camera.position + camera.w * camera.scale : camera.position - camera.w * camera.scale;
}

void InitialiseCamera(Camera& camera, AABB scene_aabb)
{
    // final state of the reference's InitialiseCamera: position = scene centre, yaw = pi/2,
    // pitch = 0, scale = length.z / 10, max_depth = 1.5 * longest extent; then the basis from UpdateCamera
    const vec3 centre = (scene_aabb.max + scene_aabb.min) * 0.5f;
    const vec3 length = scene_aabb.max - scene_aabb.min;
    camera.pitch = 0;
    camera.yaw = kPi / 2;
    camera.scale = length.z / 10.0f;
    camera.max_depth = std::max(std::max(length.x, length.y), length.z) * 1.5f;
    camera.position = centre;
    UpdateCamera(camera);
}
