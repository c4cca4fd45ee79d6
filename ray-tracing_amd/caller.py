"""Scripted replay of the reference's frame loop: the caller side of the traced path.

App::run (src/core/App.cpp:199-528) owns the frame counter and decides when it goes back to zero;
RayTracer only sees setter calls and render(frameIndex).  FrameLoop reproduces that contract so that
recorded camera paths and slider changes can be replayed through the C ABI:

  * mouse look: yaw += dx * 0.1, pitch += dy * 0.1 clamped to +-89 degrees, direction from yaw/pitch
    (:299-322);
  * WASD / space / shift fly camera at 5 units per second (:326-362);
  * any camera change -> setCamera(pos, dir) and sampleFrame = 0 BEFORE the frame is rendered (:364-368);
  * render(sampleFrame) (:403);
  * slider changes are applied AFTER the frame was recorded and also zero sampleFrame (:424-448), and
    the loop ends with ++sampleFrame (:527) -- so the first dispatch after a slider change runs with
    frame index 1 on a freshly cleared accumulator (the setter raised the reset flag);
  * a resize re-creates the images and restarts at frame 0 without rendering that iteration (:207-262).

Camera arithmetic is float32, operation for operation what the reference's GLM code evaluates
(normalize = v * (1 / sqrt(dot)), dot summed left to right, `camDir * moveSpeed * dt` from the left);
tests/golden/caller_camera.json, produced with the reference's vendored GLM, pins it bit for bit.  The
C library functions (cosf, sinf, atan2, asin) are taken as the correctly rounded binary32 results, which
is what the build container's libm returned for every value in that fixture.
"""
import math

import numpy as np

F = np.float32
_SENSITIVITY = F(0.1)   # :303
_MOVE_SPEED = F(5.0)    # :330
_RADIANS = F(0.01745329251994329576923690768489)   # glm::radians
_DEGREES = F(57.295779513082320876798154814105)    # glm::degrees


def _libm(fn, *args):
    """A float function of the C library: evaluated in double, rounded once to binary32."""
    return F(fn(*[float(a) for a in args]))


def _normalize(v):
    """glm::normalize: v * inversesqrt(dot(v, v)), inversesqrt = 1 / sqrt."""
    v = np.asarray(v, dtype=np.float32)
    d = F(F(F(v[0] * v[0]) + F(v[1] * v[1])) + F(v[2] * v[2]))
    inv = F(F(1.0) / np.sqrt(d, dtype=np.float32))
    return np.array([F(v[0] * inv), F(v[1] * inv), F(v[2] * inv)], dtype=np.float32)


def _cross(a, b):
    """glm::cross."""
    return np.array([F(F(a[1] * b[2]) - F(b[1] * a[2])), F(F(a[2] * b[0]) - F(b[2] * a[0])),
                     F(F(a[0] * b[1]) - F(b[0] * a[1]))], dtype=np.float32)


class FrameLoop:
    """One object per window: holds what App::run keeps in locals (:172-197)."""

    SLIDERS = {"samples": "set_samples_per_pixel", "aperture": "set_aperture", "focus_dist": "set_focus_distance",
               "fov": "set_fov", "max_depth": "set_max_depth"}

    def __init__(self, tracer):
        self.tracer = tracer
        self.sample_frame = 0
        self.cam_pos = np.array([13.0, 2.0, 3.0], dtype=np.float32)                                  # :180
        self.cam_dir = _normalize(np.array([0.0, 1.0, 0.0], dtype=np.float32) - self.cam_pos)        # :181
        self.yaw = F(_libm(math.atan2, self.cam_dir[2], self.cam_dir[0]) * _DEGREES)                # :182
        self.pitch = F(_libm(math.asin, self.cam_dir[1]) * _DEGREES)                                # :183
        self.rendered = []   # frame indices passed to render(), in order
        # App::run applies these right after create (:155-156)
        tracer.set_samples_per_pixel(4)
        tracer.set_aperture(0.05)

    def step(self, dt=1.0 / 60.0, mouse=(0.0, 0.0), keys=(), sliders=None, resize=None):
        """One iteration of the while loop.  mouse = (dx, dy) in pixels with dy already "lastY - cursorY";
        keys is a subset of {"W","A","S","D","SPACE","SHIFT"}; sliders = {name: value} changed this frame."""
        if resize is not None:
            # :207-262: recreate + tracer.resize + sampleFrame = 0, then `continue`
            self.tracer.resize(int(resize[0]), int(resize[1]))
            self.sample_frame = 0
            return None
        cam_changed = False
        dx, dy = float(mouse[0]), float(mouse[1])
        if dx != 0.0 or dy != 0.0:
            # :305-320
            self.yaw = F(self.yaw + F(F(dx) * _SENSITIVITY))
            self.pitch = F(self.pitch + F(F(dy) * _SENSITIVITY))
            self.pitch = F(min(max(self.pitch, F(-89.0)), F(89.0)))
            yaw_rad = F(self.yaw * _RADIANS)
            pitch_rad = F(self.pitch * _RADIANS)
            cos_pitch = _libm(math.cos, pitch_rad)
            new_dir = np.array([F(cos_pitch * _libm(math.cos, yaw_rad)), _libm(math.sin, pitch_rad),
                                F(cos_pitch * _libm(math.sin, yaw_rad))], dtype=np.float32)
            self.cam_dir = _normalize(new_dir)
            cam_changed = True
        keys = set(keys)
        if keys:
            # :326-362: camPos += camDir * moveSpeed * float(deltaTime), evaluated from the left
            right = _normalize(_cross(self.cam_dir, np.array([0.0, 1.0, 0.0], dtype=np.float32)))
            up = np.array([0.0, 1.0, 0.0], dtype=np.float32)
            fdt = F(dt)
            moves = (("W", self.cam_dir, 1), ("S", self.cam_dir, -1), ("A", right, -1), ("D", right, 1), ("SPACE", up, 1),
                     ("SHIFT", up, -1))
            for key, axis, sign in moves:   # the order of :332-361; W and S both use the direction as it is now
                if key in keys:
                    delta = np.array([F(F(c * _MOVE_SPEED) * fdt) for c in axis], dtype=np.float32)
                    self.cam_pos = (self.cam_pos + delta if sign > 0 else self.cam_pos - delta).astype(np.float32)
                    cam_changed = True
        if cam_changed:
            self.tracer.set_camera(self.cam_pos, self.cam_dir)   # :366
            self.sample_frame = 0
        frame = self.sample_frame
        self.tracer.render(frame)                                 # :403
        self.rendered.append(frame)
        for name, value in (sliders or {}).items():               # :424-448, after the frame was recorded
            getattr(self.tracer, self.SLIDERS[name])(value)
            self.sample_frame = 0
        self.sample_frame += 1                                    # :527
        return frame
