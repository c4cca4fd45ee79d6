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

Camera arithmetic is float32 like the reference's glm code; cosf/sinf come from numpy here, so a replayed
path agrees with the reference's to the last ulps of those functions, not necessarily bit for bit.
"""
import numpy as np

F = np.float32
_SENSITIVITY = F(0.1)   # :303
_MOVE_SPEED = F(5.0)    # :330


def _normalize(v):
    v = np.asarray(v, dtype=np.float32)
    d = F(v[0] * v[0] + v[1] * v[1]) + F(v[2] * v[2])
    return v * F(F(1.0) / np.sqrt(d, dtype=np.float32))


def _cross(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], dtype=np.float32)


class FrameLoop:
    """One object per window: holds what App::run keeps in locals (:172-197)."""

    SLIDERS = {"samples": "set_samples_per_pixel", "aperture": "set_aperture", "focus_dist": "set_focus_distance",
               "fov": "set_fov", "max_depth": "set_max_depth"}

    def __init__(self, tracer):
        self.tracer = tracer
        self.sample_frame = 0
        self.cam_pos = np.array([13.0, 2.0, 3.0], dtype=np.float32)                                  # :180
        self.cam_dir = _normalize(np.array([0.0, 1.0, 0.0], dtype=np.float32) - self.cam_pos)        # :181
        self.yaw = F(np.degrees(np.arctan2(self.cam_dir[2], self.cam_dir[0], dtype=np.float32)))     # :182
        self.pitch = F(np.degrees(np.arcsin(self.cam_dir[1], dtype=np.float32)))                     # :183
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
            self.yaw = F(self.yaw + F(dx) * _SENSITIVITY)
            self.pitch = F(self.pitch + F(dy) * _SENSITIVITY)
            self.pitch = F(min(max(self.pitch, F(-89.0)), F(89.0)))
            yaw_rad = F(self.yaw * F(0.01745329251994329576923690768489))
            pitch_rad = F(self.pitch * F(0.01745329251994329576923690768489))
            new_dir = np.array([np.cos(pitch_rad, dtype=np.float32) * np.cos(yaw_rad, dtype=np.float32),
                                np.sin(pitch_rad, dtype=np.float32),
                                np.cos(pitch_rad, dtype=np.float32) * np.sin(yaw_rad, dtype=np.float32)], dtype=np.float32)
            self.cam_dir = _normalize(new_dir)
            cam_changed = True
        keys = set(keys)
        if keys:
            right = _normalize(_cross(self.cam_dir, np.array([0.0, 1.0, 0.0], dtype=np.float32)))
            up = np.array([0.0, 1.0, 0.0], dtype=np.float32)
            step = _MOVE_SPEED * F(dt)
            moves = {"W": self.cam_dir, "S": -self.cam_dir, "A": -right, "D": right, "SPACE": up, "SHIFT": -up}
            for key in ("W", "S", "A", "D", "SPACE", "SHIFT"):   # the order of :332-361
                if key in keys:
                    self.cam_pos = (self.cam_pos + moves[key] * step).astype(np.float32)
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
