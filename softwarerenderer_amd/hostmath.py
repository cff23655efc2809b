"""Host-side input generators: the System.Numerics.Matrix4x4 factories the C# host uses to
build the matrices it passes to RenderMesh (Renderer.cs:406-410, Camera.cs:12-17).  They only
MAKE inputs (float32, row-major M11..M44, row-vector convention); the hot path never calls them."""
from __future__ import annotations

import numpy as np

F = np.float32


def identity() -> np.ndarray:
    return np.eye(4, dtype=F)


def create_scale(s) -> np.ndarray:
    m = np.eye(4, dtype=F)
    m[0, 0] = m[1, 1] = m[2, 2] = F(s)
    return m


def create_translation(x, y, z) -> np.ndarray:
    m = np.eye(4, dtype=F)
    m[3, 0], m[3, 1], m[3, 2] = F(x), F(y), F(z)
    return m


def create_rotation_y(rad) -> np.ndarray:
    c, s = F(np.cos(F(rad))), F(np.sin(F(rad)))
    m = np.eye(4, dtype=F)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, -s, s, c
    return m


def create_rotation_x(rad) -> np.ndarray:
    c, s = F(np.cos(F(rad))), F(np.sin(F(rad)))
    m = np.eye(4, dtype=F)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, s, -s, c
    return m


def create_perspective_fov(fov_rad, aspect, near, far) -> np.ndarray:
    """Matrix4x4.CreatePerspectiveFieldOfView (right-handed, z in [0,1])."""
    y_scale = F(1.0) / F(np.tan(F(fov_rad) * F(0.5)))
    x_scale = y_scale / F(aspect)
    m = np.zeros((4, 4), dtype=F)
    m[0, 0] = x_scale
    m[1, 1] = y_scale
    neg_far_range = F(-1.0) if np.isposinf(far) else F(far) / (F(near) - F(far))
    m[2, 2] = neg_far_range
    m[2, 3] = F(-1.0)
    m[3, 2] = F(near) * neg_far_range
    return m


def create_look_at(eye, target, up) -> np.ndarray:
    """Matrix4x4.CreateLookAt (right-handed)."""
    eye, target, up = (np.asarray(a, dtype=F) for a in (eye, target, up))
    z = eye - target
    z = z / F(np.sqrt(F(z @ z)))
    x = np.cross(up, z).astype(F)
    x = x / F(np.sqrt(F(x @ x)))
    y = np.cross(z, x).astype(F)
    m = np.eye(4, dtype=F)
    m[0, 0], m[1, 0], m[2, 0] = x
    m[0, 1], m[1, 1], m[2, 1] = y
    m[0, 2], m[1, 2], m[2, 2] = z
    m[3, 0], m[3, 1], m[3, 2] = -F(x @ eye), -F(y @ eye), -F(z @ eye)
    return m


def multiply(a, b) -> np.ndarray:
    return (np.asarray(a, dtype=F) @ np.asarray(b, dtype=F)).astype(F)
