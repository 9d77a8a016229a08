"""Spherical demo camera path (the role of python/jnerf/dataset/camera_path.py:4-28): 3x4 nerf-convention poses on a circle around the scene."""
import numpy as np


def pose_spherical(theta_deg, phi_deg, radius):
    t = np.eye(4, dtype=np.float32); t[2, 3] = radius
    p, th = np.deg2rad(phi_deg), np.deg2rad(theta_deg)
    rot_phi = np.array([[1, 0, 0, 0], [0, np.cos(p), -np.sin(p), 0], [0, np.sin(p), np.cos(p), 0], [0, 0, 0, 1]], np.float32)
    rot_theta = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], np.float32)
    c2w = rot_theta @ rot_phi @ t
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32) @ c2w
    return c2w[:3, :4]


def path_spherical(nframe=80, phi=-30.0, radius=4.0):
    return [pose_spherical(angle, phi, radius) for angle in np.linspace(-180, 180, nframe + 1)[:-1]]
