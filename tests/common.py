"""Shared fixtures for the parity tests: scene set-ups used by both the oracle and the GPU library."""
import numpy as np

from raytrace_amd import abi, scenes


def cornell_setup(res=64, spp=16, variant="diffuse", max_depth=5, filter_kind="box", radius=None, xres=None, yres=None,
                  light_strategy=abi.LIGHTS_POWER, lens_radius=0.0, focal_distance=1e30, crop=(0.0, 1.0, 0.0, 1.0)):
    xres = res if xres is None else xres
    yres = res if yres is None else yres
    scene = scenes.cornell_box(variant)
    camera = scenes.cornell_camera(xres, yres, lens_radius, focal_distance)
    film = scenes.film_desc(xres, yres, filter_kind, radius, crop=crop)
    integ = scenes.integrator_desc(film, spp, max_depth=max_depth, light_strategy=light_strategy)
    return scene, camera, film, integ


def ulp_diff(a, b):
    """Distance in units in the last place between two float32 arrays (same-sign finite values)."""
    ia = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def rays_from_camera(rays6):
    r = np.zeros((rays6.shape[0], 7), dtype=np.float32)
    r[:, :6] = rays6
    r[:, 6] = np.inf
    return r


def mean_relative_error(img, ref, floor=1e-3):
    """Mean over pixels/channels of |img - ref| / max(ref, floor)."""
    return float(np.mean(np.abs(img - ref) / np.maximum(ref, floor)))
