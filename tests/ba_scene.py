"""Seeded synthetic bundle-adjustment scenes: the generator lives in the package (bench.py --config ba uses it too)."""
from visualslam_android_amd.ba_scene import CAM, ba_scene, project  # noqa: F401
