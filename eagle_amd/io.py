"""Frame ingest with the reference's sampling semantics (eagle/utils/io.py:5-27).

``read_video`` there decodes with ``cv2.VideoCapture`` and keeps frame k iff ``k % (native_fps // fps) == 0`` (float floor
division of the container's frame rate, so 29.97 -> 24 keeps every frame and 50 -> 24 every second one; a target above the native
rate raises ZeroDivisionError, as the reference does).  Video decode itself is out of scope here (no codec library in this image):
the source is an already decoded clip — a ``.npy`` file / array of uint8 [n,h,w,3] BGR frames — plus its native frame rate."""
import os

import numpy as np


def sample_indices(n_frames: int, native_fps: float, fps: int = 24):
    """Indices ``read_video`` keeps out of ``n_frames`` decoded frames (io.py:18-25)."""
    skip = native_fps // fps
    return [k for k in range(n_frames) if k % skip == 0]


def read_clip(path_or_frames, native_fps: float, fps: int = 24):
    """-> (frames [m,h,w,3] uint8, fps): the drop-in for ``read_video(path, fps)`` on a decoded clip."""
    if isinstance(path_or_frames, (str, os.PathLike)):
        if not os.path.exists(path_or_frames):
            raise FileNotFoundError(f"File not found: {path_or_frames}")
        frames = np.load(path_or_frames, mmap_mode="r")
    else:
        frames = np.asarray(path_or_frames)
    idx = sample_indices(len(frames), native_fps, fps)
    return np.ascontiguousarray(frames[idx], np.uint8), fps
