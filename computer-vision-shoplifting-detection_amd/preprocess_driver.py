"""The reference's preprocessing loop (``/root/reference/preprocess.py:5-58``) over a pluggable frame source.

Frame decode stays on the host (north_star).  ``cv2`` is used when importable; otherwise a clip is a directory of frame
images decoded with Pillow, or a ``.npy`` stack ``[T,H,W,3]`` uint8 BGR next to the listed path (there is no video decoder
in the build image).
Semantics kept from the reference: the clip counter ``i`` counts EVERY list line, skipped ones included
(``preprocess.py:19-20``); only labels in ``videos_to_process`` are handled (``:10-13,27-29``); a clip that fails to
open is reported and skipped (``:33-35``); the frame number is ``CAP_PROP_POS_FRAMES`` read AFTER ``read()``
(1-based, ``:38-41``); the loop ends at the first failed read (``:43-44``).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional

import numpy as np

CAP_PROP_POS_FRAMES = 1
CAP_PROP_FRAME_COUNT = 7          # cv2.CAP_PROP_FRAME_COUNT: what the sweep balances its ranks by
VIDEOS_TO_PROCESS = ["Shoplifting", "Shopping"]


class NpyCapture:
    """cv2.VideoCapture look-alike over an ``.npy`` frame stack (or an in-memory array)."""

    def __init__(self, source):
        self._frames = None
        self._pos = 0
        if isinstance(source, np.ndarray):
            self._frames = source
        else:
            for cand in (source, source + ".npy", os.path.splitext(source)[0] + ".npy"):
                if cand.endswith(".npy") and os.path.exists(cand):
                    self._frames = np.load(cand, mmap_mode="r")
                    break

    def isOpened(self) -> bool:
        return self._frames is not None

    def read(self):
        if self._frames is None or self._pos >= len(self._frames):
            return False, None
        f = np.ascontiguousarray(self._frames[self._pos])
        self._pos += 1
        return True, f

    def get(self, prop):
        if prop == CAP_PROP_FRAME_COUNT:
            return float(len(self._frames)) if self._frames is not None else 0.0
        return float(self._pos) if prop == CAP_PROP_POS_FRAMES else 0.0

    def release(self):
        self._frames = None


class ImageDirCapture:
    """cv2.VideoCapture look-alike over a clip stored as a directory of frame images (``<clip>/0001.jpg`` ... -- how the
    UCF-Crime frame dumps are distributed), decoded on the host with Pillow: JPEG / PNG -> RGB -> BGR uint8, the layout
    ``cv2.VideoCapture.read()`` hands to ``/root/reference/preprocess.py:38``.  ``source`` is the directory, or the listed
    video path whose extension-less name is such a directory (``Shoplifting/Shoplifting001_x264.mp4`` ->
    ``Shoplifting/Shoplifting001_x264/``)."""

    EXTS = (".jpg", ".jpeg", ".png", ".bmp")

    def __init__(self, source: str):
        self._files: Optional[List[str]] = None
        self._pos = 0
        for cand in (source, os.path.splitext(source)[0]):
            if os.path.isdir(cand):
                names = sorted((n for n in os.listdir(cand) if n.lower().endswith(self.EXTS)), key=self._natural_key)
                if names:
                    self._files = [os.path.join(cand, n) for n in names]
                    break

    @staticmethod
    def _natural_key(name: str):
        """Frame order = the numbers in the file name, numerically (``2.jpg`` before ``10.jpg``, ``img_9`` before ``img_10``);
        text parts compare as text.  A plain string sort would feed 1, 10, 100, 2, ... to the tracker and label the CSV rows
        with the wrong frame numbers whenever the dump is not zero-padded."""
        import re
        stem = os.path.splitext(name)[0]
        return [(0, int(t), "") if t.isdigit() else (1, 0, t.lower()) for t in re.findall(r"\d+|\D+", stem)], name

    def isOpened(self) -> bool:
        return self._files is not None

    def read(self):
        if self._files is None or self._pos >= len(self._files):
            return False, None
        from PIL import Image
        with Image.open(self._files[self._pos]) as im:
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
        self._pos += 1
        return True, np.ascontiguousarray(rgb[..., ::-1])          # BGR, as cv2 decodes

    def get(self, prop):
        if prop == CAP_PROP_FRAME_COUNT:
            return float(len(self._files)) if self._files is not None else 0.0
        return float(self._pos) if prop == CAP_PROP_POS_FRAMES else 0.0

    def release(self):
        self._files = None


def open_capture(path: str):
    """cv2.VideoCapture when OpenCV is importable (mp4 decode as in the reference); otherwise a directory of frame images
    (Pillow) or an ``.npy`` frame stack next to the listed path."""
    try:
        import cv2  # noqa: WPS433  (optional; absent in the build image)
        return cv2.VideoCapture(path)
    except ImportError:
        cap = ImageDirCapture(path)
        return cap if cap.isOpened() else NpyCapture(path)


def run(people_tracker, list_path: str = "./dataset/Anomaly_Train.txt", dataset_root: str = "./dataset/",
        videos_to_process: Optional[List[str]] = None, capture: Callable = open_capture, log: Callable = print) -> int:
    """preprocess.main(); returns the number of frames handed to the tracker."""
    videos_to_process = VIDEOS_TO_PROCESS if videos_to_process is None else videos_to_process
    with open(list_path, "r") as f:
        videos = f.read().split("\n")
    i = 0
    frames_done = 0
    for video in videos:
        i += 1
        log(f"Processing video: {i}")
        label = video.split("/")[0]
        log(label)
        name = video.split("/")[1]          # a blank line raises IndexError, as in the reference
        if label not in videos_to_process:
            log(f"Skipping, {label}, {video}.")
            continue
        cap = capture(dataset_root + video)
        if not cap.isOpened():
            log(f"Failed to load video: {video}")
            continue
        while True:
            success, frame = cap.read()
            n = cap.get(CAP_PROP_POS_FRAMES)
            if not success:
                break
            people_tracker.save_to_dataset(frame, i, n, label, name)
            frames_done += 1
        cap.release()
    return frames_done
