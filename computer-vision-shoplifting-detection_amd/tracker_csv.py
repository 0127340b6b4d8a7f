"""``Tracker`` / ``BBox``: the reference's per-frame caller and CSV row format.

Mirrors ``/root/reference/model.py:14-81`` (class ``Tracker``) and ``/root/reference/dataset.py:46-60``
(dataclass ``BBox``) with the Ultralytics model replaced by the MI355X engine (``cvsd_amd.YOLO``).  Behaviour kept:

* ``get_boxes(frame)`` = ``model.track(frame, persist=True, show=False, classes=[0], verbose=False)[0].boxes``
  (``model.py:36-40``);
* a frame whose boxes carry no track ids is dropped silently (``model.py:45-46``);
* rows go to ``dataset/ucf-crime_dataset.csv`` when the label is one of the 13 anomaly names, else to
  ``dataset/ucf-crime_dataset-normal.csv`` (``model.py:20-34,48-53``), opened in append mode, no header
  (``model.py:79-81``);
* the columns called ``left``/``top`` hold the NORMALISED BOX CENTRE ``xywhn[0][0:2]`` (``model.py:61-62``);
* a row is ``dataclasses.astuple(BBox)`` written by ``csv.writer`` (excel dialect, ``\\r\\n``), which is what
  ``dataclass_csv.DataclassWriter.write(skip_header=True)`` (dataclass-csv==1.4.0) does.
"""
from __future__ import annotations

import csv
import dataclasses
import os
from dataclasses import dataclass
from typing import Iterable, List, Optional

ANOMALIES = ["Abuse", "Arrest", "Arson", "Assault", "Burglary", "Explosion", "Fighting", "RoadAccidents", "Robbery",
             "Shooting", "Shoplifting", "Stealing", "Vandalism"]


@dataclass
class BBox:
    """One CSV row (``/root/reference/dataset.py:46-60``)."""
    clip: int
    name: str
    frame: int
    person: float
    left: float
    top: float
    width: float
    height: float
    is_anomaly: bool
    anomaly: str


def write_rows(path: str, rows: Iterable[BBox]) -> None:
    """DataclassWriter(f, rows, BBox).write(skip_header=True) on a file opened with mode 'a', newline=''."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "a", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            if not isinstance(r, BBox):
                raise TypeError("The item is not an instance of BBox")
            w.writerow(dataclasses.astuple(r))


class Tracker:
    def __init__(self, model_path: str = "./models/yolov5mu.pt", model=None, out_dir: str = "dataset"):
        if model is None:
            from .engine import YOLO
            model = YOLO(model_path)
        self.model = model
        self.anomalies = list(ANOMALIES)
        self.out_dir = out_dir

    def get_boxes(self, frame):
        results = self.model.track(frame, persist=True, show=False, classes=[0], verbose=False)
        return results[0].boxes

    def rows_for(self, boxes, i: int, n, label: str, name: str) -> List[BBox]:
        is_anomaly = label in self.anomalies
        return [BBox(clip=i, name=name, frame=int(n), person=float(box.id), left=float(box.xywhn[0][0]),
                     top=float(box.xywhn[0][1]), width=float(box.xywhn[0][2]), height=float(box.xywhn[0][3]),
                     is_anomaly=is_anomaly, anomaly=label) for box in boxes]

    def csv_path(self, label: str) -> str:
        return os.path.join(self.out_dir, "ucf-crime_dataset.csv" if label in self.anomalies
                            else "ucf-crime_dataset-normal.csv")

    def save_to_dataset(self, frame, i, n, label, name) -> Optional[List[BBox]]:
        boxes = self.get_boxes(frame)
        if not boxes.is_track:
            return None
        data = self.rows_for(boxes, i, n, label, name)
        write_rows(self.csv_path(label), data)
        return data
