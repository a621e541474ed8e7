"""``Processor.process(frame) -> {players, ball, H}`` — the per-frame API BASELINE.json's north_star names.

The reference's ``eagle/processor.py`` is a whole-clip pandas post-processor and has no ``process(frame)``
(SURVEY §0); semantically this method is one iteration of the loop body of
``CoordinateModel.get_coordinates`` (eagle/models/coordinate_model.py:277-415) in the stateless configuration, with
``players`` == Coordinates["Player"] ∪ ["Goalkeeper"], ``ball`` == Coordinates["Ball"] (cm.py:369-392) and ``H`` the
3x3 homography of cm.py:355,363."""
import numpy as np

from . import records
from .coordinate_model import CoordinateModel


class Processor:
    def __init__(self, model: CoordinateModel = None, **model_kwargs):
        self.model = model or CoordinateModel(**model_kwargs)

    def process(self, frame):
        """frame: uint8 HWC BGR.  -> {"players": {id: {...}}, "ball": {k: {...}}, "H": 3x3 float64 | None, ...}"""
        rec = self.model.process_records(np.asarray(frame)[None])[0]
        return records.to_process_dict(rec)

    def process_clip(self, frames):
        return [records.to_process_dict(r) for r in self.model.process_records(frames)]
