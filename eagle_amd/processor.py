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

    def get_team_mapping(self, frames, coords):
        """The reference post-processor's ``get_team_mapping`` (eagle/processor.py:405-464; its "pretty slow" step) for a clip and the
        ``get_coordinates`` output of that clip: colour segmentation and counting of every player crop on the GPU (eagle_amd/teams.py).
        -> {player_id: 0 | 1}.  Player ids are track ids when the model was built with ``tracker=True``."""
        from . import teams
        frames = np.ascontiguousarray(frames, np.uint8)
        d = self.model.handle.upload(frames)
        try:
            return teams.get_team_mapping(self.model.handle, d, coords, n_frames=len(frames))
        finally:
            self.model.handle.free(d)
