"""Pitch landmark tables for the hot path.

Own emission of the data the reference keeps in ``eagle/utils/pitch.py:1-60`` (index -> label),
``:65`` (NOT_ON_PLANE) and ``:209-267`` (UEFA 105 x 68 m world coordinates).  One row per landmark, in
heat-map index order; ``WORLD_ORDER`` records the insertion order of the reference's GROUND_TRUTH_POINTS
dict because that order decides the line-group iteration order of the keypoint synthesis
(``eagle/models/coordinate_model.py:82-90,169-183``).  ``tests/test_pitch_weights_abi.py::test_pitch_tables_match_reference_dump`` checks every value against
``tests/golden/pitch_tables.json`` (dumped from the reference module by ``tests/golden/make_golden.py``).

The same table is compiled into the C-ABI library (``csrc/pitch_table.h`` is generated from this file by
``eagle_amd/csrc/gen_pitch_table.py``).
"""

PITCH_WIDTH = 105
PITCH_HEIGHT = 68
N_LANDMARKS = 57

_PA_Y0, _PA_Y1 = 13.84, 54.16          # penalty area
_GA_Y0, _GA_Y1 = 24.84, 43.16          # goal area
_GP_Y0, _GP_Y1 = 30.34, 37.66          # goal posts
_ARC_Y0, _ARC_Y1 = 26.687510683768487, 41.31248931623151
_CT_X0, _CT_X1 = 43.68756810653572, 61.31243189346428
_CC_X0, _CC_X1 = 46.02997295214309, 58.97002704785691
_CC_Y0, _CC_Y1 = 27.52997295214309, 40.47002704785691
_LT_X, _RT_X = 19.9906727467215, 85.0093272532785
_T_Y0, _T_Y1 = 32.29991071959168, 35.70008928040832

# (index, label, x, y, z)
LANDMARKS = (
    (0, "L_GOAL_TL_POST", 0.0, _GP_Y0, -2.44),
    (1, "L_GOAL_TR_POST", 0.0, _GP_Y1, -2.44),
    (2, "L_GOAL_BL_POST", 0.0, _GP_Y0, 0.0),
    (3, "L_GOAL_BR_POST", 0.0, _GP_Y1, 0.0),
    (4, "L_GOAL_AREA_BR_CORNER", 5.5, _GA_Y0, 0.0),
    (5, "L_GOAL_AREA_TR_CORNER", 5.5, _GA_Y1, 0.0),
    (6, "L_GOAL_AREA_BL_CORNER", 0.0, _GA_Y0, 0.0),
    (7, "L_GOAL_AREA_TL_CORNER", 0.0, _GA_Y1, 0.0),
    (8, "L_PENALTY_AREA_BR_CORNER", 16.5, _PA_Y0, 0.0),
    (9, "L_PENALTY_AREA_TR_CORNER", 16.5, _PA_Y1, 0.0),
    (10, "L_PENALTY_AREA_BL_CORNER", 0.0, _PA_Y0, 0.0),
    (11, "L_PENALTY_AREA_TL_CORNER", 0.0, _PA_Y1, 0.0),
    (12, "BL_PITCH_CORNER", 0.0, 0.0, 0.0),
    (13, "TL_PITCH_CORNER", 0.0, 68.0, 0.0),
    (14, "B_TOUCH_AND_HALFWAY_LINES_INTERSECTION", 52.5, 0.0, 0.0),
    (15, "T_TOUCH_AND_HALFWAY_LINES_INTERSECTION", 52.5, 68.0, 0.0),
    (16, "R_PENALTY_AREA_BL_CORNER", 88.5, _PA_Y0, 0.0),
    (17, "R_PENALTY_AREA_TL_CORNER", 88.5, _PA_Y1, 0.0),
    (18, "R_PENALTY_AREA_BR_CORNER", 105.0, _PA_Y0, 0.0),
    (19, "R_PENALTY_AREA_TR_CORNER", 105.0, _PA_Y1, 0.0),
    (20, "R_GOAL_AREA_BL_CORNER", 99.5, _GA_Y0, 0.0),
    (21, "R_GOAL_AREA_TL_CORNER", 99.5, _GA_Y1, 0.0),
    (22, "R_GOAL_AREA_BR_CORNER", 105.0, _GA_Y0, 0.0),
    (23, "R_GOAL_AREA_TR_CORNER", 105.0, _GA_Y1, 0.0),
    (24, "R_GOAL_TL_POST", 105.0, _GP_Y1, -2.44),
    (25, "R_GOAL_TR_POST", 105.0, _GP_Y0, -2.44),
    (26, "R_GOAL_BL_POST", 105.0, _GP_Y1, 0.0),
    (27, "R_GOAL_BR_POST", 105.0, _GP_Y0, 0.0),
    (28, "BR_PITCH_CORNER", 105.0, 0.0, 0.0),
    (29, "TR_PITCH_CORNER", 105.0, 68.0, 0.0),
    (30, "CENTER_CIRCLE_TANGENT_TR", _CT_X1, 36.462426470588234, 0.0),
    (31, "CENTER_CIRCLE_TANGENT_TL", _CT_X0, 36.46242647058824, 0.0),
    (32, "CENTER_CIRCLE_TANGENT_BR", _CT_X1, 31.537573529411766, 0.0),
    (33, "CENTER_CIRCLE_TANGENT_BL", _CT_X0, 31.53757352941176, 0.0),
    (34, "CENTER_CIRCLE_TR", _CC_X1, _CC_Y1, 0.0),
    (35, "CENTER_CIRCLE_TL", _CC_X0, _CC_Y1, 0.0),
    (36, "CENTER_CIRCLE_BR", _CC_X1, _CC_Y0, 0.0),
    (37, "CENTER_CIRCLE_BL", _CC_X0, _CC_Y0, 0.0),
    (38, "CENTER_CIRCLE_R", 61.65, 34.0, 0.0),
    (39, "CENTER_CIRCLE_L", 43.35, 34.0, 0.0),
    (40, "T_HALFWAY_LINE_AND_CENTER_CIRCLE_INTERSECTION", 52.5, 43.15, 0.0),
    (41, "B_HALFWAY_LINE_AND_CENTER_CIRCLE_INTERSECTION", 52.5, 24.85, 0.0),
    (42, "CENTER_MARK", 52.5, 34.0, 0.0),
    (43, "LEFT_CIRCLE_R", 20.15, 34.0, 0.0),
    (44, "BL_16M_LINE_AND_PENALTY_ARC_INTERSECTION", 16.5, _ARC_Y0, 0.0),
    (45, "TL_16M_LINE_AND_PENALTY_ARC_INTERSECTION", 16.5, _ARC_Y1, 0.0),
    (46, "LEFT_CIRCLE_TANGENT_T", _LT_X, _T_Y1, 0.0),
    (47, "LEFT_CIRCLE_TANGENT_B", _LT_X, _T_Y0, 0.0),
    (48, "L_PENALTY_MARK", 11.0, 34.0, 0.0),
    (49, "L_MIDDLE_PENALTY", 16.5, 34.0, 0.0),
    (50, "RIGHT_CIRCLE_L", 84.85, 34.0, 0.0),
    (51, "BR_16M_LINE_AND_PENALTY_ARC_INTERSECTION", 88.5, _ARC_Y0, 0.0),
    (52, "TR_16M_LINE_AND_PENALTY_ARC_INTERSECTION", 88.5, _ARC_Y1, 0.0),
    (53, "RIGHT_CIRCLE_TANGENT_T", _RT_X, _T_Y1, 0.0),
    (54, "RIGHT_CIRCLE_TANGENT_B", _RT_X, _T_Y0, 0.0),
    (55, "R_PENALTY_MARK", 94.0, 34.0, 0.0),
    (56, "R_MIDDLE_PENALTY", 88.5, 34.0, 0.0),
)

# Heat-map indices in the insertion order of the reference's world-coordinate dict (pitch.py:209-267).
WORLD_ORDER = (
    42, 13, 12, 29, 28, 48, 55, 11, 9, 10, 8, 17, 19, 16, 18, 7, 5, 6, 4, 21, 23, 20, 22,
    0, 1, 2, 3, 24, 25, 26, 27, 15, 14, 40, 41, 45, 44, 52, 51, 30, 31, 32, 33, 34, 35, 36, 37,
    38, 39, 43, 50, 46, 47, 49, 53, 54, 56,
)

NOT_ON_PLANE = (0, 1, 24, 25)

INTERSECTION_TO_PITCH_POINTS = {i: lab for i, lab, _, _, _ in LANDMARKS}
PITCH_POINTS_TO_INTERSECTION = {lab: i for i, lab, _, _, _ in LANDMARKS}
GROUND_TRUTH_POINTS = {LANDMARKS[i][1]: (LANDMARKS[i][2], LANDMARKS[i][3], LANDMARKS[i][4]) for i in WORLD_ORDER}


def on_plane_mask():
    """57 booleans: landmark usable for the homography (index not in NOT_ON_PLANE and world z == 0);
    coordinate_model.py:338-344."""
    return [(i not in NOT_ON_PLANE) and (z == 0.0) for i, _, _, _, z in LANDMARKS]
