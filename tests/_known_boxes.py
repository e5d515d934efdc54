"""Hand-derived known answers for the OpenCV part of utils/affutils.py:26-53,:209-212 (cv2 is absent from the reference tree and from
this image: parity unpinned, so these cases restate OpenCV's documented semantics): findContours outer contours are 8-connected
components, boundingRect is the tight box (x, y, w, h); the reference clamps x1 = min(x + w, W - 1), y1 = min(y + h, H - 1) and fills
[y0:y1, x0:x1] end-exclusive - so a box keeps its tight extent unless it touches the right / bottom image border, where it loses its
last column / row.  name -> (map rows, expected mask rows); maps hold 0 / 1 values, threshold 0.5.  Shared by the oracle test (CPU)
and the HIP kernel test (GPU)."""

KNOWN = {
    "single_pixel_centre": (["00000", "00000", "00100", "00000", "00000"], ["00000", "00000", "00100", "00000", "00000"]),
    "border_box_loses_last_col_row": (["00000", "00000", "00011", "00011", "00011"], ["00000", "00000", "00010", "00010", "00000"]),
    "corner_pixel_vanishes": (["00000", "00000", "00000", "00000", "00001"], ["00000", "00000", "00000", "00000", "00000"]),
    "hole_filled_by_box": (["11100", "10100", "11100", "00000", "00000"], ["11100", "11100", "11100", "00000", "00000"]),
    "diagonal_touch_is_one_component": (["10000", "01000", "00000", "00000", "00000"], ["11000", "11000", "00000", "00000", "00000"]),
    "anti_diagonal_touch_is_one_component": (["000100", "001000", "000000", "000000", "000000", "000000"],
                                             ["001100", "001100", "000000", "000000", "000000", "000000"]),
    "two_separate_blobs": (["10000", "00000", "00100", "00000", "00000"], ["10000", "00000", "00100", "00000", "00000"]),
    "two_blobs_keep_their_own_boxes": (["110000", "110000", "000000", "000110", "000110", "000000"],
                                       ["110000", "110000", "000000", "000110", "000110", "000000"]),
    "one_pixel_bridge_merges_the_boxes": (["110000", "111110", "000110", "000110", "000000", "000000"],
                                          ["111110", "111110", "111110", "111110", "000000", "000000"]),
    "blob_inside_a_hole_changes_nothing": (["111110", "100010", "101010", "100010", "111110", "000000"],
                                           ["111110", "111110", "111110", "111110", "111110", "000000"]),
    "l_shape_fills_its_corner": (["100000", "100000", "111000", "000000", "000000", "000000"],
                                 ["111000", "111000", "111000", "000000", "000000", "000000"]),
    "top_left_border_loses_nothing": (["110000", "110000", "000000", "000000", "000000", "000000"],
                                      ["110000", "110000", "000000", "000000", "000000", "000000"]),
    "full_image_loses_last_row_and_column": (["1111", "1111", "1111", "1111"], ["1110", "1110", "1110", "0000"]),
    "bottom_row_strip_vanishes": (["00000", "00000", "00000", "00000", "01110"], ["00000", "00000", "00000", "00000", "00000"]),
    "right_column_strip_vanishes": (["00001", "00001", "00001", "00000", "00000"], ["00000", "00000", "00000", "00000", "00000"]),
    "all_zero": (["00000", "00000", "00000", "00000", "00000"], ["00000", "00000", "00000", "00000", "00000"]),
}

# graded maps: (rows of floats, threshold, expected mask rows).  u8 truncation of map * 255, thr = int(threshold * max u8), strict >
GRADED = {
    # max 0.8 -> 204, thr = int(0.5 * 204) = 102; 0.4 -> trunc(102.0000015) = 102 (not above), 0.405 -> 103 (above)
    "knife_edge_at_the_u8_threshold": ([[0.8, 0.4, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0.0, 0.405, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0]], 0.5,
                                       ["1000", "0000", "0100", "0000"]),
    # max 1.0 -> 255, thr = int(0.79 * 255) = 201: 0.7905 -> 201 (not above), 0.7922 -> 202 (above, merges diagonally with the peak)
    "reference_default_threshold": ([[0.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.7905, 0.0], [0.0, 0.0, 0.7922, 0.0], [0.0, 0.0, 0.0, 0.0]], 0.79,
                                    ["0000", "0110", "0110", "0000"]),
}
