"""proben_amd/evalKAIST/evaluation_script.py: the KAIST miss-rate evaluator behind the reference's call
(demo/KAIST/demo_LAMR_KAIST.py:85,145; demo_train_KAIST.py:116-121).  PARITY UNPINNED - the third-party script is not in the reference's
tree - so every rule of the published protocol is pinned here by a hand-worked case with the expected numbers written out."""
import json
import math

import numpy as np
import pytest

import proben_amd  # noqa: F401
from proben_amd.evalKAIST import evaluation_script as E
from proben_amd.evalKAIST.evaluation_script import evaluate


def ann_file(tmp_path, n_images, boxes, name="KAIST_annotation.json"):
    """boxes: (image_id, x, y, w, h[, occlusion[, ignore[, category_id]]]) -> a COCO-style KAIST annotation file."""
    anns = []
    for i, b in enumerate(boxes):
        img, x, y, w, h = b[:5]
        occ = b[5] if len(b) > 5 else 0
        ig = b[6] if len(b) > 6 else 0
        cat = b[7] if len(b) > 7 else 1
        anns.append({"id": i, "image_id": img, "category_id": cat, "bbox": [x, y, w, h], "height": h, "occlusion": occ, "ignore": ig})
    ds = {"images": [{"id": i, "im_name": f"set06/V000/I{i:05d}", "height": 512, "width": 640} for i in range(n_images)],
          "annotations": anns,
          "categories": [{"id": 0, "name": "__ignore__"}, {"id": 1, "name": "person"}, {"id": 2, "name": "cyclist"},
                         {"id": 3, "name": "people"}, {"id": 4, "name": "person?"}]}
    p = tmp_path / name
    p.write_text(json.dumps(ds))
    return str(p)


def det_file(tmp_path, rows, name="KAIST_test_result.txt"):
    """rows: (1-based frame index, x, y, w, h, score) -> the text file the reference's writer produces."""
    p = tmp_path / name
    p.write_text("".join(",".join(str(v) for v in r) + "\n" for r in rows))
    return str(p)


def test_reference_call_shape(tmp_path, capsys):
    """The three calls the reference makes on the result: evaluate(...)[k].summarize(0) and 1 - eval['yy'][0][-1]."""
    gt = ann_file(tmp_path, 3, [(0, 10, 10, 30, 60), (1, 100, 100, 30, 80)])
    dt = det_file(tmp_path, [(1, 10, 10, 30, 60, 0.9), (2, 100, 100, 30, 80, 0.8)])
    res = evaluate(gt, dt, "Multispectral")
    assert sorted(res) == ["all", "day", "night"]
    assert res["all"].summarize(0) == 0.0                        # everything found, no false positive: MR 0 at every point
    assert 1 - res["all"].eval["yy"][0][-1] == 1.0               # recall_all (demo_train_KAIST.py:120)
    assert res["night"].summarize(0) == -1.0                     # 3 images: all of them "day" (first 1455 ids), nothing to evaluate at night
    assert res["all"].method == "KAIST"                          # basename up to the first underscore, as the script labels a method
    out = capsys.readouterr().out
    assert "MR_all: 0.00" in out and "recall_all: 100.00" in out


def test_log_average_over_nine_fppi_points_hand_worked(tmp_path):
    """100 images; score order FP(.95) TP(.9) FP(.8) TP(.7), one of three pedestrians never found:
    tp = 0 1 1 2, fp = 1 1 2 2, FPPI = .01 .01 .02 .02, recall = 0 1/3 1/3 2/3.
    Reference points .0100 and .0178 read the last operating point with FPPI <= them (index 1): miss rate 2/3;
    the other seven read index 3: miss rate 1/3.  MR = exp((2 ln(2/3) + 7 ln(1/3)) / 9)."""
    gt = ann_file(tmp_path, 100, [(0, 10, 10, 30, 60), (1, 100, 100, 30, 80), (2, 200, 100, 30, 80)])
    dt = det_file(tmp_path, [(1, 10, 10, 30, 60, 0.9), (2, 400, 300, 30, 60, 0.8), (2, 100, 100, 30, 80, 0.7), (4, 300, 300, 30, 60, 0.95)])
    ev = evaluate(gt, dt)["all"]
    np.testing.assert_allclose(ev.eval["xx"][0], [0.01, 0.01, 0.02, 0.02])
    np.testing.assert_allclose(ev.eval["yy"][0], [1, 2 / 3, 2 / 3, 1 / 3])
    np.testing.assert_allclose(1 - ev.eval["TP"][0, :, 0, 0], [2 / 3, 2 / 3] + [1 / 3] * 7)
    assert ev.summarize(0) == pytest.approx(math.exp((2 * math.log(2 / 3) + 7 * math.log(1 / 3)) / 9), rel=1e-12)
    np.testing.assert_allclose(ev.params.fppiThrs, 10.0 ** np.arange(-2, 0.01, 0.25), atol=5e-5)   # the 9 points, rounded to 4 digits


def test_no_operating_point_below_a_reference_reads_the_last_one(tmp_path):
    """10 images, the top-scoring detection is a false positive: FPPI starts at 0.1, so the points .01 .. .0562 have no operating point
    at or below them; the published script's index -1 then reads the LAST operating point (miss rate 0 here), not 1."""
    gt = ann_file(tmp_path, 10, [(0, 10, 10, 30, 60)])
    dt = det_file(tmp_path, [(2, 300, 300, 30, 60, 0.9), (1, 10, 10, 30, 60, 0.5)])
    ev = evaluate(gt, dt)["all"]
    np.testing.assert_allclose(ev.eval["xx"][0], [0.1, 0.1])
    np.testing.assert_allclose(1 - ev.eval["TP"][0, :, 0, 0], [0.0] * 9)
    assert ev.summarize(0) == 0.0


def test_reasonable_subset_height_rule(tmp_path):
    """Ground truth shorter than 55 px is ignored: not a miss when unfound, and a detection on it is neither TP nor FP."""
    gt = ann_file(tmp_path, 100, [(0, 10, 10, 20, 54), (0, 100, 10, 20, 55)])
    ev = evaluate(gt, det_file(tmp_path, []))["all"]
    assert [bool(v) for v in ev.evalImgs[0]["gtIgnore"]] == [False, True]        # sorted real-first: the 55-px box, then the 54-px one
    assert ev.summarize(0) == 1.0                                                 # one countable pedestrian, missed
    ev = evaluate(gt, det_file(tmp_path, [(1, 10, 10, 20, 54, 0.9)]))["all"]
    assert ev.evalImgs[0]["dtIgnore"].tolist() == [[True]] and len(ev.eval["yy"][0]) == 0   # matched the ignored box: not counted
    assert ev.summarize(0) == 1.0


def test_reasonable_subset_occlusion_rule(tmp_path):
    """Occlusion 0 (none) and 1 (partial) count, 2 (heavy) is ignored."""
    gt = ann_file(tmp_path, 100, [(0, 10, 10, 30, 60, 0), (0, 100, 10, 30, 60, 1), (0, 200, 10, 30, 60, 2)])
    ev = evaluate(gt, det_file(tmp_path, [(1, 10, 10, 30, 60, 0.9), (1, 100, 10, 30, 60, 0.8)]))["all"]
    assert ev.evalImgs[0]["gtIgnore"].tolist() == [False, False, True]
    assert ev.summarize(0) == 0.0 and ev.eval["yy"][0][-1] == 0.0                 # both countable pedestrians found


def test_reasonable_subset_image_bounds_rule(tmp_path):
    """A box must lie inside x >= 5, y >= 5, x + w <= 635, y + h <= 507 to count."""
    boxes = [(0, 4, 10, 30, 60), (0, 100, 4, 30, 60), (0, 606, 10, 30, 60), (0, 300, 448, 30, 60),      # one rule broken each
             (0, 5, 5, 30, 60), (0, 605, 447, 30, 60)]                                                    # exactly on the border: inside
    ev = evaluate(ann_file(tmp_path, 100, boxes), det_file(tmp_path, []))["all"]
    assert ev.evalImgs[0]["gtIgnore"].tolist() == [False, False, True, True, True, True]


def test_own_ignore_flag_and_category_filter(tmp_path):
    """An annotation's own `ignore` flag wins over the subset rules; categories other than 1 (person) are not loaded at all."""
    gt = ann_file(tmp_path, 100, [(0, 10, 10, 30, 60, 0, 1), (0, 100, 10, 30, 60, 0, 0, 3), (0, 200, 10, 30, 60)])
    ev = evaluate(gt, det_file(tmp_path, [(1, 100, 10, 30, 60, 0.9)]))["all"]
    assert ev.evalImgs[0]["gtIgnore"].tolist() == [False, True]       # the `people` box (category 3) is absent
    assert ev.evalImgs[0]["dtMatches"].tolist() == [[False]]          # so the detection on it is a false positive
    np.testing.assert_allclose(ev.eval["xx"][0], [0.01])


def test_detection_height_filter_uses_the_expanded_range(tmp_path):
    """Detections shorter than 55 / 1.25 = 44 px are dropped before matching; 44 px stays (and is a false positive here)."""
    gt = ann_file(tmp_path, 100, [(0, 10, 10, 30, 60)])
    ev = evaluate(gt, det_file(tmp_path, [(1, 300, 300, 20, 43.9, 0.9), (1, 400, 300, 20, 44.0, 0.8)]))["all"]
    assert ev.evalImgs[0]["dtScores"].tolist() == [0.8]
    np.testing.assert_allclose(ev.eval["xx"][0], [0.01])


def test_row_index_quirk_and_its_fix(tmp_path):
    """The published script looks a kept detection's overlaps up in row `id - id of the first kept detection` of the matrix over ALL of
    the image's detections in score order.  One pedestrian (height 80), three detections of frame 1 in file order = score order:
        id 1: score .9, 30 px tall -> dropped by the height filter (30 < 44), far from the pedestrian      matrix row 0: overlap 0
        id 2: score .8, exactly on the pedestrian                                                         matrix row 1: overlap 1
        id 3: score .7, 80 px tall, far away                                                              matrix row 2: overlap 0
    Kept: ids 2, 3 -> rows 2 - 2 = 0 and 3 - 2 = 1: the detection ON the pedestrian is judged with the dropped detection's overlaps (a
    false positive), the far one with the overlaps of the detection on the pedestrian (a true positive).  Default = the script's
    arithmetic; fix_row_index=True = each detection's own row."""
    gt = ann_file(tmp_path, 10, [(0, 100, 100, 40, 80)])
    dt = det_file(tmp_path, [(1, 400, 300, 15, 30, 0.9), (1, 100, 100, 40, 80, 0.8), (1, 300, 300, 40, 80, 0.7)])
    pub = evaluate(gt, dt, "Multispectral")["all"]
    fix = evaluate(gt, dt, "Multispectral", fix_row_index=True)["all"]
    assert pub.evalImgs[0]["dtScores"].tolist() == fix.evalImgs[0]["dtScores"].tolist() == [0.8, 0.7]
    assert pub.evalImgs[0]["dtMatches"].tolist() == [[False, True]]
    assert fix.evalImgs[0]["dtMatches"].tolist() == [[True, False]]
    np.testing.assert_allclose(pub.eval["yy"][0], [1.0, 0.0])     # miss rate after the .8 (counted false) and after the .7
    np.testing.assert_allclose(fix.eval["yy"][0], [0.0, 0.0])
    np.testing.assert_allclose(pub.eval["xx"][0], [0.1, 0.1])     # FPPI (10 images): the false positive comes first ...
    np.testing.assert_allclose(fix.eval["xx"][0], [0.0, 0.1])     # ... or last
    # reference points below 0.1 FPPI: the published lookup finds no operating point (reads the last one: miss rate 0), the fix finds
    # the true positive at FPPI 0 -> both 0 here; the curves above are where the two differ
    # no height-filtered detection in front and score-ordered rows: the two agree (the usual case)
    dt2 = det_file(tmp_path, [(1, 100, 100, 40, 80, 0.8), (1, 300, 300, 40, 80, 0.7), (1, 400, 300, 15, 30, 0.6)], name="KAIST_b_result.txt")
    a, b = evaluate(gt, dt2)["all"], evaluate(gt, dt2, fix_row_index=True)["all"]
    assert a.evalImgs[0]["dtMatches"].tolist() == b.evalImgs[0]["dtMatches"].tolist() == [[True, False]]
    # an image whose rows are not listed together: the difference leaves the image's detections -> IndexError, as in the script
    gt3 = ann_file(tmp_path, 10, [(0, 100, 100, 40, 80), (1, 100, 100, 40, 80)], name="KAIST_c_annotation.json")
    dt3 = det_file(tmp_path, [(1, 100, 100, 40, 80, 0.9), (2, 100, 100, 40, 80, 0.8), (2, 300, 300, 40, 80, 0.7), (1, 300, 300, 40, 80, 0.6)], name="KAIST_c_result.txt")
    with pytest.raises(IndexError, match="first kept id"):
        evaluate(gt3, dt3)
    assert evaluate(gt3, dt3, fix_row_index=True)["all"].evalImgs[0]["dtMatches"].tolist() == [[True, False]]


def test_overlap_against_ignored_ground_truth_is_over_the_detection_area():
    """Real box: intersection / union.  Ignored box: intersection / detection area."""
    d = [[0, 0, 10, 10]]
    g = [[5, 0, 10, 10], [5, 0, 10, 10], [20, 20, 5, 5]]
    ov = E.overlaps(d, g, [0, 1, 0])
    np.testing.assert_allclose(ov, [[50 / 150, 50 / 100, 0.0]])


def test_ignore_region_absorbs_detections_but_never_beats_a_real_match(tmp_path):
    """A large ignored region (heavily occluded) around a real pedestrian: the detection on the pedestrian is a TP although it overlaps
    the region completely (the scan stops at the first ignored box once a real match exists); two more detections inside the
    region are absorbed - neither TP nor FP -, a third one outside is the only false positive."""
    gt = ann_file(tmp_path, 100, [(0, 100, 100, 30, 60), (0, 50, 50, 300, 300, 2)])
    dt = det_file(tmp_path, [(1, 100, 100, 30, 60, 0.9), (1, 200, 200, 30, 60, 0.8), (1, 250, 120, 30, 60, 0.7), (1, 500, 50, 30, 60, 0.6)])
    ev = evaluate(gt, dt)["all"]
    im = ev.evalImgs[0]
    assert im["dtMatches"].tolist() == [[True, True, True, False]] and im["dtIgnore"].tolist() == [[False, True, True, False]]
    np.testing.assert_allclose(ev.eval["xx"][0], [0.0, 0.01])       # counted: the TP, then the FP
    np.testing.assert_allclose(ev.eval["yy"][0], [0.0, 0.0])


def test_greedy_matching_in_score_order_one_detection_per_pedestrian(tmp_path):
    """Two detections on one pedestrian: the higher-scoring one takes it, the other is a false positive; IoU exactly 0.5 matches."""
    gt = ann_file(tmp_path, 100, [(0, 100, 100, 30, 60)])
    dt = det_file(tmp_path, [(1, 100, 100, 30, 60, 0.6), (1, 101, 100, 30, 60, 0.9)])
    ev = evaluate(gt, dt)["all"]
    assert ev.evalImgs[0]["dtScores"].tolist() == [0.9, 0.6] and ev.evalImgs[0]["dtMatches"].tolist() == [[True, False]]
    # IoU exactly 1/2: boxes 30 x 60 shifted by 10 px -> inter 20 x 60 = 1200, union 3600 - 1200 = 2400
    ev = evaluate(gt, det_file(tmp_path, [(1, 110, 100, 30, 60, 0.9)]))["all"]
    assert ev.evalImgs[0]["dtMatches"].tolist() == [[True]]
    ev = evaluate(gt, det_file(tmp_path, [(1, 110.01, 100, 30, 60, 0.9)]))["all"]
    assert ev.evalImgs[0]["dtMatches"].tolist() == [[False]]


def test_day_night_split_is_the_first_1455_image_ids(tmp_path):
    """`day` = image ids 0 .. 1454, `night` = 1455 ..: a miss in frame 1455 (1-based 1456) shows at night only."""
    gt = ann_file(tmp_path, 1460, [(0, 10, 10, 30, 60), (1454, 10, 10, 30, 60), (1455, 10, 10, 30, 60), (1459, 10, 10, 30, 60)])
    dt = det_file(tmp_path, [(1, 10, 10, 30, 60, 0.9), (1455, 10, 10, 30, 60, 0.8), (1460, 10, 10, 30, 60, 0.7)])
    res = evaluate(gt, dt)
    assert len(res["day"].params.imgIds) == 1455 and len(res["night"].params.imgIds) == 5
    assert res["day"].summarize(0) == 0.0                          # both day pedestrians found
    assert res["night"].summarize(0) == pytest.approx(0.5)         # one of two night pedestrians found, no false positive
    assert res["all"].summarize(0) == pytest.approx(0.25)


def test_result_rows_are_one_based_frame_indices_and_inputs_are_checked(tmp_path):
    gt = ann_file(tmp_path, 2, [(1, 10, 10, 30, 60)])
    assert evaluate(gt, det_file(tmp_path, [(2, 10, 10, 30, 60, 0.9)]))["all"].summarize(0) == 0.0      # row "2,..." is image id 1
    with pytest.raises(ValueError, match="image ids"):
        evaluate(gt, det_file(tmp_path, [(3, 10, 10, 30, 60, 0.9)]))
    old = tmp_path / "old.json"
    old.write_text(json.dumps({"1": [[10, 10, 30, 60, 0]]}))                                            # the r04 private format: refused
    with pytest.raises(ValueError, match="not a KAIST annotation file"):
        evaluate(str(old), det_file(tmp_path, []))
    # a COCO-style result json is read as well (the script's loadRes takes both)
    rj = tmp_path / "KAIST_res.json"
    rj.write_text(json.dumps([{"image_id": 1, "category_id": 1, "bbox": [10, 10, 30, 60], "score": 0.9}]))
    assert evaluate(gt, str(rj))["all"].summarize(0) == 0.0


def test_importable_under_the_references_name():
    """`from evalKAIST.evaluation_script import evaluate` with proben_amd's directory on the path, as the reference's demo scripts do."""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(proben_amd.__file__))
    try:
        mod = importlib.import_module("evalKAIST.evaluation_script")
        assert callable(mod.evaluate)
    finally:
        sys.path.pop(0)
        sys.modules.pop("evalKAIST.evaluation_script", None)
        sys.modules.pop("evalKAIST", None)
