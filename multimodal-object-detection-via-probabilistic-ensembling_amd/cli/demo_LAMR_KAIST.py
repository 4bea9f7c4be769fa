"""Counterpart of demo/KAIST/demo_LAMR_KAIST.py:85-145: KAIST detections as text rows
`<1-based frame index>,x,y,w,h,score` + per-image variances (.npz).  The reference then calls
`evalKAIST.evaluation_script.evaluate`, a package that is NOT in its tree: the log-average miss rate
evaluator is a "next" row, parity unpinned."""
import numpy as np


def kaist_rows(frame_index, instances):
    """frame_index 0-based -> list of 'idx+1,x,y,w,h,score' strings (demo_LAMR_KAIST.py:132-142)."""
    inst = instances.to("cpu")
    b = inst.pred_boxes.tensor.numpy()
    s = inst.scores.numpy()
    return [f"{frame_index + 1},{x1:.4f},{y1:.4f},{x2 - x1:.4f},{y2 - y1:.4f},{sc:.8f}" for (x1, y1, x2, y2), sc in zip(b, s)]


def write_kaist(path_txt, path_npz, per_frame_instances):
    rows, var = [], {}
    for i, inst in enumerate(per_frame_instances):
        rows += kaist_rows(i, inst)
        if inst.has("vars"):
            var[str(i)] = inst.vars.cpu().numpy()
    with open(path_txt, "w") as f:
        f.write("\n".join(rows) + ("\n" if rows else ""))
    if path_npz:
        np.savez(path_npz, **var)
    return len(rows)
