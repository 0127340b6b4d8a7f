"""TEST INFRASTRUCTURE: what the reference's PoseLift loader makes of a bridge dict, restated compactly so that it can
run on the GPU box (where /root/reference does not exist).  Follows
/root/reference/shopformer/data/poselift_dataset.py:256-323 (person grouping, sliding window, continuity <= 5,
majority label) and :330-393 (channel selection, [-1,1] normalisation, [C,T,V] layout); with ``num_keypoints=18`` the second
loader's path, /root/reference/shopformer_2/data/poselift_dataset.py:57-91,410-564 (same windows, a synthetic neck joint = the
mean of the two shoulders appended as 18th keypoint before normalisation).  It is itself pinned by
tests/golden/poselift_fixture.npz, whose windows were produced by the reference's own classes
(tests/golden/make_poselift_fixture.py, build container only)."""
import numpy as np


def unflatten(frame_keys, row_frame, row_pid, row_bbox, row_kpts):
    """fixture arrays -> {frame: {pid: [bbox, kpts]}} in the stored iteration order"""
    data = {int(f): {} for f in frame_keys}
    for f, p, b, k in zip(row_frame, row_pid, row_bbox, row_kpts):
        data[int(f)][int(p)] = [b, k]
    return data


def _with_neck(k):
    """[17, C] -> [18, C]: neck = midpoint of the shoulders (joints 5, 6); a missing shoulder (x = y = 0) yields the other one,
    both missing yield zeros.  float64 out, as np.vstack of a float32 and a float64 row gives"""
    k = np.asarray(k)
    ls, rs = k[5], k[6]
    neck = (ls + rs) / 2.0
    l0, r0 = np.allclose(ls[:2], 0), np.allclose(rs[:2], 0)
    if l0 and r0:
        neck = np.zeros_like(ls)
    elif l0:
        neck = rs.copy()
    elif r0:
        neck = ls.copy()
    return np.vstack([k[:17], neck.reshape(1, -1)])


def windows(data, seq_len=12, stride=6, include_confidence=False, frame_labels=None, max_gap=5, num_keypoints=17):
    """-> (x [n, C, T, num_keypoints] float32, y [n] int64) in the loader's sample order"""
    per = {}
    for fnum, people in data.items():
        for pid, rec in (people or {}).items():
            k = np.array(rec[1])
            if np.isnan(k).any() or np.isinf(k).any():
                continue
            per.setdefault(pid, {})[int(fnum)] = k
    xs, ys = [], []
    c = 3 if include_confidence else 2
    for pid, fr in per.items():
        idx = sorted(fr)
        for s in range(0, len(idx) - seq_len + 1, stride):
            win = idx[s:s + seq_len]
            if any(b - a > max_gap for a, b in zip(win, win[1:])):
                continue
            seq = np.array([(_with_neck(fr[f]) if num_keypoints == 18 else fr[f][:17])[:, :c] for f in win])   # 17: float32 kept; 18: float64
            xy = seq[:, :, :2].copy()
            valid = np.any(xy != 0, axis=-1)
            if valid.sum() > 0:
                centre = xy[valid].mean(axis=0)
                scale = np.abs((xy - centre)[valid]).max() + 1e-6
            else:
                centre, scale = np.array([0.0, 0.0]), 1.0
            seq[:, :, :2] = np.nan_to_num((xy - centre) / scale, nan=0.0, posinf=0.0, neginf=0.0)
            xs.append(np.transpose(seq.astype(np.float32), (2, 0, 1)))
            if frame_labels is not None:
                lab = [frame_labels[min(f, len(frame_labels) - 1)] for f in win]
                ys.append(1 if sum(lab) > len(lab) // 2 else 0)
            else:
                ys.append(0)
    x = np.stack(xs) if xs else np.zeros((0, c, seq_len, num_keypoints), np.float32)
    return x, np.asarray(ys, np.int64)
