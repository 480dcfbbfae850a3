"""Helpers of the prior-chaining tests: split a synthetic window into the factors of a dropped landmark set (D) and the
rest (R), and turn a marginalisation result into the prior arrays of ctvio_window / Window."""
import importlib
import numpy as np

cv = importlib.import_module("ctrl-vio_amd")
PK_ROT, PK_POS, PK_BG, PK_BA, PK_LD = 0, 1, 2, 3, 4


def split_by_landmarks(w, drop):
    """drop: boolean (L,).  Returns (wD, wR, mapD, mapR): wD holds ONLY the visual blocks of the dropped landmarks (no IMU,
    no bias chain, no prior), wR everything else; map*[new landmark index] = old index."""
    drop = np.asarray(drop, bool)
    sel_v = drop[w.v_lm]

    def sub(keep_lm, keep_v, with_inertial):
        x = w.copy()
        old = np.where(keep_lm)[0]
        new_of_old = -np.ones(w.L, np.int64); new_of_old[old] = np.arange(old.size)
        x.rho = w.rho[old].copy()
        for name in ("v_lm", "v_ti", "v_tj", "v_rowi", "v_rowj", "v_pi", "v_pj"):
            setattr(x, name, getattr(w, name)[keep_v].copy())
        x.v_lm = new_of_old[x.v_lm].astype(np.int32)
        if not with_inertial:
            for name in ("imu_t", "imu_gyro", "imu_acc", "imu_bias", "bc_i", "bc_j", "bc_w"):
                a = getattr(w, name)
                setattr(x, name, a[:0].copy())
        x.pJ0 = np.zeros((0, 0)); x.pr0 = np.zeros(0)
        x.p_kind = np.zeros(0, np.int32); x.p_index = np.zeros(0, np.int32); x.p_off = np.zeros(0, np.int32); x.p_x0 = np.zeros((0, 4))
        return x.normalize(), old

    wD, mapD = sub(drop, sel_v, False)
    wR, mapR = sub(~drop, ~sel_v, True)
    return wD, wR, mapD, mapR


def prior_arrays(w, kept, J0, r0):
    """kept: ascending unknown indices (pose part only) as returned by marginalize.  Groups them into ROT / POS / BG / BA / LD
    blocks (3, 3, 3, 3, 1 consecutive unknowns) and returns (pJ0, pr0, p_kind, p_index, p_off, p_x0) at the state of w."""
    K, F, P = w.K, w.F, w.P
    kinds, idxs, offs, x0 = [], [], [], []
    kept = list(map(int, kept))
    j = 0
    while j < len(kept):
        u = kept[j]
        assert u < P, "landmarks cannot be part of a prior"
        if u == P - 1:
            kinds.append(PK_LD); idxs.append(0); offs.append(j); x0.append([w.ld, 0, 0, 0]); j += 1
            continue
        assert u % 3 == 0 and kept[j:j + 3] == [u, u + 1, u + 2], "a block must be kept or dropped as a whole"
        if u < 6 * K:
            k, part = divmod(u, 6)
            if part == 0:
                kinds.append(PK_ROT); x0.append(w.quat[k].tolist())
            else:
                kinds.append(PK_POS); x0.append(w.pos[k].tolist() + [0.0])
            idxs.append(k)
        else:
            f, part = divmod(u - 6 * K, 6)
            kinds.append(PK_BG if part == 0 else PK_BA); idxs.append(f)
            x0.append(w.bias[f, part:part + 3].tolist() + [0.0])
        offs.append(j); j += 3
    return (np.array(J0, float), np.array(r0, float), np.array(kinds, np.int32), np.array(idxs, np.int32), np.array(offs, np.int32),
            np.array(x0, float).reshape(-1, 4))


def chain_case(cfg="config1", seed=1000):
    """(w, wD, wR, role, mapR): drop the first half of the landmarks of a synthetic window."""
    w = cv.synth.make_window(cfg, seed=seed)
    drop = np.arange(w.L) < w.L // 2
    wD, wR, mapD, mapR = split_by_landmarks(w, drop)
    return w, wD, wR, mapR
