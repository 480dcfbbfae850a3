"""Three consecutive sliding windows, the way the reference chains them (TrajectoryManager::UpdateTrajectory ->
double2vector -> UpdateVIOPrior(MARGIN_OLD), src/estimator/trajectory_manager.cpp:122-286, 317-516), on a synthetic "world"
of 13 keyframes:

  for k = 0, 1, 2:   window k = frames k .. k+10 (11 frames, WINDOW_SIZE = 10)
     solve      prior_k + IMU in [first knot of frame k, t_{k+10}) + bias chain + visual blocks of the landmarks anchored in
                frames >= k (candidate rule: >= 2 observations inside the window, anchor - k < WINDOW_SIZE - 2), Solve(15)
     gauge      yaw + position of the first knot of the window put back to their pre-solve values (double2vector)
     prior      MARGIN_OLD: prior_k (dropping the knots in [ctrl_now, ctrl_later) and bias_k), the visual blocks of the
                landmarks anchored at frame k with CauchyLoss(1), the IMU samples before t_{k+1} (bias_k), the bias factor
                (k, k+1); marginalised: knots below ctrl_later = first knot of frame k+1, bias_k, those landmarks.

The same protocol is run by a `backend` (the HIP path through the C ABI, or the CPU oracle); landmarks whose anchor frame has
been marginalised are simply gone (the reference's feature manager re-anchors them: host bookkeeping outside this path).
"""
import importlib

import numpy as np

cv = importlib.import_module("ctrl-vio_amd")
PK_ROT, PK_POS, PK_BG, PK_BA, PK_LD = 0, 1, 2, 3, 4
NFRAMES, WIN = 13, 11
FRAME_DT = 100_000_000
PAD_NS = int(0.039 * 1e9)


def make_world(seed=2000, L=60, M=1440):
    """A 13-frame synthetic scenario (initial guess + the synthetic gauge prior on knots 0..3)."""
    return cv.synth.make_window("config1", seed=seed, F=NFRAMES, L=L, M=M)


def knot_index(t_ns, dt_ns):
    return int(t_ns // dt_ns)


class State:
    """The caller's live state (what the reference keeps in the trajectory deques, all_imu_bias_, para_Feature, line_delay)."""
    def __init__(self, world):
        self.quat, self.pos, self.bias = world.quat.copy(), world.pos.copy(), world.bias.copy()
        self.rho, self.ld = world.rho.copy(), float(world.ld)


def window_of(world, st, k, prior):
    """Window (index form) of frames k..k+10 at state st; prior = dict(J0, r0, kind, index (world knot / frame), off, x0) or None.
    Returns (window, info) with info = dict(kmin, lms (world landmark ids), frames)."""
    dt = world.dt_ns
    ft = np.arange(k, k + WIN, dtype=np.int64) * FRAME_DT
    kmin = knot_index(ft[0], dt)
    kmax = min(world.K - 1, knot_index(ft[-1] + PAD_NS, dt) + 3)
    opt_min = kmin * dt
    sel = (world.imu_t >= opt_min) & (world.imu_t < ft[-1])
    imu_t = world.imu_t[sel]
    anchor_t = np.zeros(world.L, np.int64)
    anchor_t[world.v_lm] = world.v_ti
    a_frame = anchor_t // FRAME_DT
    ok_v = (a_frame[world.v_lm] >= k) & (a_frame[world.v_lm] - k < 10 - 2) & (world.v_tj <= ft[-1])
    lms = np.unique(world.v_lm[ok_v])
    new_of = -np.ones(world.L, np.int64); new_of[lms] = np.arange(lms.size)
    w = cv.Window(t0_ns=kmin * dt, dt_ns=dt, quat=st.quat[kmin:kmax + 1].copy(), pos=st.pos[kmin:kmax + 1].copy(),
                  bias=st.bias[k:k + WIN].copy(), rho=st.rho[lms].copy(), ld=st.ld, ld_lo=world.ld_lo, ld_hi=world.ld_hi,
                  q_CI=world.q_CI, p_CI=world.p_CI, gravity=world.gravity, imu_w=world.imu_w, img_w=world.img_w, cauchy_a=2.0,
                  imu_t=imu_t, imu_gyro=world.imu_gyro[sel], imu_acc=world.imu_acc[sel], imu_bias=cv.packer.imu_bias_index(imu_t, ft),
                  bc_i=np.arange(WIN - 1), bc_j=np.arange(1, WIN), bc_w=cv.packer.bias_chain_sqrt_info(imu_t, ft, 2.0e-5, 4.0e-4),
                  v_lm=new_of[world.v_lm[ok_v]], v_ti=world.v_ti[ok_v], v_tj=world.v_tj[ok_v], v_rowi=world.v_rowi[ok_v],
                  v_rowj=world.v_rowj[ok_v], v_pi=world.v_pi[ok_v], v_pj=world.v_pj[ok_v])
    if prior is not None:
        idx = np.array([i - kmin if kd in (PK_ROT, PK_POS) else (i - k if kd in (PK_BG, PK_BA) else 0) for kd, i in zip(prior["kind"], prior["index"])])
        w.pJ0, w.pr0, w.p_kind, w.p_index, w.p_off, w.p_x0 = prior["J0"], prior["r0"], prior["kind"], idx, prior["off"], prior["x0"]
    return w.normalize(), dict(kmin=kmin, lms=lms, frames=ft, a_frame=a_frame)


def marg_window_of(world, st, k, w_full, info):
    """The MARGIN_OLD factor set of window k as one more window + role[] (1 marginalise, 0 keep, -1 not involved)."""
    dt = world.dt_ns
    ft, kmin, lms = info["frames"], info["kmin"], info["lms"]
    ctrl_later = knot_index(ft[1], dt)
    m = w_full.copy()
    m.quat, m.pos, m.bias, m.ld = st.quat[kmin:kmin + w_full.K].copy(), st.pos[kmin:kmin + w_full.K].copy(), st.bias[k:k + WIN].copy(), st.ld
    sel = m.imu_t < ft[1]
    m.imu_t, m.imu_gyro, m.imu_acc = m.imu_t[sel], m.imu_gyro[sel], m.imu_acc[sel]
    m.imu_bias = np.zeros(int(sel.sum()), np.int32)
    m.bc_i, m.bc_j, m.bc_w = m.bc_i[:1], m.bc_j[:1], m.bc_w[:1]
    drop_lm = np.where(info["a_frame"][lms] == k)[0]                 # landmarks anchored at the oldest frame
    new_of = -np.ones(lms.size, np.int64); new_of[drop_lm] = np.arange(drop_lm.size)
    keep_v = new_of[w_full.v_lm] >= 0
    for name in ("v_ti", "v_tj", "v_rowi", "v_rowj", "v_pi", "v_pj"):
        setattr(m, name, getattr(w_full, name)[keep_v].copy())
    m.v_lm = new_of[w_full.v_lm[keep_v]]
    m.rho = st.rho[lms[drop_lm]].copy()
    m.cauchy_a = 1.0
    m.normalize()
    K, F, P = m.K, m.F, m.P
    role = -np.ones(m.N, np.int8)

    def involve(u, n):
        role[u:u + n] = np.maximum(role[u:u + n], 0)
    for t in m.imu_t:
        involve(6 * (knot_index(t, dt) - kmin), 24)
    role[6 * K:6 * K + 6] = 1; involve(6 * K + 6, 6)               # bias_k dropped, bias_{k+1} kept
    for t in np.concatenate([m.v_ti, m.v_tj]):
        s0, s1 = knot_index(t, dt) - kmin, knot_index(t + PAD_NS, dt) - kmin
        involve(6 * s0, 6 * (s1 + 4 - s0))
    if m.V:
        involve(P - 1, 1)
    role[P:] = 1
    for kd, i in zip(m.p_kind, m.p_index):
        u = {PK_ROT: 6 * i, PK_POS: 6 * i + 3, PK_BG: 6 * K + 6 * i, PK_BA: 6 * K + 6 * i + 3, PK_LD: P - 1}[int(kd)]
        involve(u, 1 if kd == PK_LD else 3)
    for kk in range(K):
        if kmin + kk < ctrl_later:
            sl = role[6 * kk:6 * kk + 6]
            sl[sl >= 0] = 1
    return m, role


def prior_from(m, info, k, kept, J0, r0):
    """kept unknowns of the marginalisation window -> the next window's prior in world indices, linearised at m's state."""
    kmin, K = info["kmin"], m.K
    kinds, idxs, offs, x0 = [], [], [], []
    kept = [int(u) for u in kept]
    j = 0
    while j < len(kept):
        u = kept[j]
        if u == m.P - 1:
            kinds.append(PK_LD); idxs.append(0); offs.append(j); x0.append([m.ld, 0, 0, 0]); j += 1
            continue
        assert kept[j:j + 3] == [u, u + 1, u + 2]
        if u < 6 * K:
            kk, part = divmod(u, 6)
            kinds.append(PK_ROT if part == 0 else PK_POS); idxs.append(kmin + kk)
            x0.append(m.quat[kk].tolist() if part == 0 else m.pos[kk].tolist() + [0.0])
        else:
            f, part = divmod(u - 6 * K, 6)
            kinds.append(PK_BG if part == 0 else PK_BA); idxs.append(k + f)
            x0.append(m.bias[f, part:part + 3].tolist() + [0.0])
        offs.append(j); j += 3
    return dict(J0=np.array(J0, float), r0=np.array(r0, float), kind=np.array(kinds, np.int32), index=np.array(idxs, np.int64),
                off=np.array(offs, np.int32), x0=np.array(x0, float).reshape(-1, 4))


def initial_prior(world):
    return dict(J0=world.pJ0.copy(), r0=world.pr0.copy(), kind=world.p_kind.copy(), index=world.p_index.astype(np.int64),
                off=world.p_off.copy(), x0=world.p_x0.copy())


class DeviceBackend:
    def __init__(self, precision="fp64"):
        self.s = cv.Solver(precision=precision)

    def solve_and_restore(self, w, knot, iters=15):
        q0, t0 = w.quat[knot].copy(), w.pos[knot].copy()
        self.s.set_windows([w])
        sm = self.s.solve(iters, writeback=False)[0]
        self.s.gauge_restore([0], [knot], q0[None], t0[None])
        self.s.get_state(0, into=w)
        return sm

    def marginalize(self, m, role):
        self.s.set_windows([m])
        return self.s.marginalize(0, role, 1e-8)


class OracleBackend:
    def __init__(self):
        import pyctvo
        self.o = pyctvo

    def solve_and_restore(self, w, knot, iters=15):
        q0, t0 = w.quat[knot].copy(), w.pos[knot].copy()
        sm = self.o.OracleWindow(w).solve(iters)
        w.quat, w.pos = self.o.gauge_restore(w.quat, w.pos, knot, q0, t0)
        return dict(iterations=sm.iterations, final_cost=sm.final_cost)

    def marginalize(self, m, role):
        return self.o.OracleWindow(m).marginalize(role, 1e-8)


def run_slide(world, backend, nwin=3):
    """Returns (final State, list of per-window records)."""
    st = State(world)
    prior = initial_prior(world)
    rec = []
    for k in range(nwin):
        w, info = window_of(world, st, k, prior)
        sm = backend.solve_and_restore(w, 0)
        kmin, lms = info["kmin"], info["lms"]
        st.quat[kmin:kmin + w.K], st.pos[kmin:kmin + w.K] = w.quat, w.pos
        st.bias[k:k + WIN], st.rho[lms], st.ld = w.bias, w.rho, float(w.ld)
        rec.append(dict(k=k, iterations=sm["iterations"], final_cost=sm["final_cost"], K=w.K, L=w.L, V=w.V, M=w.M, pn=w.pn))
        if k + 1 < nwin:
            m, role = marg_window_of(world, st, k, w, info)
            kept, J0, r0 = backend.marginalize(m, role)
            prior = prior_from(m, info, k, kept, J0, r0)
            rec[-1].update(n_keep=int(len(kept)), n_drop=int((role == 1).sum()))
    return st, rec


def state_error(a, b):
    """relative error of State a against b over the knots / biases / depths that differ from the start, max over groups"""
    def rel(x, y):
        return float(np.linalg.norm(np.ravel(x) - np.ravel(y)) / max(np.linalg.norm(np.ravel(y)), 1e-30))
    qa = a.quat * np.sign(np.sum(a.quat * b.quat, axis=1, keepdims=True) + 1e-300)
    return dict(quat=rel(qa, b.quat), pos=rel(a.pos, b.pos), bias=rel(a.bias, b.bias), rho=rel(a.rho, b.rho), ld=abs(a.ld - b.ld) / max(abs(b.ld), 1e-30))
