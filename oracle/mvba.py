"""ORACLE (test infrastructure only - never imported by the product path): CPU restatement of the multi-view
bundle adjustment the reference delegates to Ceres.

Follows pose_optimization/multi_view/bundle_adjustment/problem/include/ba_problem.h:60-151 (the two residual
functors), problem/src/ba_problem.cpp:8-88 (CSV parser), :98-113 (WriteResult), :115-157 (Solve: DENSE_SCHUR,
squared loss, Ceres defaults).  Ceres 2.0 is an un-vendored dependency that is absent here, so the minimiser is a
restatement of its documented Levenberg-Marquardt trust-region loop (Ceres "Solving Non-linear Least Squares":
Jacobi column scaling fixed at the first iterate, lm diagonal = sqrt(clamp(diag(J^T J), 1e-6, 1e32) / radius),
step quality rho = cost change / model cost change, accept when rho > 1e-3, radius /= max(1/3, 1 - (2 rho - 1)^3)
on acceptance, radius /= 2, 4, 8, ... on rejection, initial radius 1e4, <= 50 iterations, function tolerance 1e-6,
gradient tolerance 1e-10, parameter tolerance 1e-8).  PARITY ANCHOR: the reference's own gtests
(problem/test/test_ba_problem.cpp:172-190) - known answers with tolerances - re-run in tests/test_mv_ba.py;
against Ceres itself parity is unpinned (no Ceres in the image).

Quirk kept (ba_problem.cpp:129-137, ba_problem.h:60-100): observations of the fixed camera are predicted with the
IDENTITY pose, whatever its row in the file says, and that camera's parameters are written back untouched.
"""
import numpy as np


def split_by_char(line, c=","):
    """io/src/file_utils.cpp:3-25 with allow_empty=False."""
    return [t for t in line.rstrip("\n").split(c) if t != ""]


def R_to_aa(R):
    """ceres::RotationMatrixToAngleAxis (via the quaternion); R is a proper 3x3 (row/col indexable)."""
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if tr >= 0:
        t = np.sqrt(tr + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1], q[2], q[3] = (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i + 1] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[j + 1] = (R[j, i] + R[i, j]) * t
        q[k + 1] = (R[k, i] + R[i, k]) * t
    s2 = q[1] ** 2 + q[2] ** 2 + q[3] ** 2
    if s2 > 0:
        s = np.sqrt(s2)
        two_theta = 2.0 * (np.arctan2(-s, -q[0]) if q[0] < 0 else np.arctan2(s, q[0]))
        return q[1:] * (two_theta / s)
    return 2.0 * q[1:]


def aa_to_R(w):
    """ceres::AngleAxisToRotationMatrix."""
    t2 = float(w @ w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if t2 > np.finfo(float).eps:
        th = np.sqrt(t2)
        k = K / th
        return np.eye(3) + np.sin(th) * k + (1 - np.cos(th)) * (k @ k)
    return np.eye(3) + K


def read_problem(path):
    """BaProblem::BaProblem (ba_problem.cpp:8-88): rows are classified by their field count."""
    hdr, cam_idx, pt_idx, obs, wts, cams, pts = None, [], [], [], [], [], []
    for line in open(path):
        el = split_by_char(line)
        n = len(el)
        if n == 8:
            hdr = dict(n_cams=int(el[0]), fixed=int(el[1]), n_pts=int(el[2]), n_obs=int(el[3]),
                       intr=np.array([float(x) for x in el[4:8]]))
        elif n == 3:
            pts.append([float(x) for x in el])
        elif 4 <= n <= 6:
            cam_idx.append(int(el[0]))
            pt_idx.append(int(el[1]))
            obs.append([float(el[2]), float(el[3])])
            wts.append([1.0, 1.0] if n == 4 else ([float(el[4])] * 2 if n == 5 else [float(el[4]), float(el[5])]))
        elif n == 12:
            R = np.array([float(x) for x in el[:9]]).reshape(3, 3).T  # column-major
            cams.append(np.concatenate([R_to_aa(R), [float(x) for x in el[9:]]]))
    return dict(hdr, cam_idx=np.array(cam_idx, np.int32), pt_idx=np.array(pt_idx, np.int32), obs=np.array(obs).reshape(-1, 2),
                wts=np.array(wts).reshape(-1, 2), cams=np.array(cams).reshape(-1, 6), pts=np.array(pts).reshape(-1, 3))


def write_result(path, cams):
    """BaProblem::WriteResult (ba_problem.cpp:98-113); setprecision(12) is sticky so it also covers t."""
    with open(path, "w") as f:
        for c in cams:
            R = aa_to_R(c[:3])
            f.write(",".join("%.12g" % x for x in R.T.reshape(-1)) + "," + ",".join("%.12g" % x for x in c[3:]) + "\n")


def _rotate(w, X):
    """ceres::AngleAxisRotatePoint and its derivative w.r.t. w (what autodiff yields; Gallego & Yezzi 2015 for the
    general branch, -[X]x for the first-order branch).  w [n,3], X [n,3] -> p [n,3], dp/dw [n,3,3], R [n,3,3]."""
    n = len(X)
    t2 = np.einsum("ni,ni->n", w, w)
    big = t2 > np.finfo(float).eps
    th = np.sqrt(np.where(big, t2, 1.0))
    k = w / th[:, None]

    def hat(v):
        H = np.zeros((len(v), 3, 3))
        H[:, 0, 1], H[:, 0, 2], H[:, 1, 0], H[:, 1, 2], H[:, 2, 0], H[:, 2, 1] = -v[:, 2], v[:, 1], v[:, 2], -v[:, 0], -v[:, 1], v[:, 0]
        return H

    Kh = hat(k)
    I = np.broadcast_to(np.eye(3), (n, 3, 3))
    Rb = I + np.sin(th)[:, None, None] * Kh + (1 - np.cos(th))[:, None, None] * (Kh @ Kh)
    Rs = I + hat(w)
    R = np.where(big[:, None, None], Rb, Rs)
    p = np.einsum("nij,nj->ni", R, X)
    Xh = hat(X)
    G = (np.einsum("ni,nj->nij", w, w) + (np.transpose(Rb, (0, 2, 1)) - I) @ hat(w)) / np.where(big, t2, 1.0)[:, None, None]
    Jb = -Rb @ Xh @ G
    J = np.where(big[:, None, None], Jb, -Xh)
    return p, J, R


def linearise(prob, cams, pts):
    """Residuals r [O,2] and Jacobians Jc [O,2,6] (zero rows for the fixed camera), Jp [O,2,3]."""
    ci, pi = prob["cam_idx"], prob["pt_idx"]
    fx, fy, cx, cy = prob["intr"]
    fixed = ci == prob["fixed"]
    c = cams[ci].copy()
    c[fixed] = 0.0  # identity pose for the fixed camera (ba_problem.h:66-79)
    X = pts[pi]
    p, dpdw, R = _rotate(c[:, :3], X)
    p = p + c[:, 3:]
    iz = 1.0 / p[:, 2]
    r = np.stack([fx * p[:, 0] * iz + cx, fy * p[:, 1] * iz + cy], -1) - prob["obs"]
    w = prob["wts"]
    r = r * w
    dpr = np.zeros((len(ci), 2, 3))
    dpr[:, 0, 0], dpr[:, 0, 2] = fx * iz, -fx * p[:, 0] * iz * iz
    dpr[:, 1, 1], dpr[:, 1, 2] = fy * iz, -fy * p[:, 1] * iz * iz
    dpr = dpr * w[:, :, None]
    Jc = np.concatenate([dpr @ dpdw, dpr], -1)
    Jc[fixed] = 0.0
    Jp = dpr @ R
    return r, Jc, Jp


def solve(prob, max_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8):
    """Returns (cams [C,6], pts [P,3], summary dict).  The fixed camera's parameters are returned untouched."""
    ci, pi = prob["cam_idx"], prob["pt_idx"]
    C, P = len(prob["cams"]), len(prob["pts"])
    cams, pts = prob["cams"].astype(np.float64).copy(), prob["pts"].astype(np.float64).copy()
    free = np.array([c != prob["fixed"] for c in range(C)])
    radius, decrease, invalid = 1e4, 2.0, 0
    scale_c = scale_p = None
    summary = dict(iterations=0, termination="max_iterations")

    def blocks(Jc, Jp, r):
        U = np.zeros((C, 6, 6)); gc = np.zeros((C, 6)); V = np.zeros((P, 3, 3)); gp = np.zeros((P, 3))
        np.add.at(U, ci, np.einsum("oki,okj->oij", Jc, Jc))
        np.add.at(gc, ci, np.einsum("oki,ok->oi", Jc, r))
        np.add.at(V, pi, np.einsum("oki,okj->oij", Jp, Jp))
        np.add.at(gp, pi, np.einsum("oki,ok->oi", Jp, r))
        return U, gc, V, gp

    r, Jc, Jp = linearise(prob, cams, pts)
    cost = 0.5 * float((r * r).sum())
    summary["initial_cost"] = cost
    it = 0
    while True:
        U, gc, V, gp = blocks(Jc, Jp, r)
        dc, dp = np.einsum("cii->ci", U).copy(), np.einsum("pii->pi", V).copy()
        if scale_c is None:
            scale_c, scale_p = 1.0 / (1.0 + np.sqrt(dc)), 1.0 / (1.0 + np.sqrt(dp))
        gmax = max(np.abs(gc[free]).max(initial=0.0), np.abs(gp).max(initial=0.0))
        if gmax <= gradient_tolerance:
            summary["termination"] = "gradient_tolerance"
            break
        if it >= max_iterations:
            break
        it += 1
        lam_c = np.clip(dc * scale_c ** 2, 1e-6, 1e32) / radius / scale_c ** 2
        lam_p = np.clip(dp * scale_p ** 2, 1e-6, 1e32) / radius / scale_p ** 2
        Vd = V + np.einsum("pi,ij->pij", lam_p, np.eye(3))
        Vinv = np.linalg.inv(Vd)
        W = np.einsum("oki,okj->oij", Jc, Jp)  # [O,6,3]
        Y = W @ Vinv[pi]
        n = 6 * C
        S = np.zeros((C, 6, C, 6)); rhs = -gc.copy()
        for c in range(C):
            S[c, :, c, :] = U[c] + np.diag(lam_c[c])
        # points couple the cameras that see them
        order = np.argsort(pi, kind="stable")
        starts = np.searchsorted(pi[order], np.arange(P + 1))
        for p_ in range(P):
            oo = order[starts[p_]:starts[p_ + 1]]
            for a in oo:
                rhs[ci[a]] += Y[a] @ gp[p_]
                for b in oo:
                    S[ci[a], :, ci[b], :] -= Y[a] @ W[b].T
        S = S.reshape(n, n); rhs = rhs.reshape(n)
        keep = np.repeat(free, 6)
        step_c = np.zeros(n)
        ok = True
        try:
            L = np.linalg.cholesky(S[np.ix_(keep, keep)])
            step_c[keep] = np.linalg.solve(L.T, np.linalg.solve(L, rhs[keep]))
        except np.linalg.LinAlgError:
            ok = False
        step_c = step_c.reshape(C, 6)
        if ok:
            acc = gp.copy()
            np.add.at(acc, pi, np.einsum("oij,oi->oj", W, step_c[ci]))
            step_p = -np.einsum("pij,pj->pi", Vinv, acc)
            m = np.einsum("oki,oi->ok", Jc, step_c[ci]) + np.einsum("oki,oi->ok", Jp, step_p[pi])
            model_change = -float((m * (r + 0.5 * m)).sum())
            ok = model_change > 0.0
        if not ok:
            invalid += 1
            if invalid >= 5:
                summary["termination"] = "invalid_steps"
                break
            radius /= decrease
            decrease *= 2.0
            continue
        invalid = 0
        step_norm = np.sqrt((step_c[free] ** 2).sum() + (step_p ** 2).sum())
        x_norm = np.sqrt((cams[free] ** 2).sum() + (pts ** 2).sum())
        if step_norm <= parameter_tolerance * (x_norm + parameter_tolerance):
            summary["termination"] = "parameter_tolerance"
            break
        cand_c, cand_p = cams + step_c * free[:, None], pts + step_p
        r2, Jc2, Jp2 = linearise(prob, cand_c, cand_p)
        cand_cost = 0.5 * float((r2 * r2).sum())
        change = cost - cand_cost
        # Ceres tests the function tolerance BEFORE it accepts the step (TrustRegionMinimizer::Minimize: ParameterTolerance-
        # Reached, FunctionToleranceReached, then IsStepSuccessful): the iterate stays at the previous point
        if abs(change) <= function_tolerance * cost:
            summary["termination"] = "function_tolerance"
            break
        rho = change / model_change
        if rho > 1e-3:
            cams, pts, r, Jc, Jp, cost = cand_c, cand_p, r2, Jc2, Jp2, cand_cost
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            decrease = 2.0
        else:
            radius /= decrease
            decrease *= 2.0
        if radius < 1e-32:
            summary["termination"] = "radius"
            break
    summary["iterations"] = it
    summary["final_cost"] = cost
    return cams, pts, summary


def triangulate_dlt(P0, P1, x0, x1):
    """cv2.triangulatePoints as used at bundle_adjust_io.py:226-227 (OpenCV is absent here; its algorithm is the
    homogeneous DLT: rows x*P[2]-P[0], y*P[2]-P[1] of both views, null vector by SVD).  P [3,4]; x [n,2] -> xyz [n,3]."""
    out = np.zeros((len(x0), 3))
    for i in range(len(x0)):
        A = np.stack([x0[i, 0] * P0[2] - P0[0], x0[i, 1] * P0[2] - P0[1], x1[i, 0] * P1[2] - P1[0], x1[i, 1] * P1[2] - P1[1]])
        X = np.linalg.svd(A)[2][-1]
        out[i] = X[:3] / X[3]
    return out
