"""NumPy emulation of the DEVICE algebra, driven only by eftpipe_amd.tables.build_tables output.

Test-only: lets the CPU suite prove that the real-reduced / operator-folded tables reproduce the
oracle before (and independently of) the HIP kernels, which implement exactly these contractions.
"""
import numpy as np

NH = 128


def tail_params(lnk, y):
    slope = (np.log(y[-1]) - np.log(y[-2])) / (lnk[-1] - lnk[-2])
    amp = y[-1] / np.exp(slope * lnk[-1])
    return amp, slope


def coef_half(t, Pin):
    lnk = np.log(t["kin"])
    amp, slope = tail_params(lnk, Pin)
    tail = amp * np.exp(slope * t["lnx_tail"])
    return t["Gc"] @ Pin + t["Ec"] @ tail  # [2,129]


def antidiagonal_sums(t, c):
    """c [2,129] -> S[nc, 257] complex: S[c, j'] = sum_{n+m=256+j'} c_n c_m M_c[n,m] (device: antidiag_kernel)."""
    ch = c[0] + 1j * c[1]
    full = np.concatenate([ch, np.conj(ch[:NH][::-1])])  # c_{256-n} = conj(c_n)
    AD = t["ad"]
    S = np.zeros(AD.shape[:2], dtype=complex)
    for jp in range(AD.shape[1]):
        cnt = ((256 + jp) >> 1) - jp + 1
        tt = np.arange(cnt)
        S[:, jp] = AD[:, jp, :cnt] @ (full[jp + tt] * full[256 - tt])
    return S


def rows_from_sums(Z):
    """complex [rows, 257] -> real synthesis rows [rows, 528]: (Re Z_0, Re Z_1, Im Z_1, ...) zero padded (build_rows_kernel)."""
    out = np.zeros((Z.shape[0], 528))
    out[:, 0] = Z[:, 0].real
    out[:, 1:513:2] = Z[:, 1:].real
    out[:, 2:513:2] = Z[:, 1:].imag
    return out


def rows_from_vectors(c, V):
    """single sums: X_n = c_n V_n (n = 0..128) -> rows (Re X_128, Re X_127, Im X_127, ..., Re X_0, Im X_0, 0, 0, 0)"""
    X = (c[0] + 1j * c[1])[None, :] * V
    out = np.zeros((V.shape[0], 288))
    out[:, 0] = X[:, NH].real
    out[:, 1:257:2] = X[:, NH - 1 :: -1].real
    out[:, 2:257:2] = X[:, NH - 1 :: -1].imag
    return out


def pscf(t, Pin, with_cf):
    k, s = t["k"], t["s"]
    P11 = t["Sk"] @ Pin
    c = coef_half(t, Pin)
    S = antidiagonal_sums(t, c)
    nb = t["comb22"].shape[1]
    out = dict(P11=P11)
    out["P22"] = t["comb22"] @ (rows_from_sums(S[:nb]) @ t["syn_k"])
    L = t["linvec"]
    out["P13"] = P11[None, :] * (rows_from_vectors(c, L[:10]) @ t["lin_k"])
    if with_cf:
        Nl = t["l11"].shape[0]
        # device order: synthesise the Nl*(nb + nb13) weighted basis rows, then expand (expand_kernel)
        basis = np.concatenate([rows_from_sums(t["mlj"][l][None, :] * S) @ t["syn_s"] for l in range(Nl)])
        full = t["expand_c"] @ basis
        out["C22"] = full[: Nl * 28].reshape(Nl, 28, -1)
        out["C13"] = full[Nl * 28 :].reshape(Nl, 10, -1)
        out["C11"] = rows_from_vectors(c, L[10 : 10 + Nl]) @ t["lin_s"]
        out["Cct"] = s[None, :] ** -2 * (rows_from_vectors(c, L[10 + Nl :]) @ t["lin_s"])
    return out


def regroup(t, f, T22, T13):
    out = np.zeros(T22.shape[:1] + (12,) + T22.shape[2:])
    for b, (g, p) in enumerate(t["grp22"]):
        out[:, g] += f**p * T22[:, b]
    for b, (g, p) in enumerate(t["grp13"]):
        out[:, g] += f**p * T13[:, b]
    return out


def setpscfl(t, f, st, with_cf):
    k = t["k"]
    Nl = t["l11"].shape[0]
    P11l = t["l11"][:, :, None] * st["P11"][None, None, :]
    Pctl = t["lct"][:, :, None] * (k**2 * st["P11"])[None, None, :]
    Ploopl = regroup(t, f, t["l22"][:, :, None] * st["P22"][None], t["l13"][:, :, None] * st["P13"][None])
    Ploopl = Ploopl - Ploopl[:, :, :1]
    Pstl = np.zeros((Nl, 3, k.size))
    Pstl[0, 0] = 1.0
    Pstl[0, 1] = k**2
    Pstl[1, 2] = k**2
    out = dict(P11l=P11l, Pctl=Pctl, Ploopl=Ploopl, Pstl=Pstl)
    if with_cf:
        out["Cloopl"] = regroup(t, f, t["l22"][:, :, None] * st["C22"], t["l13"][:, :, None] * st["C13"])
    return out


def ir_filters(t, Pin):
    q2 = Pin[-2:] * t["wq_last2"]
    amp, slope = tail_params(np.log(t["kin"]), np.concatenate([[1.0] * (len(t["kin"]) - 2), q2]))
    tail = amp * np.exp(slope * t["lnx_xtail"])
    return t["BX"] @ Pin + t["TX"] @ tail, t["BY"] @ Pin + t["TY"] @ tail


def resum(t, f, Pin, st):
    NIR, Na, Nklow = t["resum_dims"]
    Nl = t["l11"].shape[0]
    k = t["k"]
    Q = (t["Qpoly"] @ f ** np.arange(15))[::-1]  # Q[a] = table[1-a]
    X, Y = ir_filters(t, Pin)
    z = k[:, None] ** 2 * X[None, :]  # [Nk,Ns]
    yk = k[:, None] ** 2 * Y[None, :]
    # Z[a,l,l',v,k,s] = sum_p Q[a,l,l',p*Na+v] z^(p+1) + yk * sum_p Q[a,l,l',(NIR+p)*Na+v] z^p
    Qr = Q.reshape(2, Nl, Nl, 2 * NIR, Na)
    zp = np.stack([z**p for p in range(NIR + 1)])  # [NIR+1,Nk,Ns]
    Z = np.einsum("albpv,pks->albvks", Qr[:, :, :, :NIR], zp[1:]) + np.einsum(
        "albpv,pks->albvks", Qr[:, :, :, NIR:], zp[:NIR]
    ) * yk
    W = np.einsum("vks,albvks->albks", t["H"], Z)  # [2,Nl,Nl,Nk,Ns]
    out = dict(st)
    out["P11l"] = st["P11l"] + np.einsum("lpks,ps,pi->lik", W[0], st["C11"], t["l11"])
    out["Pctl"] = st["Pctl"] + np.einsum("lpks,ps,pi->lik", W[1], st["Cct"], t["lct"])
    out["Ploopl"] = st["Ploopl"] + np.einsum("lpks,pis->lik", W[1], st["Cloopl"])
    out["X"], out["Y"] = X, Y
    return out


def resum_mfma(t, f, Pin, st):
    """The matrix-core form of the same stage (device: resum_prep_kernel + resum_mfma_kernel / resum_mfma2_kernel): polynomials as
    A[rows, 8] x beta[8, points] in t = z / RS_ZS.  Nl = 3: the operand is built per s with X(s), Y(s)
    folded into its rows, three rows per (a, l, l') block, 64 rows; Nl = 2: 32 rows of four slots -- see tables.resum_mfma_tables."""
    from eftpipe_amd.tables import RS_ZS, resum_block

    NIR, Na, Nklow = t["resum_dims"]
    Nl = 3 if NIR == 16 else 2
    k = t["k"]
    Q = (t["Qpoly"] @ f ** np.arange(15))[::-1].reshape(-1)
    X, Y = ir_filters(t, Pin)
    z = k[:, None] ** 2 * X[None, :]
    yk = k[:, None] ** 2 * Y[None, :]
    tt = z / RS_ZS
    beta = np.einsum("rp,pks->rks", t["rs_basis"][:, :NIR], np.stack([tt**p for p in range(NIR)]))  # [8,Nk,Ns]
    H = t["H"]
    out = {n: st[n].copy() for n in ("P11l", "Pctl", "Ploopl")}
    if Nl == 3:
        rows = t["rs_rows"].reshape(2, 64)
        V8S = t["rs_basis_scaled"][:, :NIR]
        aX, aY = np.zeros((64, 8)), np.zeros((64, 8))
        for r in range(64):
            if rows[0, r] >= 0:
                aX[r] = V8S @ Q[rows[0, r] + np.arange(NIR) * Na]
            if rows[1, r] >= 0:
                aY[r] = V8S @ Q[rows[1, r] + np.arange(NIR) * Na]
        As = aX[:, :, None] * X[None, None, :] + aY[:, :, None] * Y[None, None, :]  # [64,8,Ns]
        D = np.einsum("irs,rks->iks", As, beta).reshape(4, 4, 4, *z.shape)  # [tau, slot, jg, k, s]
        k2 = k[:, None] ** 2
        for tau in range(4):
            for jg in range(4):
                for slot in range(4):
                    blk = resum_block(tau, jg, slot)
                    if blk is None:
                        assert not np.any(D[tau, slot, jg])
                        continue
                    a, l, lp, v = blk
                    W = k2 * H[v] * D[tau, slot, jg]  # [Nk,Ns]
                    if a == 0:
                        out["P11l"][l] += np.einsum("ks,s,i->ik", W, st["C11"][lp], t["l11"][lp])
                    else:
                        out["Pctl"][l] += np.einsum("ks,s,i->ik", W, st["Cct"][lp], t["lct"][lp])
                        out["Ploopl"][l] += np.einsum("ks,is->ik", W, st["Cloopl"][lp])
    else:
        rows = t["rs_rows"]
        A = np.zeros((80, 8))
        for r in np.nonzero(rows[:80] >= 0)[0]:
            A[r] = t["rs_basis_scaled"][:, :NIR] @ Q[rows[r] + np.arange(NIR) * Na]
        D = np.einsum("ir,rks->iks", A, beta).reshape(5, 4, 4, *z.shape)  # [tau, slot, jg, k, s]
        for tau in range(2):
            for jg in range(4):
                lp = tau
                a, l = ((1, 0), (1, 1), (0, 0), (0, 1))[jg]
                d = D[tau, :, jg]
                W = z * H[lp] * d[0] + yk * (H[0] * d[1] + H[1] * d[2])  # [Nk,Ns]
                if a == 0:
                    out["P11l"][l] += np.einsum("ks,s,i->ik", W, st["C11"][lp], t["l11"][lp])
                else:
                    out["Pctl"][l] += np.einsum("ks,s,i->ik", W, st["Cct"][lp], t["lct"][lp])
                    out["Ploopl"][l] += np.einsum("ks,is->ik", W, st["Cloopl"][lp])
    out = dict(st, **out)
    out["X"], out["Y"] = X, Y
    return out


def resum_plk(t, f, Pin, st, bias):
    """The resummation correction of a direct-P_l run (device: resum_prep_plk / resum_plk kernels, Nl = 3): with the bias contraction taken first,
    dP_l(k) = k^2 sum_s sum_v H_v(k,s) D_lv(k^2 X(s) / RS_ZS; s),  D_lv = sum_p t^p RS_ZS^p sum_l' sum_a g_a[l'](s) (delta(v,l') X(s) Q_a[l,l',(0,p,v)] +
    Y(s) Q_a[l,l',(1,p,v)]),  g_0 = C11 (b . l11),  g_1 = Cct (b . lct) + b . Cloopl.   -> [Nl, Nk]"""
    from eftpipe_amd.tables import RS_ZS

    NIR, Na, _ = t["resum_dims"]
    Nl = 3
    k = t["k"]
    Qd = (t["Qpoly"] @ f ** np.arange(15))[::-1].reshape(2, Nl, Nl, 2, NIR, Na)  # device a (0: the C11 series), l, l', half, p, v
    X, Y = ir_filters(t, Pin)
    b = np.asarray(bias)
    g = np.stack([st["C11"] * (t["l11"] @ b[0:3])[:, None],
                  st["Cct"] * (t["lct"] @ b[3:9])[:, None] + np.einsum("i,lis->ls", b[9:21], st["Cloopl"])])  # [a, l', s]
    term = Y[None, None, None, None, None, :] * Qd[:, :, :, 1, :, :, None]  # [a, l, l', p, v, s]
    for lp in range(Nl):
        term[:, :, lp, :, lp, :] += X[None, None, None, :] * Qd[:, :, lp, 0, :, lp, None]
    coef = np.einsum("aps,alpqvs->lvqs", g, term) * (RS_ZS ** np.arange(NIR))[None, None, :, None]  # [l, v, p, s]
    tt = k[:, None] ** 2 * X[None, :] / RS_ZS
    D = np.zeros((Nl, Na) + tt.shape)
    for p in range(NIR - 1, -1, -1):  # Horner, as the kernel does
        D = D * tt + coef[:, :, p, None, :]
    return k[None, :] ** 2 * np.einsum("vks,lvks->lk", t["H"], D)


def spline_derivs(t, y):
    """y [..., Nk] -> knot derivatives via the banded operator (device: spline_kernel)."""
    band = t["sp_band"]
    hb = (band.shape[0] - 1) // 2
    n = y.shape[-1]
    yp = np.concatenate([np.zeros(y.shape[:-1] + (hb,)), y, np.zeros(y.shape[:-1] + (hb,))], axis=-1)
    sd = np.zeros_like(y)
    for d in range(band.shape[0]):
        sd += band[d] * yp[..., d : d + n]
    dx = np.diff(t["k"])
    return sd, np.diff(y, axis=-1) / dx


def spline_eval(t, y, sd, slope, xe):
    k = t["k"]
    dx = np.diff(k)
    i = np.clip(np.searchsorted(k, xe, side="right") - 1, 0, k.size - 2)
    tt = xe - k[i]
    c3 = (sd[..., i] + sd[..., i + 1] - 2 * slope[..., i]) / dx[i] ** 2
    c2 = (slope[..., i] - sd[..., i]) / dx[i] - c3 * dx[i]
    return y[..., i] + tt * (sd[..., i] + tt * (c2 + tt * c3))


def ap(t, DA, H, st, names):
    Nl = t["l11"].shape[0]
    k, mu, wmu = t["k"], t["mu"], t["wmu"]
    DAf, Hf = t["ap_fid"]
    qperp, qpar = DA / DAf, Hf / H
    F = qpar / qperp
    root = np.sqrt(1 + mu**2 * (F**-2 - 1))
    kp = k[:, None] / qperp * root[None, :]
    mup = mu / F / root
    x2 = mup**2
    legp = np.stack([np.ones_like(x2), 0.5 * (3 * x2 - 1), (35 * x2**2 - 30 * x2 + 3) / 8.0])[:Nl]
    out = dict(st)
    for n in names:
        y = st[n]
        sd, slope = spline_derivs(t, y)
        val = spline_eval(t, y, sd, slope, kp)  # [Nl,rows,Nk,Nmu]
        pkmu = np.einsum("lrkm,lm->rkm", val, legp)
        out[n] = 2.0 / (qperp**2 * qpar) * np.einsum("rkm,lm,m->lrk", pkmu, t["legmu"], wmu)
    return out
