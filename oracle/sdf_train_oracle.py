"""CPU oracle for the fused training-mode SDF network (gens_sdf_train_{fwd,bwd}) -- TEST INFRASTRUCTURE.

Two restatements of what one training step asks of SDFNetwork (/root/reference/models/modules/sdf_network.py:98-154, driven from
implicit_surface.py:179-191,257,305,490): the value y = sdf(x), its gradient g = dy/dx, the `smooth` vector s = d(sum_k g_k)/dx, and
the derivatives of a scalar loss L(y, g, s) with respect to every layer's (weight-normed) matrix, bias and the volume pyramid:

* `by_autograd`  -- torch autograd over the functional MLP of render_oracle with the sampler truncated at second order exactly like
  the reference's Function pair (gens_oracle.lookup_volume_truncated; cuda_gridsample.py:110-123).  This is the reference's own
  computation, pinned by goldens g17 / g17b / g18 / g18b (scripts/probe/oracle_vs_reference_training.py: 2e-5 of the reference).
* `by_sweeps`    -- the same quantities written out as the eight layer sweeps the HIP kernels execute:
  forward value / tangent (a, a'), reverse adjoint / its tangent (lambda, mu), and for the loss backward the tangent along the
  cotangent of s (nu), the mixed second tangent (kappa), and the two reverse sweeps (rho, omega).  Checked against `by_autograd`
  in tests/test_sdf_train_oracle.py; the device kernels are checked against both.

Shapes: pts (N,3); weights W[l] (out_l, in_l) effective (weight norm applied), biases b[l]; volumes list of (1,4,X,Y,Z).
"""
import math

import torch
import torch.nn.functional as F

from . import gens_oracle as K

SQ2 = 1.0 / math.sqrt(2.0)


def _embed(x, n_freq):
    parts = [x]
    for i in range(n_freq):
        parts += [torch.sin(x * 2.0 ** i), torch.cos(x * 2.0 ** i)]
    return torch.cat(parts, -1)


def sdf_functional(W, b, pts, volumes, lookup, skip_in=(3,)):
    """SDFNetwork.sdf (sdf_network.py:98-126, scale = 1) from effective weights: -> (N,1)."""
    fe = _embed(lookup(volumes, pts), 2)
    xe = _embed(pts, 4)
    h = xe
    n_lin = len(W)
    for l in range(n_lin):
        if l in skip_in:
            h = torch.cat([h, xe], -1) * SQ2
        if l > 0:
            h = torch.cat([h, fe], -1)
        h = h @ W[l].t() + b[l]
        if l < n_lin - 1:
            h = F.softplus(h, beta=100)
    return h[:, :1]


def by_autograd(W, b, volumes, pts, y_bar, g_bar, s_bar):
    """-> dict(y, g, s, dW [7], db [7], dvol [L]) with the reference's truncation (third order through the sampler dropped)."""
    W = [w.detach().clone().requires_grad_(True) for w in W]
    b = [v.detach().clone().requires_grad_(True) for v in b]
    vols = [v.detach().clone().requires_grad_(True) for v in volumes]
    x = pts.detach().clone().requires_grad_(True)
    y = sdf_functional(W, b, x, vols, K.lookup_volume_truncated)
    g = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)[0]
    s = torch.autograd.grad(g, x, torch.ones_like(g), create_graph=True)[0]
    loss = (y * y_bar).sum() + (g * g_bar).sum() + (s * s_bar).sum()
    grads = torch.autograd.grad(loss, W + b + vols, allow_unused=True)
    grads = [torch.zeros_like(t) if gr is None else gr for gr, t in zip(grads, W + b + vols)]
    n = len(W)
    return dict(y=y.detach(), g=g.detach(), s=s.detach(), dW=grads[:n], db=grads[n:2 * n], dvol=grads[2 * n:])


# ----------------------------------------------------------------------------------------------------------------------
# the sweeps
# ----------------------------------------------------------------------------------------------------------------------
def _softplus_derivs(a):
    """softplus(beta = 100, threshold 20) and its first three derivatives as torch's autograd forms them: above the threshold
    the function is the identity (derivatives 1, 0, 0)."""
    lin = a * 100.0 > 20.0
    t = torch.sigmoid(100.0 * a)
    h = torch.where(lin, a, F.softplus(a, beta=100))
    d1 = torch.where(lin, torch.ones_like(a), t)
    d2 = torch.where(lin, torch.zeros_like(a), 100.0 * t * (1 - t))
    d3 = torch.where(lin, torch.zeros_like(a), 1.0e4 * t * (1 - t) * (1 - 2 * t))
    return h, d1, d2, d3


def _pe_derivs(x):
    """P(x) (27), and per encoding entry: its coordinate a(i), P_i'(x_a), P_i''(x_a)."""
    axis = torch.arange(3).repeat(9)                       # entries come in blocks of 3 coordinates
    xa = x[:, axis]
    freq = torch.tensor([1.0] * 3 + [f for k in range(4) for f in [2.0 ** k] * 6])
    kind = torch.tensor([0] * 3 + [k for _ in range(4) for k in (1, 1, 1, 2, 2, 2)])     # 0 identity, 1 sin, 2 cos
    arg = xa * freq
    p = torch.where(kind == 0, xa, torch.where(kind == 1, torch.sin(arg), torch.cos(arg)))
    d1 = torch.where(kind == 0, torch.ones_like(xa), torch.where(kind == 1, freq * torch.cos(arg), -freq * torch.sin(arg)))
    d2 = torch.where(kind == 0, torch.zeros_like(xa), torch.where(kind == 1, -freq * freq * torch.sin(arg), -freq * freq * torch.cos(arg)))
    return p, d1, d2, axis


def _fe_derivs(f):
    """E(f) = [f, sin f, cos f, sin 2f, cos 2f] and the factors of E', E'', E''' per block (applied elementwise to a CF-vector)."""
    s1, c1, s2, c2 = torch.sin(f), torch.cos(f), torch.sin(2 * f), torch.cos(2 * f)
    e = torch.cat([f, s1, c1, s2, c2], -1)
    d1 = [torch.ones_like(f), c1, -s1, 2 * c2, -2 * s2]
    d2 = [torch.zeros_like(f), -s1, -c1, -4 * s2, -4 * c2]
    d3 = [torch.zeros_like(f), -c1, s1, -8 * c2, 8 * s2]
    return e, d1, d2, d3


def _blocks(v, cf):
    return [v[:, i * cf:(i + 1) * cf] for i in range(5)]


def by_sweeps(W, b, volumes, pts, y_bar, g_bar, s_bar):
    """The eight sweeps, layer by layer, as the kernels run them.  Same result dict as `by_autograd`."""
    n = pts.shape[0]
    dt = pts.dtype
    cf = 4 * len(volumes)
    one = torch.ones(n, 3, dtype=dt)
    nl = len(W) - 1                                        # hidden layers 0..nl-1, then the output row
    w6, b6 = W[nl][0], b[nl][0]
    hid = W[1].shape[0]

    # ---- prologue: encodings, sampler derivatives along v = (1,1,1), s_bar and g_bar ---------------------------------------------
    pe, pe1, pe2, axis = _pe_derivs(pts)
    f = K.lookup_volume(volumes, pts)
    zero_go = torch.zeros(n, cf, dtype=dt)
    f_dot = K.lookup_volume_bwd2(None, one, zero_go, volumes, pts)[0]                    # J v
    nu_f = K.lookup_volume_bwd2(None, s_bar, zero_go, volumes, pts)[0]                   # J s_bar
    kap_f0 = K.lookup_volume_bwd2(None, g_bar, zero_go, volumes, pts)[0]                 # J g_bar
    e, e1, e2, e3 = _fe_derivs(f)
    cat = lambda parts: torch.cat(parts, -1)  # noqa: E731
    e_dot = cat([d * f_dot for d in e1])
    nu_e = cat([d * nu_f for d in e1])
    kap_e = cat([d * kap_f0 for d in e1]) + cat([d * f_dot * nu_f for d in e2])
    pe_dot = pe1                                                                          # P' v  (v = 1 on every axis)
    nu_pe = pe1 * s_bar[:, axis]
    kap_pe = pe2 * s_bar[:, axis] + pe1 * g_bar[:, axis]

    # ---- forward sweeps: value, tangent along v, tangent along s_bar (nu), second tangent (kappa) --------------------------------
    z, zd, zn, zk = pe, pe_dot, nu_pe, kap_pe
    Z, A = [], []                                          # per layer: inputs (z, zd, zk, zn) and (d1, d2, d3, a_dot, nu_a, kap_a)
    for l in range(nl):
        if l in (3,):
            z, zd, zn, zk = (cat([t, u]) * SQ2 for t, u in ((z, pe), (zd, pe_dot), (zn, nu_pe), (zk, kap_pe)))
        if l > 0:
            z, zd, zn, zk = cat([z, e]), cat([zd, e_dot]), cat([zn, nu_e]), cat([zk, kap_e])
        Z.append((z, zd, zk, zn))
        a = z @ W[l].t() + b[l]
        ad, an, ak = zd @ W[l].t(), zn @ W[l].t(), zk @ W[l].t()
        h, d1, d2, d3 = _softplus_derivs(a)
        A.append((d1, d2, d3, ad, an, ak))
        z, zd, zn, zk = h, d1 * ad, d1 * an, d1 * ak + d2 * ad * an
    z6 = cat([z, e])
    kap_z6 = cat([zk, kap_e])
    y = (z6 @ w6 + b6)[:, None]

    # ---- reverse sweeps: lambda, mu (forward pass' g and s) and rho, omega (loss backward) ---------------------------------------
    lam_h = w6[:hid][None].expand(n, hid)
    mu_h = torch.zeros(n, hid, dtype=dt)
    rho_h = torch.zeros(n, hid, dtype=dt)
    om_h = y_bar * w6[:hid][None]
    lam_e = w6[hid:][None].expand(n, -1).clone()
    mu_e, rho_e = torch.zeros_like(lam_e), torch.zeros_like(lam_e)
    om_e = y_bar * w6[hid:][None]
    lam_pe = torch.zeros(n, 27, dtype=dt)
    mu_pe = torch.zeros(n, 27, dtype=dt)
    dW = [None] * (nl + 1)
    db = [None] * (nl + 1)
    dW[nl] = torch.zeros_like(W[nl])
    db[nl] = torch.zeros_like(b[nl])
    dW[nl][0] = (y_bar * z6 + kap_z6).sum(0)
    db[nl][0] = y_bar.sum()
    for l in range(nl - 1, -1, -1):
        d1, d2, d3, ad, an, ak = A[l]
        lam_a = d1 * lam_h
        mu_a = d2 * ad * lam_h + d1 * mu_h
        rho_a = d2 * an * lam_h + d1 * rho_h
        om_a = d1 * om_h + d3 * ad * an * lam_h + d2 * (mu_h * an + ad * rho_h + lam_h * ak)
        zl, zdl, zkl, znl = Z[l]
        dW[l] = om_a.t() @ zl + rho_a.t() @ zdl + lam_a.t() @ zkl + mu_a.t() @ znl
        db[l] = om_a.sum(0)
        lam_z, mu_z, rho_z, om_z = lam_a @ W[l], mu_a @ W[l], rho_a @ W[l], om_a @ W[l]
        if l == 0:
            lam_pe = lam_pe + lam_z
            mu_pe = mu_pe + mu_z
            break
        lam_e, mu_e, rho_e, om_e = lam_e + lam_z[:, hid:], mu_e + mu_z[:, hid:], rho_e + rho_z[:, hid:], om_e + om_z[:, hid:]
        lam_h, mu_h, rho_h, om_h = lam_z[:, :hid], mu_z[:, :hid], rho_z[:, :hid], om_z[:, :hid]
        if l == 3:
            k = W[2].shape[0]
            lam_pe, mu_pe = lam_h[:, k:] * SQ2, mu_h[:, k:] * SQ2
            lam_h, mu_h, rho_h, om_h = (t[:, :k] * SQ2 for t in (lam_h, mu_h, rho_h, om_h))

    # ---- epilogue: back through the encodings and the sampler ----------------------------------------------------------------------
    bl = lambda v: _blocks(v, cf)  # noqa: E731
    lam_f = sum(d * t for d, t in zip(e1, bl(lam_e)))
    mu_f = sum(d * t for d, t in zip(e1, bl(mu_e))) + f_dot * sum(d * t for d, t in zip(e2, bl(lam_e)))
    proj = lambda q: torch.stack([(q * (axis == a)).sum(-1) for a in range(3)], -1)  # noqa: E731
    gv1, jt_lam = K.lookup_volume_bwd(lam_f, volumes, pts)
    g = proj(pe1 * lam_pe) + jt_lam
    _, gv_lam_g, jd_lam = K.lookup_volume_bwd2(None, one, lam_f, volumes, pts)          # d/dp <J^T lam_f, v>   (constant in the backward)
    _, jt_mu = K.lookup_volume_bwd(mu_f, volumes, pts)
    s = proj(pe2 * lam_pe) + proj(pe1 * mu_pe) + jd_lam + jt_mu
    # volume gradient: three scatters
    f_hat = (sum(d * t for d, t in zip(e1, bl(om_e)))
             + f_dot * sum(d * t for d, t in zip(e2, bl(rho_e)))
             + nu_f * sum(d * t for d, t in zip(e2, bl(mu_e)))
             + nu_f * f_dot * sum(d * t for d, t in zip(e3, bl(lam_e)))
             + kap_f0 * sum(d * t for d, t in zip(e2, bl(lam_e))))
    dvol, _ = K.lookup_volume_bwd(f_hat, volumes, pts)
    dv_mu = K.lookup_volume_bwd2(None, s_bar, mu_f, volumes, pts)[1]
    dv_lam = K.lookup_volume_bwd2(None, g_bar, lam_f, volumes, pts)[1]
    dvol = [a + b_ + c for a, b_, c in zip(dvol, dv_mu, dv_lam)]
    return dict(y=y, g=g, s=s, dW=dW, db=db, dvol=dvol)


def shipped_weights(n_levels, seed=0, scale=0.3, dtype=torch.float32):
    """Random effective weights of the shipped architecture (confs/gens.conf:69-86): 27 -> 128, then (128 + 20 L) -> 128 with the
    101-wide layer 2, and the 129-row output layer of which only row 0 matters here."""
    g = torch.Generator().manual_seed(seed)
    fe = 20 * n_levels
    shapes = [(128, 27), (128, 128 + fe), (101, 128 + fe), (128, 128 + fe), (128, 128 + fe), (128, 128 + fe), (129, 128 + fe)]
    W = [(torch.randn(s, generator=g, dtype=dtype) * scale / math.sqrt(s[1])) for s in shapes]
    b = [torch.randn(s[0], generator=g, dtype=dtype) * 0.02 for s in shapes]
    return W, b
