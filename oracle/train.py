"""ORACLE — TEST INFRASTRUCTURE ONLY (not the product path).

CPU restatement of one optimisation step of the reference's train_hyper.py (174-214) with torch autograd
standing in for tf.GradientTape:

  forward   train_hyper.py:184-191   y=A(x), z=HE(y), z~=z+U, (loc,scale)=HD(z~), scale=max(scale,lb), y~=y+U, x~=S(y~)
  rates     train_hyper.py:193-196   bpp = sum(log p) / (-ln2 * num_points)
  BCE       loss.py:8-33             empty / full means of -log(1-o), -log(o), o = clip(sigmoid(x~), 1e-7, 1-1e-7)
  loss      train_hyper.py:197-199   alpha*(beta*empty + full) + delta*bpp_y + gamma*bpp_z
  Adam      tf.train.AdamOptimizer defaults (TF1 form): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps)

Gradient conventions restated from TensorFlow: tf.maximum passes the gradient to its first argument where
x >= y; tf.abs -> sign(x); tf.sign has zero gradient; clip_by_value passes gradient only strictly inside...
(TF: where (x >= min) & (x <= max)); tf.math.abs in the likelihood -> sign().  *** PARITY UNPINNED ***
(TensorFlow absent; autograd of the restated formulas is the oracle).
"""
import numpy as np
import torch
import torch.nn.functional as F



def _conv(w, name, x, stride=1, relu=False, tconv=False):
    k = w[name + "/kernel"]
    b = w.get(name + "/bias")
    wt = k.permute(4, 3, 0, 1, 2)
    ks = k.shape[0]
    pb = max(ks - 2, 0) // 2                        # TF 'SAME' front padding of the stride-2 pair (oracle/nets.py)
    pa = max(ks - 2, 0) - pb
    if tconv:
        n = x.shape[2]
        y = F.conv_transpose3d(x, wt, None, stride=2)[:, :, pb:pb + 2 * n, pb:pb + 2 * n, pb:pb + 2 * n]
        if b is not None:
            y = y + b.reshape(1, -1, 1, 1, 1)
    elif stride == 2:
        y = F.conv3d(F.pad(x, (pb, pa, pb, pa, pb, pa)), wt, b, stride=2)
    else:
        y = F.conv3d(x, wt, b, padding=(ks - 1) // 2)
    return torch.relu(y) if relu else y


def _vrn(w, p, x):
    t12 = _conv(w, p + "/conv1_2", _conv(w, p + "/conv1_1", x, relu=True), relu=True)
    t23 = _conv(w, p + "/conv2_3", _conv(w, p + "/conv2_2", _conv(w, p + "/conv2_1", x, relu=True), relu=True), relu=True)
    return torch.relu(x + torch.cat([t12, t23], 1))


def _analysis(w, x):
    p = "analysis_transform/"
    f = _conv(w, p + "conv_in", x, relu=True)
    for i in (1, 2, 3):
        f = _vrn(w, p + "vrn1_%d" % i, f)
    f = _conv(w, p + "down_1", f, stride=2, relu=True)
    for i in (1, 2, 3):
        f = _vrn(w, p + "vrn2_%d" % i, f)
    f = _conv(w, p + "down_2", f, stride=2, relu=True)
    for i in (1, 2, 3):
        f = _vrn(w, p + "vrn3_%d" % i, f)
    return _conv(w, p + "conv_out", f)


def _synthesis(w, y):
    p = "synthesis_transform/"
    f = _conv(w, p + "deconv_in", y, relu=True)
    for i in (1, 2, 3):
        f = _vrn(w, p + "vrn1_%d" % i, f)
    f = _conv(w, p + "up_1", f, relu=True, tconv=True)
    for i in (1, 2, 3):
        f = _vrn(w, p + "vrn2_%d" % i, f)
    f = _conv(w, p + "up_2", f, relu=True, tconv=True)
    for i in (1, 2, 3):
        f = _vrn(w, p + "vrn3_%d" % i, f)
    return _conv(w, p + "deconv_out", f)


def _eb_logits(w, x):          # x [C,1,n]
    logits = x
    for i in range(4):
        logits = torch.matmul(F.softplus(w["estimator/matrix_%d" % i]), logits) + w["estimator/bais_%d" % i]
        logits = logits + torch.tanh(w["estimator/factor_%d" % i]) * torch.tanh(logits)
    return logits


def _eb_likelihood(w, z):      # z [B,C,d,h,w] -> likelihood same shape
    C = z.shape[1]
    flat = z.transpose(0, 1).reshape(C, 1, -1)
    lower, upper = _eb_logits(w, flat - 0.5), _eb_logits(w, flat + 0.5)
    sign = -torch.sign(lower + upper).detach()
    lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
    return lik.reshape(C, z.shape[0], *z.shape[2:]).transpose(0, 1)


def _laplace_cdf(x, loc, scale):
    e = torch.exp(-torch.abs(x - loc) / scale)
    return torch.where(x <= loc, 0.5 * e, 1.0 - 0.5 * e)


def _sc_likelihood(y, loc, scale):
    upper, lower = y + 0.5, y - 0.5
    sign = torch.sign(upper + lower - loc).detach()
    upper = -sign * (upper - loc) + loc
    lower = -sign * (lower - loc) + loc
    return torch.abs(_laplace_cdf(upper, loc, scale) - _laplace_cdf(lower, loc, scale))


def forward_loss(weights, x, noise_y, noise_z, alpha, beta, gamma=1.0, delta=1.0, lower_bound=1e-9, requires_grad=True):
    """weights: flat dict of numpy arrays (checkpoint keys).  x [B,N,N,N,1] numpy.  Returns (loss terms dict,
    dict of torch leaf tensors with .grad filled when requires_grad)."""
    w = {k: torch.tensor(np.asarray(v, np.float32), requires_grad=requires_grad) for k, v in weights.items()}
    xt = torch.from_numpy(np.ascontiguousarray(x, np.float32)).permute(0, 4, 1, 2, 3)
    ny = torch.from_numpy(np.ascontiguousarray(noise_y, np.float32)).permute(0, 4, 1, 2, 3)
    nz = torch.from_numpy(np.ascontiguousarray(noise_z, np.float32)).permute(0, 4, 1, 2, 3)
    y = _analysis(w, xt)
    hp = "hyper_encoder/"
    z = _conv(w, hp + "conv3", _conv(w, hp + "conv2", _conv(w, hp + "conv1", y, relu=True), stride=2, relu=True))
    z_t = z + nz
    lik_z = torch.clamp_min(_eb_likelihood(w, z_t), 1e-9)
    hd = "hyper_decoder/"
    f = _conv(w, hd + "conv3", _conv(w, hd + "conv2", _conv(w, hd + "conv1", z_t, relu=True), relu=True, tconv=True), relu=True)
    loc = _conv(w, hd + "conv4_1", f)
    scale = torch.maximum(torch.abs(_conv(w, hd + "conv4_2", f)), torch.tensor(np.float32(lower_bound)))
    y_t = y + ny
    lik_y = torch.clamp_min(_sc_likelihood(y_t, loc, scale), 1e-9)
    x_t = _synthesis(w, y_t)
    num_points = (xt.sum(1) > 0).float().sum()
    bpp_y = torch.log(lik_y).sum() / (-np.log(2.0) * num_points)
    bpp_z = torch.log(lik_z).sum() / (-np.log(2.0) * num_points)
    occ = torch.clamp(torch.sigmoid(x_t), 1e-7, 1.0 - 1e-7)
    lab = xt.amax(1, keepdim=True)
    empty = (-torch.log(1.0 - occ))[lab == 0].mean()
    full = (-torch.log(occ))[lab > 0].mean()
    loss = alpha * (beta * empty + full) + delta * bpp_y + gamma * bpp_z
    if requires_grad:
        loss.backward()
    terms = {k: float(v.detach()) for k, v in dict(loss=loss, bpp_y=bpp_y, bpp_z=bpp_z, empty=empty, full=full).items()}
    return terms, w


def forward_loss_factorized(weights, x, noise_y, alpha, beta, model="model_voxception", requires_grad=True):
    """train_factorized.py:160-170 with torch autograd: y = A(x); y~ = y + U; p = EB(y~); x~ = S(y~);
    loss = alpha * (beta * empty + full) + sum(log p) / (-ln2 * num_points).  model_simple.py layers for model="model_simple"."""
    w = {k: torch.tensor(np.asarray(v, np.float32), requires_grad=requires_grad) for k, v in weights.items()}
    xt = torch.from_numpy(np.ascontiguousarray(x, np.float32)).permute(0, 4, 1, 2, 3)
    ny = torch.from_numpy(np.ascontiguousarray(noise_y, np.float32)).permute(0, 4, 1, 2, 3)
    if model == "model_simple":
        a, s_ = "analysis_transform/", "synthesis_transform/"
        y = _conv(w, a + "conv_3", _conv(w, a + "conv_2", _conv(w, a + "conv_1", xt, stride=2, relu=True), stride=2, relu=True), stride=2)
    else:
        y = _analysis(w, xt)
    y_t = y + ny
    lik = torch.clamp_min(_eb_likelihood(w, y_t), 1e-9)
    if model == "model_simple":
        x_t = _conv(w, s_ + "deconv_3", _conv(w, s_ + "deconv_2", _conv(w, s_ + "deconv_1", y_t, relu=True, tconv=True), relu=True, tconv=True),
                    tconv=True)
    else:
        x_t = _synthesis(w, y_t)
    num_points = (xt.sum(1) > 0).float().sum()
    bpp = torch.log(lik).sum() / (-np.log(2.0) * num_points)
    occ = torch.clamp(torch.sigmoid(x_t), 1e-7, 1.0 - 1e-7)
    lab = xt.amax(1, keepdim=True)
    empty = (-torch.log(1.0 - occ))[lab == 0].mean()
    full = (-torch.log(occ))[lab > 0].mean()
    loss = alpha * (beta * empty + full) + bpp
    if requires_grad:
        loss.backward()
    return {k: float(v.detach()) for k, v in dict(loss=loss, bpp=bpp, empty=empty, full=full).items()}, w


def adam_step(param, grad, m, v, t, lr=1e-5, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (TF1): returns (param, m, v) after step t (1-based)."""
    m = b1 * m + (1 - b1) * grad
    v = b2 * v + (1 - b2) * grad * grad
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return (param - lr_t * m / (np.sqrt(v) + eps)).astype(np.float32), m.astype(np.float32), v.astype(np.float32)
